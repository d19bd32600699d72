"""Nested attribute dict with recursive update: the part of `ever.ERModule`'s config handling the
reference relies on (`self.config.backbone.resnet_type`, `**self.config.ppm`; Encoder.py:88-110,167-186)."""


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def update(self, other=(), **kw):
        for k, v in dict(other, **kw).items():
            if isinstance(v, dict):
                cur = self.get(k)
                if not isinstance(cur, AttrDict):
                    cur = AttrDict()
                    dict.__setitem__(self, k, cur)
                cur.update(v)
            else:
                dict.__setitem__(self, k, v)
