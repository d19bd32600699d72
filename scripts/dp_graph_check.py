#!/usr/bin/env python3
"""The data-parallel train_ssl_uem step captured in ONE hipGraph (uemda_amd.step.GraphedStep(dp=wrapper)) against the same step run
eagerly, replay by replay, from the same complete state.  Launched with the torchrun environment (RANK / WORLD_SIZE / MASTER_*);
tests/test_gpu_dp.py runs it as one rank with UEM_DP_FORCE=1 -- real RCCL launches inside the graph, a trivial sum -- because RCCL
refuses two ranks on one device; on a multi-GPU node the same file runs under `python -m torch.distributed.run --nproc-per-node N`.
UEM_DP_BACKEND=gloo: the capture must be refused.  Prints DP_GRAPH_OK / DP_GRAPH_REFUSED on rank 0."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from uemda_amd import dp as udp
from uemda_amd.gast.alignment import Aligner
from uemda_amd.models.Encoder import Deeplabv2
from uemda_amd.ops import UemError
from uemda_amd.optim import FusedSGD
from uemda_amd.step import HYPER, GraphedStep, StepState, ssl_step
from uemda_amd.utils import synth
from uemda_amd.utils.tools import seed_torch

C = 6


def main():
    backend = os.environ.get("UEM_DP_BACKEND", "nccl")
    rank, world, local = udp.init(backend, device=0 if backend == "gloo" else None)
    torch.cuda.set_device(0 if backend == "gloo" else local)
    seed_torch(2333)
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333 + rank).items()}
    protos = synth.make_batch(B=1, H=32, W=32, C=C, k=2048, seed=2333)["prototypes"].cuda()

    def fresh():
        model = Deeplabv2(cfg).cuda()
        wrap = udp.DataParallel(model)
        al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        al.prototypes = protos.clone()
        return model, wrap, al, FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4), StepState(C)
    m1, w1, a1, o1, s1 = fresh()
    try:
        gs = GraphedStep(ssl_step, m1, a1, o1, s1, batch, warmup=2, lr=1e-3, sup_ignore_id=256, dp=w1)
    except UemError as e:
        assert backend != "nccl", e
        if rank == 0:
            print("DP_GRAPH_REFUSED", str(e)[:80])
        dist.destroy_process_group()
        return
    assert backend == "nccl"
    m2, w2, a2, o2, s2 = fresh()
    for lr in (3e-3, 3e-4):
        m2.load_state_dict({k: v.detach().clone() for k, v in m1.state_dict().items()})
        o2.momentum_buffer.copy_(o1.momentum_buffer)
        o2._steps = o1._steps
        a2.prototypes = a1.prototypes.clone()
        n = m1.flat_parameters()[2]
        before = m1.flat_parameters()[0][:n].clone()
        out = gs(lr)
        ref = ssl_step(m2, a2, o2, s2, batch, lr, sup_ignore_id=256, dp=w2)
        torch.cuda.synchronize()
        assert abs(float(out["loss_source"]) - float(ref["loss_source"])) <= 2e-6 * abs(float(ref["loss_source"]))
        assert torch.equal(out["label_t_hard"], ref["label_t_hard"])
        a, b = m1.flat_parameters()[0][:n], m2.flat_parameters()[0][:n]
        moved = float((a - before).norm())
        assert float((a - b).norm()) < 2e-3 * moved, (float((a - b).norm()), moved)
        torch.testing.assert_close(a1.prototypes, a2.prototypes, rtol=1e-5, atol=1e-6)
    gs.check()
    if world > 1:                                         # replicas stayed one model
        mine = torch.stack([m1.flat_parameters()[0][:n].double().sum(), a1.prototypes.double().sum()])
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        assert all(torch.equal(t, allr[0]) for t in allr)
    if rank == 0:
        print("DP_GRAPH_OK")
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
