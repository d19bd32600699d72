#!/usr/bin/env python3
"""What the memory system gives a pure write stream, a pure read stream and a copy (cold operands: a pool of buffers larger than
the 256 MB infinity cache): the ceilings the write-heavy conv epilogues (64 -> 256 channels) and the elementwise passes sit under."""
import torch


def run(fn, pool, reps=3):
    torch.cuda.synchronize()
    for b in pool:
        fn(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for b in pool:
            fn(b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(pool))


def main():
    n = 512 * 1024 * 1024 // 4                      # 512 MB fp32 per buffer
    pool = [torch.empty(n, device="cuda") for _ in range(6)]
    dst = [torch.empty(n, device="cuda") for _ in range(6)]
    t_w = run(lambda b: b.fill_(1.0), pool)
    t_r = run(lambda b: b.sum(), pool)
    pairs = list(zip(pool, dst))
    t_c = run(lambda p: p[1].copy_(p[0]), pairs)
    gb = n * 4 / 1e9
    print(f"write-only {gb / t_w * 1e3:7.0f} GB/s   read-only (sum) {gb / t_r * 1e3:7.0f} GB/s   copy {2 * gb / t_c * 1e3:7.0f} GB/s (read + write)")


if __name__ == "__main__":
    main()
