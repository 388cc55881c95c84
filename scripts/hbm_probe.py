#!/usr/bin/env python3
"""What this box's HBM delivers to plain streaming kernels (torch ops): the practical ceiling next to the 8 TB/s peak."""
import json
import torch

dev = torch.device("cuda", 0)
out = {}
for mb in (410, 1640):
    n = mb * 1024 * 1024 // 2
    x = torch.ones(n, dtype=torch.bfloat16, device=dev)
    y = torch.empty_like(x)
    xf = x.view(torch.float32)

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    t = timeit(lambda: xf.sum())
    out["read_sum_f32_%dMB" % mb] = round(n * 2 / t / 1e9, 1)
    t = timeit(lambda: y.copy_(x))
    out["copy_%dMB_rw" % mb] = round(2 * n * 2 / t / 1e9, 1)
    t = timeit(lambda: y.zero_())
    out["fill_%dMB" % mb] = round(n * 2 / t / 1e9, 1)
    del x, y, xf
print(json.dumps(out))
