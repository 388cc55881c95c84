"""Host-side cost of EpochWalkSampler epoch starts (C2 shape): per-call host time of sample(), outliers listed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fashionvisualexpl_recommend_amd.engine import EpochWalkSampler
dev = torch.device("cuda", 0)
U, I, npu, B = 100_000, 50_000, 20, 65_536
g = torch.Generator(device=dev); g.manual_seed(1)
items = torch.randint(I, (U, npu), generator=g, device=dev, dtype=torch.int32).sort(dim=1).values
indptr = torch.arange(U + 1, device=dev, dtype=torch.int64) * npu
pos_user = torch.arange(U, device=dev, dtype=torch.int32).repeat_interleave(npu)
s = EpochWalkSampler.from_csr(indptr, items.reshape(-1), pos_user, I, seed=3)
bufs = tuple(torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for it in range(5):
        s.sample(B, out=bufs)
    st.synchronize()
    ts = []
    t_all = time.perf_counter()
    for it in range(400):
        t0 = time.perf_counter(); s.sample(B, out=bufs); ts.append(time.perf_counter() - t0)
    st.synchronize()
    t_all = time.perf_counter() - t_all
ts = np.array(ts) * 1e3
print("total %.1f ms for 400 samples; host per call: median %.3f ms, max %.1f ms; calls > 1 ms: %s" % (
    t_all * 1e3, np.median(ts), ts.max(), [(i, round(float(t), 1)) for i, t in enumerate(ts) if t > 1.0][:20]))
