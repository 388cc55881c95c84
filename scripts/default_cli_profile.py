"""Per-kernel times of the step at the reference's CLI defaults (batch 256, adam_tf23, k=128, d=20, fp32 features D=4096)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fashionvisualexpl_recommend_amd import synth
from fashionvisualexpl_recommend_amd.engine import Engine
U, I, k, d, D, B = 20000, 10000, 128, 20, 4096, 256
rs = np.random.RandomState(0)
for dtype, opt in (("fp32", "adam_tf23"), ("fp32", "sgd"), ("bf16", "adam_tf23"), ("bf16", "sgd")):
    F = np.abs(rs.standard_normal((I, D))).astype(np.float32); F /= F.max()
    t = dict(Gu=synth.glorot_uniform(rs, U, k), Gi=synth.glorot_uniform(rs, I, k), Bi=np.zeros(I, np.float32),
             Tu=synth.glorot_uniform(rs, U, d), F=F, E=synth.glorot_uniform(rs, D, d), Bp=synth.glorot_uniform(rs, D, 1).reshape(-1))
    e = Engine(model="vbpr", num_users=U, num_items=I, embed_k=k, embed_d=d, feat_dim=D, feat_dtype=dtype, optimizer=opt, lr=1e-3,
               reg=0.0, max_batch=B, device=0).bind(**{n: torch.as_tensor(v) for n, v in t.items()})
    st = torch.cuda.Stream()
    bt = [tuple(torch.as_tensor(rs.randint(n, size=B).astype(np.int32), device="cuda") for n in (U, I, I)) for _ in range(50)]
    with torch.cuda.stream(st):
        for b in bt[:10]:
            e.step(*b, want_loss=False)
        st.synchronize(); t0 = time.perf_counter()
        for r in range(4):
            for b in bt:
                e.step(*b, want_loss=False)
        st.synchronize(); dt = (time.perf_counter() - t0) / 200
        e.profile(True)
        for b in bt:
            e.step(*b, want_loss=False)
        st.synchronize()
        prof = e.profile_read(); e.profile(False)
    print(dtype, opt, "%.1f us/step" % (dt * 1e6), {p: round(ms / n * 1e3, 1) for p, (ms, n) in prof.items()}, flush=True)
    e.close()
