"""Does a table survive in the Infinity Cache next to an nt stream?  Loop: read A (a MB, default policy), read B (b MB, default
or nt); report A's and B's rates.  python scripts/mall_probe.py"""
import ctypes as C, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fashionvisualexpl_recommend_amd import _ffi
L = _ffi.lib()
dev = torch.device("cuda:0")
st = torch.cuda.Stream()
sink = torch.zeros(4096, device=dev)
def rd(buf, nt):
    f = L.bprx_probe_stream_read_nt if nt else L.bprx_probe_stream_read
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    got = f(C.c_void_p(buf.data_ptr()), buf.numel() * 4, C.c_void_p(sink.data_ptr()), C.c_void_p(st.cuda_stream))
    b.record(st)
    assert got > 0
    return got, a, b
with torch.cuda.stream(st):
    for a_mb in (64, 100, 128, 160, 200):
        for b_mb in (320, 512):
            A = torch.ones(a_mb * (1 << 20) // 4, device=dev)
            Bf = torch.ones(b_mb * (1 << 20) // 4, device=dev)
            for a_nt, b_nt in ((False, False), (False, True), (True, True)):
                recs = []
                for it in range(8):
                    ga, a0, a1 = rd(A, a_nt)
                    gb, b0, b1 = rd(Bf, b_nt)
                    b1.synchronize()
                    if it >= 3:
                        recs.append((ga / a0.elapsed_time(a1) / 1e6, gb / b0.elapsed_time(b1) / 1e6))
                ra = sorted(r[0] for r in recs)[len(recs) // 2]
                rb = sorted(r[1] for r in recs)[len(recs) // 2]
                print(json.dumps({"A_MB": a_mb, "B_MB": b_mb, "A_policy": "nt" if a_nt else "default", "B_policy": "nt" if b_nt else "default",
                                  "A_GBps": round(ra, 1), "B_GBps": round(rb, 1)}), flush=True)
            del A, Bf
