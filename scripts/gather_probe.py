"""Random-row gather / update rates of this device (bprx_probe_row_gather): the practical roofline of the sparse kernels.
   python scripts/gather_probe.py  -> one JSON line per (row bytes, table size, mode)."""
import ctypes as C, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fashionvisualexpl_recommend_amd import _ffi

L = _ffi.lib()
dev = torch.device("cuda:0")
st = torch.cuda.Stream()
sink = torch.zeros(64, device=dev)
out = []
with torch.cuda.stream(st):
    sp = C.c_void_p(st.cuda_stream)
    for row_floats in (64, 128, 256):
        for rows in (1 << 16, 1 << 18, 1 << 20, 1 << 21):
            if rows * row_floats * 4 > (6 << 30):
                continue
            table = torch.ones((rows, row_floats), device=dev)
            for n in (131072, 196608):
                if n > rows:
                    continue
                for mode in (0, 1):
                    best = 0.0
                    for it in range(8):
                        idx = torch.randperm(rows, device=dev)[:n].to(torch.int32)
                        torch.cuda.synchronize()
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(st)
                        got = L.bprx_probe_row_gather(C.c_void_p(table.data_ptr()), rows, row_floats, C.c_void_p(idx.data_ptr()), n,
                                                      mode, C.c_void_p(sink.data_ptr()), sp)
                        b.record(st)
                        b.synchronize()
                        assert got > 0, got
                        if it >= 2:
                            best = max(best, got / (a.elapsed_time(b) * 1e-3) / 1e9)
                    rec = {"row_bytes": row_floats * 4, "table_MB": rows * row_floats * 4 / 1e6, "rows_per_launch": n,
                           "mode": "read+write" if mode else "read", "GBps": round(best, 1),
                           "us": round(n * row_floats * 4 * (2 if mode else 1) / best / 1e3, 2)}
                    out.append(rec)
                    print(json.dumps(rec), flush=True)
            del table
