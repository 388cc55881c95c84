#!/usr/bin/env python3
"""Per-kernel mean of each PMC counter from rocprofv3 --pmc CSV output directories (bprx kernels only), plus
(--json FILE) the per-launch figures bench.py quotes: HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB (gfx950 tallies the
128-B requests of a wide coalesced read as 64 B: MI355X_MICROARCH.md, HBM) and the MFMA busy share
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CU_CYCLES-equivalent), see mfma_busy() below."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

args = sys.argv[1:]
json_out = None
if args and args[0] == "--json":
    json_out, args = args[1], args[2:]
acc = defaultdict(lambda: defaultdict(list))
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+(<[^>]*>)?)", r.get("Kernel_Name", ""))
            if not m:
                continue
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
for k in sorted(mean):
    print(k)
    for c in sorted(mean[k]):
        print("   %-30s mean %.6g  (n=%d)" % (c, mean[k][c], len(acc[k][c])))


def mfma_busy(m):
    """Share of the chip's matrix-pipe cycles that were busy during the kernel: SQ_VALU_MFMA_BUSY_CYCLES counts cycles per
    SIMD pipe summed over the chip (MI355X_MICROARCH.md: = 32 x N for N v_mfma_f32_32x32x16_bf16); the denominator is the
    kernel's duration in shader cycles on every one of the 1024 SIMDs: GRBM_GUI_ACTIVE is reported summed over the 8 XCDs,
    so cycles = GRBM_GUI_ACTIVE / 8."""
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in m or not m.get("GRBM_GUI_ACTIVE"):
        return None
    return m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)


if json_out:
    phases = {"proj_fwd": ("k_proj_fwd_bf16_v10", "k_proj_fwd_f8s"), "proj_bwd": ("k_proj_bwd_bf16_v3",),
              "triplet_grad": ("k_triplet_seg", "k_triplet_grad"), "item_seg": ("k_item_seg",), "dense_update": ("k_dense_update",),
              "row_count": ("k_index_seg", "k_row_count"), "apply": ("k_apply_sgd",), "sampler": ("k_sample_epoch", "k_sample_philox")}
    out = {}
    for ph, pat in phases.items():
        ks = [k for k in mean if any(k.startswith(p_) for p_ in pat)]
        if not ks:
            continue
        k = max(ks, key=lambda q: len(acc[q].get("FETCH_SIZE", [])) + len(acc[q].get("SQ_WAVES", [])))
        m = mean[k]
        e = {"kernel": k}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e.update(FETCH_SIZE_KB=m["FETCH_SIZE"], WRITE_SIZE_KB=m["WRITE_SIZE"],
                     hbm_bytes=(2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0)
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
                  "SQ_INSTS_VALU_MFMA_MOPS_F8", "TCC_EA0_ATOMIC_sum", "SQ_LDS_BANK_CONFLICT"):
            if c in m:
                e[c] = m[c]
        b = mfma_busy(m)
        if b is not None:
            e["mfma_busy_frac"] = b
        out[ph] = e
    json.dump(out, open(json_out, "w"), indent=1)
