#!/usr/bin/env python3
"""Per-kernel mean of each PMC counter from rocprofv3 --pmc CSV output directories (bprx kernels only)."""
import csv
import glob
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+(<[^>]*>)?)", r.get("Kernel_Name", ""))
            if not m:
                continue
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-24s mean %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
