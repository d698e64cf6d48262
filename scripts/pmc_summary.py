#!/usr/bin/env python
"""Summarise a rocprofv3 counter_collection.csv: per kernel name, mean of every counter over its dispatches."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "?")
    if "conv_igemm" not in k and "conv3x3_fused" not in k and len(sys.argv) < 3:
        continue
    key = (k[:70], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in acc.items():
    print(key)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} mean {sum(v)/len(v):.4g}  n={len(v)}")
