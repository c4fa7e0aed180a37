#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel name over the passes in a directory (p1/, p2/, ...)."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "p*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "conv" not in k and len(sys.argv) < 3:
        continue
    print("==", k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
