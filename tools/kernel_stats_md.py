#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats kernel_stats.csv -> the markdown table kept under profiles/.
usage: kernel_stats_md.py kernel_stats.csv "<title line>" > profiles/rNN_..._kernel_stats.md"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)


print(f"# {sys.argv[2]}\n")
print(f"total kernel time {tot / 1e6:.2f} ms over {calls} dispatches\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
for r in rows:
    if int(r["TotalDurationNs"]) / tot < 0.001:
        continue
    print(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.2f} | "
          f"{int(r['MinNs']) / 1e3:.2f} | {int(r['MaxNs']) / 1e3:.2f} | {100 * int(r['TotalDurationNs']) / tot:.1f} |")
