#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 rocpd database (--kernel-trace): how much of the wall time of a
single-stream run is NOT covered by any kernel (launch / dependency latency between graph nodes).
Usage: rocpd_gaps.py results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
ncol = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select start, end, {ncol} from kernels order by start").fetchall()
busy = gap = 0
hist = {}
last_end = rows[0][0]
n_gaps = 0
pairs = {}
prev_name = ""
for s, e, nm in rows:
    if s > last_end:
        g = s - last_end
        if g < 200_000:  # ignore host-side pauses between phases of the script
            gap += g
            n_gaps += 1
            b = min(int(g / 1000), 20)
            hist[b] = hist.get(b, 0) + 1
            if g >= 2000:
                k = (prev_name.split("(")[0][-60:], nm.split("(")[0][-60:])
                pairs[k] = (pairs.get(k, (0, 0))[0] + 1, pairs.get(k, (0, 0))[1] + g)
    prev_name = nm
    busy += max(0, e - max(s, last_end))
    last_end = max(last_end, e)
print(f"{len(rows)} kernels; covered by kernels {busy/1e6:.1f} ms; idle between kernels (gaps < 200 us) {gap/1e6:.1f} ms in {n_gaps} gaps "
      f"= {gap/max(n_gaps,1)/1e3:.2f} us per gap, {100*gap/(gap+busy):.1f} % of busy+idle")
print("gap histogram (us: count):", ", ".join(f"{k}{'+' if k == 20 else ''}: {v}" for k, v in sorted(hist.items())))
print("gaps >= 2 us by (previous kernel -> next kernel), top 12 by total time:")
for k, (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {n:6d} x {t / n / 1e3:5.1f} us  {k[0]}  ->  {k[1]}")
