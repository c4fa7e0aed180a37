#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 rocpd database (--kernel-trace): how much of the wall time of a
single-stream run is NOT covered by any kernel (launch / dependency latency between graph nodes).
Usage: rocpd_gaps.py results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select start, end from kernels order by start").fetchall()
busy = gap = 0
hist = {}
last_end = rows[0][0]
n_gaps = 0
for s, e in rows:
    if s > last_end:
        g = s - last_end
        if g < 200_000:  # ignore host-side pauses between phases of the script
            gap += g
            n_gaps += 1
            b = min(int(g / 1000), 20)
            hist[b] = hist.get(b, 0) + 1
    busy += max(0, e - max(s, last_end))
    last_end = max(last_end, e)
print(f"{len(rows)} kernels; covered by kernels {busy/1e6:.1f} ms; idle between kernels (gaps < 200 us) {gap/1e6:.1f} ms in {n_gaps} gaps "
      f"= {gap/max(n_gaps,1)/1e3:.2f} us per gap, {100*gap/(gap+busy):.1f} % of busy+idle")
print("gap histogram (us: count):", ", ".join(f"{k}{'+' if k == 20 else ''}: {v}" for k, v in sorted(hist.items())))
