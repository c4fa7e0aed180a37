#!/usr/bin/env python3
"""One score evaluation as a list of dispatches in launch order, from a rocprofv3 rocpd database (--kernel-trace) of a
single-stream run: the median duration of every position of the repeating launch sequence.
Usage: rocpd_timeline.py results.db period_start_kernel_substring [out.md]"""
import re
import sqlite3
import statistics
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("unsigned short", "bf16")[:70]


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {ncol}, grid_x, grid_y, grid_z, start, end from kernels order by start").fetchall()
    key = sys.argv[2]
    starts = [i for i, r in enumerate(rows) if key in r[0]]
    periods = [(a, b) for a, b in zip(starts, starts[1:])]
    n = statistics.mode(b - a for a, b in periods)
    periods = [(a, b) for a, b in periods if b - a == n][-40:]
    lines = [f"# one score evaluation in launch order: median over {len(periods)} evaluations of `{sys.argv[1]}`", "",
             "| # | kernel | grid | us | cumulative us |", "|---:|---|---|---:|---:|"]
    tot = 0.0
    for j in range(n):
        d = statistics.median((rows[a + j][5] - rows[a + j][4]) / 1e3 for a, _ in periods)
        r = rows[periods[0][0] + j]
        tot += d
        lines.append(f"| {j} | `{short(r[0])}` | {r[1]}x{r[2]}x{r[3]} | {d:.1f} | {tot:.0f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
