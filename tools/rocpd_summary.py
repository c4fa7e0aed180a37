#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel table:
calls, total ms, average us, share of GPU kernel time.  Usage: rocpd_summary.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {ncol}, count(*), sum(end-start), min(end-start), max(end-start) from kernels "
                     f"group by {ncol} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    span = c.execute("select min(start), max(end) from kernels").fetchone()
    lines = [f"# rocprofv3 --kernel-trace summary of `{db}`", "",
             f"total kernel time {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches; "
             f"first-to-last dispatch span {(span[1]-span[0])/1e6:.2f} ms", "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for n, k, t, mn, mx in rows:
        lines.append(f"| `{short(n)}` | {k} | {t/1e6:.2f} | {t/k/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*t/tot:.1f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
