#!/usr/bin/env python3
"""Per (kernel, grid) table from a rocprofv3 rocpd SQLite database (--kernel-trace): which SHAPES of a kernel template
take the time.  Usage: rocpd_by_shape.py results.db [out.md] [top_n]"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "")[:90]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    gcols = [k for k in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z", "grid_size") if k in cols]
    wcols = [k for k in ("workgroup_x", "workgroup_size_x", "workgroup_size") if k in cols]
    sel = ", ".join([ncol] + gcols + wcols)
    rows = c.execute(f"select {sel}, count(*), sum(end-start) from kernels group by {sel} order by sum(end-start) desc").fetchall()
    tot = sum(r[-1] for r in rows)
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = [f"# rocprofv3 --kernel-trace, per (kernel, grid) — `{db}`", "", f"columns: {gcols + wcols}; total kernel time {tot/1e6:.2f} ms", "",
             "| kernel | grid / block | calls | total ms | avg us | % |", "|---|---|---:|---:|---:|---:|"]
    for r in rows[:top]:
        n, g, k, t = r[0], r[1:-2], r[-2], r[-1]
        lines.append(f"| `{short(n)}` | {'x'.join(str(v) for v in g)} | {k} | {t/1e6:.2f} | {t/k/1e3:.1f} | {100*t/tot:.1f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
