#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace as a per-kernel stats table (what
`rocprofv3 --stats` prints as *_kernel_stats.csv).  usage: summarize_rocpd.py results.db [out.md]"""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = c.execute(f"""select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
                         from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc""").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, t, a, mn, mx in rows:
        short = name.split("(")[0][:80]
        lines.append(f"| `{short}` | {n} | {t/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*t/tot:.1f} |")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "a").write(txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:3])
