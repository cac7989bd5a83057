#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
usage: tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for n, k, s, a, mn, mx in rows:
        short = n if len(n) < 90 else n[:87] + "..."
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (short, k, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
