#!/usr/bin/env python3
"""Kernel timeline of ONE re-neighboring from a rocprofv3 (rocpd sqlite) kernel trace: start offset, duration and the idle gap
before each kernel, from the last force kernel before the rebuild to the first force kernel after it.
usage: tools/rocpd_timeline.py <results.db> [which rebuild, default -1 = last]"""
import sqlite3
import sys

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_build_rows" in r[0] or "k_build_tiles" in r[0]]
b = idx[which]
lo = b
while lo > 0 and not any(k in rows[lo][0] for k in ("k_lj_", "k_eam_force", "k_eam_half_force")):
    lo -= 1
hi = b
while hi < len(rows) - 1 and not any(k in rows[hi][0] for k in ("k_lj_", "k_eam_force", "k_eam_half_force")):
    hi += 1
t0 = rows[lo][2]
prev_end = t0
busy = 0
print("re-neighboring window: %d kernels, %.1f us from the end of the previous force kernel to the start of the next" % (hi - lo - 1, (rows[hi][1] - t0) / 1e3))
for n, s, e in rows[lo + 1:hi + 1]:
    print("%9.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n.split("(")[0][:70]))
    busy += e - s
    prev_end = max(prev_end, e)
print("kernel time inside the window %.1f us" % ((busy - (rows[hi][2] - rows[hi][1])) / 1e3))
