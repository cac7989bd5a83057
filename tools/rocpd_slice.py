#!/usr/bin/env python3
"""One run slice (mmd_integrate_run call) from a rocprofv3 (rocpd sqlite) kernel trace of bench.py: slices start with k_initial_integrate. Prints the
kernels of slice number k (default: the 4th from the end = bench.py's first timed window when three windows follow the warm-up) with every gap above
1 us, and the sums. usage: tools/rocpd_slice.py <results.db> [k_from_end=3]"""
import sqlite3
import sys

db = sys.argv[1]
kfe = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if r[0].startswith("k_initial_integrate")]
lo = starts[-kfe]
hi = starts[-kfe + 1] if kfe > 1 else len(rows)
t0 = rows[lo][1]
prev_end = t0
busy = 0.0
gaps = 0.0
for n, s, e in rows[lo:hi]:
    g = (s - prev_end) / 1e3
    if g > 1.0 or not ("k_lj_full_tile" in n):
        print("%9.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, g, (e - s) / 1e3, n.split("(")[0][:70]))
    busy += (e - s) / 1e3
    gaps += max(g, 0.0)
    prev_end = max(prev_end, e)
print("slice: %d kernels, %.1f us from first start to last end, kernel time %.1f us, gaps %.1f us" % (hi - lo, (prev_end - t0) / 1e3, busy, gaps))
