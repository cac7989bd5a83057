#!/usr/bin/env python3
"""Kernel timeline of a few PLAIN steps (between two re-neighborings) from a rocprofv3 (rocpd sqlite) kernel trace: every kernel between the
k-th and the (k+n)-th force launch after the last neighbor build but one. usage: tools/rocpd_steps.py <results.db> [nsteps=2]"""
import sqlite3
import sys

db = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
builds = [i for i, r in enumerate(rows) if "k_build_rows" in r[0]]
b = builds[-2] if len(builds) > 1 else builds[-1]
# first kernel after the 3rd "last force launch of a step" behind the build: steps are delimited by the integrator-carrying force launch
forces = [i for i in range(b, len(rows)) if "k_lj_full_tile" in rows[i][0] or "k_eam_force" in rows[i][0] or "k_lj_half_tile" in rows[i][0]]
lo = forces[6]
t0 = rows[lo][2]
prev_end = t0
cnt = 0
for n, s, e in rows[lo + 1:]:
    print("%9.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n.split("(")[0][:80]))
    prev_end = max(prev_end, e)
    cnt += 1
    if cnt > 14 * nsteps:
        break
