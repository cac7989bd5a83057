#!/usr/bin/env python3
"""Cross-check of bench.py's roofline.frac against the profiler, like for like: from a rocprofv3 kernel trace (rocpd sqlite) of `bench.py --steps K --windows W`
take the force launches of the run's step loop (every k_lj_full_tile launch except the force-only instantiation <0, 0> that mmd_profile_kernel / the clock
warm-up use) — the last W*K of them are the W timed windows — and print the average duration per window next to the line's own figures.
usage: tools/rocpd_timed_region.py <results.db> <bench.json> """
import json, re, sqlite3, sys
db, bj = sys.argv[1], sys.argv[2]
d = json.loads([l for l in open(bj) if l.startswith("{")][-1])
K, W = d["steps"], len(d["value_windows"])
rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
step = [(s, e) for n, s, e in rows if "k_lj_full_tile" in n and not re.search(r"k_lj_full_tile<0, 0>", n)]
r = d["roofline"]
print("line: kernel_ms %.5f (frac %.4f), sampled events %.5f (frac_sampled %.4f), device-clock span %.5f, %d launches" % (
    r["kernel_ms"], r["frac"], r["kernel_ms_sampled"], r["frac_sampled"], r.get("kernel_span_ms_all_launches") or 0, r["launches"]))
for w in range(W):
    seg = step[len(step) - (W - w) * K: len(step) - (W - w - 1) * K]
    avg = sum(e - s for s, e in seg) / len(seg) / 1e6
    print("trace: timed window %d: %d force launches, average duration %.5f ms%s" % (w + 1, len(seg), avg, ("  -> line / trace = %.4f" % (r["kernel_ms"] / avg)) if w == 0 else ""))
allf = [(e - s) for n, s, e in rows if re.search(r"k_lj_full_tile<0, 1>", n)]
print("trace: all %d launches of the fused instantiation in the process (equilibration, warm-up, all windows): average %.5f ms" % (len(allf), sum(allf) / len(allf) / 1e6))
