#!/usr/bin/env python3
"""How many of k_build_rows' pair tests could a candidate stream per PART of the wavefront save? A tile's 64 atoms are sorted by x; the kernel tests every
atom against every candidate that survives the cull (within the cutoff of the tile's bounding box). On the host, for a sample of real tiles: the culled
candidates (all atoms, ghosts included, within cutneigh of the box of the tile's atoms), the tile's union (candidates some atom keeps), and — for the
wavefront cut into 2 halves / 4 quarters of consecutive lanes — the candidates within cutneigh of the PART's own bounding box (what a part would have to
test if the buffered candidates were sorted by x and walked per part). Prints tests per tile as a fraction of the present 64 x culled.
    usage: tools/build_window_probe.py [size] [ntiles_sampled]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd

size = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 200
s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100], quiet=True)
s.initial()
s.run_steps(60)
h = s.handle
nt = h.neighbor_tile_stats()["tiles"]
d = h.download()
x = d["x"].astype(np.float64)
cut = 2.8                       # in.lj.miniMD: force cutoff 2.5 + skin 0.3
rng = np.random.default_rng(2)
tiles = rng.choice(nt, size=min(nsample, nt), replace=False)


def within_box(p, lo, hi, r):
    dd = np.maximum(np.maximum(lo - p, p - hi), 0.0)
    return (dd * dd).sum(axis=1) <= r * r


tot = {"culled": 0, "union": 0, "hits": 0, "halves": 0, "quarters": 0, "eighths": 0, "tiles": 0}
for t in tiles:
    rows, atoms, cand = h.neighbor_tile_rows(int(t))
    a = atoms[atoms >= 0]
    a = a[a < d["nlocal"]]
    if len(a) == 0:
        continue
    pa = x[a]
    lo, hi = pa.min(axis=0), pa.max(axis=0)
    # coarse pre-selection of all atoms near the box, then the exact box distance
    near = np.all((x >= lo - cut) & (x <= hi + cut), axis=1)
    idx = np.nonzero(near)[0]
    culled = idx[within_box(x[idx], lo, hi, cut)]
    pc = x[culled]
    tot["culled"] += len(culled) * 64
    tot["union"] += len(cand) * 64
    d2 = ((pa[:, None, :] - pc[None, :, :]) ** 2).sum(axis=2)
    tot["hits"] += int((d2 <= cut * cut).sum()) - len(a)
    order = np.argsort(pa[:, 0], kind="stable")          # (lanes are in x order to a quarter of a bin: take them exactly sorted)
    for name, parts in (("halves", 2), ("quarters", 4), ("eighths", 8)):
        n = 0
        for part in np.array_split(order, parts):
            if len(part) == 0:
                continue
            q = pa[part]
            n += int(within_box(pc, q.min(axis=0), q.max(axis=0), cut).sum()) * (64 // parts)
        tot[name] += n
    tot["tiles"] += 1
n = max(tot["tiles"], 1)
print("-s %d, %d tiles: culled candidates per tile %.0f, union %.0f, hits per atom %.1f" % (size, n, tot["culled"] / 64 / n, tot["union"] / 64 / n, tot["hits"] / 64 / n))
for k in ("union", "halves", "quarters", "eighths"):
    print("  pair tests with %-9s %.3f of the present 64 x culled" % (k + ":", tot[k] / max(tot["culled"], 1)))
s.close()
