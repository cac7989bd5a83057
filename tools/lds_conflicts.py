#!/usr/bin/env python3
"""LDS bank conflicts of the tile force kernels' position gathers, priced on the host from the real tile lists of a run.
A pair's three `ds_read_b64` (x, y, z of record `slot`, 24-byte records, DP) are serviced per 32-lane group; two lanes of a group that
read DIFFERENT records whose 8-byte words fall on the same bank pair ((3 slot + c) mod 32) cost one extra LDS cycle each
(MI355X_MICROARCH.md, LDS). For a sample of tiles this prints the mean LDS cycles per (row, lane group) — 1.0 = conflict-free — for
  * the list as built (slots in candidate-stream order, rows in the same order),
  * the same rows with the slots renumbered by alternative rules (what a different union order in the build would give),
  * every lane's row re-ordered by bank class rotated by the lane number (what a class-major expansion in the build would give).
    usage: tools/lds_conflicts.py [size] [ntiles_sampled]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd

size = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 300
REC = 24                          # bytes per {x,y,z} record (DP)
s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100])
s.initial()
s.run_steps(60)
h = s.handle
st = h.neighbor_tile_stats()
nt = st["tiles"]
d = h.download()
x = d["x"]
rng = np.random.default_rng(1)
tiles = rng.choice(nt, size=min(nsample, nt), replace=False)


def cycles(slots_row):
    """LDS cycles of one ds_read_b64 over a 64-lane row of record numbers (dummy = -1 entries still read a record: kept as they are)"""
    tot = 0
    for g in (0, 1):
        v = np.unique(slots_row[32 * g:32 * g + 32])
        cls = (3 * v) % 32
        tot += np.bincount(cls, minlength=32).max()
    return tot / 2.0


acc = {}


def add(name, val, n):
    a = acc.setdefault(name, [0.0, 0])
    a[0] += val
    a[1] += n


for t in tiles:
    rows, atoms, cand = h.neighbor_tile_rows(int(t))
    km = rows.shape[0]
    if km == 0:
        continue
    slots = rows.astype(np.int64) // REC            # record numbers; the dummy record is number len(cand)
    nc = len(cand)
    # ---- as built
    add("as built (stream order)", sum(cycles(slots[k]) for k in range(km)), km)
    # ---- slots renumbered: by x of the candidate / random / x-rank with a stride-11 shuffle
    xc = x[np.minimum(cand, len(x) - 1), 0]
    for name, perm in (("slots renumbered by x", np.argsort(np.argsort(xc, kind="stable"))),
                       ("slots renumbered at random", rng.permutation(nc)),
                       ("slots renumbered: x rank * 11 mod 32 classes", None)):
        if perm is None:
            xr = np.argsort(np.argsort(xc, kind="stable"))
            # class = (11 * xrank) mod 32, unique record number = class + 32 * (how many earlier candidates share the class)
            cls = (11 * xr) % 32
            order = np.argsort(xr)
            cnt = np.zeros(32, np.int64)
            perm = np.zeros(nc, np.int64)
            for c_ in order:
                # record numbers whose bank class (3 r mod 32) equals cls: r = inv3 * cls mod 32 (inv3 = 11)
                r0 = (11 * cls[c_]) % 32
                perm[c_] = r0 + 32 * cnt[cls[c_]]
                cnt[cls[c_]] += 1
        pm = np.concatenate([perm, [nc + 64]])          # dummy keeps a number of its own
        add(name, sum(cycles(pm[np.minimum(slots[k], nc)]) for k in range(km)), km)
    # ---- rows re-ordered: every lane walks its entries by bank class, starting at the class of its lane number
    re = np.full((km, 64), nc, np.int64)
    for l in range(64):
        e = slots[:, l]
        e = e[e < nc]
        key = ((3 * e) % 32 - l) % 32
        e = e[np.argsort(key, kind="stable")]
        re[:len(e), l] = e
    add("rows re-ordered by bank class, rotated by lane", sum(cycles(re[k]) for k in range(km)), km)
    # ---- rows re-ordered: strict diagonal — in row k lane l takes an entry of class (l + k) mod 32 if it still has one, otherwise of the
    # class it has most entries left of ("diagonal with fallback": what a class-indexed walk in the build could do, lanes independent)
    re2 = np.full((km, 64), nc, np.int64)
    for l in range(64):
        e = slots[:, l]
        e = e[e < nc]
        buckets = [list(e[((3 * e) % 32) == c]) for c in range(32)]
        for k in range(len(e)):
            c = (l + k) % 32
            if not buckets[c]:
                c = max(range(32), key=lambda q: len(buckets[q]))
            re2[k, l] = buckets[c].pop()
    add("rows re-ordered: diagonal class walk with fallback", sum(cycles(re2[k]) for k in range(km)), km)
    # ---- rows re-ordered greedily with knowledge of the other lanes (an upper bound on what any ordering can reach): row by row, lanes in turn
    # take an entry of a class nobody of their 32-lane group has taken in this row, preferring the class they have most entries of
    re3 = np.full((km, 64), nc, np.int64)
    bk = []
    for l in range(64):
        e = slots[:, l]
        e = e[e < nc]
        bk.append([list(e[((3 * e) % 32) == c]) for c in range(32)])
    for k in range(km):
        for g in (0, 1):
            taken = set()
            for l in range(32 * g, 32 * g + 32):
                b = bk[l]
                avail = [c for c in range(32) if b[c]]
                if not avail:
                    continue
                free = [c for c in avail if c not in taken]
                c = max(free or avail, key=lambda q: len(b[q]))
                taken.add(c)
                re3[k, l] = b[c].pop()
    add("rows re-ordered greedily across the lanes (bound)", sum(cycles(re3[k]) for k in range(km)), km)
    # the same with the padding spread: a lane whose row is shorter than the tile's idles at the END (as now)
print("-s %d, %d of %d tiles sampled; mean LDS cycles per (row, 32-lane group) of one ds_read_b64 (1.0 = no bank conflict)" % (size, len(tiles), nt))
for k_, (v, n) in acc.items():
    print("  %-52s %.3f" % (k_, v / max(n, 1)))
s.close()
