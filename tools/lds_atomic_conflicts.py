#!/usr/bin/env python3
"""Half lists: how the partners of one wave instruction collide in the LDS accumulators (k_lj_half_tile). A `ds_add_f64` is serviced in 16-lane groups
on 32 banks (a double = one bank pair); lanes of a group on the same bank pair take turns — whether they name the SAME accumulator (adjacent atoms share
partners: same-address read-modify-writes cannot be merged) or different ones. Printed: mean LDS turns per (row, 16-lane group) as built, with the
same-address lanes counted once (what a gather of the same slots costs), and with every lane's row rotated by a lane-dependent offset.
    usage: tools/lds_atomic_conflicts.py [size] [ntiles_sampled]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd

size = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 200
REC = 24
s = minimd_amd.Sim(["-s", size, "--half_neigh", 1, "-n", 100])
s.initial()
s.run_steps(60)
h = s.handle
nt = h.neighbor_tile_stats()["tiles"]
rng = np.random.default_rng(1)
tiles = rng.choice(nt, size=min(nsample, nt), replace=False)


def turns(slots_row, nc, merge_same):
    tot = 0.0
    for g in range(4):
        v = slots_row[16 * g:16 * g + 16]
        v = v[v < nc]                              # (padding: masked out of the atomics)
        if len(v) == 0:
            continue
        if merge_same:
            v = np.unique(v)
        tot += np.bincount((3 * v) % 16, minlength=16).max()
    return tot / 4.0


acc = {}
for t in tiles:
    rows, atoms, cand = h.neighbor_tile_rows(int(t))
    km, nc = rows.shape[0], len(cand)
    if km == 0:
        continue
    slots = rows.astype(np.int64) // REC
    for name, sl, merge in (("as built", slots, False), ("as built, same-address lanes counted once", slots, True)):
        a = acc.setdefault(name, [0.0, 0])
        a[0] += sum(turns(sl[k], nc, merge) for k in range(km)); a[1] += km
    rot = np.full_like(slots, nc)
    for l in range(64):
        e = slots[:, l]
        e = e[e < nc]
        if len(e):
            r = (l * 7) % len(e)
            e = np.concatenate([e[r:], e[:r]])
            rot[:len(e), l] = e
    a = acc.setdefault("rows rotated by 7 x lane", [0.0, 0])
    a[0] += sum(turns(rot[k], nc, False) for k in range(km)); a[1] += km
print("-s %d half lists, %d of %d tiles; mean LDS turns per (row, 16-lane group) of one ds_add_f64 (1.0 = no collision)" % (size, len(tiles), nt))
for k_, (v, n) in acc.items():
    print("  %-48s %.3f" % (k_, v / max(n, 1)))
s.close()
