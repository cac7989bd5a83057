#!/usr/bin/env python3
"""Prices two proposals on REAL tile lists (mmd_neighbor_tile_rows), on the host:
 (1) half lists: one workgroup walks a run of L consecutive tiles of a pencil with a shared accumulator window and flushes a partner's accumulator only
     when the window slides past it -> global atomics per launch = sum over runs of |union of the run's candidate unions| instead of sum over tiles of |union|;
 (2) EAM (round-4 verdict, item 5): fuse density sweep -> fp -> force sweep inside one workgroup for tiles whose WHOLE candidate union is owned by the workgroup's own
     run of tiles -> fraction of tiles that qualify, for runs of L consecutive tiles (and for whole pencils).
usage: tools/window_probe.py <deck> <size> <half 0/1> [first tile] [ntiles]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minimd_amd
deck, size, half = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
s = minimd_amd.Sim(["-i", deck, "-s", size, "--half_neigh", half, "-n", 100] + (["-gn", 0] if "eam" in deck else []))
s.initial(); s.run_steps(100)
h = s.handle
st = h.neighbor_tile_stats()
nt = st["tiles"]
first = int(sys.argv[4]) if len(sys.argv) > 4 else nt // 2
n = min(int(sys.argv[5]) if len(sys.argv) > 5 else 1536, nt - first)
unions, owned = [], []
for t in range(first, first + n):
    rows, atoms, cand = h.neighbor_tile_rows(t)
    unions.append(np.unique(cand)); owned.append(atoms[atoms >= 0])
# a run breaks where consecutive tiles stop sharing candidates (the end of a pencil)
share = [len(np.intersect1d(unions[k], unions[k + 1])) / max(len(unions[k]), 1) for k in range(n - 1)]
breaks = [0] + [k + 1 for k, v in enumerate(share) if v < 0.15] + [n]
pencils = [(breaks[k], breaks[k + 1]) for k in range(len(breaks) - 1) if breaks[k + 1] > breaks[k]]
tot = sum(len(u) for u in unions)
print("%s -s %d half %d: %d tiles sampled (%d..%d of %d), %d pencil pieces of %.1f tiles, mean union %.1f, consecutive tiles of a pencil share %.0f %% of their unions" % (
    deck, size, half, n, first, first + n, nt, len(pencils), n / len(pencils), tot / n, 100 * np.mean([v for v in share if v >= 0.15])))
for L in (1, 2, 4, 8, 16, 10 ** 6):
    flushed, runs, qual = 0, 0, 0
    for a, b in pencils:
        for r0 in range(a, b, L):
            r1 = min(r0 + L, b)
            u = np.unique(np.concatenate(unions[r0:r1]))
            flushed += len(u); runs += 1
            own = np.unique(np.concatenate(owned[r0:r1]))
            for t in range(r0, r1):
                qual += int(np.all(np.isin(unions[t], own)))
    print("  runs of %-8s tiles: %6d runs, accumulators flushed %.3f of today's (global atomics per launch), tiles whose whole union the run owns: %d of %d" % (
        "whole-pencil" if L > 1000 else L, runs, flushed / tot, qual, n))
s.close()
