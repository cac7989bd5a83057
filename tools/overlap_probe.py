#!/usr/bin/env python3
"""How much of Force::compute fits UNDER Neighbor::build when the two run at the same time on one GPU? Two independent systems of the same size
(two handles, two streams, two host threads): the time of nb builds of one and nf force launches of the other, alone and together. The build is
bound by instruction issue, the force kernel by HBM and the LDS: `together` well below `alone + alone` would pay for building a
re-neighboring step's lists in chunks with the force kernel of the finished chunk under the build of the next.
    usage: tools/overlap_probe.py [size] [nbuilds]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd

size = int(sys.argv[1]) if len(sys.argv) > 1 else 80
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sims = []
for _ in range(2):
    s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100], quiet=True)
    s.initial()
    s.run_steps(40)
    sims.append(s)
A, B = sims[0].handle, sims[1].handle
tb = A.profile_kernel(1, nb)
tf = B.profile_kernel(0, 4 * nb)
nf = max(int(round(nb * tb / tf)), 1)
print("-s %d alone: build %.4f ms, force %.4f ms; %d builds ~ %d force launches" % (size, tb, tf, nb, nf))
for rnd in range(3):
    t0 = time.perf_counter(); A.profile_kernel(1, nb); A.sync(); t_b = time.perf_counter() - t0
    t0 = time.perf_counter(); B.profile_kernel(0, nf); B.sync(); t_f = time.perf_counter() - t0
    th = threading.Thread(target=lambda: (A.profile_kernel(1, nb), A.sync()))
    t0 = time.perf_counter()
    th.start()
    B.profile_kernel(0, nf); B.sync()
    t_fin_f = time.perf_counter() - t0
    th.join()
    t_both = time.perf_counter() - t0
    print("round %d: builds alone %.2f ms, force alone %.2f ms, sum %.2f ms; together %.2f ms (force side done after %.2f) -> %.1f %% of the sum" % (
        rnd, t_b * 1e3, t_f * 1e3, (t_b + t_f) * 1e3, t_both * 1e3, t_fin_f * 1e3, 100 * t_both / (t_b + t_f)))
for s in sims:
    s.close()
