#!/usr/bin/env python3
"""What does the second candidate list (ghosts named by owner + image code, written by k_build_rows for the tiles near a box face) cost the build?
mmd_profile_kernel(1) with ghost_resolve on / off, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import minimd_amd
size = int(sys.argv[1]) if len(sys.argv) > 1 else 80
sim = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 40]); sim.initial(); sim.run_steps(40)
h = sim.handle
for rnd in range(3):
    for v in (1, 0):
        h.set_option("ghost_resolve", v)
        h.neighbor_build()
        print("ghost_resolve=%d  neighbor build + binning %.4f ms" % (v, h.profile_kernel(1, 12)))
