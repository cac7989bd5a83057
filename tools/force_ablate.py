#!/usr/bin/env python3
"""Phases of the tile force kernels switched off one at a time (profiling build only: tools/build_variant.sh profile all -DMMD_PROFILE, run
with MMD_LIB_DIR=variants/profile): ablate 1 = no staging of the candidates, 2 = no pair loop, 3 = neither (what is left: launch, the tiles'
load round trips, barriers, epilogue), 4 = EAM: the core part of the rows whatever the displacement. Results of an ablated run are invalid;
note that without staging the LDS holds stale positions, so the EAM knot gathers turn conflict-free (1 overstates the staging).
    usage: tools/force_ablate.py [lj|eam] [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
style = sys.argv[1] if len(sys.argv) > 1 else "lj"
size = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if style == "eam" else 80)
args = (["-i", "in.eam.miniMD"] if style == "eam" else []) + ["-s", size, "--half_neigh", 0, "-n", 100]
s = minimd_amd.Sim(args)
s.initial(); s.run_steps(45)
h = s.handle
for ab in (0, 1, 2, 3, 4, 0):
    h.set_option("ablate", ab)
    print("%s -s %d  ablate=%d  Force::compute %.4f ms" % (style, size, ab, h.profile_kernel(0, 10)), flush=True)
s.close()
