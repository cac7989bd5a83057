#!/usr/bin/env python3
"""How much of k_lj_full_tile waits for its slot stream (profiling build: tools/build_variant.sh profile all -DMMD_PROFILE, MMD_LIB_DIR=variants/profile):
ablate 4 lets every tile walk the rows of one of 64 tiles, so the 2-byte slots come from the L2 instead of the HBM; ablate 8 replaces the
slots by a pattern without LDS bank conflicts (results invalid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
size = int(sys.argv[1]) if len(sys.argv) > 1 else 80
s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100])
s.initial(); s.run_steps(45)
h = s.handle
for rnd in range(3):
    for ab in (0, 4, 8):
        h.set_option("ablate", ab)
        print("-s %d  ablate=%d  Force::compute %.4f ms" % (size, ab, h.profile_kernel(0, 20)), flush=True)
h.set_option("ablate", 0)
s.close()
