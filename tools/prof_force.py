#!/usr/bin/env python3
"""Quick A/B timing of the hot kernels on the BASELINE workload (hipEvents on the compute stream).
usage: python tools/prof_force.py [--size 80] [--steps 100] [--ab tiles]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=80)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--half", type=int, default=0)
ap.add_argument("--deck", default="in.lj.miniMD")
ap.add_argument("--prec", default="dp")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--kernels", default="0,1,2,3,4")
ap.add_argument("--ab", default="", help="option to flip for an A/B of the force kernel, e.g. tiles")
a = ap.parse_args()

sim = minimd_amd.Sim(["-i", a.deck, "-s", a.size, "--half_neigh", a.half, "-n", a.steps], precision=a.prec)
sim.initial()
sim.run_steps(a.steps)
h = sim.handle
nl, ng, _ = h.counts()
info = h.neighbor_info()
print("atoms", nl, "ghosts", ng, "kbar %.2f" % (info["total"] / nl), "maxneighs", info["maxneighs"], "max_row", info["max_row"])
print("tile stats", h.neighbor_tile_stats())
names = {0: "force", 1: "neighbor_build(+binning)", 2: "initial_integrate", 3: "final_integrate", 4: "communicate"}
for k in [int(q) for q in a.kernels.split(",")]:
    ms = h.profile_kernel(k, a.reps if k != 1 else 12)
    print("%-28s %.4f ms" % (names[k], ms))
if a.ab:
    for rnd in range(3):
        for v in (1, 0):
            h.set_option(a.ab, v)
            print("round %d  %s=%d  force %.4f ms" % (rnd, a.ab, v, h.profile_kernel(0, a.reps)))
if os.environ.get("BUILDABLATE"):
    for ab in (0, 1, 2, 0):
        h.set_option("ablate", ab)
        print("ablate=%d  k_build_rows path: neighbor build %.4f ms" % (ab, h.profile_kernel(1, 12)))
    h.set_option("ablate", 0)
    h.neighbor_build()
if os.environ.get("BUILDAB"):
    for ab in (0, 1, 8, 9, 13, 16):
        h.set_option("ablate", ab)
        try:
            print("ablate=%d  neighbor build %.4f ms" % (ab, h.profile_kernel(1, 12)))
        except Exception as e:
            print("ablate=%d failed: %s" % (ab, e))
    h.set_option("ablate", 0)
    h.neighbor_build()
if os.environ.get("ABLATE"):
    h.set_option("tiles", 1)
    for ab in (0, 1, 2, 3, 4, 8, 24, 0):
        h.set_option("ablate", ab)
        print("ablate=%d  tile force %.4f ms" % (ab, h.profile_kernel(0, a.reps)))
    h.set_option("tiles", 0)
    for ab in (0, 4, 8, 16, 32):
        h.set_option("ablate", ab)
        print("ablate=%d  generic force %.4f ms" % (ab, h.profile_kernel(0, a.reps)))
    h.set_option("ablate", 0)
sim.close()
