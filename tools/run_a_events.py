#!/usr/bin/env python3
"""-s 32 / -s 80 with and without the per-step force-kernel event pairs (how much the two stream markers cost a step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
for size in (32, 80):
    for ev in (1, 0):
        for resolve in (1, 0):
            s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100], precision="dp")
            s.handle.set_option("time_force_events", ev)
            s.handle.set_option("ghost_resolve", resolve)
            s.initial()
            s.run_steps(20)
            best = min(s.run_steps(100) for _ in range(3))
            print("-s %d events %d resolve %d: %.1f Matom-steps/s  %.4f ms/step" % (size, ev, resolve, s.natoms() * 100 / best / 1e6, best * 10), flush=True)
            s.close()
