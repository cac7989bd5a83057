#!/usr/bin/env python3
"""Why is k_build_rows inside a run ~90 us slower than the same build repeated on resident data (mmd_profile_kernel)? Under rocprofv3 --kernel-trace:
19 plain steps, then Neighbor::build three times back to back, then a slice with a real in-run re-neighboring; tools/probes/build_cold_list.py prints the
durations of every k_build_rows launch with the kernel in front of it.
  (cd /tmp && rocprofv3 --kernel-trace -d <out> -o t -- python tools/probes/build_cold_probe.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import minimd_amd
sim = minimd_amd.Sim(["-s", 80, "--half_neigh", 0, "-n", 100])
sim.initial()
sim.run_steps(59)
h = sim.handle
for i in range(3):
    h.neighbor_build()
sim.run_steps(41)
for i in range(3):
    h.neighbor_build()
h.profile_kernel(1, 4)
