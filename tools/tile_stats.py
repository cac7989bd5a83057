#!/usr/bin/env python3
"""tile statistics of a configuration: tools/tile_stats.py <size> <half 0/1> [deck]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
size, half = int(sys.argv[1]), int(sys.argv[2])
deck = sys.argv[3] if len(sys.argv) > 3 else "in.lj.miniMD"
s = minimd_amd.Sim(["-i", deck, "-s", size, "--half_neigh", half, "-n", 100])
s.initial()
for label, n in (("step 0", 0), ("after 100 steps", 100)):
    if n: s.run_steps(n)
    st = s.handle.neighbor_tile_stats()
    info = s.handle.neighbor_info()
    print(label, st, "total", info["total"], flush=True)
    hc, hr = s.handle.neighbor_tile_histogram(64, 16)
    print("  union size histogram (bins of 16):", {16 * k: v for k, v in enumerate(hc) if v})
    print("  padded rows histogram (bins of 4): ", {4 * k: v for k, v in enumerate(s.handle.neighbor_tile_histogram(64, 4)[1]) if v}, flush=True)
s.close()
