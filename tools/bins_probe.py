#!/usr/bin/env python3
"""force-kernel time of in.lj.miniMD -s <size> full lists as a function of the bin count (-b): how full the 64-lane tiles are"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
size = int(sys.argv[1]) if len(sys.argv) > 1 else 80
for rep in range(2):
    for b in [int(a) for a in sys.argv[2:]] or [62, 63, 64, 65, 66, 67]:
        s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100, "-b", b])
        s.initial()
        s.run_steps(20)
        sec = s.run_steps(100)
        tm = s.handle.timers()
        st = s.handle.neighbor_tile_stats()
        k = s.handle.profile_kernel(0, 20)
        print("-b %d: %.1f Matom-steps/s  force %.4f ms/launch (kernel only %.4f)  neigh %.4f s  tiles %d  atoms/tile %.1f  rows %.1f  union %.0f" % (
            b, s.natoms() * 100 / sec / 1e6, tm["force_kernel_ms"] / max(tm["force_launches"], 1), k, tm["neigh"], st["tiles"], st["sum_atoms"] / st["tiles"],
            st["sum_padded_rows"] / st["tiles"], st["sum_candidates"] / st["tiles"]), flush=True)
        s.close()
