#!/usr/bin/env python3
"""Run ONE BASELINE.json configuration (A, B, Bh, C, E, Ef) for 100 + 20 steps — the command rocprofv3 wraps for the per-configuration kernel statistics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd

CONFIGS = {
    "A": (["-s", 32, "--half_neigh", 0, "-n", 100], "dp"),
    "B": (["-s", 80, "--half_neigh", 0, "-n", 100], "dp"),
    "Bh": (["-s", 80, "--half_neigh", 1, "-n", 100], "dp"),
    "C": (["-i", "in.eam.miniMD", "-s", 64, "--half_neigh", 0, "-n", 100], "dp"),
    "Ch": (["-i", "in.eam.miniMD", "-s", 64, "--half_neigh", 1, "-n", 100], "dp"),
    "E": (["-s", 160, "--half_neigh", 1, "-n", 100], "sp"),
    "Ef": (["-s", 160, "--half_neigh", 0, "-n", 100], "sp"),
}
args, prec = CONFIGS[sys.argv[1]]
s = minimd_amd.Sim(args, precision=prec)
s.initial()
s.run_steps(20)
sec = s.run_steps(100)
tm = s.handle.timers()
print("%s: %.1f Matom-steps/s  %.3f ms/step  force %.4f ms/launch  last row %s" % (
    sys.argv[1], s.natoms() * 100 / sec / 1e6, sec * 10, tm["force_kernel_ms"] / max(tm["force_launches"], 1), s.rows()[-1]))
s.close()
