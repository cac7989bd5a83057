#!/usr/bin/env python3
"""Run the BASELINE.json configurations once (20 steps + 0.4 s of clock warm-up, then 100 timed steps) and print Matom-steps/s + phase split + force-kernel time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd

CONFIGS = [
    ("A  lj -s 32 full DP", ["-s", 32, "--half_neigh", 0, "-n", 100], "dp"),
    ("B  lj -s 80 full DP", ["-s", 80, "--half_neigh", 0, "-n", 100], "dp"),
    ("B' lj -s 80 half DP", ["-s", 80, "--half_neigh", 1, "-n", 100], "dp"),
    ("C  eam -s 64 full DP", ["-i", "in.eam.miniMD", "-s", 64, "--half_neigh", 0, "-n", 100], "dp"),
    ("C' eam -s 64 half DP", ["-i", "in.eam.miniMD", "-s", 64, "--half_neigh", 1, "-n", 100], "dp"),
    ("E  lj -s 160 half SP", ["-s", 160, "--half_neigh", 1, "-n", 100], "sp"),
    ("E' lj -s 160 full SP", ["-s", 160, "--half_neigh", 0, "-n", 100], "sp"),
]
only = sys.argv[1:] 
for name, args, prec in CONFIGS:
    if only and not any(name.startswith(o) for o in only):
        continue
    t0 = time.time()
    s = minimd_amd.Sim(args, precision=prec)
    for kv in filter(None, os.environ.get("MMD_SIM_OPTIONS", "").split(",")):      # A/B knobs, e.g. MMD_SIM_OPTIONS=core_pct=30
        k, v = kv.split("=")
        s.handle.set_option(k, int(v))
    s.initial()
    s.run_steps(20 if s.natoms() > 500000 else 200)      # (small systems: list capacities and the device-resident borders settle over the first re-neighborings)
    # the chip's clocks keep ramping for >100 ms after idling and a small configuration is over in a few ms: keep the GPU busy with
    # force-kernel launches that leave the state untouched first (the set-up bench.py uses), then time 100 steps
    t_w = time.time()
    while (time.time() - t_w) < float(os.environ.get("MMD_CLOCK_WARM_S", "0.4")):
        s.handle.profile_kernel(0, 50)
    sec = s.run_steps(100)
    tm = s.handle.timers()
    rows = s.rows()
    print("%-22s %8.1f Matom-steps/s  %.3f ms/step  force %.3f ms/launch  neigh %.1f ms  comm %.1f ms  | last row %s  (wall %.1fs)" % (
        name, s.natoms() * 100 / sec / 1e6, sec * 10, tm["force_kernel_ms"] / max(tm["force_launches"], 1), tm["neigh"] * 1e3, tm["comm"] * 1e3,
        "%d %.6e %.6e %.6e" % rows[-1] if rows else "-", time.time() - t0), flush=True)
    s.close()
