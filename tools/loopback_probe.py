#!/usr/bin/env python3
"""What the multi-rank step costs a rank apart from the wire: ONE rank whose periodic self swaps are routed through the RCCL transport
(option force_transport: every halo, border and exchange message is packed, sent to itself with ncclSend/ncclRecv and unpacked — the code path
of a rank inside an 8-GPU run, xGMI transfer time excepted) against the one-rank fast path.
usage: tools/loopback_probe.py [size] ["opt=val,opt=val" ...]   (each option set is one loop-back run; default: the shipped defaults, overlap 0 and 1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
size = int(sys.argv[1]) if len(sys.argv) > 1 else 80
sets = sys.argv[2:] or ["overlap=1", "overlap=0"]
for mode in ["one rank"] + sets:
    s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100])
    h = s.handle
    if mode != "one rank":
        h.init_rccl(h.unique_id(), 0, 1)
        h.set_option("force_transport", 1)
        for kv in filter(None, mode.split(",")):
            k, v = kv.split("=")
            h.set_option(k, int(v))
    s.initial()
    s.run_steps(40)
    t_w = time.time()
    while time.time() - t_w < 0.4:
        h.profile_kernel(0, 50)
    best = 1e9
    for rep in range(3):
        sec = s.run_steps(100)
        best = min(best, sec)
    tm = h.timers()
    st = h.run_stats()
    print("-s %d  %-44s %8.1f Matom-steps/s  %.4f ms/step  comm %.2f ms neigh %.2f ms per 100 steps  host syncs %d  bytes sent %.1f MB  direct borders %d" % (
        size, mode if mode == "one rank" else "loop-back " + mode, s.natoms() * 100 / best / 1e6, best * 10, tm["comm"] * 1e3, tm["neigh"] * 1e3, st["host_syncs"], st["bytes_sent"] / 1e6,
        h.counter("borders_direct")), flush=True)
    s.close()
