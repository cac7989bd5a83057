#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
size = int(sys.argv[1]) if len(sys.argv) > 1 else 80
ov = int(sys.argv[2]) if len(sys.argv) > 2 else 1
s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100])
h = s.handle
h.init_rccl(h.unique_id(), 0, 1)
h.set_option("force_transport", 1)
h.set_option("overlap", ov)
for kv in filter(None, os.environ.get("MMD_TRACE_OPTIONS", "").split(",")):
    k, v = kv.split("="); h.set_option(k, int(v))
s.initial()
s.run_steps(60)
s.close()
