import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
build = int(sys.argv[1])
for size in (32, 80):
    a = minimd_amd.Sim(["-s", size, "-n", 40, "--half_neigh", 0]); a.handle.set_option("build", build); a.initial(); a.run(); print("full", size, a.rows()[-1], flush=True); a.close()
s = minimd_amd.Sim(["-s", 80, "-n", 20, "--half_neigh", 1])
h = s.handle
h.set_option("build", build)
print("created", flush=True)
h.exchange(); h.borders(); h.sync()
print("borders ok", h.counts(), flush=True)
h.neighbor_build(); h.sync()
print("build ok", h.neighbor_info(), h.neighbor_tile_stats(), flush=True)
e = h.force_compute(1); h.sync()
print("force ok", e, flush=True)
h.reverse_communicate(); h.sync()
print("reverse ok", flush=True)
s.initial(); s.run()
print(s.rows(), flush=True)
