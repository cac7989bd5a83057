import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd
size, build, gn = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = minimd_amd.Sim(["-s", size, "-n", 20, "--half_neigh", 1, "-gn", gn])
h = s.handle
h.set_option("build", build)
print("created", flush=True)
h.exchange(); h.borders(); h.sync()
print("borders ok", h.counts(), flush=True)
h.neighbor_build(); h.sync()
print("build ok", h.neighbor_info(), h.neighbor_tile_stats(), flush=True)
e = h.force_compute(1); h.sync()
print("force ok", e, flush=True)
s.initial(); s.run()
print(s.rows(), flush=True)
