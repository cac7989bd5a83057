import os, sys
sys.path.insert(0, os.getcwd())
import minimd_amd
s = minimd_amd.Sim(["-i", "in.eam.miniMD", "-s", 64, "--half_neigh", 0, "-n", 100])
s.initial(); s.run_steps(45)
h = s.handle
for ab in (0, 1, 2, 3, 4, 0):
    h.set_option("ablate", ab)
    print("ablate=%d  EAM Force::compute %.4f ms" % (ab, h.profile_kernel(0, 10)), flush=True)
s.close()
