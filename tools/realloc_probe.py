import os, sys
sys.path.insert(0, os.getcwd())
import minimd_amd
s = minimd_amd.Sim(["-s", "80", "--half_neigh", "0", "-n", "100"], quiet=True)
s.initial()
sys.stderr.write("=== setup done\n"); sys.stderr.flush()
for k in range(5):
    s.run_steps(20)
    sys.stderr.write("=== slice %d done\n" % k); sys.stderr.flush()
s.close()
