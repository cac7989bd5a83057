import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import minimd_amd
size = int(sys.argv[1])
s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100])
h = s.handle
h.init_rccl(h.unique_id(), 0, 1)
h.set_option("force_transport", 1); h.set_option("overlap", 1)
s.initial()
s.run_steps(20)
for k in range(6):
    t0 = time.time()
    try:
        s.run_steps(2)
    except Exception as e:
        print("step", 20 + 2 * k, "ERROR", e, "%.3f s" % (time.time() - t0), flush=True); break
    print("steps", 20 + 2 * k + 2, "%.4f s" % (time.time() - t0), "flag steps", h.counter("halo_flag_steps"), "in_x", h.counter("halo_in_x_steps"), flush=True)
s.close()
