import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import minimd_amd
size = int(sys.argv[1]); recv = int(sys.argv[2]); chunks = [int(c) for c in sys.argv[3].split(",")]
s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100])
h = s.handle
h.init_rccl(h.unique_id(), 0, 1)
h.set_option("force_transport", 1); h.set_option("overlap", 0); h.set_option("halo_recv", recv)
s.initial()
done = 0
for chunk in chunks:
    s.run_steps(chunk); done += chunk
    d = h.download()
    x = d["x"]; nl = d["nlocal"]
    print("step", done, "in_x", h.counter("halo_in_x_steps"), "direct", h.counter("borders_direct"), h.counts(), "recv", h.counter("dh_total_recv"), "R", h.counter("dh_R"),
          "src", h.counter("cand_src_halo"), "tiles", h.counter("tiles_ready"), "x own [%.3f %.3f] nan %d  ghosts [%.3f %.3f] nan %d  |v|max %.3f" % (
              x[:nl].min(), x[:nl].max(), np.isnan(x[:nl]).sum(), x[nl:].min() if len(x) > nl else 0, x[nl:].max() if len(x) > nl else 0, np.isnan(x[nl:]).sum(), np.abs(d["v"]).max()), flush=True)
s.close()
