import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import minimd_amd
size = int(sys.argv[1]); chunks = [int(c) for c in sys.argv[2].split(",")]
res = {}
for recv in (1, 3):
    s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100])
    h = s.handle
    h.init_rccl(h.unique_id(), 0, 1)
    h.set_option("force_transport", 1); h.set_option("overlap", 0); h.set_option("halo_recv", recv)
    s.initial()
    out = []
    for c in chunks:
        s.run_steps(c)
        d = h.download()
        out.append((d["x"].copy(), d["v"].copy(), d["f"].copy(), d["nlocal"], [h.swap_info(q)["recvnum"] for q in range(6)]))
    res[recv] = out
    s.close()
for k, c in enumerate(chunks):
    xa, va, fa, nl, rn = res[1][k]; xb, vb, fb, _, _ = res[3][k]
    bad_o = np.where(~np.all(xa[:nl] == xb[:nl], axis=1))[0]
    bad_g = np.where(~np.all(xa[nl:] == xb[nl:], axis=1))[0]
    bad_f = np.where(~np.all(fa[:nl] == fb[:nl], axis=1))[0] if fa.ndim == 2 else []
    print("after chunk", k, "steps", sum(chunks[:k + 1]), "owned x differ", len(bad_o), "ghost x differ", len(bad_g), bad_g[:20], "f differ", len(bad_f), bad_f[:10], "recvnum", rn, "cum", np.cumsum(rn))
    if len(bad_g):
        print("  ghost", bad_g[0], "a", xa[nl + bad_g[0]], "b", xb[nl + bad_g[0]])
