import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import minimd_amd
for prec in ("sp", "dp"):
    for nsteps in (1, 2, 3, 19, 21, 41):
        res = {}
        for fuse in (2, 1, 0):
            s = minimd_amd.Sim(["-s", "8", "-n", str(nsteps), "--half_neigh", "0"], precision=prec)
            s.handle.set_option("fuse", fuse)
            s.initial(); s.run()
            d = s.handle.download()
            o = np.argsort(d["tag"])
            res[fuse] = (d["x"][:d["nlocal"]][o], d["v"][o], d["f"][o], s.rows())
            s.close()
        for a, b in ((2, 1), (1, 0)):
            print(prec, nsteps, a, b, "x", np.abs(res[a][0] - res[b][0]).max(), "v", np.abs(res[a][1] - res[b][1]).max(), "f", np.abs(res[a][2] - res[b][2]).max(), res[a][3][-1] == res[b][3][-1])
