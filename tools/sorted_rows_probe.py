#!/usr/bin/env python3
"""Would neighbor rows ORDERED BY DISTANCE speed the tile force kernels up? In a wavefront the 64 lanes work on the k-th entries of their rows at the same time; in rows
sorted by r the k-th neighbours of the 64 atoms sit at nearly the same distance, so (EAM) the spline knots they gather from LDS are the same few records (broadcast
instead of bank conflicts) and (LJ) the cutoff branch is uniform. Measured without writing the sort on the GPU: the device-built rows are downloaded, re-ordered on the
host and uploaded again (mmd_neighbor_upload keeps the order of the uploaded rows in the tile form); Force::compute alone is timed (mmd_profile_kernel) on
 (a) the device-built lists, (b) the same rows uploaded as they are (control: the upload path itself), (c) uploaded sorted by distance, (d) uploaded in random order.
usage: tools/sorted_rows_probe.py <deck> <size> [half]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minimd_amd
deck, size = sys.argv[1], int(sys.argv[2])
half = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = minimd_amd.Sim(["-i", deck, "-s", size, "--half_neigh", half, "-n", 100] + (["-gn", 0] if "eam" in deck else []))
s.initial(); s.run_steps(100)
h = s.handle
h.set_option("core_pct", 0)
def t(label):
    h.profile_kernel(0, 10)
    ms = min(h.profile_kernel(0, 20) for _ in range(3))
    print("%-46s Force::compute %.4f ms" % (label, ms), flush=True)
    return ms
a = t("device-built lists")
nb, nn = h.neighbor_download()
d = h.download()
x = d["x"]
nl = d["nlocal"]
K = nb.shape[1]
valid = np.arange(K)[None, :] < nn[:, None]
h.neighbor_upload(nb, nn)
b = t("uploaded as they are")
r2 = np.full(nb.shape, np.inf)
for c0 in range(0, nl, 1 << 17):
    c1 = min(c0 + (1 << 17), nl)
    j = np.where(valid[c0:c1], nb[c0:c1], 0)
    dx = x[j] - x[c0:c1, None, :]
    r2[c0:c1] = np.where(valid[c0:c1], (dx * dx).sum(-1), np.inf)
order = np.argsort(r2, axis=1, kind="stable")
nb_sorted = np.take_along_axis(nb, order, axis=1)
h.neighbor_upload(nb_sorted, nn)
c = t("uploaded sorted by distance")
rng = np.random.default_rng(1)
key = np.where(valid, rng.random(nb.shape), np.inf)
nb_rand = np.take_along_axis(nb, np.argsort(key, axis=1), axis=1)
h.neighbor_upload(nb_rand, nn)
e = t("uploaded in random order")
print("sorted / as-they-are = %.3f   random / as-they-are = %.3f   upload path / device-built = %.3f" % (c / b, e / b, b / a))
s.close()
