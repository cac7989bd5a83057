#!/usr/bin/env python3
"""Large-scale equivalence of two builds of the neighbor kernel (e.g. the product library with the MFMA pre-test against variants/nomfma): the SAME thermalised positions
(first call writes them to /tmp/build_equiv_x.npy, later calls load them) are uploaded as owned atoms without ghosts, Neighbor::build runs, the rows come back through
mmd_neighbor_download and are reduced to (number of pairs, order-independent checksum per row summed) — two libraries agree iff every row holds the same set.
usage: [MMD_LIB_DIR=variants/<name>] python tools/probes/build_equiv_probe.py [size]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import minimd_amd
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "dp"
fn = "/tmp/build_equiv_x_%d_%s.npy" % (size, prec)
if not os.path.exists(fn):
    sim = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", 100], precision=prec)
    sim.initial(); sim.run_steps(100)
    h = sim.handle
    nl = h.counts()[0]
    x = h.download()["x"][:nl]
    box = h.get_box()
    np.save(fn, x); np.save(fn + ".box.npy", np.array(box[0], dtype=np.float64))
    sim.close()
x = np.load(fn); prd = np.load(fn + ".box.npy")
x = np.ascontiguousarray(np.mod(x, prd.astype(x.dtype)))          # (between two re-neighborings atoms sit a little outside the box)
x[x >= prd.astype(x.dtype)] = 0
n = len(x)
h = minimd_amd.Handle(prec)
h.set_box(list(prd))
h.set_mass(1.0)
h.upload(x, np.zeros_like(x), np.zeros(n, np.int32), np.arange(1, n + 1, dtype=np.int32), nlocal=n)
nb = [max(1, int(p / 1.68)) for p in prd]
h.neighbor_setup(nb, 2.8, 0, 1, 1)
h.neighbor_build()
tiles = h.neighbor_tile_stats()["tiles"]
rows, nn = h.neighbor_download()
m = np.arange(rows.shape[1])[None, :] < nn[:, None]
r = rows.astype(np.uint64)
hsh = (r * np.uint64(0x9E3779B97F4A7C15)) ^ (r >> np.uint64(7)) * np.uint64(0xC2B2AE3D27D4EB4F)
rowsum = np.where(m, hsh, np.uint64(0)).sum(axis=1, dtype=np.uint64)
tot = int((rowsum * (np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))
print("lib %-18s size %d %s: %d atoms, %d tiles, %d pairs, checksum %016x" % (os.environ.get("MMD_LIB_DIR", "product"), size, prec, n, tiles, int(nn.sum()), tot))
