"""Deterministic LAMMPS data files for the `-f / --data_file` path (read_lammps_data, ref/setup.cpp:55-301).

The files are INPUT fixtures made by this module (not reference files): a jittered FCC crystal in a non-cubic
box with shuffled atom ids, random velocities (zero total momentum), a Masses section, comments and blank
lines.  Only +,-,* on Python floats and an integer LCG are used, and numbers are printed with %.17g, so the
bytes are identical on every machine; tests/golden/datafile_runs.json records their sha256 next to the thermo
rows the unmodified reference printed for them (tests/golden/make_datafile_golden.py).
"""
import hashlib

import numpy as np


class _Lcg:
    """Park-Miller style 31-bit generator on Python ints (platform independent)."""

    def __init__(self, seed):
        self.s = seed % 2147483647 or 1

    def u(self):            # in (0,1)
        self.s = (self.s * 48271) % 2147483647
        return self.s * (1.0 / 2147483647)


CASES = {
    # name: (cells, lattice constant, position jitter / a, velocity scale, mass or None, seed)
    "lj_5x6x7": dict(cells=(5, 6, 7), a=1.6795962, jitter=0.04, vscale=1.9, mass=1.5, seed=20240917),
    "eam_4x5x5": dict(cells=(4, 5, 5), a=3.615, jitter=0.02, vscale=6.0, mass=None, seed=777),
    # one cell compressed to 0.8 a along x: the box is thinner than half the neighbor cutoff (1.34 < 1.4), Comm::setup asks for THREE
    # ghost layers in x (need = 3, ref/comm.cpp:150-152) and an atom sees up to three images of itself
    "lj_thin_1x6x7": dict(cells=(1, 6, 7), a=1.6795962, jitter=0.02, vscale=1.2, mass=1.0, seed=4711, scale=(0.8, 1.0, 1.0)),
}


def make_arrays(cells, a, jitter, vscale, seed, scale=None, **_):
    """positions/velocities indexed by (file id - 1), box lengths"""
    nx, ny, nz = cells
    rng = _Lcg(seed)
    basis = ((0.0, 0.0, 0.0), (0.5, 0.5, 0.0), (0.5, 0.0, 0.5), (0.0, 0.5, 0.5))
    prd = (nx * a, ny * a, nz * a)
    if scale is not None:                                # anisotropic cell (cases without `scale` keep their bytes)
        return _make_arrays_scaled(cells, a, jitter, vscale, seed, scale)
    pos = []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                for b in basis:
                    p = []
                    for d, (c, off) in enumerate(zip((i, j, k), b)):
                        q = (c + off) * a + (rng.u() - 0.5) * 2.0 * jitter * a
                        if q < 0.0:
                            q += prd[d]
                        if q >= prd[d]:
                            q -= prd[d]
                        p.append(q)
                    pos.append(p)
    n = len(pos)
    vel = [[(rng.u() - 0.5) * 2.0 * vscale for _ in range(3)] for _ in range(n)]
    for d in range(3):                                   # zero total momentum (summation order fixed)
        m = 0.0
        for i in range(n):
            m += vel[i][d]
        m = m * (1.0 / n)
        for i in range(n):
            vel[i][d] -= m
    # shuffle: ids[k] = file id (1-based) of lattice atom k
    ids = list(range(1, n + 1))
    for k in range(n - 1, 0, -1):
        r = int(rng.u() * (k + 1))
        ids[k], ids[r] = ids[r], ids[k]
    x = np.zeros((n, 3))
    v = np.zeros((n, 3))
    for k in range(n):
        x[ids[k] - 1] = pos[k]
        v[ids[k] - 1] = vel[k]
    return x, v, np.array(prd)


def _make_arrays_scaled(cells, a, jitter, vscale, seed, scale):
    nx, ny, nz = cells
    rng = _Lcg(seed)
    basis = ((0.0, 0.0, 0.0), (0.5, 0.5, 0.0), (0.5, 0.0, 0.5), (0.0, 0.5, 0.5))
    ad = (a * scale[0], a * scale[1], a * scale[2])
    prd = (nx * ad[0], ny * ad[1], nz * ad[2])
    pos = []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                for b in basis:
                    p = []
                    for d, (c, off) in enumerate(zip((i, j, k), b)):
                        q = (c + off) * ad[d] + (rng.u() - 0.5) * 2.0 * jitter * ad[d]
                        if q < 0.0:
                            q += prd[d]
                        if q >= prd[d]:
                            q -= prd[d]
                        p.append(q)
                    pos.append(p)
    n = len(pos)
    vel = [[(rng.u() - 0.5) * 2.0 * vscale for _ in range(3)] for _ in range(n)]
    for d in range(3):
        m = 0.0
        for i in range(n):
            m += vel[i][d]
        m = m * (1.0 / n)
        for i in range(n):
            vel[i][d] -= m
    ids = list(range(1, n + 1))
    for k in range(n - 1, 0, -1):
        r = int(rng.u() * (k + 1))
        ids[k], ids[r] = ids[r], ids[k]
    x = np.zeros((n, 3))
    v = np.zeros((n, 3))
    for k in range(n):
        x[ids[k] - 1] = pos[k]
        v[ids[k] - 1] = vel[k]
    return x, v, np.array(prd)


def write_case(name, path):
    """write the data file of CASES[name]; returns (x, v, prd, mass, sha256)"""
    c = CASES[name]
    x, v, prd = make_arrays(**c)
    n = len(x)
    rng = _Lcg(c["seed"] + 1)
    order = list(range(n))                               # line order differs from id order
    for k in range(n - 1, 0, -1):
        r = int(rng.u() * (k + 1))
        order[k], order[r] = order[r], order[k]
    out = ["LAMMPS data file made by tests/datafile_fixture.py: %s\n" % name, "\n",
           "%d atoms   # jittered FCC, shuffled ids\n" % n, "1 atom types\n", "\n",
           "# box (must start at 0, like the reference assumes)\n",
           "0.0 %.17g xlo xhi\n" % prd[0], "0.0 %.17g ylo yhi\n" % prd[1], "0.0 %.17g zlo zhi\n" % prd[2], "\n"]
    if c["mass"] is not None:
        out += ["Masses\n", "\n", "1 %.17g\n" % c["mass"], "\n"]
    out += ["Atoms\n", "\n"]
    out += ["%d 1 %.17g %.17g %.17g\n" % (i + 1, x[i, 0], x[i, 1], x[i, 2]) for i in order]
    out += ["\n", "Velocities\n", "\n"]
    out += ["%d %.17g %.17g %.17g\n" % (i + 1, v[i, 0], v[i, 1], v[i, 2]) for i in reversed(order)]
    text = "".join(out)
    with open(path, "w") as f:
        f.write(text)
    return x, v, prd, c["mass"], hashlib.sha256(text.encode()).hexdigest()
