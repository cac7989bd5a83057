"""`-m gpu` parity tests proper: the HIP path, called through the C-ABI (minimd_amd.api -> libmmd_hip_*.so),
against the CPU oracle on identical seeded inputs, against the committed golden vectors, and — at the
BASELINE.json sizes — through size-independent properties. Tolerances are stated per test.

DP tolerance rationale: the reference's own acceptance rule (ref/run_one_test:121-138) tolerates
|dT| <= 0.4/sqrt(N)*x + 1e-5 (x ~ 0.03 at step 100 in DP), i.e. ~7 printed digits for the first ~1400 steps.
Our force kernels use FMA contraction and a Newton reciprocal, so per-atom forces agree with the
uncontracted oracle to ~1e-13 relative, and thermo rows to the printed 7 digits for the first few hundred steps.
"""
import json
import os

import numpy as np
import pytest

from oracle_lib import Oracle, fmt7, ref_pass_rule

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REFRUNS = json.load(open(os.path.join(GOLD, "ref_runs.json")))
PUBLISHED = json.load(open(os.path.join(GOLD, "reference_output.json")))


def mm():
    import minimd_amd
    return minimd_amd


def handle_from_oracle(o, precision="dp", with_ghosts=True, neighbor_setup=True):
    """device handle holding exactly the oracle's current atoms (rank 0)"""
    m = mm()
    h = m.Handle(precision)
    box = o.box()
    h.set_box(box[0:3], [box[3], box[5], box[7]], [box[4], box[6], box[8]])
    h.set_mass(o.param("mass"))
    nl = o.nlocal()
    x = o.x() if with_ghosts else o.x()[:nl]
    t = o.type() if with_ghosts else o.type()[:nl]
    h.upload(x, o.v(), t, o.tag(), nlocal=nl)
    if neighbor_setup:
        h.neighbor_setup(o.nbins(), o.param("cutneigh"), int(o.param("halfneigh")), int(o.param("ghost_newton")), o.ntypes())
    return h


def rows_close(rows, ref, rtol):
    assert [r[0] for r in rows] == [int(r[0]) for r in ref]
    for a, b in zip(rows, ref):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= rtol * max(1.0, abs(b[k])), (a, b)


# ---------------------------------------------------------------------------------------------------
# kernel level: ForceLJ::compute on the oracle's own neighbor list
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size,ntypes", [(4, 4), (6, 4), (7, 1)])
def test_lj_force_full_matches_oracle(size, ntypes):
    o = Oracle(["-s", size, "-n", 20, "--half_neigh", 0, "--ntypes", ntypes])
    o.initial()
    o.run()                                   # thermalised positions, list rebuilt at step 20
    h = handle_from_oracle(o)
    h.force_lj_setup(*o.lj_tables())
    nb = o.neighbors()
    h.neighbor_upload(nb, o.numneigh())
    # an uploaded list is turned into the tile form (k_rows_to_tiles) and served by k_lj_full_tile; with tiles off the same rows run
    # on the general row kernel k_lj_full
    assert h.counter("tiles_ready") == 1 and h.counter("rows_uploaded") == 1 and h.neighbor_tile_stats()["tiles"] > 0
    for tiles in (1, 0):
        h.set_option("tiles", tiles)
        eng, vir = h.force_compute(1)
        f = h.download()["f"]
        fo = o.f()
        scale = np.abs(fo).max()
        # tolerance: 1e-12 of the largest force component (FMA + Newton reciprocal vs strict IEEE order)
        assert np.abs(f - fo).max() <= 1e-12 * scale
        assert abs(eng - o.eng_vdwl()) <= 1e-12 * abs(o.eng_vdwl())
        assert abs(vir - o.virial()) <= 1e-11 * max(1.0, abs(o.virial()))
    # what went in comes back out (rows as the reference has them, in its order)
    nb2, nn2 = h.neighbor_download()
    np.testing.assert_array_equal(nn2, o.numneigh())
    for i in range(o.nlocal()):
        assert list(nb2[i, :nn2[i]]) == list(nb[i, :nn2[i]])
    h.close(); o.close()


@pytest.mark.parametrize("args", [["-s", 4], ["-s", 7], ["-nx", 9, "-ny", 5, "-nz", 6], ["-s", 6, "-b", 9]])
def test_lj_force_full_tile_path_matches_oracle(args):
    """device-built list -> LDS tile kernel (the production LJ path) vs the oracle's forces on the same atoms;
    and the tile kernel vs the generic global-gather kernel on the same device list"""
    o = Oracle(args + ["-n", 20, "--half_neigh", 0])
    o.initial(); o.run()
    h = handle_from_oracle(o)
    h.force_lj_setup(*o.lj_tables())
    h.neighbor_build()
    fo = o.f()
    scale = np.abs(fo).max()
    out = {}
    for tiles in (1, 0):
        h.set_option("tiles", tiles)
        eng, vir = h.force_compute(1)
        f = h.download()["f"]
        assert np.abs(f - fo).max() <= 1e-12 * scale
        assert abs(eng - o.eng_vdwl()) <= 1e-12 * abs(o.eng_vdwl())
        assert abs(vir - o.virial()) <= 1e-11 * max(1.0, abs(o.virial()))
        out[tiles] = f
    assert np.abs(out[0] - out[1]).max() <= 1e-13 * scale
    h.close(); o.close()


def test_lj_force_nonuniform_type_tables():
    """per type-pair tables (the general path; miniMD itself always fills them uniformly)"""
    o = Oracle(["-s", 5, "-n", 20, "--half_neigh", 0, "--ntypes", 3])
    o.initial(); o.run()
    cut, s6, eps = o.lj_tables()
    rng = np.random.default_rng(7)
    sym = lambda a: (a + a.reshape(3, 3).T.ravel()) / 2
    eps2 = sym(eps * (1 + 0.3 * rng.random(9)))
    s62 = sym(s6 * (1 + 0.2 * rng.random(9)))
    cut2 = sym(cut * (1 - 0.2 * rng.random(9)))
    x, t = o.x(), o.type()
    fo, eo, vo = o.lj_force_full(x, t, o.nlocal(), o.neighbors(), o.numneigh(), cut2, s62, eps2, 1)
    h = handle_from_oracle(o)
    h.force_lj_setup(cut2, s62, eps2)
    h.neighbor_upload(o.neighbors(), o.numneigh())
    eng, vir = h.force_compute(1)
    f = h.download()["f"]
    assert np.abs(f - fo).max() <= 1e-12 * np.abs(fo).max()
    assert abs(eng - eo) <= 1e-12 * abs(eo) and abs(vir - vo) <= 1e-11 * max(1.0, abs(vo))
    h.close(); o.close()


@pytest.mark.parametrize("gn", [1, 0])
def test_lj_force_half_matches_oracle(gn):
    o = Oracle(["-s", 6, "-n", 20, "--half_neigh", 1, "-gn", gn])
    o.initial(); o.run()
    L = o.lib
    L.orc_force_compute(o.w, 1)               # forces incl. ghost contributions, before reverse comm
    h = handle_from_oracle(o)
    h.force_lj_setup(*o.lj_tables())
    h.neighbor_upload(o.neighbors(), o.numneigh())
    assert h.counter("tiles_ready") == 1 and h.neighbor_tile_stats()["tiles"] > 0
    for tiles in (1, 0):              # the reference's half list in tile form (k_lj_half_tile), then on the row kernel (k_lj_half)
        h.set_option("tiles", tiles)
        eng, vir = h.force_compute(1)
        f = h.download(halfneigh=True)["f"]
        fo = o.f(with_ghosts=True)
        # atomics reorder the per-atom sums: 1e-11 of the largest component
        assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
        assert abs(eng - o.eng_vdwl()) <= 1e-11 * abs(o.eng_vdwl())
        assert abs(vir - o.virial()) <= 1e-10 * max(1.0, abs(o.virial()))
    h.close(); o.close()


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_lj_force_half_tile_kernel_matches_oracle(prec):
    """device-built half list in tile form -> k_lj_half_tile (partner forces accumulated in LDS, one global atomic per
    candidate): forces on owned atoms, energy and virial against the oracle's half-list force on the same atoms"""
    o = Oracle(["-s", 6, "-n", 20, "--half_neigh", 1, "-gn", 0], precision=prec)
    o.initial(); o.run()
    o.lib.orc_force_compute(o.w, 1)
    h = handle_from_oracle(o, precision=prec)
    h.force_lj_setup(*o.lj_tables())
    h.neighbor_build()
    assert h.neighbor_tile_stats()["tiles"] > 0
    eng, vir = h.force_compute(1)
    nl = o.nlocal()
    f = h.download(halfneigh=True)["f"][:nl]
    fo = o.f()[:nl]
    tol = 1e-11 if prec == "dp" else 2e-5
    assert np.abs(f - fo).max() <= tol * np.abs(fo).max()
    assert abs(eng - o.eng_vdwl()) <= tol * abs(o.eng_vdwl())
    assert abs(vir - o.virial()) <= 10 * tol * max(1.0, abs(o.virial()))
    h.close(); o.close()


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_lj_force_half_ghost_newton_tile_kernel_matches_oracle(prec):
    """same with ghost newton (the device stores every pair once on its lower atom in (z,y,x) order, the reference by its half-stencil bin order):
    whole setup through the driver, forces compared on owned atoms after the reverse halo, matched by tag"""
    args = ["-s", "6", "-n", "20", "--half_neigh", "1", "-gn", "1"]
    o = Oracle(args, precision=prec)
    o.initial(); o.run()                   # (the initial lattice has zero forces: compare a thermalised state)
    s = mm().Sim(args, precision=prec)
    s.initial(); s.run()
    d = s.handle.download(halfneigh=True)
    nl = o.nlocal()
    assert d["nlocal"] == nl and s.handle.neighbor_tile_stats()["tiles"] > 0
    fo = o.f()[:nl][np.argsort(o.tag()[:nl])]
    f = d["f"][:nl][np.argsort(d["tag"])]
    tol = 1e-9 if prec == "dp" else 1e-3   # two independent 20-step trajectories (summation order differs)
    assert np.abs(f - fo).max() <= tol * np.abs(fo).max()
    s.close(); o.close()


# ---------------------------------------------------------------------------------------------------
# Neighbor::build — rows equal the oracle's as SETS, counts exactly (index work: bit-exact)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("args", [["-s", 4], ["-s", 6], ["-nx", 7, "-ny", 5, "-nz", 6], ["-s", 6, "-b", 9], ["-s", 8, "-b", 3]])
def test_neighbor_build_full_equals_oracle(args):
    o = Oracle(args + ["-n", 20, "--half_neigh", 0])
    o.initial(); o.run()
    h = handle_from_oracle(o)
    h.neighbor_build()
    nb, nn = h.neighbor_download()
    np.testing.assert_array_equal(nn, o.numneigh())
    assert h.neighbor_info()["total"] == int(o.numneigh().sum())
    rows = o.neighbor_rows()
    for i in range(len(nn)):
        np.testing.assert_array_equal(np.sort(nb[i, :nn[i]]), np.sort(rows[i]))
    h.close(); o.close()


def test_neighbor_build_half_gn0_equals_oracle():
    o = Oracle(["-s", 6, "-n", 20, "--half_neigh", 1, "-gn", 0])
    o.initial(); o.run()
    h = handle_from_oracle(o)
    h.neighbor_build()
    nb, nn = h.neighbor_download()
    np.testing.assert_array_equal(nn, o.numneigh())
    rows = o.neighbor_rows()
    for i in range(len(nn)):
        np.testing.assert_array_equal(np.sort(nb[i, :nn[i]]), np.sort(rows[i]))
    h.close(); o.close()


def test_neighbor_build_half_gn1_download_equals_oracle():
    """half lists WITH ghost newton: the device list partitions the pairs by position order, the reference by its half stencil of
    bins and the same-bin rules (ref/neighbor.cpp:143-182, :424-441). What crosses the C-ABI is the reference's list:
    mmd_neighbor_download rebuilds the rows with the reference's rule (k_build<3>) — counts and rows (as sets) equal the oracle's."""
    o = Oracle(["-s", 6, "-n", 20, "--half_neigh", 1, "-gn", 1])
    o.initial(); o.run()
    h = handle_from_oracle(o)
    h.neighbor_build()
    assert h.neighbor_info()["total"] == int(o.numneigh().sum())
    nb, nn = h.neighbor_download()
    np.testing.assert_array_equal(nn, o.numneigh())
    rows = o.neighbor_rows()
    for i in range(len(nn)):
        np.testing.assert_array_equal(np.sort(nb[i, :nn[i]]), np.sort(rows[i]))
    h.close(); o.close()


@pytest.mark.parametrize("state", ["s0pre", "s1pre"])
def test_neighbor_half_gn1_download_equals_the_reference_arrays(state):
    """the same against the unmodified reference's OWN arrays (tests/golden/arrays_lj_s4_half_gn1.npz, dumped by oracle/ref_dump.cpp
    from the reference's Neighbor object): its owned + ghost atoms uploaded as they are, our build, our download — the reference's
    rows as sets. s0pre is the perfect lattice (whole planes share z and y: the equality branches of ref/neighbor.cpp:155-157),
    s1pre the system after the run."""
    d = np.load(os.path.join(GOLD, "arrays_lj_s4_half_gn1.npz"))
    nl, ng = int(d[state + ".nlocal"][0]), int(d[state + ".nghost"][0])
    x = d[state + ".x"].reshape(-1, 3)[: nl + ng]
    m = mm()
    h = m.Handle()
    h.set_box(d["prd"])
    h.set_mass(1.0)
    h.upload(x, np.zeros((nl, 3)), d[state + ".type"][: nl + ng], None, nlocal=nl)
    h.neighbor_setup(d["nbin"], float(d["cutneigh"][0]), 1, 1, int(d["ntypes"][0]))
    h.neighbor_build()
    nb, nn = h.neighbor_download()
    ref_nn = d[state + ".numneigh"]
    ref_rows = np.split(d[state + ".neighbors"], np.cumsum(ref_nn)[:-1])          # (the fixture stores the rows back to back)
    np.testing.assert_array_equal(nn, ref_nn)
    for i in range(nl):
        np.testing.assert_array_equal(np.sort(nb[i, :nn[i]]), np.sort(ref_rows[i]))
    h.close()


def test_neighbor_overflow_regrow():
    """rows longer than maxneighs trigger the reference's grow-and-retry protocol (ref/neighbor.cpp:186-208)"""
    o = Oracle(["-s", 5, "-n", 1, "--half_neigh", 0])
    o.initial()
    h = handle_from_oracle(o)
    h.set_option("maxneighs", 16)
    h.neighbor_build()
    info = h.neighbor_info()
    assert info["max_row"] == int(o.numneigh().max()) and info["maxneighs"] > info["max_row"]
    nb, nn = h.neighbor_download()
    np.testing.assert_array_equal(nn, o.numneigh())
    h.close(); o.close()


# ---------------------------------------------------------------------------------------------------
# Integrate / Thermo kernels — compiled without contraction: bit-exact
# ---------------------------------------------------------------------------------------------------
def test_integrate_bit_exact():
    o = Oracle(["-s", 5, "-n", 3, "--half_neigh", 0])
    o.initial()
    h = handle_from_oracle(o, neighbor_setup=False)
    h.upload_f(o.f())
    dt, dtf = o.param("dt"), o.param("dtforce")
    h.integrate_setup(dt, dtf, 20, 20)
    h.initial_integrate()
    o.lib.orc_initial_integrate(o.w)
    d = h.download()
    nl = o.nlocal()
    np.testing.assert_array_equal(d["x"][:nl], o.x()[:nl])
    np.testing.assert_array_equal(d["v"], o.v())
    h.final_integrate()
    o.lib.orc_final_integrate(o.w)
    np.testing.assert_array_equal(h.download()["v"], o.v())
    # kinetic energy sum: same terms, different summation order
    t_dev = h.temperature_sum() * o.param("t_scale")
    assert abs(t_dev - o.thermo()[0]) <= 1e-13 * o.thermo()[0]
    h.close(); o.close()


# ---------------------------------------------------------------------------------------------------
# Comm: exchange(pbc) / borders / communicate on one rank — ghost atoms bit-exact, lists identical
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("args", [["-s", 4], ["-s", 6], ["-nx", 7, "-ny", 5, "-nz", 6]])
def test_borders_and_communicate_equal_oracle(args):
    o = Oracle(args + ["-n", 30, "--half_neigh", 0])
    o.initial(); o.run()                      # positions drifted; last borders at step 20
    # move the oracle one more half step so that x != x at the last borders, then redo exchange+borders
    o.lib.orc_initial_integrate(o.w)
    h = handle_from_oracle(o, with_ghosts=False)
    h.comm_setup(o.param("cutneigh"), 0, 1)
    h.exchange(); h.borders()
    o.lib.orc_exchange(o.w); o.lib.orc_borders(o.w)
    d = h.download()
    assert (d["nlocal"], d["nghost"]) == (o.nlocal(), o.nghost())
    np.testing.assert_array_equal(d["x"], o.x())
    np.testing.assert_array_equal(d["type"], o.type())
    for s in range(o.nswap()):
        si, so = h.swap_info(s), o.swap_info(0, s)
        assert (si["sendnum"], si["recvnum"], si["firstrecv"]) == (o.sendnum()[s], o.recvnum()[s], o.firstrecv()[s])
        assert (si["slablo"], si["slabhi"], si["pbc_any"], si["pbc"]) == (so["slablo"], so["slabhi"], so["pbc_any"], so["pbc"])
        np.testing.assert_array_equal(h.sendlist(s), o.sendlist(0, s))
    # forward halo after another drift
    h.upload_f(np.zeros((o.nlocal(), 3)))
    o.lib.orc_initial_integrate(o.w)
    # mirror the same move on the device: upload the oracle's new owned positions, keep ghosts stale
    xo = o.x()
    h2 = d["x"].copy(); h2[: o.nlocal()] = xo[: o.nlocal()]
    h.upload(h2, o.v(), o.type(), o.tag(), nlocal=o.nlocal())      # ghosts still the old ones
    h.communicate()
    o.lib.orc_communicate(o.w)
    np.testing.assert_array_equal(h.download()["x"], o.x())
    h.close(); o.close()


def test_sort_is_a_permutation_and_keeps_physics():
    o = Oracle(["-s", 6, "-n", 20, "--half_neigh", 0])
    o.initial(); o.run()
    h = handle_from_oracle(o, with_ghosts=False)
    h.comm_setup(o.param("cutneigh"), 0, 1)
    h.sort()
    d = h.download()
    order = np.argsort(d["tag"])
    oo = np.argsort(o.tag())
    np.testing.assert_array_equal(d["x"][:o.nlocal()][order], o.x()[:o.nlocal()][oo])
    np.testing.assert_array_equal(d["v"][order], o.v()[oo])
    np.testing.assert_array_equal(d["type"][:o.nlocal()][order], o.type()[:o.nlocal()][oo])
    h.close(); o.close()


# ---------------------------------------------------------------------------------------------------
# whole runs: thermo rows against the golden rows of the reference
# ---------------------------------------------------------------------------------------------------
def sim_rows(args, precision="dp"):
    s = mm().Sim(args, precision=precision)
    s.initial(); s.run()
    rows = s.rows()
    s.close()
    return rows


@pytest.mark.parametrize("name", ["lj_s10_full_n1000", "lj_s16_full_n300", "lj_nx12_ny8_nz10_full_n200", "lj_s32_full_n100", "lj_s32_full_n20"])
def test_run_lj_full_rows_match_reference(name):
    ent = REFRUNS[name]
    rows = sim_rows([a for a in ent["args"]])
    # printed digits: 7 significant; allow 1.5 units of the 6th digit up to step 1000 (summation order + FMA)
    rows_close(rows, ent["rows"], 1.5e-5)
    early = [(a, b) for a, b in zip(rows, ent["rows"]) if a[0] <= 300]
    for a, b in early:
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 2e-6 * max(1.0, abs(b[k])), (a, b)
    ok, frac = ref_pass_rule(ent["rows"], rows, ent["natoms"], 8)
    assert ok, frac


def test_run_matches_published_4k_log():
    """the reference's own known-good log tests/reference_output/4k.lj, first 1000 steps"""
    ref = [r for r in PUBLISHED["4k.lj"]["rows"] if r[0] <= 1000]
    rows = sim_rows(["-s", 10, "-n", 1000, "--half_neigh", 0])
    rows_close(rows, ref, 1.5e-5)
    assert ref_pass_rule(ref, rows, 4000, 8)[0]


def test_step0_row_is_digit_exact():
    rows = sim_rows(["-s", 20, "-n", 1, "--half_neigh", 0])
    ref = PUBLISHED["32k.lj"]["rows"][0]
    assert (fmt7(rows[0][1]), fmt7(rows[0][2]), fmt7(rows[0][3])) == (fmt7(ref[1]), fmt7(ref[2]), fmt7(ref[3]))


def test_structural_counts_s32():
    """nghost / total neighbors of the last rebuild equal the reference's YAML report (6 printed digits)"""
    ent = REFRUNS["lj_s32_full_n20"]
    s = mm().Sim(ent["args"])
    s.initial(); s.run()
    nl, ng, _ = s.handle.counts()
    assert nl == ent["natoms"] and ng == int(ent["nghost"])
    tot = s.handle.neighbor_info()["total"]
    assert abs(tot - ent["neigh_total"]) <= 5e-6 * ent["neigh_total"]
    s.close()


def test_device_resident_borders_equal_the_swap_by_swap_path():
    """one rank, 16 384 atoms: Comm::borders three ways on the same atoms — swap by swap with host-read counts (first call: nothing
    to size the device path by), device-resident (second call: count/scatter pairs, one read-back), and device-resident with
    arrays sized too small (overflow flag -> swap-by-swap fallback): identical counts, send lists and ghost atoms"""
    o = Oracle(["-s", 16, "-n", 1, "--half_neigh", 0])
    o.initial()
    h = handle_from_oracle(o, with_ghosts=False)
    h.comm_setup(o.param("cutneigh"), 0, 1)
    h.exchange()
    res = []
    for mode in ("swap-by-swap", "device", "device-overflow"):
        if mode == "device-overflow":
            h.set_option("borders_est", 40)
        h.borders()
        d = h.download()
        nsw = h.comm_info()["nswap"]
        res.append((d["nghost"], d["x"].copy(), d["type"].copy(), [(h.swap_info(s_)["sendnum"], h.swap_info(s_)["recvnum"], h.swap_info(s_)["firstrecv"]) for s_ in range(nsw)], [h.sendlist(s_).copy() for s_ in range(nsw)]))
    h.set_option("borders_est", 150)
    assert res[0][0] == o.nghost() > 0
    np.testing.assert_array_equal(res[0][1], o.x())
    for other in res[1:]:
        assert other[0] == res[0][0] and other[3] == res[0][3]
        np.testing.assert_array_equal(other[1], res[0][1])
        np.testing.assert_array_equal(other[2], res[0][2])
        for a, b in zip(other[4], res[0][4]):
            np.testing.assert_array_equal(a, b)
    h.close(); o.close()
