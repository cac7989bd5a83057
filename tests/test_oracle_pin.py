"""Pin the CPU oracle (oracle/mmd_oracle.c) against the reference BEFORE trusting it as the checker:
 (a) the reference's own published logs tests/reference_output/*.{lj,eam}  (tests/golden/reference_output.json)
 (b) thermo rows printed by the unmodified reference built in the build container (tests/golden/ref_runs.json)
 (c) per-atom arrays dumped from the reference objects (tests/golden/arrays_*.npz) — bit-exact.
All CPU-only."""
import glob
import json
import os

import numpy as np
import pytest

from oracle_lib import Oracle, fmt7, ref_pass_rule

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
PUBLISHED = json.load(open(os.path.join(GOLD, "reference_output.json")))
REFRUNS = json.load(open(os.path.join(GOLD, "ref_runs.json")))


def rows_as_text(rows):
    return [(int(r[0]), fmt7(r[1]), fmt7(r[2]), fmt7(r[3])) for r in rows]


def run_oracle(args, nprocs=1, precision="dp"):
    o = Oracle(args, nprocs=nprocs, precision=precision)
    o.initial()
    o.run()
    rows = o.rows()
    o.close()
    return rows


# ---- (a) published logs ------------------------------------------------------------------------

@pytest.mark.parametrize("half", [1, 0])
def test_published_4k_lj_digit_for_digit(half):
    ref = [r for r in PUBLISHED["4k.lj"]["rows"] if r[0] <= 1000]
    rows = run_oracle(["-s", 10, "-n", 1000, "--half_neigh", half])
    assert rows_as_text(rows) == rows_as_text(ref)


def test_published_4k_eam_digit_for_digit():
    ref = [r for r in PUBLISHED["4k.eam"]["rows"] if r[0] <= 500]
    rows = run_oracle(["-i", "in.eam.miniMD", "-s", 10, "-n", 500, "--half_neigh", 1])
    assert rows_as_text(rows) == rows_as_text(ref)


def test_published_16k_lj_on_8_virtual_ranks():
    """the published logs were produced on 16 MPI ranks; 8 virtual ranks (2x2x2) must agree to the
    printed digits up to summation order (<= 2 units in the 7th digit) for the first rows."""
    ref = [r for r in PUBLISHED["16k.lj"]["rows"] if r[0] <= 300]
    rows = run_oracle(["-s", 16, "-n", 300, "--half_neigh", 1], nprocs=8)
    assert [r[0] for r in rows] == [r[0] for r in ref]
    for a, b in zip(rows, ref):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 2.5e-6 * max(1.0, abs(b[k])), (a, b)
    ok, frac = ref_pass_rule(ref, rows, PUBLISHED["16k.lj"]["natoms"], 8)
    assert ok, frac


def test_published_step0_rows_all_sizes():
    """step-0 row of every published LJ log depends only on the lattice + velocity generator."""
    for name in ("32k.lj",):
        ent = PUBLISHED[name]
        o = Oracle(["-s", ent["size"][0], "-n", 0 + 1, "--half_neigh", 0])
        o.initial()
        assert rows_as_text(o.rows()[:1]) == rows_as_text(ent["rows"][:1])
        o.close()


# ---- (b) rows printed by the reference built here ----------------------------------------------

SMALL_REF_CASES = ["lj_s10_full_n1000", "lj_s10_half_gn1_n1000", "lj_s10_half_gn0_n1000", "lj_s16_full_n300",
                   "lj_nx12_ny8_nz10_full_n200", "eam_s10_full_n1000", "eam_s10_half_n300", "lj_s10_full_n1000_sp",
                   # boxes thinner than the neighbor cutoff: two ghost layers per dimension (need = 2, ref/comm.cpp:150-152)
                   "lj_s1_full_n60", "lj_1x3x2_half_n60", "lj_1x3x2_full_n60", "eam_2x1x3_full_n60"]


@pytest.mark.parametrize("name", SMALL_REF_CASES)
def test_rows_equal_reference_binary(name):
    ent = REFRUNS[name]
    rows = run_oracle(ent["args"], precision=ent["precision"])
    assert rows_as_text(rows) == rows_as_text(ent["rows"])


@pytest.mark.parametrize("name", ["lj_s1_full_n60", "lj_1x3x2_half_n60", "lj_1x3x2_full_n60", "eam_2x1x3_full_n60"])
def test_two_ghost_layers_counts_equal_reference(name):
    """need = 2: ghost and neighbor counts of the reference's YAML report"""
    ent = REFRUNS[name]
    o = Oracle(ent["args"])
    o.initial(); o.run()
    assert o.nghost() == int(ent["nghost"]) and int(o.numneigh().sum()) == int(ent["neigh_total"])
    o.close()


@pytest.mark.parametrize("name", ["lj_s32_full_n20", "lj_s20_full_n200"])
def test_rows_equal_reference_binary_threaded_ref(name):
    """the reference ran with 8 OpenMP threads here: rows agree to the printed digits up to the
    summation order of the energy/virial reduction (<= 1 unit in the 7th digit)."""
    ent = REFRUNS[name]
    args = [a for a in ent["args"]]
    o = Oracle(args)
    o.initial()
    o.run()
    rows = o.rows()
    for a, b in zip(rows, ent["rows"]):
        assert a[0] == b[0]
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1.5e-6 * max(1.0, abs(b[k])), (a, b)
    # structural counts of the last neighbor build (reference YAML report)
    assert o.nghost() == int(ent["nghost"])
    # the YAML report prints 6 significant digits
    assert abs(int(o.numneigh().sum()) - ent["neigh_total"]) <= 5e-6 * ent["neigh_total"]
    assert o.nbins() == ent["nbin"]
    o.close()


# ---- (c) per-atom arrays, bit-exact --------------------------------------------------------------

ARRAY_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "arrays_*.npz")))


def args_from_fixture(d, deck):
    a = ["-i", deck, "-s", int(d["size"][0]), "-n", int(d["nsteps"][0]), "--half_neigh", int(d["halfneigh"][0]),
         "-gn", int(d["ghost_newton"][0]), "--ntypes", int(d["ntypes"][0])]
    nb = d["nbin"]
    default = int(np.float64(5.0 / 6.0) * int(d["size"][0])) or 1
    if int(nb[0]) != default:
        a += ["-b", int(nb[0])]
    return a


def check_state(o, d, tag, with_lists):
    nl, ng = int(d[tag + ".nlocal"][0]), int(d[tag + ".nghost"][0])
    assert (o.nlocal(), o.nghost()) == (nl, ng)
    np.testing.assert_array_equal(o.x().ravel(), d[tag + ".x"])
    np.testing.assert_array_equal(o.v().ravel(), d[tag + ".v"])
    np.testing.assert_array_equal(o.type(), d[tag + ".type"])
    nf = len(d[tag + ".f"]) // 3
    np.testing.assert_array_equal(o.f(with_ghosts=(nf > nl)).ravel(), d[tag + ".f"])
    assert o.eng_vdwl() == d[tag + ".eng_vdwl"][0]
    assert o.virial() == d[tag + ".virial"][0]
    if with_lists:
        np.testing.assert_array_equal(o.numneigh(), d[tag + ".numneigh"])
        assert o.maxneighs() == int(d[tag + ".maxneighs"][0])
        flat = np.concatenate(o.neighbor_rows()) if nl else np.zeros(0, np.int32)
        np.testing.assert_array_equal(flat, d[tag + ".neighbors"])
    if (tag + ".fp") in d.files:
        np.testing.assert_array_equal(o.eam_fp()[:nl], d[tag + ".fp"][:nl])


@pytest.mark.parametrize("case", ARRAY_CASES)
def test_arrays_bit_exact(case):
    d = np.load(os.path.join(GOLD, case + ".npz"))
    prec = "sp" if case.endswith("_sp") else "dp"
    deck = "in.eam.miniMD" if "eam" in case else "in.lj.miniMD"
    o = Oracle(args_from_fixture(d, deck), precision=prec)
    # setup: lattice, velocities, types, thermo scales
    assert o.natoms() == int(d["natoms"][0])
    np.testing.assert_array_equal(o.x()[: o.nlocal()].ravel(), d["created.x"])
    np.testing.assert_array_equal(o.v().ravel(), d["created.v"])
    np.testing.assert_array_equal(o.type()[: o.nlocal()], d["created.type"])
    for k in ("t_scale", "e_scale", "p_scale", "dof_boltz", "mvv2e", "mass", "dt"):
        assert o.param(k) == d[k][0], k
    # initial exchange/borders/build/force
    L = o.lib
    L.orc_exchange(o.w)
    L.orc_borders(o.w)
    np.testing.assert_array_equal(o.sendnum(), d["sendnum"])
    np.testing.assert_array_equal(o.recvnum(), d["recvnum"])
    np.testing.assert_array_equal(o.firstrecv(), d["firstrecv"])
    L.orc_neighbor_build(o.w)
    L.orc_force_compute(o.w, 1)
    check_state(o, d, "s0pre", True)
    if int(d["halfneigh"][0]) and int(d["ghost_newton"][0]):
        L.orc_reverse_communicate(o.w)
    check_state(o, d, "s0", False)
    assert o.thermo()[0] == d["s0.T"][0]
    # dynamics: ntimes steps incl. one re-neighboring, then the final force
    o.run()
    L.orc_force_compute(o.w, 1)
    check_state(o, d, "s1pre", True)
    if int(d["halfneigh"][0]) and int(d["ghost_newton"][0]):
        L.orc_reverse_communicate(o.w)
    check_state(o, d, "s1", False)
    assert o.thermo()[0] == d["s1.T"][0]
    o.close()


# ---- independent checks of the binned build -------------------------------------------------------

def test_binned_build_equals_brute_force():
    o = Oracle(["-s", 5, "-n", 20, "--half_neigh", 0])
    o.initial()
    o.run()
    x = o.x()
    nb, nn = o.neighbor_brute_full(x, o.nlocal(), o.param("cutneigh") ** 2)
    np.testing.assert_array_equal(nn, o.numneigh())
    for i, row in enumerate(o.neighbor_rows()):
        np.testing.assert_array_equal(np.sort(row), nb[i, : nn[i]])
    o.close()


@pytest.mark.parametrize("nprocs", [2, 3, 4, 8])
def test_virtual_ranks_match_single_rank(nprocs):
    """spatial decomposition must not change the physics (tests/reference_output/README:3-5)"""
    base = run_oracle(["-s", 8, "-n", 100, "--half_neigh", 0])
    rows = run_oracle(["-s", 8, "-n", 100, "--half_neigh", 0], nprocs=nprocs)
    for a, b in zip(rows, base):
        assert a[0] == b[0]
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (a, b)


def test_oracle_exchange_all_equals_exchange_on_a_healthy_run():
    """Comm::exchange_all (ref/comm.cpp:599-689) offers the leavers of a dimension to every rank within `need` sub-domains; while
    no atom moves further than one sub-domain between two re-neighborings every rank keeps exactly the atoms, in the order, the
    plain Comm::exchange gives it — rows, per-rank counts and positions of the two oracle runs are identical (4 virtual ranks,
    1x1x4 grid of sub-domains thinner than the cutoff, need = 2, and a 2x2x2 grid with need = 1)."""
    for args, nprocs in ((["-nx", "3", "-ny", "3", "-nz", "6", "-n", "60", "--half_neigh", "0"], 4), (["-s", "6", "-n", "60", "--half_neigh", "1"], 8)):
        a = Oracle(args, nprocs=nprocs)
        b = Oracle(args + ["--safe_exchange"], nprocs=nprocs)
        for o in (a, b):
            o.initial(); o.run()
        assert a.rows() == b.rows()
        for p in range(nprocs):
            assert a.nlocal(p) == b.nlocal(p) and a.nghost(p) == b.nghost(p)
            assert np.array_equal(a.x(p), b.x(p))
        a.close(); b.close()
