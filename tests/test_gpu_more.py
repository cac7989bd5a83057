"""`-m gpu` parity tests, part 2: EAM, single precision, half neighbor lists, two ranks on one GPU, and
size-independent properties at the BASELINE.json sizes (-s 80 LJ, -s 64 EAM)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle_lib import Oracle, fmt7, ref_pass_rule
from conftest import free_port
from test_gpu_parity import handle_from_oracle, mm, rows_close, sim_rows, REFRUNS, PUBLISHED

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


# ---- ForceEAM::compute_fullneigh (ref/force_eam.cpp:274-449) -----------------------------------------
@pytest.mark.parametrize("size,ntypes", [(4, 4), (5, 1)])
def test_eam_force_full_matches_oracle(size, ntypes):
    o = Oracle(["-i", "in.eam.miniMD", "-s", size, "-n", 20, "--half_neigh", 0, "--ntypes", ntypes])
    o.initial(); o.run()
    h = handle_from_oracle(o)
    h.comm_setup(o.param("cutneigh"), 0, 1)
    h.exchange(); h.borders()                          # the fp halo needs the send lists (ghosts are rebuilt identically)
    np.testing.assert_array_equal(h.download()["x"], o.x())
    from minimd_amd import api
    t = api.eam_tables_from_file(os.path.join(REPO, "data", "Cu_u6.eam"), ntypes)
    h.force_eam_setup(ntypes, t)
    h.neighbor_upload(o.neighbors(), o.numneigh())
    assert h.counter("tiles_ready") == 1
    nl = o.nlocal()
    fo, fpo = o.f(), o.eam_fp()
    for tiles in (1, 0):              # the oracle's rows in tile form (tile sweeps), then on the row kernels
        h.set_option("tiles", tiles)
        eng, vir = h.force_compute(1)
        f = h.download()["f"]
        # tolerance: 1e-11 of the largest component (FMA contraction in the spline Horner forms)
        assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
        fp = h.eam_fp()
        assert np.abs(fp - fpo).max() <= 1e-12 * np.abs(fpo).max()      # owned AND ghost fp (halo)
        assert abs(eng - o.eng_vdwl()) <= 1e-12 * abs(o.eng_vdwl())
        assert abs(vir - o.virial()) <= 1e-10 * max(1.0, abs(o.virial()))
    h.set_option("tiles", 1)
    # device-built list (tile kernels: knot window in LDS, derived z2r derivative) gives the same physics ...
    h.neighbor_build()
    eng2, vir2 = h.force_compute(1)
    assert abs(eng2 - eng) <= 1e-12 * abs(eng) and np.abs(h.download()["f"] - f).max() <= 1e-11 * np.abs(fo).max()
    # ... also when every pair takes the global-memory path meant for pairs closer than the window's first knot
    h.set_option("eam_mlo", 100000)
    eng3, vir3 = h.force_compute(1)
    assert abs(eng3 - o.eng_vdwl()) <= 1e-12 * abs(o.eng_vdwl()) and np.abs(h.download()["f"] - fo).max() <= 1e-11 * np.abs(fo).max()
    assert np.abs(h.eam_fp() - fpo).max() <= 1e-12 * np.abs(fpo).max()
    h.close(); o.close()


# ---- ForceEAM::compute_halfneigh (ref/force_eam.cpp:94-270): third-law scatter of rho_j and f_j with atomics ---------------
@pytest.mark.parametrize("size,ntypes", [(4, 4), (5, 1)])
def test_eam_force_half_matches_oracle(size, ntypes):
    """the oracle's half-list sweep is pinned bit-exactly to the reference's arrays (tests/golden/arrays_eam_s4_half.npz,
    test_oracle_pin); the device kernels on the SAME half list: fp of owned and ghost atoms, forces over owned + ghost
    atoms (ghosts get none: ref :251-255), energy and virial. Then the device-built half list: same physics, rows as sets."""
    o = Oracle(["-i", "in.eam.miniMD", "-s", size, "-n", 20, "--half_neigh", 1, "--ntypes", ntypes])
    o.initial(); o.run()
    assert int(o.param("halfneigh")) == 1 and int(o.param("ghost_newton")) == 0
    h = handle_from_oracle(o)
    h.comm_setup(o.param("cutneigh"), 0, 1)
    h.exchange(); h.borders()
    np.testing.assert_array_equal(h.download()["x"], o.x())
    from minimd_amd import api
    t = api.eam_tables_from_file(os.path.join(REPO, "data", "Cu_u6.eam"), ntypes)
    h.force_eam_setup(ntypes, t)
    h.neighbor_upload(o.neighbors(), o.numneigh())
    assert h.counter("tiles_ready") == 1
    nl, ng = o.nlocal(), o.nghost()
    fo, fpo = o.f(with_ghosts=True), o.eam_fp()
    for tiles in (1, 0):              # the oracle's half rows in tile form, then on the global-atomic row kernels
        h.set_option("tiles", tiles)
        eng, vir = h.force_compute(1)
        f = h.download(halfneigh=True)["f"]
        assert f.shape == fo.shape == (nl + ng, 3)
        assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
        assert not f[nl:].any()
        fp = h.eam_fp()
        assert np.abs(fp - fpo).max() <= 1e-12 * np.abs(fpo).max()
        assert abs(eng - o.eng_vdwl()) <= 1e-12 * abs(o.eng_vdwl())
        assert abs(vir - o.virial()) <= 1e-10 * max(1.0, abs(o.virial()))
    h.set_option("tiles", 1)
    h.neighbor_build()
    nb, nn = h.neighbor_download()
    np.testing.assert_array_equal(nn, o.numneigh())                 # half rows come back (not the full-list stand-in)
    onb = o.neighbors()
    for i in range(nl):
        assert sorted(nb[i, :nn[i]]) == sorted(onb[i, :nn[i]])
    # device-built list: uniform tables run the TILE kernels (partner shares summed in LDS), the others the row kernels again
    eng2, vir2 = h.force_compute(1)
    f2 = h.download(halfneigh=True)["f"]
    assert abs(eng2 - eng) <= 1e-12 * abs(eng) and np.abs(f2 - fo).max() <= 1e-11 * np.abs(fo).max()
    assert abs(vir2 - o.virial()) <= 1e-10 * max(1.0, abs(o.virial())) and not f2[nl:].any()
    assert np.abs(h.eam_fp() - fpo).max() <= 1e-12 * np.abs(fpo).max()
    eng3, vir3 = h.force_compute(0)                                   # (no energy/virial: another instantiation)
    assert np.abs(h.download(halfneigh=True)["f"] - fo).max() <= 1e-11 * np.abs(fo).max()
    h.set_option("tiles", 0)                                          # the row kernels on the device-built list
    eng4, vir4 = h.force_compute(1)
    assert abs(eng4 - eng) <= 1e-12 * abs(eng) and np.abs(h.download(halfneigh=True)["f"] - fo).max() <= 1e-11 * np.abs(fo).max()
    h.close(); o.close()


def test_eam_half_golden_arrays_of_the_reference():
    """arrays dumped from the REFERENCE objects after an EAM half-list run (-s 4, 20 steps incl. one re-neighboring; fixture
    `s1pre`: the state right after the final ForceEAM::compute_halfneigh): the device kernels on the reference's own
    positions, types and half list reproduce its f (owned + ghost), fp, eng_vdwl and virial"""
    d = np.load(os.path.join(GOLD, "arrays_eam_s4_half.npz"))
    tag = "s1pre"
    nl, ng = int(d[tag + ".nlocal"][0]), int(d[tag + ".nghost"][0])
    ntypes, cutneigh = int(d["ntypes"][0]), float(d["cutneigh"][0])
    assert int(d["halfneigh"][0]) == 1 and int(d["ghost_newton"][0]) == 0
    h = mm().Handle()
    h.set_box(d["prd"])
    h.set_mass(float(d["mass"][0]))
    x = d[tag + ".x"].reshape(-1, 3)
    h.upload(x, d[tag + ".v"].reshape(-1, 3), d[tag + ".type"].astype(np.int32), nlocal=nl)
    h.neighbor_setup([int(v) for v in d["nbin"]], cutneigh, 1, 0, ntypes)
    h.comm_setup(cutneigh, 0, 1)
    from minimd_amd import api
    h.force_eam_setup(ntypes, api.eam_tables_from_file(os.path.join(REPO, "data", "Cu_u6.eam"), ntypes))
    h.exchange(); h.borders()                    # send lists for the fp halo; ghosts are rebuilt in the reference's order
    np.testing.assert_array_equal(h.download()["x"], x)
    nn = d[tag + ".numneigh"].astype(np.int32)
    maxn = int(d[tag + ".maxneighs"][0])
    rows = np.zeros((nl, maxn), np.int32)
    flat, off = d[tag + ".neighbors"], 0
    for i in range(nl):
        rows[i, :nn[i]] = flat[off:off + nn[i]]
        off += nn[i]
    h.neighbor_upload(rows, nn)
    eng, vir = h.force_compute(1)
    f = h.download(halfneigh=True)["f"]
    fo = d[tag + ".f"].reshape(-1, 3)
    assert f.shape == fo.shape and np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
    fpo = d[tag + ".fp"]
    assert np.abs(h.eam_fp()[:nl] - fpo[:nl]).max() <= 1e-12 * np.abs(fpo[:nl]).max()
    assert abs(eng - d[tag + ".eng_vdwl"][0]) <= 1e-12 * abs(d[tag + ".eng_vdwl"][0])
    assert abs(vir - d[tag + ".virial"][0]) <= 1e-10 * abs(d[tag + ".virial"][0])
    h.close()


@pytest.mark.parametrize("name", ["eam_s10_full_n1000", "eam_s16_full_n200"])
def test_run_eam_rows_match_reference(name):
    ent = REFRUNS[name]
    rows = sim_rows([a for a in ent["args"]])
    rows_close(rows, ent["rows"], 1.5e-5)
    for a, b in zip(rows, ent["rows"]):
        if a[0] <= 300:
            for k in (1, 2, 3):
                assert abs(a[k] - b[k]) <= 2e-6 * max(1.0, abs(b[k])), (a, b)
    assert ref_pass_rule(ent["rows"], rows, ent["natoms"], 8, eam=True)[0]


def test_eam_published_log_4k():
    """tests/reference_output/4k.eam was produced with half lists; full lists give the same rows (README there)"""
    ref = [r for r in PUBLISHED["4k.eam"]["rows"] if r[0] <= 500]
    rows = sim_rows(["-i", "in.eam.miniMD", "-s", 10, "-n", 500, "--half_neigh", 0])
    rows_close(rows, ref, 1.5e-5)


# ---- half neighbor lists (ref/force_lj.cpp:271-357): device lists + atomics + reverse halo -----------------
@pytest.mark.parametrize("name", ["lj_s10_half_gn1_n1000", "lj_s10_half_gn0_n1000", "lj_s32_half_n100"])
def test_run_lj_half_rows_match_reference(name):
    ent = REFRUNS[name]
    args = [a for a in ent["args"] if a not in ("-t", "8")]
    s = mm().Sim(args)
    s.initial(); s.run()
    rows = s.rows()
    rows_close(rows, ent["rows"], 1.5e-5)
    assert ref_pass_rule(ent["rows"], rows, ent["natoms"], 8)[0]
    # every pair is stored exactly once (gn=1) / local-ghost pairs twice (gn=0): same totals as the reference
    tot = s.handle.neighbor_info()["total"]
    assert abs(tot - ent["neigh_total"]) <= 5e-6 * ent["neigh_total"]
    s.close()


@pytest.mark.parametrize("build", [1, -1])
def test_half_gn1_device_list_pairs_once(build):
    """our ghost-newton partitions (tiles: every pair on its lower atom in (z,y,x) order; row builders: owned j > i, ghosts by
    that order) differ from the reference's half-stencil bin rule (ref/neighbor.cpp:150-170 + its half stencil) but must store
    each pair once: the half-list forces after reverse communication equal the full-list forces, the list holds half the
    full list's entries. build 1 = k_build_rows (tile form), -1 = k_build (option tiles 0: global rows)."""
    o = Oracle(["-s", 6, "-n", 20, "--half_neigh", 0])
    o.initial(); o.run()
    fo = o.f()
    m = mm()
    h = m.Handle()
    box = o.box()
    h.set_box(box[0:3])
    h.set_mass(1.0)
    h.upload(o.x()[: o.nlocal()], o.v(), o.type()[: o.nlocal()], o.tag())
    h.comm_setup(o.param("cutneigh"), 0, 1)
    h.neighbor_setup(o.nbins(), o.param("cutneigh"), 1, 1, o.ntypes())
    h.force_lj_setup(*o.lj_tables())
    if build < 0:
        h.set_option("tiles", 0)
    h.exchange(); h.borders(); h.neighbor_build()
    assert h.neighbor_info()["total"] * 2 == int(o.numneigh().sum())
    nb, nn = h.neighbor_download()                    # the downloaded rows hold every pair once, too (tile builds: in the reference's partition)
    assert int(nn.sum()) * 2 == int(o.numneigh().sum())
    eng, vir = h.force_compute(1)
    h.reverse_communicate()
    f = h.download(halfneigh=True)["f"][: o.nlocal()]
    assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
    # energy conventions: half lists report half of the full-list double sum (thermo doubles it back, thermo.cpp:123-125)
    assert abs(2 * eng - o.eng_vdwl()) <= 1e-11 * abs(o.eng_vdwl())
    assert abs(vir - o.virial()) <= 1e-10 * abs(o.virial())
    h.close(); o.close()


# ---- single precision build (PRECISION=1, ref/types.h:61-66) ---------------------------------------------
def test_sp_force_matches_sp_oracle():
    o = Oracle(["-s", 6, "-n", 20, "--half_neigh", 0], precision="sp")
    o.initial(); o.run()
    h = handle_from_oracle(o, precision="sp")
    h.force_lj_setup(*o.lj_tables())
    h.neighbor_upload(o.neighbors(), o.numneigh())
    eng, vir = h.force_compute(1)
    f = h.download()["f"]
    fo = o.f()
    # float arithmetic: 2e-6 of the largest component; energy accumulated in double on the device (the
    # reference's float accumulation error is ~3e-3) -> compare against a double recomputation
    assert np.abs(f - fo).max() <= 4e-6 * np.abs(fo).max()
    od = Oracle(["-s", 6, "-n", 1, "--half_neigh", 0])
    fd, ed, vd = od.lj_force_full(o.x().astype(np.float64), o.type(), o.nlocal(), o.neighbors(), o.numneigh(), *od.lj_tables(), 1)
    assert abs(eng - ed) <= 2e-6 * abs(ed) and abs(vir - vd) <= 2e-5 * max(1.0, abs(vd))
    h.neighbor_build()
    nb, nn = h.neighbor_download()
    np.testing.assert_array_equal(nn, o.numneigh())
    h.close(); o.close(); od.close()


def test_sp_run_passes_reference_rule_and_target_known_answer():
    """SP -s 32: judged against the DP rows with the reference's SP tolerance (prec=4 in ref/run_one_test:124),
    and against the known answer of target/run-offload-tests.sh:7-10 (T,U,P at step 100, the script's tol is 1e-1)"""
    ref = REFRUNS["lj_s32_full_n100"]
    rows = sim_rows(["-s", 32, "-n", 100, "--half_neigh", 0], precision="sp")
    assert ref_pass_rule(ref["rows"], rows, ref["natoms"], 4)[0]
    t, u, p = rows[-1][1:]
    assert abs(t - 8.200912e-01) <= 1e-4 * 8.200912e-01
    assert abs(u - (-5.852703e+00)) <= 1e-4 * 5.852703e+00
    assert abs(p - (-1.873937e-01)) <= 5e-3 * 1.873937e-01
    rows_h = sim_rows(["-s", 32, "-n", 100, "--half_neigh", 1], precision="sp")
    assert ref_pass_rule(ref["rows"], rows_h, ref["natoms"], 4)[0]


# ---- two ranks sharing this GPU, halo over the gloo host transport ----------------------------------------
@pytest.mark.parametrize("half", [0, 1])
def test_two_ranks_match_one_rank(half, port, tmp_path):
    args = ["-s", "8", "-n", "100", "--half_neigh", str(half)]
    base = sim_rows(args)
    out = str(tmp_path / "mp.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py"), "sim", out, "dp"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    assert res["natoms"] == 4 * 8 ** 3 and sum(c[0] for c in res["counts"]) == res["natoms"]
    rows = [tuple(x) for x in res["rows"]]
    assert [r_[0] for r_ in rows] == [b[0] for b in base]
    for a, b in zip(rows, base):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (a, b)      # summation order only
    # the oracle on 2 virtual ranks owns the same atoms per rank
    o = Oracle(args, nprocs=2)
    o.initial(); o.run()
    assert [o.nlocal(0), o.nlocal(1)] == [c[0] for c in res["counts"]]
    assert [o.nghost(0), o.nghost(1)] == [c[1] for c in res["counts"]]
    o.close()


@pytest.mark.parametrize("nprocs,size,half", [(4, ["-nx", "8", "-ny", "9", "-nz", "10"], 0), (8, ["-s", "10"], 0),
                                              (8, ["-s", "10"], 1),
                                              # 8 atoms on a 1x1x4 grid: one rank owns NO atom (it still relays ghost forces and takes part in
                                              # every exchange), sub-boxes of 0.3 cutoffs need four ghost layers
                                              (4, ["-nx", "1", "-ny", "1", "-nz", "2"], 0), (4, ["-nx", "1", "-ny", "1", "-nz", "2"], 1)])
def test_four_and_eight_ranks_match_one_rank(nprocs, size, half, port, tmp_path):
    """the decompositions the 4- and 8-GPU runs use (2x2x1 / 2x2x2: both neighbours of a dimension are the same rank,
    corner ghosts travel through chained swaps, atoms migrate in every dimension), here with all ranks sharing this GPU
    over the gloo host transport: rows equal the one-rank run to summation order, per-rank owned/ghost counts equal the
    oracle's virtual ranks"""
    args = size + ["-n", "100", "--half_neigh", str(half)]
    base = sim_rows(args)
    out = str(tmp_path / "mp.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nprocs), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py"), "sim", out, "dp"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    assert sum(c[0] for c in res["counts"]) == res["natoms"]
    rows = [tuple(x) for x in res["rows"]]
    assert [r_[0] for r_ in rows] == [b[0] for b in base]
    for a, b in zip(rows, base):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (a, b)
    o = Oracle(args, nprocs=nprocs)
    o.initial(); o.run()
    assert [o.nlocal(r_) for r_ in range(nprocs)] == [c[0] for c in res["counts"]]
    assert [o.nghost(r_) for r_ in range(nprocs)] == [c[1] for c in res["counts"]]
    o.close()
    # re-neighboring on several ranks without count handshakes (fixed-size messages sized by the previous counts, counts left on the
    # device): the host waits for the GPU twice per re-neighboring — the new nlocal after Comm::exchange, the build's results — plus once
    # per thermo row; the waits of the host-staged test transport are counted apart (RCCL has none). 100 steps = 5 re-neighborings.
    for st in res["stats"]:
        if res["natoms"] >= 1000:                                    # (sub-boxes thinner than half a cutoff take the swap-by-swap borders with their count handshakes)
            # (+ thermo row, + a build that sized its lists again; + the FIRST exchange of the run, which takes the count-handshake path — three
            #  compaction counts and a count handshake per split dimension — because the set-up exchange moves nobody and cannot size its messages)
            ndim_split = 2 if nprocs == 4 else 3
            assert st["host_syncs"] <= 2 * 5 + 4 + 4 * ndim_split, res["stats"]
            assert st["exchange_fast"] == 4 and st["exchange_overflows"] == 0, res["stats"]
        assert st["bytes_sent"] > 0 and st["transport_syncs"] > 0


@pytest.mark.parametrize("prec,deck,nprocs", [("dp", "in.eam.miniMD", 2), ("sp", "in.lj.miniMD", 4), ("dp", "in.eam.miniMD", 8)])
def test_eam_and_sp_on_several_ranks(prec, deck, nprocs, port, tmp_path):
    """EAM needs a second halo per step (fp of the ghosts, ForceEAM::communicate ref/force_eam.cpp:851-913) and the SP build
    moves float4 halos: both on several ranks sharing this GPU against the one-rank run"""
    size = ["-s", "8"] if "eam" in deck else ["-nx", "8", "-ny", "9", "-nz", "10"]
    args = ["-i", deck] + size + ["-n", "60", "--half_neigh", "0"]
    base = sim_rows(args, precision=prec)
    out = str(tmp_path / "mp.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nprocs), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py"), "sim", out, prec] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=os.path.join(REPO, "data"))
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    assert sum(c[0] for c in res["counts"]) == res["natoms"]
    rows = [tuple(x) for x in res["rows"]]
    tol = 1e-9 if prec == "dp" else 2e-5
    assert [r_[0] for r_ in rows] == [b[0] for b in base]
    for a, b in zip(rows, base):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= tol * max(1.0, abs(b[k])), (a, b)


@pytest.mark.parametrize("lists", [["--half_neigh", 1, "-gn", 1], ["--half_neigh", 0]])
def test_rccl_loopback_single_rank(lists):
    """exercise the production transport calls (ncclCommInitRank, grouped ncclSend/ncclRecv, ncclAllReduce) on ONE
    GPU: every periodic self-swap of borders / communicate / reverse_communicate is forced through RCCL; with full
    lists this also runs the overlapped step (halo on the comm stream under the interior tiles)"""
    o = Oracle(["-s", 8, "-n", 40] + lists)
    o.initial(); o.run()
    ref = o.rows()
    m = mm()
    s = m.Sim(["-s", 8, "-n", 40] + lists)
    h = s.handle
    h.init_rccl(h.unique_id(), 0, 1)
    h.set_option("force_transport", 1)
    s.initial(); s.run()
    rows = s.rows()
    for a, b in zip(rows, ref):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (a, b)
    nl, ng, _ = h.counts()
    assert (nl, ng) == (o.nlocal(), o.nghost())
    s.close(); o.close()


def test_overlap_split_half_lists_equal_single_launch():
    """half lists: the interior tiles of the third-law kernel run under the position halo, the boundary tiles behind it, then the
    reverse halo. The scatter uses floating-point atomics, so the comparison is to summation order (1e-9) instead of bit-wise."""
    m = mm()
    rows = {}
    for ov in (1, 0):
        s = m.Sim(["-s", 12, "-n", 60, "--half_neigh", 1])
        h = s.handle
        h.init_rccl(h.unique_id(), 0, 1)
        h.set_option("force_transport", 1)
        h.set_option("overlap", ov)
        s.initial(); s.run()
        rows[ov] = s.rows()
        s.close()
    for a, b in zip(rows[0], rows[1]):
        assert a[0] == b[0]
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (a, b)


@pytest.mark.parametrize("deck", ["in.lj.miniMD", "in.eam.miniMD"])
def test_overlap_split_equals_single_launch(deck):
    """interior + boundary tile launches (the multi-GPU overlap path) give bit-identical results to one launch. LJ: the
    position halo runs on the communication stream under the interior tiles. EAM: the position halo under the interior tiles
    of the density sweep, the fp halo (ForceEAM::communicate) under the interior tiles of the force sweep."""
    m = mm()
    rows = {}
    for ov in (1, 0):
        s = m.Sim(["-i", deck, "-s", 12 if "lj" in deck else 8, "-n", 60, "--half_neigh", 0])
        h = s.handle
        h.init_rccl(h.unique_id(), 0, 1)
        h.set_option("force_transport", 1)
        h.set_option("overlap", ov)
        s.initial(); s.run()
        rows[ov] = (s.rows(), h.download()["f"].copy(), h.download()["x"].copy())
        s.close()
    assert rows[0][0] == rows[1][0]
    np.testing.assert_array_equal(rows[0][1], rows[1][1])
    np.testing.assert_array_equal(rows[0][2], rows[1][2])


@pytest.mark.parametrize("deck,half", [("in.lj.miniMD", 0), ("in.lj.miniMD", 1), ("in.eam.miniMD", 0)])
@pytest.mark.parametrize("overlap", [0, 1])
def test_direct_halo_equals_the_swap_by_swap_halo(deck, half, overlap):
    """Several ranks (here: one rank whose periodic self swaps are forced through RCCL, the code path of a rank inside a multi-GPU run): the forward
    communication of a step — positions, and fp between the EAM sweeps — as ONE exchange with the up to 26 neighbours (26 lists of owned atoms per
    rank, one message per partner, option direct_halo) against the three dependent rounds of ref/comm.cpp:276-317. Same ghosts in the same slots:
    bit-identical rows, positions and forces (half lists: to the order of the atomics) after 60 steps with 2 re-neighborings."""
    m = mm()
    out = {}
    for dh in (0, 1):
        s = m.Sim(["-i", deck, "-s", 12 if "lj" in deck else 8, "-n", 60, "--half_neigh", half])
        h = s.handle
        h.init_rccl(h.unique_id(), 0, 1)
        h.set_option("force_transport", 1)
        h.set_option("overlap", overlap)
        h.set_option("direct_halo", dh)
        s.initial(); s.run()
        d = h.download()
        out[dh] = (s.rows(), d["x"].copy(), d["f"].copy(), h.run_stats()["bytes_sent"])
        s.close()
    assert out[0][3] > 0 and out[1][3] > 0
    if half:
        rows_close(out[0][0], out[1][0], 1e-10)
        assert np.abs(out[0][1] - out[1][1]).max() <= 1e-9
    else:
        assert out[0][0] == out[1][0]
        np.testing.assert_array_equal(out[0][1], out[1][1])
        np.testing.assert_array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("args", [["-s", 12], ["-s", 3], ["-nx", 2, "-ny", 5, "-nz", 3]])
def test_ghosts_staged_from_their_owners_equal_the_ghost_update(args):
    """One rank: the tile kernel reads a ghost candidate as owner position + box shift (no per-step Comm::communicate) — the
    same bits as the ghost-update kernel writes into the ghost slots, so thermo rows, forces and the final state (ghost
    positions included: refreshed when the run returns) are identical. -s 3 and 2x5x3 have images of images."""
    m = mm()
    out = {}
    for mode in (0, 2):
        s = m.Sim(args + ["-n", 60, "--half_neigh", 0])
        s.handle.set_option("ghost_resolve", mode)
        s.initial(); s.run()
        d = s.handle.download()
        out[mode] = (s.rows(), d["x"].copy(), d["v"].copy(), d["f"].copy())
        s.close()
    assert out[0][0] == out[2][0]
    for k in (1, 2, 3):
        np.testing.assert_array_equal(out[0][k], out[2][k])


@pytest.mark.parametrize("prec", ["dp", "sp"])
@pytest.mark.parametrize("gn", [0, 1])
@pytest.mark.parametrize("args", [["-s", 12], ["-s", 3], ["-nx", 2, "-ny", 5, "-nz", 3]])
def test_half_list_ghosts_staged_from_their_owners(args, gn, prec):
    """LJ over half lists on one rank: on steps without re-neighboring the tile kernel stages the ghosts from their owners (second candidate
    list of the build: owner + image code) — no Comm::communicate launch; with ghost newton a ghost's share goes to its owner at the flush,
    without it a ghost partner gets none and the pair counts half. Same sums in the same order as with the ghost update (ghost_resolve 0):
    rows to 1e-10 (atomics: the order inside a launch is not fixed), positions of the owned atoms to 1e-9 after 60 steps."""
    m = mm()
    out = {}
    for mode in (0, 1):
        s = m.Sim(args + ["-n", 60, "--half_neigh", 1, "-gn", gn], precision=prec)
        s.handle.set_option("ghost_resolve", mode)
        s.initial(); s.run()
        d = s.handle.download()
        nl = s.handle.counts()[0]
        out[mode] = (s.rows(), d["x"][:nl].copy(), d["f"][:nl].copy(), d["tag"].copy())
        s.close()
    tol = 1e-10 if prec == "dp" else 2e-5
    rows_close(out[0][0], out[1][0], tol)
    np.testing.assert_array_equal(out[0][3], out[1][3])
    assert np.abs(out[0][1] - out[1][1]).max() <= (1e-9 if prec == "dp" else 1e-3)


@pytest.mark.parametrize("prec", ["dp", "sp"])
@pytest.mark.parametrize("args", [["-s", 10], ["-s", 4], ["-nx", 3, "-ny", 6, "-nz", 4]])
def test_eam_ghosts_staged_from_their_owners_equal_the_halos(args, prec):
    """EAM over full lists on one rank: the build leaves the candidate lists a second time with every ghost named by its owner and image code,
    and on steps without re-neighboring both sweeps stage the ghosts from their owners (position + box shift; fp of the owner) — no
    Comm::communicate launch, no ForceEAM::communicate between the sweeps. Same bits as with the two halos (ghost_resolve 0): thermo rows,
    positions (ghosts refreshed when the run returns), velocities, forces. -s 4 and 3x6x4 cells have images of images."""
    m = mm()
    out = {}
    for mode in (0, 1):
        s = m.Sim(["-i", "in.eam.miniMD"] + args + ["-n", 60, "--half_neigh", 0], precision=prec)
        s.handle.set_option("ghost_resolve", mode)
        s.initial(); s.run()
        d = s.handle.download()
        out[mode] = (s.rows(), d["x"].copy(), d["v"].copy(), d["f"].copy())
        s.close()
    assert out[0][0] == out[1][0]
    for k in (1, 2, 3):
        np.testing.assert_array_equal(out[0][k], out[1][k])


@pytest.mark.parametrize("args", [["-s", 12], ["-s", 3], ["-nx", 2, "-ny", 5, "-nz", 3]])
def test_reverse_communicate_folded_into_the_half_kernel(args):
    """One rank, half lists with ghost newton: the tile kernel adds a ghost's share of a pair to the ghost's owner directly
    instead of summing it on the ghost and sending it home (Comm::reverse_communicate, ref/comm.cpp:321-355) — the same sums
    in another order."""
    m = mm()
    out = {}
    for mode in (0, 1):
        s = m.Sim(args + ["-n", 60, "--half_neigh", 1])
        s.handle.set_option("fold_reverse", mode)
        s.initial(); s.run()
        d = s.handle.download()
        nl = s.handle.counts()[0]
        out[mode] = (s.rows(), d["x"][:nl].copy(), d["f"][:nl].copy())
        s.close()
    rows_close(out[0][0], out[1][0], 1e-10)
    fmax = np.abs(out[0][2]).max()
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-9
    assert np.abs(out[0][2] - out[1][2]).max() <= 1e-8 * fmax


@pytest.mark.parametrize("half", [0, 1])
def test_reneighboring_with_counts_left_on_the_device(half):
    """inside Integrate::run a one-rank re-neighboring reads no count before the neighbor build has run: list sizes come from the
    previous build, the ghost and tile counts return with the build's result flags. With estimates that are too small (borders_est 60 %) the
    swap-by-swap path redoes the borders and the build runs again: same rows, same final state."""
    m = mm()
    out = {}
    for mode in ("async", "async_overflow"):
        s = m.Sim(["-s", 14, "-n", 100, "--half_neigh", half])
        if mode == "async_overflow":
            s.handle.set_option("borders_est", 60)
        s.initial(); s.run()
        d = s.handle.download()
        nl, ng, _ = s.handle.counts()
        out[mode] = (s.rows(), nl, ng, s.handle.neighbor_info()["total"], d["x"][:nl].copy(), d["tag"].copy(), s.handle.counter("borders_general"))
        s.close()
    assert out["async_overflow"][6] > out["async"][6]                # (the fall-back really ran)
    assert out["async_overflow"][1:4] == out["async"][1:4]
    if half:
        rows_close(out["async_overflow"][0], out["async"][0], 1e-10)          # (atomics: summation order)
    else:
        assert out["async_overflow"][0] == out["async"][0]
        np.testing.assert_array_equal(out["async_overflow"][4], out["async"][4])
        np.testing.assert_array_equal(out["async_overflow"][5], out["async"][5])


def test_eam_rows_in_two_parts_give_the_same_run():
    """EAM full lists on one rank: rows are written core-first (pairs closer than cutforce + 30 % of the skin at the build), the
    force kernels stop after the core part for as long as the tracked displacement since the build stays below half that margin.
    A pair of the rest cannot be inside the force cutoff then, so the run is the same run: thermo rows, positions and velocities
    after 100 steps (5 re-neighborings, a thermo step) against whole rows (core_pct 0) and against a margin so small (2 %) that the
    kernels must fall back to whole rows after a few steps of every window."""
    m = mm()
    out = {}
    for pct in (0, 30, 2):
        s = m.Sim(["-i", "in.eam.miniMD", "-s", 10, "-n", 100, "--half_neigh", 0])
        s.handle.set_option("core_pct", pct)
        s.initial(); s.run()
        d = s.handle.download()
        nl = s.handle.counts()[0]
        out[pct] = (s.rows(), d["x"][:nl].copy(), d["v"][:nl].copy(), s.handle.neighbor_info()["total"], d["tag"].copy())
        s.close()
    for pct in (30, 2):
        rows_close(out[pct][0], out[0][0], 1e-11)
        assert out[pct][3] == out[0][3]
        np.testing.assert_array_equal(out[pct][4], out[0][4])
        # (the two parts change the order of the sums, nothing else: 1e-10 A after 100 steps)
        assert np.abs(out[pct][1] - out[0][1]).max() <= 1e-10
        assert np.abs(out[pct][2] - out[0][2]).max() <= 1e-9


@pytest.mark.parametrize("half", [0, 1])
def test_profiling_the_force_kernel_between_two_slices_leaves_the_run_untouched(half):
    """mmd_profile_kernel(0) = Force::compute launches on the current state (bench.py and tools/run_configs.py warm the clocks with
    it): the run continues as if nothing had happened — with half lists and ghost newton that needs the ghosts' shares sent home
    before the next initialIntegrate"""
    m = mm()
    rows = {}
    for probe in (0, 1):
        s = m.Sim(["-s", 10, "-n", 100, "--half_neigh", half])
        s.initial()
        s.run_steps(37)
        if probe:
            s.handle.profile_kernel(0, 3)
        s.run_steps(63)
        rows[probe] = s.rows()
        s.close()
    rows_close(rows[1], rows[0], 1e-11 if half else 0.0)


# ---- BASELINE.json sizes: golden rows + size-independent properties -----------------------------------------
def test_baseline_s80_full_and_half():
    ent = REFRUNS["lj_s80_full_n100"]
    s = mm().Sim(["-s", 80, "-n", 100, "--half_neigh", 0])
    s.initial(); s.run()
    rows = s.rows()
    rows_close(rows, ent["rows"], 2e-6)
    assert fmt7(rows[0][1]) == fmt7(ent["rows"][0][1]) and fmt7(rows[0][2]) == fmt7(ent["rows"][0][2])
    nl, ng, _ = s.handle.counts()
    assert nl == 2048000 and ng == int(ent["nghost"])
    assert abs(s.handle.neighbor_info()["total"] - ent["neigh_total"]) <= 5e-6 * ent["neigh_total"]
    d = s.handle.download()
    # Newton's third law over the periodic system: total force vanishes; atoms stay inside the box after PBC
    assert np.abs(d["f"].sum(axis=0)).max() <= 1e-7 * np.abs(d["f"]).max() * np.sqrt(nl)
    assert len(np.unique(d["tag"])) == nl
    s.close()
    rows_h = sim_rows(["-s", 80, "-n", 100, "--half_neigh", 1])
    rows_close(rows_h, ent["rows"], 2e-6)


def test_beyond_baseline_s144_rows_equal_the_reference():
    """11.9 M atoms (5.8x the BASELINE -s 80): index arithmetic, tile counts and list sizes far past the tested sizes"""
    ent = REFRUNS["lj_s144_full_n100"]
    rows = sim_rows(ent["args"][2:-2])
    rows_close(rows, ent["rows"], 2e-6)
    assert [fmt7(v) for v in rows[-1][1:4]] == [fmt7(v) for v in ent["rows"][-1][1:4]]


def test_baseline_eam_s64():
    ent = REFRUNS["eam_s64_full_n100"]
    s = mm().Sim(["-i", "in.eam.miniMD", "-s", 64, "-n", 100, "--half_neigh", 0])
    s.initial(); s.run()
    rows_close(s.rows(), ent["rows"], 2e-6)
    nl, ng, _ = s.handle.counts()
    assert nl == 1048576 and ng == int(ent["nghost"])
    assert abs(s.handle.neighbor_info()["total"] - ent["neigh_total"]) <= 5e-6 * ent["neigh_total"]
    s.close()


def test_config_e_full_size_dp_equals_the_reference_row_and_sp_follows_dp():
    """BASELINE configs[4] at its real size: -s 160 (16 384 000 atoms), half neighbor lists with the third-law scatter, 100 steps.
    DOUBLE precision: the rows equal the row the UNMODIFIED reference printed for this configuration (tests/golden/ref_runs.json
    `lj_s160_half_n100`, a 10-minute run of oracle/_ref/miniMD_ref_dp -s 160 --half_neigh 1 -t 8) to the printed digits, ghost count and
    neighbor total equal its YAML report. SINGLE precision (the configuration BASELINE names): no SP row of the reference is pinned at
    this size (its float sums have lost their digits), so the SP run is judged by size-independent properties — atoms
    conserved with unique tags, total force ~ 0, the initial lists hold exactly the pairs the DP build finds — and against the DP rows
    within float-trajectory bounds."""
    natoms = 4 * 160 ** 3
    ent = REFRUNS["lj_s160_half_n100"]
    assert ent["natoms"] == natoms
    out = {}
    for prec in ("dp", "sp"):
        s = mm().Sim(["-s", 160, "-n", 100, "--half_neigh", 1], precision=prec)
        s.initial()
        tot0 = s.handle.neighbor_info()["total"]
        s.run()
        nl, ng, _ = s.handle.counts()
        assert nl == natoms == s.natoms()
        tot = s.handle.neighbor_info()["total"]
        if prec == "sp":
            d = s.handle.download(halfneigh=True)
            assert len(np.unique(d["tag"])) == natoms
            f = d["f"][:nl].astype(np.float64)
            assert np.abs(f.sum(axis=0)).max() <= 2e-6 * np.abs(f).max() * np.sqrt(nl)       # Newton's third law, float sums
            x = d["x"][:nl]
            prd = s.handle.get_box()[0]
            assert (x.min(axis=0) >= -0.6).all() and (x.max(axis=0) <= prd + 0.6).all()       # within a skin of the box between exchanges
            del d, f, x
        out[prec] = (s.rows(), tot0, tot, ng)
        s.close()
    # ---- DP against the reference's own row
    rows_close(out["dp"][0], ent["rows"], 2e-6)
    for a, b in zip(out["dp"][0], ent["rows"]):
        assert [fmt7(v) for v in a[1:4]] == [fmt7(v) for v in b[1:4]], (a, b)               # digit for digit
    assert abs(out["dp"][3] - ent["nghost"]) <= 5e-6 * ent["nghost"]                     # (6 printed digits)
    assert abs(out["dp"][2] - ent["neigh_total"]) <= 5e-6 * ent["neigh_total"]            # (the YAML report prints 6 digits)
    # ---- SP against DP
    assert out["sp"][1] == out["dp"][1]                       # the lattice has no pair within float rounding of the cutoff
    assert abs(out["sp"][2] - out["dp"][2]) <= 2e-4 * out["dp"][2] and abs(out["sp"][3] - out["dp"][3]) <= 2e-4 * out["dp"][3]    # (float trajectories drift: other pairs sit inside the skin after 100 steps)
    assert [r[0] for r in out["sp"][0]] == [0, 100]
    # (the reference's statistical pass rule shrinks with 1/sqrt(natoms): at 16 M atoms it is tighter than float rounding, and
    #  the reference's own SP build misses it by far; compare the rows directly instead. The SP build scales the initial
    #  velocities with float sums over 16 M atoms exactly like the reference (ref/setup.cpp:485-517 in MMD_float), so its
    #  starting temperature is already 0.4 % off 1.44 and the trajectories are different members of the same ensemble)
    for a, b in zip(out["sp"][0], out["dp"][0]):
        assert abs(a[1] - b[1]) <= 1e-2 * abs(b[1]) and abs(a[2] - b[2]) <= 2e-3 * abs(b[2]) and abs(a[3] - b[3]) <= 5e-2, (a, b)
    t, u, p = out["sp"][0][-1][1:]
    assert abs(u - (-5.652)) < 5e-3 and abs(t - 0.695) < 5e-3


# ---- the reference's CoMD-parameter decks (ref/in.lj.miniMD_comd, ref/in.eam.miniMD_comd; values in data/in.*.miniMD_comd) ------------------
@pytest.mark.parametrize("name", ["lj_comd_s10_full_n1000", "lj_comd_s10_half_n1000", "lj_comd_s10_full_n300_sp", "eam_comd_s10_full_n300",
                                  "eam_comd_s10_half_n300"])
def test_comd_decks_match_the_reference_rows(name):
    """in.lj.miniMD_comd is the only deck with epsilon, sigma != 1 (0.167 / 2.315, cutoff 4.59, dt 5e-5): it exercises the folded constant
    c_out = 48 eps sigma^6 of the LJ tile kernels away from 48 and a 704 000-entry list on 4000 atoms (176 neighbors per atom); in.eam.miniMD_comd
    runs EAM at another density with skin 0.5 and a thermo row every 10 steps. Rows of the unmodified reference (tests/golden/ref_runs.json):
    <= 2e-6 relative to step 300, <= 1.5e-5 to step 1000, plus the reference's own pass rule; neighbor totals and ghost counts of its YAML report."""
    ent = REFRUNS[name]
    prec = ent["precision"]
    s = mm().Sim([a for a in ent["args"]], precision=prec, cwd=os.path.join(REPO, "data"))
    s.initial(); s.run()
    rows = s.rows()
    eam = name.startswith("eam")
    if prec == "dp":
        rows_close(rows, ent["rows"], 1.5e-5)
        for a, b in zip(rows, ent["rows"]):
            if a[0] <= 300:
                for k in (1, 2, 3):
                    assert abs(a[k] - b[k]) <= 2e-6 * max(1.0, abs(b[k])), (a, b)
    else:
        # single precision: the reference sums 704 000 pair energies of ~+0.9 each in FLOAT — its own step-0 row reads U = 166.5229, P = 221.5665
        # where the exact lattice values (its DP build, and this code in either precision: sums in double, rounded once) are 166.3043 / 221.2786.
        # The SP run is therefore judged against the reference's DOUBLE rows with the reference's SP pass rule, and against the reference's SP
        # rows only as far as their own summation error allows (2.5e-3).
        dp_rows = [r for r in REFRUNS["lj_comd_s10_full_n1000"]["rows"] if r[0] <= 300]
        rows_close(rows, dp_rows, 2e-4)
        for k in (1, 2, 3):
            assert abs(rows[0][k] - dp_rows[0][k]) <= 2e-5 * max(1.0, abs(dp_rows[0][k])), (rows[0], dp_rows[0])
        rows_close(rows, ent["rows"], 2.5e-3)
        assert ref_pass_rule(dp_rows, rows, ent["natoms"], 4, eam=eam)[0]
    if prec == "dp":
        assert ref_pass_rule(ent["rows"], rows, ent["natoms"], 8, eam=eam)[0]
    nl, ng, _ = s.handle.counts()
    assert nl == ent["natoms"] and ng == int(ent["nghost"])
    assert abs(s.handle.neighbor_info()["total"] - ent["neigh_total"]) <= 5e-6 * ent["neigh_total"] + (2 if prec == "sp" else 0)
    s.close()


@pytest.mark.parametrize("name", ["lj_s1_full_n60", "lj_1x3x2_half_n60", "lj_1x3x2_full_n60", "eam_2x1x3_full_n60"])
def test_two_ghost_layers_one_rank(name):
    """boxes thinner than the neighbor cutoff need TWO ghost layers per dimension (need = 2, ref/comm.cpp:150-152,208-269):
    4 swaps per dimension, images of images, ghost chains of length 2 in the one-kernel halo. Rows equal the
    unmodified reference's, ghost / neighbor counts equal its YAML report."""
    ent = REFRUNS[name]
    s = mm().Sim([a for a in ent["args"]], cwd=os.path.join(REPO, "data"))
    assert max(s.handle.comm_info()["need"]) == 2
    s.initial(); s.run()
    rows_close(s.rows(), ent["rows"], 2e-6)
    nl, ng, _ = s.handle.counts()
    assert (nl, ng) == (int(ent["nlocal"]), int(ent["nghost"]))
    assert s.handle.neighbor_info()["total"] == int(ent["neigh_total"])
    s.close()


@pytest.mark.parametrize("size,nprocs,half", [(["-s", "3"], 2, 0), (["-nx", "3", "-ny", "3", "-nz", "6"], 4, 1),
                                                   (["-nx", "4", "-ny", "3", "-nz", "3"], 3, 0)])
def test_two_ghost_layers_several_ranks(size, nprocs, half, port, tmp_path):
    """sub-domains thinner than the cutoff on several ranks (sharing this GPU, gloo host transport): a rank's ghosts come
    from its neighbor AND from the rank beyond it (second swap pair of the dimension, ref/comm.cpp:208-269). Rows equal the
    one-rank run to summation order; owned / ghost counts per rank equal the oracle's virtual ranks."""
    args = size + ["-n", "60", "--half_neigh", str(half)]
    base = sim_rows(args)
    out = str(tmp_path / "mp.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nprocs), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py"), "sim", out, "dp"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    assert sum(c[0] for c in res["counts"]) == res["natoms"]
    rows = [tuple(x) for x in res["rows"]]
    assert [r_[0] for r_ in rows] == [b[0] for b in base]
    for a, b in zip(rows, base):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (a, b)
    o = Oracle(args, nprocs=nprocs)
    o.initial(); o.run()
    assert o.nswap(0) > 6                                   # more than one swap pair in some dimension
    assert [o.nlocal(r_) for r_ in range(nprocs)] == [c[0] for c in res["counts"]]
    assert [o.nghost(r_) for r_ in range(nprocs)] == [c[1] for c in res["counts"]]
    o.close()


def test_config_e_sp_half_scaled_down():
    """BASELINE configs[4] (-s 160 SP half lists) at -s 48: runs, conserves atoms, passes the SP rule vs the DP run"""
    ref = sim_rows(["-s", 48, "-n", 100, "--half_neigh", 0])
    rows = sim_rows(["-s", 48, "-n", 100, "--half_neigh", 1], precision="sp")
    assert ref_pass_rule(ref, rows, 4 * 48 ** 3, 4)[0]


# ---- the drop-in executable: same CLI / stdout grammar as ref/ljs.cpp ---------------------------------------
def parse_thermo(text):
    rows, on = [], False
    for line in text.splitlines():
        if line.startswith("# Timestep"):
            on = True
            continue
        if line.startswith("# Performance Summary"):
            break
        if on and len(line.split()) >= 4:
            p = line.split()
            rows.append((int(p[0]), float(p[1]), float(p[2]), float(p[3])))
    return rows


@pytest.mark.parametrize("exe,deck,extra", [("miniMD_dp", "in.lj.miniMD", ["--half_neigh", "0"]), ("miniMD_dp", "in.eam.miniMD", ["--half_neigh", "0"]),
                                            ("miniMD_sp", "in.lj.miniMD", ["--half_neigh", "1"])])
def test_executable_stdout_grammar_and_rows(exe, deck, extra):
    """what ref/run_one_test greps: '# Atoms:' $3, 'System size' field 10, '# Size of float' $5, the thermo block
    between '# Timestep T' and '# Performance Summary', and the PERF_SUMMARY row (unknown flags such as -dm ignored)"""
    path = os.path.join(REPO, "minimd_amd", "bin", exe)
    r = subprocess.run([path, "-t", "1", "-s", "10", "-n", "200", "--yaml_output", "0", "-dm", "-i", deck] + extra,
                       cwd=os.path.join(REPO, "data"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    atoms = [l for l in out.splitlines() if "# Atoms:" in l][0].split()
    assert atoms[2] == "4000"
    size = [l for l in out.splitlines() if "System size" in l][0].split()
    assert size[9] == "10"
    assert [l for l in out.splitlines() if "Size of float" in l][0].split()[4] == ("8" if exe.endswith("dp") else "4")
    rows = parse_thermo(out)
    assert [r_[0] for r_ in rows] == [0, 100, 200]
    perf = [l for l in out.splitlines() if "PERF_SUMMARY" in l and not l.startswith("#")][0].split()
    assert perf[2] == "200" and perf[3] == "4000" and float(perf[9]) > 0
    key = "4k.eam" if "eam" in deck else "4k.lj"
    ref = [tuple(x) for x in PUBLISHED[key]["rows"] if x[0] <= 200]
    ok, frac = ref_pass_rule(ref, rows, 4000, 8 if exe.endswith("dp") else 4, eam="eam" in deck)
    assert ok, frac
    if exe.endswith("dp"):
        rows_close(rows, ref, 2e-6)


def test_executable_reports_errors_like_the_reference():
    path = os.path.join(REPO, "minimd_amd", "bin", "miniMD_dp")
    r = subprocess.run([path, "-i", "does_not_exist.miniMD"], cwd=os.path.join(REPO, "data"), capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "Cannot open" in r.stdout          # ref/input.cpp:62-66 prints and exits 0


# ---- edge cases: tiny / ragged boxes, unusual bin counts, type counts ------------------------------------------
@pytest.mark.parametrize("args", [["-s", 2], ["-s", 3], ["-nx", 2, "-ny", 5, "-nz", 3], ["-s", 6, "-b", 1], ["-s", 6, "-b", 2],
                                  ["-s", 6, "-b", 20], ["-s", 6, "-b", 40], ["-s", 5, "-b", 64], ["-s", 5, "--ntypes", 1], ["-s", 5, "--ntypes", 8], ["-s", 5, "--ntypes", 40], ["-s", 4, "--sort", 0],
                                  ["-s", 4, "--sort", 7]])
@pytest.mark.parametrize("half", [0, 1])
def test_edge_case_runs_match_oracle(args, half):
    """same flags through the oracle and the device path; rows agree to summation order (1e-9) over 60 steps
    (3 re-neighborings), counts of owned/ghost atoms and neighbor totals agree exactly. `-b 40` / `-b 64` ask for bins of a fifteenth of the
    cutoff and finer: the device then bins on its own coarser grid (mmd_neighbor_setup) — the lists do not depend on the bins."""
    full = [str(a) for a in args] + ["-n", "60", "--half_neigh", str(half)]
    o = Oracle(full)
    o.initial(); o.run()
    s = mm().Sim(full)
    s.initial(); s.run()
    rows, ref = s.rows(), o.rows()
    assert [r_[0] for r_ in rows] == [r_[0] for r_ in ref]
    for a, b in zip(rows, ref):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 2e-9 * max(1.0, abs(b[k])), (a, b)
    nl, ng, _ = s.handle.counts()
    assert (nl, ng) == (o.nlocal(), o.nghost())
    if not (half and "-b" in full):       # half lists with ghost newton partition pairs differently, totals still equal
        assert s.handle.neighbor_info()["total"] == int(o.numneigh().sum())
    s.close(); o.close()


def test_thermo_every_step_and_odd_lengths():
    """thermo_nstat from the deck (100) with a run length that is not a multiple: final row printed (thermo.cpp:80)"""
    o = Oracle(["-s", 5, "-n", 130, "--half_neigh", 0]); o.initial(); o.run()
    rows = sim_rows(["-s", 5, "-n", 130, "--half_neigh", 0])
    assert [r_[0] for r_ in rows] == [0, 100, 130] == [r_[0] for r_ in o.rows()]
    for a, b in zip(rows, o.rows()):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 2e-9 * max(1.0, abs(b[k]))
    o.close()


def test_stale_list_and_bad_arguments_are_errors():
    m = mm()
    h = m.Handle()
    with pytest.raises(m.MMDError, match="box not set"):
        h.neighbor_setup([4, 4, 4], 2.8, 0, 1, 4)
    h.set_box([10.0, 10.0, 10.0])
    h.neighbor_setup([4, 4, 4], 2.8, 0, 1, 4)
    x = np.random.default_rng(1).random((100, 3)) * 10
    h.upload(x, None, np.zeros(100, np.int32))
    h.force_lj_setup(np.full(16, 6.25), np.ones(16), np.ones(16))
    with pytest.raises(m.MMDError, match="stale"):
        h.force_compute(1)
    with pytest.raises(m.MMDError):
        h.set_option("no_such_option", 1)
    h.close()


def test_eam_half_request_and_original_force_alias():
    """EAM with the reference's default --half_neigh 1 (serial-only path there) runs the device half-list kernels; with
    --eam_half_full the full-list kernels stand in (energy converted to the half-list convention): both reproduce the rows
    of the reference's half-list run; --half_neigh -1 (ForceLJ::compute_original, ref/force_lj.cpp:118-176) runs the half +
    ghost-newton lists through the un-tiled row kernel k_lj_half (a path of its own: its timers show no tile kernel), rows as
    the reference's half-list run"""
    ent = REFRUNS["eam_s10_half_n300"]
    for extra in ([], ["--eam_half_full"]):
        s = mm().Sim([a for a in ent["args"]] + extra)
        s.initial(); s.run()
        rows_close(s.rows(), ent["rows"], 2e-6)
        tot = s.handle.neighbor_info()["total"]
        if extra:       # full rows: owned-owned pairs twice, owned-ghost pairs once
            assert ent["neigh_total"] < tot < 2 * ent["neigh_total"]
        else:           # the half list of the reference's YAML report (printed with 6 digits)
            assert abs(tot - ent["neigh_total"]) <= 5e-6 * ent["neigh_total"]
        s.close()
    ref = REFRUNS["lj_s10_half_gn1_n1000"]
    rows = sim_rows(["-s", 10, "-n", 300, "--half_neigh", -1])
    rows_close(rows, [r_ for r_ in ref["rows"] if r_[0] <= 300], 2e-6)
    # the same request through the tile kernel (option lj_original 0) and through the row kernel: same physics
    out = {}
    for orig in (1, 0):
        s = mm().Sim(["-s", 10, "-n", 100, "--half_neigh", -1])
        s.handle.set_option("lj_original", orig)
        s.initial(); s.run()
        out[orig] = s.rows()
        s.close()
    rows_close(out[1], out[0], 1e-9)


def test_yaml_report_matches_reference_counts(tmp_path):
    """`-o 1 --yaml_screen` (ref/output.cpp): same keys; the structural counts equal the reference's own report"""
    ent = REFRUNS["lj_s16_full_n300"]
    path = os.path.join(REPO, "minimd_amd", "bin", "miniMD_dp")
    for f in ("in.lj.miniMD",):
        os.symlink(os.path.join(REPO, "data", f), str(tmp_path / f))
    r = subprocess.run([path, "-i", "in.lj.miniMD", "-s", "16", "-n", "300", "--half_neigh", "0", "-o", "1", "--yaml_screen"],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    out = r.stdout
    for key in ("run_configuration:", "  variant:", "  atoms: 16384", "  unit_cells: 16 16 16", "  neighbor_type: 0", "thermodynamic_output:",
                "      Conservation:", "time:", "    performance:", "  force:", "  neigh:", "  comm:", "  other:", "# Timing histograms",
                "# Nlocal:", "# Nghost:", "# Nswaps:", "# Neighs:", "# Total # of neighbors ="):
        assert key in out, key
    import re
    assert float(re.search(r"# Nghost:\s+([-+.\deE]+) ave", out).group(1)) == ent["nghost"]
    assert float(re.search(r"# Total # of neighbors = ([-+.\deE]+)", out).group(1)) == ent["neigh_total"]
    cons = [float(m) for m in re.findall(r"Conservation:\s+([-+.\deE]+)", out)]
    assert cons[0] == 1.0 and all(abs(c - 1.0) < 5e-3 for c in cons)
    yamls = [f for f in os.listdir(str(tmp_path)) if f.startswith("miniMD-") and f.endswith(".yaml")]
    assert len(yamls) == 1 and "thermodynamic_output:" in open(str(tmp_path / yamls[0])).read()


# ---- --check_exchange (ref/integrate.cpp:112-151) ----------------------------------------------------------------------------
@pytest.mark.gpu
def test_max_move_since_mark_equals_numpy():
    """positions are marked, 19 steps run (no re-neighboring, so no reordering), the device maximum equals the numpy one;
    then one atom is displaced by more than a box length to exercise the reference's single +-prd correction"""
    s = mm().Sim(["-s", "6", "-n", "19", "--half_neigh", "0"])
    s.initial()
    h = s.handle
    h.mark_positions()
    before = h.download()
    s.run_steps(19)
    after = h.download()
    nl = before["nlocal"]
    assert np.array_equal(before["tag"], after["tag"])
    d = np.sqrt((((after["x"][:nl] - before["x"][:nl]) ** 2).sum(1)).max())
    assert 0 < d < 1.0 and abs(h.max_move() - d) <= 1e-14
    prd = h.get_box()[0]
    x = after["x"][:nl].copy()
    x[7, 0] += 1.25 * prd[0]                      # dx > prd: corrected once by -prd (ref :120)
    x[11, 2] -= 0.75 * prd[2]                     # |dz| < prd: taken as it is
    h.upload(x, after["v"], after["type"][:nl], after["tag"])
    dx = x - before["x"][:nl]
    dx[7, 0] -= prd[0]
    assert abs(h.max_move() - np.sqrt((dx ** 2).sum(1).max())) <= 1e-12
    s.close()


@pytest.mark.gpu
def test_check_exchange_flag_is_silent_on_a_healthy_run_and_changes_nothing():
    path = os.path.join(REPO, "minimd_amd", "bin", "miniMD_dp")
    outs = []
    for extra in ([], ["--check_exchange"]):
        r = subprocess.run([path, "-s", "6", "-n", "100", "--half_neigh", "0"] + extra, cwd=os.path.join(REPO, "data"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    assert "Warning: Atoms move further" not in outs[1]          # (the reference's own check misfires here: DESIGN.md §7)
    assert [r_[:4] for r_ in parse_thermo(outs[0])] == [r_[:4] for r_ in parse_thermo(outs[1])]


@pytest.mark.gpu
@pytest.mark.parametrize("deck", ["in.lj.miniMD", "in.eam.miniMD"])
@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_integrator_inside_the_force_kernel_is_bit_identical(prec, deck):
    """fuse=2 (default: finalIntegrate(n)+initialIntegrate(n+1) at the end of the LJ / EAM tile force kernel, positions double
    buffered) against fuse=1 (separate k_final_initial_integrate) and fuse=0 (reference call order): same bits after
    130 steps with 6 re-neighborings, thermo rows included. (EAM: whole rows in all three — only the fused kernel tracks the
    displacement that lets it stop after the core part of the rows, which changes the order of the sums: core_pct 0 here,
    test_eam_rows_in_two_parts_give_the_same_run covers that feature.)"""
    res = []
    for fuse in (2, 1, 0):
        s = mm().Sim(["-i", deck, "-s", "8", "-n", "130", "--half_neigh", "0"], precision=prec)
        s.handle.set_option("fuse", fuse)
        s.handle.set_option("core_pct", 0)
        s.initial(); s.run()
        d = s.handle.download()
        res.append((s.rows(), d["x"][:d["nlocal"]].copy(), d["v"].copy(), d["f"].copy(), d["tag"].copy()))
        s.close()
    for other in res[1:]:
        assert res[0][0] == other[0]
        for a, b in zip(res[0][1:], other[1:]):
            assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_final_integrate_inside_the_last_force_launch(prec):
    """the last step of a run has no next step to fuse initialIntegrate with; its LJ tile force launch still carries finalIntegrate (v += dtf f with the
    force in registers) and stores the forces — no k_final_integrate pass behind it. Same bits as the separate kernels (option fuse 1: ref/integrate.cpp:59-68
    as launches of their own), for a run cut into slices of 1, 7 and 20 steps (every slice ends with such a step; the re-neighboring at step 20 is the last
    step of a slice once, launched behind the build) and in one piece."""
    res = []
    for fuse, cuts in ((1, [47]), (2, [47]), (2, [1, 6, 13, 20, 7])):
        s = mm().Sim(["-s", "12", "-n", "47", "--half_neigh", "0"], precision=prec)
        s.handle.set_option("fuse", fuse)
        s.initial()
        for c in cuts:
            s.run_steps(c)
        d = s.handle.download()
        res.append((d["x"][:d["nlocal"]].copy(), d["v"].copy(), d["f"][:3 * d["nlocal"]].copy() if d["f"].ndim == 1 else d["f"][:d["nlocal"]].copy(), d["tag"].copy()))
        s.close()
    for other in res[1:]:
        for a_, b_ in zip(res[0], other):
            assert np.array_equal(a_, b_)


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["-s", "14", "-n", "130", "--half_neigh", "0"], ["-s", "12", "-n", "90", "--half_neigh", "1"],
                                  ["-i", "in.eam.miniMD", "-s", "8", "-n", "70", "--half_neigh", "0"], ["-s", "6", "-b", "1", "-n", "50", "--half_neigh", "0"]])
def test_build_binning_that_places_the_ghosts_only(args):
    """inside a re-neighboring the owned atoms are in bin order once Atom::sort has run, and the histogram still holds their counts; the build's
    Neighbor::binatoms (ref/neighbor.cpp:215-268) then counts and places the ghosts only and writes the owned part of every bin from the sort's bin starts
    (counter bin_reuses; `-b 1` = every atom in one bin: the long-bin path, which never reuses). The lists are those of binning everything: thermo rows and
    neighbor totals of the oracle (round 4 compared the two binnings bit for bit while both existed)."""
    s = mm().Sim(args)
    s.initial(); s.run()
    reuses, total = s.handle.counter("bin_reuses"), s.handle.neighbor_info()["total"]
    rows = s.rows()
    s.close()
    assert reuses == 0 if "-b" in args else reuses >= 3
    o = Oracle(args)
    o.initial(); o.run()
    assert total == int(o.numneigh().sum())
    rows_close(rows, o.rows(), 1e-9)
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_force_launch_behind_the_build_changes_nothing(prec):
    """option spec (default on): on a re-neighboring step of a one-rank LJ full-list run Force::compute goes onto the stream behind the
    neighbor build, before the build's result words have reached the host, gated on the device by the build's own verdict. Same bits as
    the ordinary launch order (spec 0) after 130 steps with 6 re-neighborings; with ghost estimates that are too small (borders_est 60)
    or no LDS margin for a larger union (spec 1 on a melting lattice) the verdict is "no", the gated launch does nothing and the step
    loop launches again — still the same bits."""
    res = []
    for mode in ("off", "on", "on_borders_overflow", "on_tight_union"):
        s = mm().Sim(["-s", "14", "-n", "130", "--half_neigh", "0"], precision=prec)
        s.handle.set_option("spec", {"off": 0, "on_tight_union": 1}.get(mode, 16))
        if mode == "on_borders_overflow":
            s.handle.set_option("borders_est", 60)
        s.initial(); s.run()
        d = s.handle.download()
        runs, noop = s.handle.counter("spec_runs"), s.handle.counter("spec_fails")
        res.append((s.rows(), d["x"][:d["nlocal"]].copy(), d["v"].copy(), d["f"].copy(), d["tag"].copy()))
        s.close()
        if mode == "off":
            assert runs == 0
        elif mode == "on":
            assert runs >= 4 and noop <= 1, (runs, noop)       # (thermo steps are launched the ordinary way: step 100 is one of the six)
        elif mode == "on_borders_overflow":
            assert runs >= 1 and noop >= 1, (runs, noop)
        else:
            assert runs >= 4 and noop >= 1, (runs, noop)       # the largest union grows while the lattice melts (steps 20...60)
    for other in res[1:]:
        assert res[0][0] == other[0]
        for a, b in zip(res[0][1:], other[1:]):
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_bench_launches_itself_for_two_ranks(tmp_path):
    """`python bench.py --gpus 2` as a PLAIN process (the way the driver starts it): bench.py re-launches itself through
    torch.distributed.run on a free loop-back port; with one visible GPU the two ranks share it and the halos take the
    host-staged transport (the JSON line says so), with two GPUs they use RCCL. One JSON line from rank 0."""
    import torch
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("MMD_BENCH_TRANSPORT", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "20", "--size", "16", "--equil", "20"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 40 and d["warmup"] == 20 and d["scaling"] == "weak" and d["value"] > 0
    assert "32x16x16" in d["config"]["workload"] and d["cpu_baseline"] is None      # (CPU baseline only at N=1)
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["config"]["transport"] == ("rccl" if torch.cuda.device_count() >= 2 else "host") and d["config"]["transport_ranks"] == 2
    # a line whose halos did not travel over RCCL says so unmistakably (it is a debug-transport figure, not a scaling one)
    assert d["valid"] == (d["config"]["transport"] == "rccl") and (d["valid"] or "debug transport" in d["reason"])
    assert len(d["value_windows"]) == 3 and d["value_windows"][0] == d["value"]


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["-s", "12", "-n", "100", "--half_neigh", "0"], ["-s", "12", "-n", "100", "--half_neigh", "1"],
                                  ["-i", "in.eam.miniMD", "-s", "8", "-n", "60", "--half_neigh", "0"]])
def test_rccl_two_gpus_match_one_rank(args, port, tmp_path):
    """the PRODUCTION transport between two real GPUs (skipped on a one-GPU box): RCCL count handshakes and payloads of
    exchange / borders (both swaps of a ghost layer in one group), the per-step halo pair of a 2-wide dimension (two sends to
    and two receives from the same peer in one group, comm.hip mmd_comm_communicate), the reverse halo (half lists), the EAM
    fp halo and the thermo all-reduce. Rows equal the one-rank run to summation order; per-rank counts equal the oracle's."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    cwd = os.path.join(REPO, "data")
    base = sim_rows(args)
    out = str(tmp_path / "mp.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py"), "simrccl", out, "dp"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=cwd)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    assert sum(c[0] for c in res["counts"]) == res["natoms"]
    rows = [tuple(x) for x in res["rows"]]
    assert [r_[0] for r_ in rows] == [b[0] for b in base]
    for a, b in zip(rows, base):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (a, b)
    o = Oracle(args, nprocs=2)
    o.initial(); o.run()
    assert [o.nlocal(0), o.nlocal(1)] == [c[0] for c in res["counts"]]
    assert [o.nghost(0), o.nghost(1)] == [c[1] for c in res["counts"]]
    o.close()


@pytest.mark.gpu
def test_bench_contract_single_gpu():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "40", "--warmup", "20", "--size", "24", "--cpu-steps", "20"],
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["metric"].startswith("Matom-steps/sec") and d["unit"] == "Matom-steps/s" and d["dtype"] == "f64"
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["unit"] == "GB/s"
    cb = d["cpu_baseline"]
    assert cb and cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert "host:" in cb["sample"] and "hardware threads" in cb["sample"]            # which CPU the baseline ran on
    assert d["valid"] is True and d["reason"] is None
    assert len(d["value_windows"]) == 3 and d["value_windows"][0] == d["value"] and min(d["value_windows"]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("inp,half", [("lj", 0), ("lj", 1), ("eam", 0)])
def test_reference_harness_scope0_passes(inp, half):
    """the reference's own validation procedure (ref/run_tests scope 0 -> ref/run_one_test) on the drop-in executable:
    tools/run_one_test.py runs miniMD_dp like `make test` would, cuts the thermo block and applies the PASS rule against
    the published log of that system size"""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "run_one_test.py"), "--scope", "0", "--input", inp, "--halfneigh", str(half)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PASSED" in r.stdout and "Failed" not in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_run_stats_one_rank_has_no_host_syncs_between_rebuilds():
    """mmd_run_stats: a one-rank run blocks the host once per re-neighboring (the build's read-back) and on thermo rows only;
    nothing is sent to other ranks"""
    import minimd_amd
    s = minimd_amd.Sim(["-s", "16", "-n", "100", "--half_neigh", "0"], quiet=True)
    s.initial()
    s.run_steps(19)                      # steps 1..19: no re-neighboring, no thermo row
    st = s.handle.run_stats()
    assert st["host_syncs"] == 0 and st["bytes_sent"] == 0, st
    s.run_steps(41)                      # steps 20..60: three re-neighborings, one read-back each (the first may size its lists twice)
    st = s.handle.run_stats()
    assert 3 <= st["host_syncs"] <= 5 and st["bytes_sent"] == 0, st
    s.close()


def _thermo_rows(stdout):
    rows, on = [], False
    for line in stdout.splitlines():
        if line.startswith("# Timestep T"):
            on = True
            continue
        if line.startswith("# Performance Summary"):
            break
        f = line.split()
        if on and len(f) >= 4 and f[0].isdigit():
            rows.append((int(f[0]), float(f[1]), float(f[2]), float(f[3])))
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("prec,lists", [("dp", ["--half_neigh", "0"]), ("dp", ["--half_neigh", "1", "-gn", "1"]), ("dp", ["--half_neigh", "1", "-gn", "0"]),
                                        ("sp", ["--half_neigh", "0"])])
def test_reference_program_runs_on_the_plugin(prec, lists):
    """SURVEY §8(b)'s in-process plugin point exercised for real: oracle/_ref/ref_hip_<prec> is the UNMODIFIED reference program
    (its Atom, Neighbor::build, Comm, Thermo and Integrate::run, compiled from /root/reference by oracle/Makefile `ref_hip`) with
    ONE substitution — where ref/ljs.cpp:285 constructs ForceLJ it constructs ForceHIP (tests/integration/force_hip.h), linked
    against libmmd_hip_<prec>.so. The reference's Integrate::run calls force->compute through the vtable (ref/integrate.cpp:183)
    on the lists ITS Neighbor built; with half lists + ghost newton ITS Comm::reverse_communicate folds the ghost forces the
    plugin hands back. Rows must equal the published 4k.lj log (the reference's own output for this system)."""
    exe = os.path.join(REPO, "oracle", "_ref", "ref_hip_" + prec)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_hip_%s is built where the reference tree exists (make -C oracle ref_hip)" % prec)
    r = subprocess.run([exe, "-i", "in.lj.miniMD", "-s", "10", "-n", "200", "-t", "1"] + lists, cwd=os.path.join(REPO, "data"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = _thermo_rows(r.stdout)
    ref = [tuple(x) for x in PUBLISHED["4k.lj"]["rows"] if x[0] <= 200]
    assert [x[0] for x in rows] == [0, 100, 200], r.stdout[-2000:]
    tol = 2e-6 if prec == "dp" else 2e-4
    for a, b in zip(rows, ref):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= tol * max(1.0, abs(b[k])), (a, b)
    assert ref_pass_rule(ref, rows, 4000, 8 if prec == "dp" else 4)[0]
    assert "ForceHIP:" not in r.stderr                   # (the plugin reports C-ABI errors there)


@pytest.mark.gpu
@pytest.mark.parametrize("half", [0, 1])
def test_uploaded_list_older_than_the_positions(half):
    """a list handed over by mmd_neighbor_upload need not be fresh: the oracle's list of step 20 with its positions of step 27 (atoms have
    drifted: a partner may lie outside the candidate runs of its tile, then k_rows_to_tiles leaves the list to the row kernels). Whichever
    kernels serve it, forces, energy and virial are those of the oracle on the same list and positions; then only the positions move
    (mmd_atom_upload_x) and the same list keeps serving them."""
    o = Oracle(["-s", 8, "-n", 27, "--half_neigh", half, "-gn", 0])
    o.initial(); o.run()
    o.lib.orc_force_compute(o.w, 1)
    h = handle_from_oracle(o)
    h.force_lj_setup(*o.lj_tables())
    h.neighbor_upload(o.neighbors(), o.numneigh())
    for tiles in (1, 0):
        h.set_option("tiles", tiles)
        eng, vir = h.force_compute(1)
        f = h.download(halfneigh=bool(half))["f"]
        fo = o.f(with_ghosts=bool(half))
        assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
        assert abs(eng - o.eng_vdwl()) <= 1e-11 * abs(o.eng_vdwl()) and abs(vir - o.virial()) <= 1e-10 * max(1.0, abs(o.virial()))
    h.set_option("tiles", 1)
    if not half:
        # positions only: every owned atom nudged, ghosts follow their owners' shift is NOT required by the API (the caller's Comm does that):
        # here the whole array is shifted rigidly, which leaves every pair distance — and so every force — unchanged
        x = o.x() + np.array([0.01, -0.02, 0.005])
        h.upload_x(x)
        np.testing.assert_array_equal(h.download()["x"], x)
        eng2, vir2 = h.force_compute(1)
        f2 = h.download()["f"]
        assert np.abs(f2 - o.f()).max() <= 1e-10 * np.abs(o.f()).max() and abs(eng2 - eng) <= 1e-11 * abs(eng)
        with pytest.raises(Exception):
            h.upload_x(x[:-1])                               # another atom count is an error, not a silent truncation
    h.close(); o.close()


@pytest.mark.gpu
def test_eam_with_uploaded_ghosts_needs_the_callers_fp_halo():
    """ghost atoms that came through mmd_atom_upload (a reference Atom incl. its ghosts) have no send lists in this handle: ForceEAM::compute
    fails loudly unless the caller's ForceEAM::communicate is installed (mmd_force_eam_set_fp_halo, what ForceEAMHIP does with the reference's
    Comm); with it — here the oracle's fp of the ghosts — forces, fp and energy are the oracle's."""
    ntypes = 1
    o = Oracle(["-i", "in.eam.miniMD", "-s", 5, "-n", 20, "--half_neigh", 0, "--ntypes", ntypes])
    o.initial(); o.run()
    h = handle_from_oracle(o)                         # atoms AND ghosts uploaded; no comm_setup / borders on this handle
    from minimd_amd import api
    h.force_eam_setup(ntypes, api.eam_tables_from_file(os.path.join(REPO, "data", "Cu_u6.eam"), ntypes))
    h.neighbor_upload(o.neighbors(), o.numneigh())
    with pytest.raises(Exception, match="fp_halo|borders"):
        h.force_compute(1)
    fpo = o.eam_fp()
    seen = {}

    def halo(fp, nlocal, nghost):
        seen["owned_err"] = float(np.abs(fp[:nlocal] - fpo[:nlocal]).max())
        fp[nlocal:] = fpo[nlocal:nlocal + nghost]
    h.set_fp_halo(halo)
    for tiles in (1, 0):
        h.set_option("tiles", tiles)
        eng, vir = h.force_compute(1)
        assert seen["owned_err"] <= 1e-12 * np.abs(fpo).max()
        f, fo = h.download()["f"], o.f()
        assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
        assert abs(eng - o.eng_vdwl()) <= 1e-12 * abs(o.eng_vdwl())
    h.set_fp_halo(None)
    with pytest.raises(Exception):
        h.force_compute(1)
    h.close(); o.close()


def _run_ref_program(exe_name, deck, lists, nsteps=200, size=10):
    exe = os.path.join(REPO, "oracle", "_ref", exe_name)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/%s is built where the reference tree exists (make -C oracle ref_hip ref_hipnb)" % exe_name)
    r = subprocess.run([exe, "-i", deck, "-s", str(size), "-n", str(nsteps), "-t", "1"] + lists, cwd=os.path.join(REPO, "data"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "HIP:" not in r.stderr, r.stderr[-2000:]           # (the plugins report C-ABI errors there: ForceHIP: / ForceEAMHIP: / NeighborHIP:)
    return _thermo_rows(r.stdout), r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("lists", [["--half_neigh", "0"], ["--half_neigh", "1"]])
def test_reference_program_runs_on_the_plugin_eam(lists):
    """the EAM plug point (ref/ljs.cpp:274-283): oracle/_ref/ref_hip_dp constructs ForceEAMHIP (tests/integration/force_eam_hip.h, a
    ForceEAM whose setup() is the reference's — Cu_u6.eam, its own spline tables, handed over through mmd_force_eam_setup — and whose
    compute() is the library's two sweeps with the REFERENCE's ForceEAM::communicate called back between them for the fp halo, on
    ITS Comm's send lists). Rows must equal the published 4k.eam log (the reference's own output: half lists, and full lists agree)."""
    rows, out = _run_ref_program("ref_hip_dp", "in.eam.miniMD", lists)
    ref = [tuple(x) for x in PUBLISHED["4k.eam"]["rows"] if x[0] <= 200]
    assert [x[0] for x in rows] == [0, 100, 200], out[-2000:]
    for a, b in zip(rows, ref):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 2e-6 * max(1.0, abs(b[k])), (a, b)
    assert ref_pass_rule(ref, rows, 4000, 8, eam=True)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("deck,lists", [("in.lj.miniMD", ["--half_neigh", "0"]), ("in.lj.miniMD", ["--half_neigh", "1", "-gn", "0"]),
                                        ("in.lj.miniMD", ["--half_neigh", "1", "-gn", "1"]), ("in.eam.miniMD", ["--half_neigh", "0"]),
                                        ("in.eam.miniMD", ["--half_neigh", "1"])])
def test_reference_program_runs_on_the_neighbor_plugin(deck, lists):
    """the Neighbor plug point (ref/neighbor.h:59): oracle/_ref/ref_hipnb_dp is the reference program — its own ForceLJ / ForceEAM, Atom,
    Comm, Integrate — with the BODY of Neighbor::build replaced by tests/integration/neighbor_hip.cpp (mmd_neighbor_build +
    mmd_neighbor_download into the reference's own neighbors[] / numneigh[]). The reference's force loops run on rows the device
    built: full lists, half lists (`j > i` re-homed), half lists with ghost newton (re-derived with the reference's half-stencil rule)."""
    rows, out = _run_ref_program("ref_hipnb_dp", deck, lists)
    name = "4k.eam" if "eam" in deck else "4k.lj"
    ref = [tuple(x) for x in PUBLISHED[name]["rows"] if x[0] <= 200]
    assert [x[0] for x in rows] == [0, 100, 200], out[-2000:]
    for a, b in zip(rows, ref):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 2e-6 * max(1.0, abs(b[k])), (a, b)
    assert ref_pass_rule(ref, rows, 4000, 8, eam="eam" in deck)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("half", [0, 1])
def test_uploaded_lists_run_at_the_speed_of_device_lists(half):
    """what a reference build that adopts the plugin gets per Force::compute: the reference's rows (here the oracle's, -s 24, 55 k atoms,
    thermalised) handed over by mmd_neighbor_upload are converted into the tile form once and then served by the tile kernels — GPU time
    per Force::compute within 1.3x of a list the device built itself (round 3: the row kernels, 3x slower), same forces."""
    o = Oracle(["-s", 24, "-n", 20, "--half_neigh", half, "-gn", 1])
    o.initial(); o.run()
    h = handle_from_oracle(o)
    h.force_lj_setup(*o.lj_tables())
    h.neighbor_upload(o.neighbors(), o.numneigh())
    assert h.counter("tiles_ready") == 1
    h.force_compute(0)
    f_up = h.download(halfneigh=bool(half))["f"][: o.nlocal()].copy()
    t_up = min(h.profile_kernel(0, 50) for _ in range(3))
    st_up = h.neighbor_tile_stats()
    h.set_option("tiles", 0)
    t_rows = min(h.profile_kernel(0, 20) for _ in range(2))
    h.set_option("tiles", 1)
    h.neighbor_build()
    h.force_compute(0)
    f_dev = h.download(halfneigh=bool(half))["f"][: o.nlocal()]
    t_dev = min(h.profile_kernel(0, 50) for _ in range(3))
    st_dev = h.neighbor_tile_stats()
    if not half:
        assert np.abs(f_up - f_dev).max() <= 1e-12 * np.abs(f_dev).max()
        assert st_up["sum_candidates"] == st_dev["sum_candidates"]          # same tiles, same unions
    print("uploaded tiles %.4f ms, uploaded rows %.4f ms, device tiles %.4f ms" % (t_up, t_rows, t_dev))
    assert t_up <= 1.3 * t_dev, (t_up, t_dev, t_rows)
    h.close(); o.close()


@pytest.mark.gpu
def test_exchange_all_moves_atoms_two_subdomains_like_the_oracle(port, tmp_path):
    """Comm::exchange_all (ref/comm.cpp:599-689, `--safe_exchange`): 4 ranks in a 1x1x4 grid of sub-domains thinner than the
    cutoff (need = 2); two fifths of every rank's atoms are pushed TWO sub-domains up / down before the exchange. Every rank must
    end up owning exactly the atoms, in exactly the order, the oracle's virtual ranks own after the same displacement; the plain
    exchange (offers leavers to the direct neighbours only) loses them."""
    import ctypes
    args = ["-nx", "3", "-ny", "3", "-nz", "6", "--sort", "0", "-n", "20", "--half_neigh", "0"]
    out = str(tmp_path / "ex.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py"), "exchall", out, "dp"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    assert res[0]["procgrid"] == [1, 1, 4] and res[0]["need"][2] == 2 and res[0]["dim"] == 2
    o = Oracle(args + ["--safe_exchange"], nprocs=4)
    o.initial()
    for p_ in range(4):
        nl = o.nlocal(p_)
        xv = np.ctypeslib.as_array(o.lib.orc_x(o.w, p_), shape=(3 * nl,)).reshape(nl, 3)       # (a view: written in place)
        box = o.box(p_)
        w = box["zhi"] - box["zlo"] if isinstance(box, dict) else box[8] - box[7]
        idx = np.arange(nl)
        xv[idx % 5 == 0, 2] += 2 * w
        xv[idx % 5 == 1, 2] -= 2 * w
    o.lib.orc_exchange(o.w)
    total = 0
    for p_ in range(4):
        nl = o.nlocal(p_)
        total += nl
        assert len(res[p_]["x"]) == nl, (p_, len(res[p_]["x"]), nl)
        assert np.array_equal(np.array(res[p_]["x"]), o.x(p_)[:nl])
        # (velocities: the centre-of-mass removal of create_velocity sums over the ranks in a different order — last digits only)
        assert np.allclose(np.array(res[p_]["v"]), o.v(p_), rtol=1e-12, atol=1e-13)
    assert total == o.natoms()
    o.close()
    # the plain exchange on the same displaced system loses the far movers (that is what the option is for)
    env["MMD_TEST_SAFE"] = "0"
    out2 = str(tmp_path / "ex0.json")
    cmd[cmd.index(out)] = out2
    cmd[cmd.index("--master-port") + 1] = str(free_port())           # (a port probed now: port + 1 may be somebody's)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert sum(len(q["x"]) for q in json.load(open(out2))) < total


@pytest.mark.gpu
def test_pencil_tiles_are_full_wavefronts():
    """the production tiles are 64-atom pieces of an x-sorted row of blocks: at LJ liquid density more than 60 of the 64
    lanes of a tile hold an atom (one-block tiles: 57), every owned atom sits in exactly one tile, and the list is the oracle's"""
    import minimd_amd
    s = minimd_amd.Sim(["-s", "24", "-n", "40", "--half_neigh", "0"], quiet=True)
    s.initial(); s.run()
    st = s.handle.neighbor_tile_stats()
    nl, ng, _ = s.handle.counts()
    assert st["sum_atoms"] >= nl and st["sum_atoms"] <= nl + ng
    assert st["sum_atoms"] / st["tiles"] >= 59.0, st          # (9 tiles per 564-atom pencil at this size; 63 at -s 80)
    hc, hr = s.handle.neighbor_tile_histogram(64, 16)
    assert sum(hc) == st["tiles"] == sum(hr)
    o = Oracle(["-s", "24", "-n", "40", "--half_neigh", "0"])
    o.initial(); o.run()
    assert s.handle.neighbor_info()["total"] == int(o.numneigh().sum())
    nb, nn = s.handle.neighbor_download()
    assert int(nn.sum()) == int(o.numneigh().sum())
    s.close(); o.close()


@pytest.mark.gpu
def test_undersized_ghost_arrays_on_several_ranks_fall_back_together(port, tmp_path):
    """the device-resident borders of several ranks with ghost arrays sized for 60 % of the previous count: the overflow flag is
    raised on the device, max-reduced over the ranks, and EVERY rank redoes the borders swap by swap (a mix of paths would dead-lock
    or mis-match messages). Rows and per-rank counts must be those of the normal run."""
    args = ["-s", "8", "-n", "60", "--half_neigh", "0"]
    base = sim_rows(args)
    out = str(tmp_path / "mp.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MMD_TEST_OPTIONS="borders_est=60")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py"), "sim", out, "dp"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    rows = [tuple(x) for x in res["rows"]]
    assert [r_[0] for r_ in rows] == [b[0] for b in base]
    for a, b in zip(rows, base):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (a, b)
    o = Oracle(args, nprocs=2)
    o.initial(); o.run()
    assert [o.nlocal(0), o.nlocal(1)] == [c[0] for c in res["counts"]]
    assert [o.nghost(0), o.nghost(1)] == [c[1] for c in res["counts"]]
    o.close()
    # (every re-neighboring took the fall-back: the swap-by-swap path waits for its counts)
    assert all(st["host_syncs"] > 3 * 6 for st in res["stats"]), res["stats"]


def _two_rank_run(args, port, tmp_path, options="", nprocs=2):
    out = str(tmp_path / "mp.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MMD_TEST_OPTIONS=options)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nprocs), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py"), "sim", out, "dp"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.load(open(out))


@pytest.mark.gpu
@pytest.mark.parametrize("nprocs,args", [(2, ["-s", "12", "-n", "100", "--half_neigh", "0"]), (8, ["-s", "14", "-n", "60", "--half_neigh", "0"]),
                                         (3, ["-i", "in.eam.miniMD", "-s", "10", "-n", "60", "--half_neigh", "0"]), (4, ["-s", "12", "-n", "60", "--half_neigh", "1"])])
def test_direct_halo_on_several_ranks_equals_the_swap_by_swap_halo(nprocs, args, port, tmp_path):
    """2 / 8 / 3 / 4 ranks (2x1x1, 2x2x2, 3x1x1, 2x2x1: partners that are the same rank twice, diagonal partners, dimensions that wrap onto the rank
    itself): the per-step halo as one exchange with the distinct partners (direct_halo, default) against the three forwarding rounds — the same ghosts in
    the same slots, so the same rows (half lists: to the order of the atomics) and the same atoms per rank."""
    a = _two_rank_run(args, port, tmp_path, options="direct_halo=1", nprocs=nprocs)
    b = _two_rank_run(args, free_port(), tmp_path, options="direct_halo=0", nprocs=nprocs)
    assert a["counts"] == b["counts"]
    if "1" == args[-1]:
        rows_close([tuple(r) for r in a["rows"]], [tuple(r) for r in b["rows"]], 1e-10)
    else:
        assert a["rows"] == b["rows"]


@pytest.mark.gpu
@pytest.mark.parametrize("nprocs,cap", [(2, 8), (4, 2)])
def test_overflowing_exchange_messages_fall_back_together(nprocs, cap, port, tmp_path):
    """the handshake-free Comm::exchange with messages far too small (exchange_cap records): the sender notices before any atom has
    been moved, the flag is max-reduced over the ranks, every rank skips the mutating kernels of that dimension and the later ones, and
    the count-handshake path finishes them (ref/comm.cpp:364-597). Nobody is lost, rows and per-rank counts are those of the oracle's
    virtual ranks, and the run does not abort (round 3: 'more atoms migrated than the fixed-size messages hold')."""
    args = ["-s", "8", "-n", "100", "--half_neigh", "0"]
    base = sim_rows(args)
    res = _two_rank_run(args, port, tmp_path, "exchange_cap=%d" % cap, nprocs)
    rows = [tuple(x) for x in res["rows"]]
    assert [r_[0] for r_ in rows] == [b[0] for b in base]
    for a, b in zip(rows, base):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (a, b)
    o = Oracle(args, nprocs=nprocs)
    o.initial(); o.run()
    assert [o.nlocal(r) for r in range(nprocs)] == [c[0] for c in res["counts"]]
    assert [o.nghost(r) for r in range(nprocs)] == [c[1] for c in res["counts"]]
    o.close()
    assert sum(c[0] for c in res["counts"]) == res["natoms"]
    # every rank saw the same number of overflows (the decision is collective) and at least one
    ov = [st["exchange_overflows"] for st in res["stats"]]
    assert len(set(ov)) == 1 and ov[0] >= 1, res["stats"]


@pytest.mark.gpu
def test_first_exchange_of_a_run_at_s80_on_two_ranks_does_not_overflow(port, tmp_path):
    """two ranks of -s 80 (1 M atoms each): FCC atoms are created exactly on the sub-domain faces, so at the first re-neighboring about
    half a lattice plane (~6400 atoms) leaves through each face — more than the 4096 records a fixed-size message sized from the set-up
    exchange (nobody moves) would hold. The first exchange of a run therefore takes the count-handshake path and sizes the messages
    of the second; no overflow, nobody lost, rows of the one-rank run."""
    args = ["-s", "80", "-n", "40", "--half_neigh", "0"]
    base = sim_rows(args)
    res = _two_rank_run(args, port, tmp_path)
    assert sum(c[0] for c in res["counts"]) == res["natoms"] == 2048000
    for a, b in zip(res["rows"], base):
        assert a[0] == b[0]
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (a, b)
    for st in res["stats"]:
        assert st["exchange_overflows"] == 0 and st["exchange_fast"] == 1 and st["borders_fast"] + st["borders_direct"] >= 1, res["stats"]


@pytest.mark.gpu
@pytest.mark.parametrize("style,half", [("eam", 0), ("eam", 1), ("lj", 0), ("lj", 1)])
def test_unions_beyond_one_staging_round(style, half, tmp_path):
    """The tile kernels stage the first 512 candidates of a tile's union in one batched round and the rest in a plain loop;
    at the EAM deck's density (unions of 300-480 atoms) that loop never runs, at the LJ deck's for 2 % of the tiles. Lattices compressed to 1.4x
    (copper) / 1.25x (LJ) the density have unions of 530-700 atoms: the runs must still follow the oracle (ForceEAM::compute_fullneigh /
    compute_halfneigh, ref/force_eam.cpp:94-449; ForceLJ::compute_fullneigh / compute_halfneigh, ref/force_lj.cpp:271-449)."""
    deck = open(os.path.join(REPO, "data", "in.%s.miniMD" % style)).read()
    old_rho, new_rho = ("0.07041125", "0.0985758 ") if style == "eam" else ("0.8442", "1.0552")
    assert old_rho in deck and "100            thermo" in deck
    deck = deck.replace(old_rho, new_rho).replace("100            thermo", "10             thermo")
    if style == "lj" and half:                  # (half lists with ghost newton keep the upper half shell only: a wider skin as well)
        assert "2.5 0.30" in deck
        deck = deck.replace("2.5 0.30", "2.5 0.80")
    p = str(tmp_path / ("in.%s_dense.miniMD" % style))
    open(p, "w").write(deck)
    args = ["-i", p, "-s", "8", "-n", "40", "--half_neigh", str(half)]
    import minimd_amd
    s = minimd_amd.Sim(args, quiet=True)
    s.initial(); s.run()
    st = s.handle.neighbor_tile_stats()
    assert st["max_candidates"] > 512, st                  # (otherwise this test does not reach the loop it is for)
    rows = s.rows()
    s.close()
    o = Oracle(args)
    o.initial(); o.run()
    ref = o.rows()
    o.close()
    assert len(rows) == len(ref) == 5
    rows_close(rows, ref, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("half", [0, 1])
def test_eam_single_precision_run_follows_the_oracle(half):
    """ForceEAM in MMD_float (ref -DPRECISION=1): the SP library's tile sweeps (float knot tables, ds_read_b32 position records) against the
    oracle's SP run of the same deck — thermo rows of 40 steps within float summation noise (5e-4 relative, the sums run in another order)."""
    args = ["-i", "in.eam.miniMD", "-s", "6", "-n", "40", "--half_neigh", str(half)]
    rows = sim_rows(args, precision="sp")
    o = Oracle(args, precision="sp")
    o.initial(); o.run()
    ref = o.rows()
    o.close()
    assert len(rows) == len(ref) and len(rows) >= 2
    rows_close(rows, ref, 5e-4)
    dp = sim_rows(args, precision="dp")
    for a, b in zip(rows, dp):
        assert abs(a[2] - b[2]) <= 5e-4 * abs(b[2]), (a, b)          # and next to the DP library's run
