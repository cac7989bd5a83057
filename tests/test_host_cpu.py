"""CPU-only tests of the product's host side (no GPU, no compute calls): the C-ABI library loads and exports
every symbol include/mmd.h declares; host logic (input deck, lattice, EAM tables, Comm::setup and
Neighbor::setup geometry) is bit-identical to the oracle's; the product refuses to run without a GPU."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import minimd_amd
from minimd_amd import api
from oracle_lib import Oracle

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(REPO, "data")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not (os.path.exists(api.lib_path("dp")) and os.path.exists(api.lib_path("sp"))):
        api.build()


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_library_exports_every_declared_symbol(prec):
    hdr = open(os.path.join(REPO, "include", "mmd.h")).read()
    declared = set(re.findall(r"\b(mmd_[a-z0-9_]+)\s*\(", hdr)) - {"mmd_sendrecv_fn", "mmd_allreduce_fn", "mmd_thermo_fn"}
    L = api.load_library(prec)
    import ctypes
    raw = ctypes.CDLL(api.lib_path(prec))
    missing = [s for s in sorted(declared) if not hasattr(raw, s)]
    assert not missing, missing
    assert L.mmd_float_size() == (8 if prec == "dp" else 4)
    assert len(declared) >= 55


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(minimd_amd.MMDError, match="no HIP device"):
        minimd_amd.Handle()
    with pytest.raises(minimd_amd.MMDError):
        minimd_amd.Sim(["-s", 4, "-n", 1])


def test_product_never_references_the_oracle():
    """the oracle is test infrastructure: nothing under minimd_amd/ or include/ may mention it"""
    bad = []
    for root in ("minimd_amd", "include"):
        for dp, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle_lib|libmmd_oracle|mmd_oracle\.|orc_[a-z]+\(", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


@pytest.mark.parametrize("prec,deck", [("dp", "in.lj.miniMD"), ("dp", "in.eam.miniMD"), ("sp", "in.lj.miniMD")])
def test_input_deck_equals_oracle(prec, deck):
    inp = api.input_read(os.path.join(DATA, deck), prec)
    o = Oracle(["-i", deck, "-s", 4, "-n", 1, "--half_neigh", 0], precision=prec)
    assert (inp.nx, inp.ny, inp.nz) == (32, 32, 32)
    assert inp.dt == o.param("dt") and inp.rho == o.param("rho") and inp.t_request == o.param("t_request")
    assert inp.neigh_cut == o.param("cutneigh") and inp.neigh_every == int(o.param("neigh_every"))
    assert inp.thermo_nstat == int(o.param("nstat")) and inp.units == int(o.param("units")) and inp.forcetype == int(o.param("forcetype"))
    if deck == "in.lj.miniMD":
        assert inp.force_cut == o.param("cutforce")
    o.close()


def test_input_errors_are_reported_not_fatal(tmp_path):
    with pytest.raises(minimd_amd.MMDError, match="Cannot open"):
        api.input_read(str(tmp_path / "nope"))
    p = tmp_path / "bad.deck"
    p.write_text("t\n\nfurlongs\nnone\nlj\n1 1\n4 4 4\n1\n0.005\n1.44\n0.8\n20\n2.5 0.3\n100\n")
    with pytest.raises(minimd_amd.MMDError, match="Unknown units"):
        api.input_read(str(p))
    p.write_text("short\n")
    with pytest.raises(minimd_amd.MMDError):
        api.input_read(str(p))


@pytest.mark.parametrize("prec", ["dp", "sp"])
@pytest.mark.parametrize("dims", [(4, 4, 4), (5, 3, 6)])
def test_create_atoms_equals_oracle(prec, dims):
    o = Oracle(["-nx", dims[0], "-ny", dims[1], "-nz", dims[2], "-n", 1, "--half_neigh", 0], precision=prec)
    prd = api.create_box(*dims, o.param("rho"), prec)
    box = o.box()
    np.testing.assert_array_equal(prd, np.array(box[:3], dtype=prd.dtype))
    x, v, t, tag = api.create_atoms(*dims, o.param("rho"), [0, 0, 0], prd, o.ntypes(), prec)
    assert len(x) == o.natoms() == 4 * dims[0] * dims[1] * dims[2]
    np.testing.assert_array_equal(x, o.x()[: o.nlocal()])
    np.testing.assert_array_equal(t, o.type()[: o.nlocal()])
    np.testing.assert_array_equal(tag, o.tag())
    # raw velocities -> the oracle's after centre-of-mass removal and rescaling (same order of operations)
    vv = v.astype(np.float64)
    vtot = np.array([sum(vv[:, d].tolist()) for d in range(3)]) / len(x)       # sequential sums like the reference
    vv = (v.astype(np.float64) - vtot).astype(v.dtype)
    np.testing.assert_allclose(vv / np.abs(vv).max(), o.v() / np.abs(o.v()).max(), rtol=0, atol=2e-6 if prec == "sp" else 1e-13)
    o.close()


def test_create_atoms_subboxes_partition_the_lattice():
    """every rank's sub-box creates its own atoms; together they are exactly the single-rank set"""
    dims, rho = (6, 4, 5), 0.8442
    prd = api.create_box(*dims, rho)
    x_all, _, _, tag_all = api.create_atoms(*dims, rho, [0, 0, 0], prd)
    tags = []
    for ix in range(2):
        for iz in range(3):
            lo = [ix * prd[0] / 2, 0.0, iz * prd[2] / 3]
            hi = [(ix + 1) * prd[0] / 2, prd[1], (iz + 1) * prd[2] / 3]
            x, v, t, tag = api.create_atoms(*dims, rho, lo, hi)
            assert np.all((x >= np.array(lo)) & (x < np.array(hi)))
            tags.append(tag)
    tags = np.concatenate(tags)
    assert len(tags) == len(tag_all) and set(tags.tolist()) == set(tag_all.tolist())


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_eam_tables_equal_oracle(prec):
    t = api.eam_tables_from_file(os.path.join(DATA, "Cu_u6.eam"), 4, prec)
    o = Oracle(["-i", "in.eam.miniMD", "-s", 4, "-n", 1, "--half_neigh", 0], precision=prec)
    ot = o.eam_tables()
    for k in ("nr", "nrho", "nr_tot", "nrho_tot"):
        assert t[k] == ot[k], k
    assert t["rdr"] == ot["rdr"] and t["rdrho"] == ot["rdrho"] and t["mass"] == ot["mass"]
    for k in ("rhor_spline", "z2r_spline", "frho_spline", "cutforcesq"):
        np.testing.assert_array_equal(t[k], ot[k])
    o.close()


def test_eam_missing_file_is_an_error():
    with pytest.raises(minimd_amd.MMDError, match="Can't open EAM"):
        api.eam_tables_from_file("/nonexistent/Cu_u6.eam")


@pytest.mark.parametrize("nprocs", [1, 2, 3, 4, 6, 8, 12])
@pytest.mark.parametrize("dims", [(8, 8, 8), (16, 8, 8), (6, 10, 14)])
def test_comm_setup_geometry_equals_oracle(nprocs, dims):
    """Comm::setup (ref/comm.cpp:60-272): grid, neighbors, sub-boxes, slabs, PBC flags — every rank"""
    o = Oracle(["-nx", dims[0], "-ny", dims[1], "-nz", dims[2], "-n", 1, "--half_neigh", 0], nprocs=nprocs)
    cut = o.param("cutneigh")
    for me in range(nprocs):
        h = minimd_amd.Handle(device=-2)
        box = o.box(me)
        h.set_box(box[:3])
        h.comm_setup(cut, me, nprocs)
        info = h.comm_info()
        assert list(info["procgrid"]) == o.procgrid()
        prd, lo, hi = h.get_box()
        assert [lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]] == box[3:]
        assert info["nswap"] == o.nswap(me)
        for s in range(info["nswap"]):
            a, b = h.swap_info(s), o.swap_info(me, s)
            for k in ("slablo", "slabhi", "pbc_any", "pbc", "sendproc", "recvproc"):
                assert a[k] == b[k], (me, s, k, a, b)
        h.close()
    o.close()


@pytest.mark.parametrize("args", [["-s", 4], ["-s", 10], ["-s", 32], ["-s", 80], ["-nx", 7, "-ny", 5, "-nz", 6], ["-s", 8, "-b", 3],
                                  ["-i", "in.eam.miniMD", "-s", 8]])
def test_neighbor_geometry_equals_oracle(args):
    """Neighbor::setup bin grid (ref/neighbor.cpp:349-391); plus: the block stencil covers the cutoff"""
    o = Oracle(args + ["-n", 1, "--half_neigh", 0])
    import ctypes as C
    mb, lo, ns = (C.c_int * 3)(), (C.c_int * 3)(), C.c_int()
    o.lib.orc_bin_geometry(o.w, 0, mb, lo, C.byref(ns))
    h = minimd_amd.Handle(device=-2)
    box = o.box()
    h.set_box(box[:3])
    h.comm_setup(o.param("cutneigh"), 0, 1)
    h.neighbor_setup(o.nbins(), o.param("cutneigh"), 0, 1, 4)
    g = h.neighbor_geometry()
    assert list(g["mbin"]) == list(mb) and list(g["mbinlo"]) == list(lo)
    for d in range(3):
        binsize = box[d] / o.nbins()[d]
        # a block is 2 bins wide: reach blocks on each side must span >= cutneigh beyond any atom of the block
        assert g["reach"][d] * 2 * binsize >= o.param("cutneigh") - 1e-12 or (2 * g["reach"][d] - 1) * binsize >= o.param("cutneigh")
        assert (2 * g["reach"][d] - 1) * binsize + binsize >= o.param("cutneigh")
    h.close(); o.close()


@pytest.mark.skipif(not os.path.isdir("/root/reference/ref"), reason="needs the reference headers (build container only)")
def test_documented_force_plugin_compiles_against_the_reference_headers():
    """tests/integration/force_hip.h (class ForceHIP : public Force) against ref/force.h:40-69, both precisions"""
    import subprocess
    r = subprocess.run(["make", "-s", "-C", os.path.join(REPO, "tests", "integration"), "check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    for fn in ("tests/integration/force_hip.h", "tests/integration/force_eam_hip.h", "tests/integration/neighbor_hip.cpp"):
        assert fn in text and os.path.exists(os.path.join(REPO, fn))


# ---- the user-facing validation harness (ref/run_one_test, ref/run_tests restated as tools/run_one_test.py) -------------------
def test_harness_pass_rule_and_a_run_of_the_cpu_oracle():
    """tools/run_one_test.py applies the reference's statistical PASS rule (ref/run_one_test:121-138). Here (no GPU) it judges the
    CPU oracle's executable — same CLI and stdout grammar as the drop-in — against the published 4k.lj log, and must reject rows
    that drift by more than the rule allows."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("run_one_test", os.path.join(REPO, "tools", "run_one_test.py"))
    rot = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rot)
    from oracle_lib import ref_pass_rule
    gold = json.load(open(os.path.join(REPO, "tests", "golden", "reference_output.json")))["4k.lj"]
    rows = [tuple(r) for r in gold["rows"]]
    assert rot.pass_rule(rows, rows, gold["natoms"], 8, False)[0]
    bad = [(s, t * 1.05, u * 1.05, p) for s, t, u, p in rows]          # (one whole column may miss: the rule sums the three)
    assert not rot.pass_rule(rows, bad, gold["natoms"], 8, False)[0]
    assert rot.pass_rule(rows, bad, gold["natoms"], 8, False)[0] == ref_pass_rule(rows, bad, gold["natoms"], 8)[0]
    nsteps, threads, runs = rot.scope_runs(1)
    assert (nsteps, threads) == (1000, 1) and runs == [(1, 10), (3, 10), (8, 10)]          # ref/run_tests:62-66,116-145
    exe = os.path.join(REPO, "oracle", "mmd_oracle_dp")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "run_one_test.py"), exe, "1", "1", "10", "100", "0", "0", "lj"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PASSED" in r.stdout and "Testfile: tests/reference_output/4k.lj" in r.stdout, r.stdout + r.stderr


def test_register_budgets_of_the_hot_kernels():
    """Occupancy on gfx950 goes in steps of 8 VGPRs (512 // VGPRs wavefronts per SIMD): the production instantiations must stay inside the budget they were tuned
    for — the fused LJ tile kernel at 96 (5 wavefronts; 100 cost 2.4 % in round 5), the neighbor build at <= 128 (its 9.9 KB of LDS allow 4 per SIMD anyway).
    Read from the built objects' AMDGPU metadata (tools/kernel_regs.py); no GPU needed."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_regs", os.path.join(REPO, "tools", "kernel_regs.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    ks = kr.kernels("dp")
    assert len(ks) > 100, "kernel metadata not found in minimd_amd/build/dp (run __graft_entry__.build())"

    def one(prefix):
        m = [k for k in ks if k.startswith(prefix)]
        assert len(m) == 1, (prefix, m)
        return ks[m[0]]
    for fuse in (0, 1, 2):        # k_lj_full_tile<0, FUSE>: force only / fused integrator / fused finalIntegrate
        k = one("_Z14k_lj_full_tileILi0ELi%dEE" % fuse)
        assert k["vgpr"] <= 96 and k["agpr"] == 0, k
    b = one("_Z12k_build_rowsILi0ELi0EE")
    assert b["vgpr"] <= 128 and b["agpr"] == 0 and b["lds"] <= 10240, b      # (agpr == 0: the MFMA accumulators are read by VALU instructions, -amdgpu-mfma-vgpr-form)


def test_error_model_of_the_mfma_pretest():
    """The neighbor build's MFMA pre-test (minimd_amd/csrc/neighbor.hip, comment in front of k_build_rows) trusts the sign of d = |b|^2 - 2 a.b - (cutneigh^2 - |a|^2), evaluated on
    hi/lo half pairs, iff |d| >= E_i = 40 x 2^-24 x T_i. 24 of the 40 units are the allowance for the hardware's fp32 accumulation (measured on the GPU: tools/probes/mfma_f16_probe);
    the other 16 must cover everything that happens BEFORE the matrix core: float rounding of the local coordinates, the fp32 |b|^2 chain, the threshold through float, the
    hi/lo splits, the dropped a_lo . m_lo. That part is arithmetic and is checked here in numpy, with the kernel's operations restated step by step (RN conversions, fma = exact
    product + one rounding) and the 13 products summed exactly: |d_model - d_exact| <= 16 x 2^-24 x T_i for pairs ON the cutoff sphere of every atom of long pencil tiles."""
    rng = np.random.default_rng(3)
    f32, f16, f64 = np.float32, np.float16, np.float64
    cutneigh = f64(2.8)
    cutsq = cutneigh * cutneigh
    n = 200000
    # tile atoms: a pencil piece 9.7 x 2.8 x 2.8 somewhere in a 134-wide box; partners on the cutoff sphere (+- 1e-6 relative)
    origin = rng.uniform(0.0, 120.0, (n, 3))                                   # the tile's bounding-box centre (double; what the kernel subtracts)
    a = origin + rng.uniform(-1.0, 1.0, (n, 3)) * np.array([4.85, 1.4, 1.4])
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    b = a + u * (cutneigh * (1.0 + rng.uniform(-1.0e-6, 1.0e-6, n)))[:, None]
    ox = f32(origin).astype(f64)                                               # (the kernel's origin is a float promoted to real)
    fa = f32(a - ox)                                                           # (float)(pme - o)
    lf = f32(b - ox)

    def fma32(x, y, z):                                                        # fmaf: exact product (24 + 24 bits fit a double), one rounding
        return f32(x.astype(f64) * y.astype(f64) + z.astype(f64))
    m = f32(-2.0) * lf
    bb = fma32(lf[:, 2], lf[:, 2], fma32(lf[:, 1], lf[:, 1], f32(lf[:, 0] * lf[:, 0])))
    mh = f16(m); ml = f16(m - f32(mh))
    bbh = f16(bb); bbl = f16(bb - f32(bbh))
    aa_d = (fa.astype(f64) ** 2).sum(axis=1)
    thf = f32(cutsq - aa_d)
    th = f16(thf); tl = f16(thf - f32(th))
    ah = f16(fa); al = f16(fa - f32(ah))
    F = lambda q: q.astype(f64)
    d_model = (F(ah) * F(mh) + F(ah) * F(ml) + F(al) * F(mh)).sum(axis=1) + F(bbh) + F(bbl) - F(th) - F(tl)
    d_exact = ((a - b) ** 2).sum(axis=1) - cutsq
    cutp = f32(1.001) * f32(cutneigh) + f32(0.01)
    s = np.abs(fa) + cutp
    T = (s * s).sum(axis=1) + np.abs(thf) + f32(2.0) * (s * np.abs(fa)).sum(axis=1) + f32(cutsq)
    unit = 2.0 ** -24 * T.astype(f64)
    worst = float(np.max(np.abs(d_model - d_exact) / unit))
    assert worst <= 16.0, worst
    assert worst > 0.5, worst                                                  # (the model is not vacuous: the arithmetic does lose a few units)
