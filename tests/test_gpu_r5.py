"""`-m gpu` parity tests, part 3 (round 5): option combinations the step loop must survive, the launcher contract of the drop-in
executable (ref/run_one_test:50: `${MPISTART} -np N ./exe ...`), direct borders (ref/comm.cpp:700-883 as one exchange)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import free_port
from test_gpu_parity import mm, rows_close

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


# every option mmd_set_option still knows that changes HOW a one-rank run is computed, never what: each legal value runs to the last step (sliced, so that
# every slice ends with a finalIntegrate-only launch) and lands on the default's rows. (Round 6 cut the table from 40 entries to 20: what lost its A/B is gone —
# tile_read, tile_waves, tile_unroll, exact_div, build 0, fuse_final, kernel_dummy, bin_reuse, async_counts, spin_readback, borders_fast, overlap_join, halo_recv 2, ...;
# the remaining multi-rank knobs — overlap, direct_halo, direct_borders, halo_recv, borders_est, exchange_cap, force_transport, safe_exchange — have their tests below and
# in test_gpu_more.py, maxneighs / eam_mlo / lj_original / check_exchange theirs in test_gpu_parity.py / test_gpu_more.py.)
ONE_RANK_KNOBS = [("tiles", [0, 1]), ("fuse", [0, 1, 2]), ("ghost_resolve", [0, 1, 2]), ("spec", [0, 1, 16]), ("time_force_sample", [1, 3, 7]), ("check_exchange", [0, 1]),
                  ("fold_reverse", [0, 1]), ("core_pct", [0, 2, 30])]


@pytest.mark.parametrize("knob,values", ONE_RANK_KNOBS)
def test_every_value_of_every_one_rank_knob_lands_on_the_defaults_rows(knob, values):
    deck, half = ("in.eam.miniMD", 0) if knob == "core_pct" else ("in.lj.miniMD", 1 if knob == "fold_reverse" else 0)
    args = ["-i", deck, "-s", "8" if "eam" in deck else "10", "-n", "47", "--half_neigh", str(half)]

    def run(opt):
        s = mm().Sim(args)
        for k, v in opt.items():
            s.handle.set_option(k, v)
        s.initial()
        for c in (1, 6, 20, 20):
            s.run_steps(c)
        s.handle.force_compute(1)
        d = s.handle.download()
        out = (s.rows(), d["x"][:d["nlocal"]].copy())
        s.close()
        return out
    base = run({})
    for v in values:
        got = run({knob: v})
        # same physics: bit-identical where the summation order is the same, to rounding where it is not (half lists: atomics; core_pct: the order of a row's pairs)
        assert [r[0] for r in got[0]] == [r[0] for r in base[0]]
        rows_close(got[0], base[0], 1e-9)
        assert np.allclose(got[1], base[1], rtol=0, atol=1e-8), (knob, v)


def test_unknown_and_illegal_option_values_are_errors():
    h = mm().Handle()
    for name, v in (("tile_read", 0), ("build", 0), ("exact_div", 1), ("halo_recv", 2), ("no_such_option", 1)):
        with pytest.raises(Exception):
            h.set_option(name, v)
    h.close()


# ---- the launcher contract of the drop-in executable (ref/run_one_test:50, ref/ljs.cpp:63-68) ---------------------------------------------------
LAUNCH_VARS = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "PMI_RANK", "PMI_SIZE",
               "MMD_TRANSPORT")


def _thermo(stdout):
    rows, on = [], False
    for line in stdout.splitlines():
        if line.startswith("# Timestep T"):
            on = True
            continue
        if line.startswith("# Performance Summary"):
            break
        f = line.split()
        if on and len(f) >= 4 and f[0].lstrip("-").isdigit():
            rows.append((int(f[0]), float(f[1]), float(f[2]), float(f[3])))
    return rows


def _mpiexec():
    import shutil
    for cand in (shutil.which("mpiexec"), "/opt/conda/bin/mpiexec"):
        if cand and os.path.exists(cand):
            return cand
    return None


@pytest.mark.parametrize("how", ["openmpi_env", "mpiexec"])
def test_executable_as_three_plain_processes_reproduces_4k_lj(how):
    """`miniMD_dp -s 10 -n 1000 --half_neigh 0` as THREE ranks started the way the reference's harness starts them (ref/run_one_test:50, np = 3 of
    ref/run_tests:116-151): plain processes that learn their rank from the launcher's environment, no MASTER_* anywhere. On a one-GPU box the ranks agree
    on the built-in TCP mesh (banner: DEBUG transport); the thermo rows are those of the reference's published 4k.lj log (its own mode-independence
    claim, tests/reference_output/README:3-5) and pass the reference's statistical rule."""
    import json
    from oracle_lib import ref_pass_rule
    exe = os.path.join(REPO, "minimd_amd", "bin", "miniMD_dp")
    argv = [exe, "-s", "10", "-n", "1000", "--half_neigh", "0", "--yaml_output", "0", "-dm", "-i", "in.lj.miniMD"]
    env0 = {k: v for k, v in os.environ.items() if k not in LAUNCH_VARS}
    cwd = os.path.join(REPO, "data")
    if how == "mpiexec":
        if _mpiexec() is None:
            pytest.skip("no mpiexec in this image")
        r = subprocess.run([_mpiexec(), "-np", "3"] + argv, cwd=cwd, env=env0, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        out = r.stdout
    else:
        procs = [subprocess.Popen(argv, cwd=cwd, env=dict(env0, OMPI_COMM_WORLD_RANK=str(k), OMPI_COMM_WORLD_SIZE="3", OMPI_COMM_WORLD_LOCAL_RANK=str(k)),
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(3)]
        outs = [p.communicate(timeout=900) for p in procs]
        assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
        out = outs[0][0]
        assert all(_thermo(o[0]) == [] for o in outs[1:])          # (only rank 0 prints, ref/thermo.cpp:106-112)
    assert "# MPI processes: 3" in out
    ngpu = mm().load_library("dp").mmd_device_count()
    assert ("# Transport: TCP mesh" in out) == (ngpu < 3) and ("# Transport: RCCL" in out) == (ngpu >= 3), out[:1500]
    rows = _thermo(out)
    ref = [r for r in json.load(open(os.path.join(GOLD, "reference_output.json")))["4k.lj"]["rows"] if r[0] <= 1000]
    rows_close(rows, ref, 1.5e-5)
    assert ref_pass_rule(ref, rows, 4000, 8)[0]


def test_harness_scope_1_runs_np_3_and_8():
    """ref/run_tests scope 1 = sizes 10 at np 1, 3 and 8, 1000 steps: tools/run_one_test.py starts the executable through the launcher like `make test`
    does and applies the reference's PASS rule — all three on whatever this box has (one GPU: the np > 1 runs share it over the TCP mesh)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "run_one_test.py"), "--scope", "1", "--input", "lj", "--halfneigh", "0"],
                       capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.count("PASSED") == 3 and "np=8" in r.stdout, r.stdout[-3000:]


# ---- direct borders: Comm::borders on several ranks as one exchange (ref/comm.cpp:700-883) --------------------------------------------------------
def _loopback(args, options, prec="dp"):
    """one rank whose periodic self swaps are forced through RCCL (the code path of a rank inside a multi-GPU run)"""
    s = mm().Sim(args, precision=prec)
    h = s.handle
    h.init_rccl(h.unique_id(), 0, 1)
    h.set_option("force_transport", 1)
    for k, v in options.items():
        h.set_option(k, v)
    s.initial(); s.run()
    d = h.download()
    stats = h.run_stats()
    info = h.comm_info()
    swaps = [h.swap_info(q) for q in range(info["nswap"])]
    lists = [h.sendlist(q).copy() for q in range(info["nswap"])]
    out = {"rows": s.rows(), "x": d["x"].copy(), "f": d["f"].copy(), "v": d["v"].copy(), "counts": h.counts(), "swaps": swaps, "lists": lists,
           "direct": h.counter("borders_direct"), "general": h.counter("borders_general"), "fast": h.counter("borders_fast"), "stats": stats, "in_x": h.counter("halo_in_x_steps")}
    s.close()
    return out


@pytest.mark.parametrize("deck,half,gn", [("in.lj.miniMD", 0, 0), ("in.lj.miniMD", 1, 1), ("in.lj.miniMD", 1, 0), ("in.eam.miniMD", 0, 0)])
@pytest.mark.parametrize("recv", [1, 3])
def test_direct_borders_equal_the_swap_by_swap_borders(deck, half, gn, recv):
    """option direct_borders (default on): from the second re-neighboring of a run on, the ghosts of a rank are made by ONE exchange of the 26 image lists
    (every owner decides from its own coordinates which later swaps forward its atoms) instead of the three dependent forwarding rounds of
    ref/comm.cpp:700-883; halo_recv 1 unpacks the per-step halo with a kernel of its own; halo_recv 3 (default) leaves the partners'
    messages where they land, behind the ghost slots of the position buffer, and the LJ full-list force kernel stages the boundary tiles' ghosts from there
    (no k_dh_unpack on the step; the ghost slots are brought up to date when the run ends). Same ghosts in the same slots with the same
    positions, the same swap counts and — derived on demand from the slab bits that travelled along — the same six send lists; so the same rows, positions
    and forces, bit for bit (half lists: to the order of the atomics) after 70 steps with 3 re-neighborings."""
    args = ["-i", deck, "-s", "12" if "lj" in deck else "8", "-n", "70", "--half_neigh", str(half), "-gn", str(gn)]
    a = _loopback(args, {"direct_borders": 0, "halo_recv": 1, "overlap": 0})
    b = _loopback(args, {"direct_borders": 1, "halo_recv": recv, "overlap": 0})
    assert a["direct"] == 0 and b["direct"] >= 2, (a["direct"], b["direct"], b["general"], b["fast"])
    assert a["in_x"] == 0 and (b["in_x"] > 30) == (recv == 3 and half == 0 and "lj" in deck), (recv, b["in_x"])
    assert a["counts"][:2] == b["counts"][:2]
    assert [(s_["sendnum"], s_["recvnum"], s_["firstrecv"]) for s_ in a["swaps"]] == [(s_["sendnum"], s_["recvnum"], s_["firstrecv"]) for s_ in b["swaps"]]
    for la, lb in zip(a["lists"], b["lists"]):
        np.testing.assert_array_equal(la, lb)
    if half:
        rows_close(a["rows"], b["rows"], 1e-10)
        assert np.abs(a["x"] - b["x"]).max() <= 1e-9
    else:
        assert a["rows"] == b["rows"]
        np.testing.assert_array_equal(a["x"], b["x"])
        np.testing.assert_array_equal(a["f"], b["f"])
        np.testing.assert_array_equal(a["v"], b["v"])
    # one dependent transfer per re-neighboring instead of three: fewer bytes are not expected (the same records travel), fewer synchronisations are not
    # needed either way (both forms defer their counts to the build's read-back)
    assert b["stats"]["host_syncs"] <= a["stats"]["host_syncs"]


def test_direct_borders_with_undersized_messages_fall_back(tmp_path):
    """message capacities of 60 % of the previous lengths (borders_est 60): k_db_pack notices, the flag is max-reduced, and the borders are redone swap by
    swap — the run is the one with the normal capacities."""
    args = ["-s", "12", "-n", "70", "--half_neigh", "0"]
    a = _loopback(args, {"direct_borders": 1})
    b = _loopback(args, {"direct_borders": 1, "borders_est": 60})
    assert a["direct"] >= 2 and b["direct"] == 0 and b["general"] > a["general"]
    assert a["rows"] == b["rows"]
    np.testing.assert_array_equal(a["x"], b["x"])


def _mp_run(args, port, tmp_path, options="", nprocs=2, prec="dp"):
    import json
    out = str(tmp_path / ("mp_%d.json" % port))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MMD_TEST_OPTIONS=options)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nprocs), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py"), "sim", out, prec] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.load(open(out))


@pytest.mark.parametrize("nprocs,args", [(2, ["-s", "12", "-n", "100", "--half_neigh", "0"]), (8, ["-s", "14", "-n", "60", "--half_neigh", "0"]),
                                         (3, ["-i", "in.eam.miniMD", "-s", "10", "-n", "60", "--half_neigh", "0"]), (4, ["-s", "12", "-n", "60", "--half_neigh", "1"]),
                                         (4, ["-s", "12", "-n", "60", "--half_neigh", "1", "-gn", "0"])])
def test_direct_borders_on_several_ranks(nprocs, args, port, tmp_path):
    """2 / 8 / 3 / 4 ranks sharing the GPU (2x1x1, 2x2x2, 3x1x1, 2x2x1: partners that are the same rank twice, diagonal partners, dimensions that wrap onto
    the rank itself and stay local): direct borders against the swap-by-swap form — the same atoms and ghosts per rank, the same rows (half lists: to the
    order of the atomics; with ghost newton the reverse communication runs on the send lists derived from the slab bits)."""
    a = _mp_run(args, port, tmp_path, options="direct_borders=1", nprocs=nprocs)
    b = _mp_run(args, free_port(), tmp_path, options="direct_borders=0", nprocs=nprocs)
    assert all(st["borders_direct"] >= 2 for st in a["stats"]) and all(st["borders_direct"] == 0 for st in b["stats"]), (a["stats"], b["stats"])
    assert a["counts"] == b["counts"]
    if "--half_neigh" in args and args[args.index("--half_neigh") + 1] == "1":
        rows_close([tuple(r) for r in a["rows"]], [tuple(r) for r in b["rows"]], 1e-10)
    else:
        assert a["rows"] == b["rows"]


@pytest.mark.parametrize("deck,half", [("in.lj.miniMD", 0), ("in.lj.miniMD", 1), ("in.eam.miniMD", 0)])
def test_overlap_chosen_by_measurement_changes_nothing(deck, half):
    """option overlap -1 (default): behind the first re-neighboring of a run 2 x 8 steps are timed with and without the halo under the interior tiles, the
    times are summed over the ranks and every rank keeps the faster form. Whatever the choice, and across the switch inside the run, the results are those of
    a run that never overlapped (full lists: bit for bit)."""
    args = ["-i", deck, "-s", "12" if "lj" in deck else "8", "-n", "70", "--half_neigh", str(half)]
    a = _loopback(args, {"overlap": 0})
    m = mm()
    s = m.Sim(args)
    h = s.handle
    h.init_rccl(h.unique_id(), 0, 1)
    h.set_option("force_transport", 1)
    s.initial(); s.run()
    d = h.download()
    assert h.counter("overlap_choice") in (0, 1) and h.counter("overlap_trial_on_ns") > 0 and h.counter("overlap_trial_off_ns") > 0
    if half:
        rows_close(a["rows"], s.rows(), 1e-10)
    else:
        assert a["rows"] == s.rows()
        np.testing.assert_array_equal(a["x"], d["x"])
        np.testing.assert_array_equal(a["f"], d["f"])
    s.close()


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_boundary_tiles_on_the_communication_stream_change_nothing(prec):
    """overlap 1 on several ranks, LJ over full lists (here: RCCL loop-back): the boundary tiles are launched on the communication stream right behind the transfer
    (they run under the tail of the interior tiles and read the received records where they landed; the compute stream joins at the end of the step; a thermo step keeps
    the boundary tiles on the compute stream behind a wait: its energy sum needs both launches finished). Same bits as the run without overlap, thermo step included."""
    args = ["-s", "14", "-n", "130", "--half_neigh", "0"]
    res = {}
    for name, opts in (("none", {"overlap": 0}), ("join", {"overlap": 1})):
        s = mm().Sim(args, precision=prec)
        h = s.handle
        h.init_rccl(h.unique_id(), 0, 1)
        h.set_option("force_transport", 1)
        for k, v in opts.items():
            h.set_option(k, v)
        s.initial(); s.run()
        d = h.download()
        res[name] = (s.rows(), d["x"].copy(), d["v"].copy(), d["f"].copy(), h.counter("overlap_join_steps"), h.counter("halo_in_x_steps"))
        s.close()
    assert res["none"][4] == 0 and res["join"][4] > 100, [r[4:] for r in res.values()]
    assert res["join"][5] > 80             # (from the second re-neighboring on the joined steps need no unpack kernel)
    for other in ("join",):
        assert res["none"][0] == res[other][0]
        for a_, b_ in zip(res["none"][1:4], res[other][1:4]):
            np.testing.assert_array_equal(a_, b_)


# ---------------------------------------------------------------------------------------------------
# Neighbor::build with the pre-test on the matrix cores (round 5): rows stay those of an all-double build
# ---------------------------------------------------------------------------------------------------
def _brute_force_rows(x, cutneighsq):
    """full neighbor rows of owned atoms without ghosts, the reference's arithmetic (ref/neighbor.cpp:160-166: delx*delx + dely*dely + delz*delz <= cutneighsq)"""
    d = x[:, None, :] - x[None, :, :]
    rsq = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
    hit = rsq <= cutneighsq
    np.fill_diagonal(hit, False)
    return [np.nonzero(hit[i])[0] for i in range(len(x))]


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_neighbor_build_with_pairs_on_the_cutoff_sphere(prec):
    """pairs placed ON the cutoff (rsq within a few ulps of cutneighsq, both sides) are exactly what the half-precision MFMA pre-test cannot decide: its band must
    send them to the exact re-test. Rows against a brute-force list in the reference's arithmetic, as sets."""
    rng = np.random.default_rng(11)
    real = np.float64 if prec == "dp" else np.float32
    cutneigh = real(2.8)
    cutsq = real(cutneigh * cutneigh)
    L = 33.6
    pts = []
    for c in rng.uniform(4.0, L - 4.0, (150, 3)):
        pts.append(c)
        for _ in range(6):                       # partners at distance cutneigh * (1 +- a few eps) in random directions
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            pts.append(c + u * float(cutneigh) * (1.0 + rng.integers(-6, 7) * (2.0e-16 if prec == "dp" else 1.0e-7)))
    x = np.ascontiguousarray(np.array(pts), real)
    h = mm().Handle(prec)
    h.set_box([L, L, L])
    h.set_mass(1.0)
    n = len(x)
    h.upload(x, np.zeros_like(x), np.zeros(n, np.int32), np.arange(1, n + 1, dtype=np.int32), nlocal=n)
    h.neighbor_setup([20, 20, 20], float(cutneigh), 0, 1, 1)
    h.neighbor_build()
    assert h.neighbor_tile_stats()["tiles"] > 0
    nb, nn = h.neighbor_download()
    rows = _brute_force_rows(x, cutsq)
    close = sum(int(np.sum(np.abs(((x[i] - x[rows[i]]) ** 2).sum(axis=1) / float(cutsq) - 1.0) < 1.0e-5)) for i in range(n) if len(rows[i]))
    assert close > 300, close                    # (the construction did put pairs on the sphere)
    for i in range(n):
        assert nn[i] == len(rows[i]), (i, nn[i], len(rows[i]))
        np.testing.assert_array_equal(np.sort(nb[i, :nn[i]]), rows[i])
    h.close()


def test_neighbor_build_of_stretched_tiles_falls_back_to_the_row_kernel():
    """clusters strung along x in a long box: a tile of 64 atoms of one pencil spans hundreds of length units, |b|^2 of its candidates would leave the half range of
    the MFMA pre-test — such a build is handed to the row kernel (no tile lists), with the same rows."""
    rng = np.random.default_rng(5)
    L = (640.0, 8.4, 8.4)
    pts = []
    for cx in np.arange(20.0, 620.0, 60.0):
        for cy, cz in ((1.4, 1.4), (7.0, 4.2)):
            pts.append(np.array([cx, cy, cz]) + rng.uniform(-1.0, 1.0, (12, 3)))
    x = np.ascontiguousarray(np.concatenate(pts))
    n = len(x)
    h = mm().Handle("dp")
    h.set_box(list(L))
    h.set_mass(1.0)
    h.upload(x, np.zeros_like(x), np.zeros(n, np.int32), np.arange(1, n + 1, dtype=np.int32), nlocal=n)
    h.neighbor_setup([456, 6, 6], 2.8, 0, 1, 1)
    h.neighbor_build()
    assert h.neighbor_tile_stats()["tiles"] == 0
    nb, nn = h.neighbor_download()
    rows = _brute_force_rows(x, 2.8 * 2.8)
    assert sum(len(r) for r in rows) > 1000
    for i in range(n):
        assert nn[i] == len(rows[i])
        np.testing.assert_array_equal(np.sort(nb[i, :nn[i]]), rows[i])
    h.close()
