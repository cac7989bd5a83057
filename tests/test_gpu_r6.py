"""`-m gpu` parity tests, part 4 (round 6): BASELINE configs[3] dress-rehearsed at its REAL geometry on whatever box this is — 8 ranks, 2x2x2,
`-s 80` per rank (160^3 cells, 16.4 M atoms) — against the reference's own rows for that box; the driver's multi-GPU bench command on a one-GPU box;
the RCCL bring-up self-check; PERF_SUMMARY's buckets on several ranks (ref/integrate.cpp:100-107, 155-207; ref/ljs.cpp:485-495)."""
import json
import os
import subprocess
import sys
import time

import pytest

from test_gpu_parity import mm, rows_close

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
REFRUNS = json.load(open(os.path.join(GOLD, "ref_runs.json")))
LAUNCH_VARS = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE",
               "OMPI_COMM_WORLD_LOCAL_RANK", "PMI_RANK", "PMI_SIZE", "PMIX_RANK", "SLURM_PROCID", "SLURM_STEP_ID", "MMD_TRANSPORT", "MMD_LAUNCHER", "MMD_NRANKS")


def _mp_run(args, port, tmp_path, options="", nprocs=8, prec="dp", timeout=1500):
    out = str(tmp_path / ("mp_%d.json" % port))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MMD_TEST_OPTIONS=options, MMD_TEST_STEADY="40")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nprocs), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "mp_worker.py"), "sim", out, prec] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.load(open(out))


# what DESIGN.md §5.5 tells the first 8-GPU lease to expect per rank at `-s 80` per rank on a 2x2x2 grid
D_OWNED, D_GHOSTS, D_HALO_BYTES = 2048000, 277000, 9.0e6


def _check_rank_geometry(res, nsteps, owned, ghosts, halo_bytes, rebuild_every=20):
    nreb = nsteps // rebuild_every
    for rk, ((nl, ng, _tot), st) in enumerate(zip(res["counts"], res["stats"])):
        assert abs(nl - owned) <= 1e-3 * owned, (rk, nl)
        assert abs(ng - ghosts) <= 0.04 * ghosts, (rk, ng)
        # every re-neighboring of the run but (at most) the first went through the one-exchange borders and the handshake-free exchange
        assert st["borders_direct"] >= nreb - 1 and st["exchange_overflows"] == 0, (rk, st)
        # host synchronisations: 2 per re-neighboring in the steady state (the 40-step slice behind the run: plans exist, nothing falls back), <= 3 with a thermo row
        # or the overlap trial's one; the run's FIRST re-neighboring has no previous counts to size its messages from and pays the count handshakes once
        sd = st["steady"]
        assert sd["host_syncs"] <= 3 * (sd["steps"] // rebuild_every), (rk, st)
        assert sd["exchange_overflows"] == 0 and sd["borders_general"] == st["borders_general"], (rk, st)
        assert st["host_syncs"] <= 3 * nreb + 16, (rk, st)
        if halo_bytes:
            assert abs(st["bytes_sent"] / nsteps - halo_bytes) <= 0.05 * halo_bytes, (rk, st["bytes_sent"] / nsteps)
            assert abs(sd["bytes_sent"] / sd["steps"] - halo_bytes) <= 0.05 * halo_bytes, (rk, sd)
    assert sum(c[0] for c in res["counts"]) == res["natoms"]


@pytest.mark.slow
def test_config_d_dress_rehearsal_lj_full(port, tmp_path):
    """BASELINE configs[3] = `in.lj.miniMD -s 80 per rank`, 8 ranks 2x2x2 = 160^3 cells, 16 384 000 atoms, full lists, DP, the deck's 100 steps: every rank has 7
    distinct partners, ~277 k ghosts, ~9 MB of halo per step, fixed-size border / exchange messages derived at that size. On fewer than 8 GPUs the ranks share
    devices over the debug transport — the code path (direct halo, direct borders, handshake-free exchange) is the one RCCL runs. The thermo rows are the
    reference's own for this box (tests/golden/ref_runs.json: lj_s160_half_n100, printed by the unmodified reference; rows do not depend on the list style or the
    rank count to the printed digits: tests/reference_output/README:3-5)."""
    ref = [tuple(r) for r in REFRUNS["lj_s160_half_n100"]["rows"]]
    res = _mp_run(["-i", "in.lj.miniMD", "-nx", "160", "-ny", "160", "-nz", "160", "--half_neigh", "0", "-n", "100"], port, tmp_path)
    assert res["natoms"] == 16384000 and len(res["counts"]) == 8
    rows_close([tuple(r) for r in res["rows"]], ref, 2e-6)
    _check_rank_geometry(res, 100, D_OWNED, D_GHOSTS, D_HALO_BYTES)
    # full lists: 2 x the reference's half-list total, to its 6 printed digits
    tot = sum(c[2] for c in res["counts"])
    assert abs(tot - 2 * REFRUNS["lj_s160_half_n100"]["neigh_total"]) <= 2e-6 * tot, tot


@pytest.mark.slow
def test_config_d_dress_rehearsal_lj_half_ghost_newton(port, tmp_path):
    """the same box over HALF lists with ghost newton (the reference's default `--half_neigh 1`, ref/force_lj.cpp:271-357 + Comm::reverse_communicate,
    ref/comm.cpp:321-355): the reverse halo derives its six send lists from the slab bits at every re-neighboring. Rows = the reference's row of exactly this run."""
    ref = [tuple(r) for r in REFRUNS["lj_s160_half_n100"]["rows"]]
    res = _mp_run(["-i", "in.lj.miniMD", "-nx", "160", "-ny", "160", "-nz", "160", "--half_neigh", "1", "-n", "100"], port, tmp_path)
    rows_close([tuple(r) for r in res["rows"]], ref, 2e-6)
    _check_rank_geometry(res, 100, D_OWNED, D_GHOSTS, None)
    tot = sum(c[2] for c in res["counts"])
    assert abs(tot - REFRUNS["lj_s160_half_n100"]["neigh_total"]) <= 2e-6 * tot, tot


@pytest.mark.slow
def test_config_c_weak_scaled_over_8_ranks_eam(port, tmp_path):
    """BASELINE configs[2] weak-scaled the same way: `in.eam.miniMD -s 64 per rank`, 8 ranks = 128^3 cells, 8 388 608 atoms, 40 steps (two re-neighborings);
    ForceEAM::communicate (ref/force_eam.cpp:851-887) rides the direct-halo plan between the two sweeps. Rows = the unmodified reference's for that box
    (ref_runs.json: eam_s128_full_n40)."""
    ent = REFRUNS["eam_s128_full_n40"]
    res = _mp_run(["-i", "in.eam.miniMD", "-nx", "128", "-ny", "128", "-nz", "128", "--half_neigh", "0", "-n", "40"], port, tmp_path)
    assert res["natoms"] == ent["natoms"] == 8388608
    rows_close([tuple(r) for r in res["rows"]], [tuple(r) for r in ent["rows"]], 2e-6)
    _check_rank_geometry(res, 40, 1048576, 180000, None)          # (one rank at -s 64: 175.5 k ghosts at step 100, 182 k at step 40)
    tot = sum(c[2] for c in res["counts"])
    assert abs(tot - ent["neigh_total"]) <= 2e-6 * tot, tot


def _perf_summary(stdout):
    line = [l for l in stdout.splitlines() if "PERF_SUMMARY" in l and not l.startswith("#")][0].split()
    return {"nprocs": int(line[0]), "nsteps": int(line[2]), "natoms": int(line[3]), "total": float(line[4]), "force": float(line[5]), "neigh": float(line[6]),
            "comm": float(line[7]), "other": float(line[8])}


def _plain_ranks(nranks, argv, timeout=1500):
    from conftest import free_port
    exe = os.path.join(REPO, "minimd_amd", "bin", "miniMD_dp")
    env0 = {k: v for k, v in os.environ.items() if k not in LAUNCH_VARS}
    port = free_port()
    procs = [subprocess.Popen([exe] + argv, cwd=os.path.join(REPO, "data"),
                              env=dict(env0, OMPI_COMM_WORLD_RANK=str(k), OMPI_COMM_WORLD_SIZE=str(nranks), OMPI_COMM_WORLD_LOCAL_RANK=str(k), MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(nranks)]
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    return outs[0][0]


@pytest.mark.slow
@pytest.mark.parametrize("nranks,argv", [(8, ["-i", "in.lj.miniMD", "-nx", "160", "-ny", "160", "-nz", "160", "--half_neigh", "0"]),
                                         (2, ["-i", "in.lj.miniMD", "-nx", "80", "-ny", "40", "-nz", "40", "--half_neigh", "0"]),
                                         (4, ["-i", "in.eam.miniMD", "-nx", "32", "-ny", "32", "-nz", "16", "--half_neigh", "0"])])
def test_perf_summary_buckets_partition_the_wall_clock_on_several_ranks(nranks, argv):
    """The drop-in executable as plain ranks started the way `mpirun -np N` starts them (config D's box among them: its own TCP mesh at that size). The reference's
    PERF_SUMMARY columns partition t_total — t_other = t_total - t_force - t_neigh - t_comm >= 0 (ref/ljs.cpp:485-495; the buckets are consecutive host intervals,
    ref/integrate.cpp:100-107, 155-207). Round 5 printed t_comm > t_total and a negative t_other on several ranks (GPU times of concurrent streams summed)."""
    from test_gpu_r5 import _thermo
    out = _plain_ranks(nranks, argv)
    assert "# MPI processes: %d" % nranks in out
    ps = _perf_summary(out)
    assert ps["nprocs"] == nranks
    assert min(ps["force"], ps["neigh"], ps["comm"]) >= 0 and ps["other"] >= -3e-6, ps
    assert ps["force"] + ps["neigh"] + ps["comm"] <= ps["total"] + 3e-6, ps                 # (six printed decimals each)
    assert abs(ps["total"] - ps["force"] - ps["neigh"] - ps["comm"] - ps["other"]) <= 3e-6, ps
    if nranks == 8:
        rows_close(_thermo(out), [tuple(r) for r in REFRUNS["lj_s160_half_n100"]["rows"]], 2e-6)


@pytest.mark.slow
def test_the_drivers_multi_gpu_bench_command_on_this_box():
    """`python3 bench.py --gpus 8 --steps 20 --warmup 5` — the driver's exact command for the scaling run — as a plain process on whatever this box has: ONE JSON line on
    stdout, n_gpus 8, 8 ranks of config D's geometry; on fewer than 8 GPUs `valid` is false and `reason` says the halos did not travel over RCCL."""
    t0 = time.time()
    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_VARS}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], cwd=REPO, env=env, capture_output=True, text=True, timeout=1700)
    took = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    d = json.loads(lines[0])
    ngpu = mm().load_library("dp").mmd_device_count()
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["transport_ranks"] == 8
    assert d["valid"] == (ngpu >= 8) and (d["reason"] is None) == (ngpu >= 8), (d["valid"], d["reason"])
    assert d["config"]["transport"] == ("rccl" if ngpu >= 8 else "host")
    assert d["atoms_per_rank"]["owned_min"] >= 0.999 * D_OWNED and d["atoms_per_rank"]["owned_max"] <= 1.001 * D_OWNED
    assert abs(d["halo_bytes_per_step"]["max_rank"] - D_HALO_BYTES) <= 0.05 * D_HALO_BYTES
    assert d["host_syncs_per_rebuild"] <= 3
    assert abs(d["value"] - 16384000 * 20 / (d["ms_per_step"] * 20e-3) / 1e6) <= 1e-6 * d["value"]
    assert took < 900, took


# ---- RCCL bring-up (mmd_comm_init_rccl) ------------------------------------------------------------------------------------------------------------
def test_rccl_bring_up_self_check_runs_in_loop_back():
    """mmd_comm_init_rccl ends with ONE grouped send/recv of a known pattern with every distinct partner + one all-reduce, verified (here: one rank, itself the
    partner — the code a rank of an 8-GPU run executes with its 7 partners)."""
    h = mm().Handle(precision="dp")
    h.init_rccl(h.unique_id(), 0, 1)
    assert h.counter("rccl_check_partners") == 1 and h.counter("rccl_check_us") > 0
    assert h.transport_info()["kind"] == "rccl"
    h.close()


def test_rccl_bring_up_fails_with_a_diagnosis_when_a_rank_never_arrives():
    """rank 0 of 2 whose partner never starts: within MMD_RCCL_TIMEOUT seconds the call returns an error that names rank, size and the device — instead of a process
    that sits in ncclCommInitRank until the lease ends."""
    code = ("import os, sys; sys.path.insert(0, %r); import minimd_amd\n"
            "h = minimd_amd.Handle(precision='dp')\n"
            "try:\n"
            "    h.init_rccl(h.unique_id(), 0, 2)\n"
            "    print('UNEXPECTED: returned')\n"
            "except Exception as e:\n"
            "    print('ERR', e)\n"
            "sys.stdout.flush(); os._exit(0)\n") % REPO
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MMD_RCCL_TIMEOUT="4"), capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert "ERR" in r.stdout and "rank 0 of 2" in r.stdout and "did not return within" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
    assert "RCCL bring-up FAILED on rank 0 of 2" in r.stderr
    assert took < 120, took


# ---- neighbor SETS at BASELINE sizes against the oracle (round-5 verdict: exact sets had only been compared at <= 864 atoms; at 1 M+ only totals) ---------------------
def _rows_sorted(nb, nn, width):
    """rows as a dense (n, width) array, every row sorted, the unused tail filled with a sentinel above every atom index"""
    import numpy as np
    out = np.full((len(nn), width), np.iinfo(np.int32).max, np.int32)
    k = min(width, nb.shape[1])
    out[:, :k] = nb[:, :k]
    out[np.arange(width)[None, :] >= nn[:, None]] = np.iinfo(np.int32).max
    out.sort(axis=1)
    return out


@pytest.mark.slow
@pytest.mark.parametrize("size,half,prec", [(32, 0, "dp"), (32, 1, "dp"), (80, 0, "dp"), (80, 1, "dp"), (32, 0, "sp"), (48, 1, "sp")])
def test_neighbor_sets_at_baseline_sizes_equal_the_oracles(size, half, prec):
    """BASELINE configs[0] (-s 32, 131 072 atoms) over full and half lists and configs[1] (-s 80, 2 048 000 atoms, full lists): the oracle (pinned bit for bit to the
    reference's arrays, tests/test_oracle_pin.py; single precision: the reference's own float expression) runs 20 steps — a melting lattice, one re-neighboring —, its owned + ghost atoms go to the device as they are, the device
    builds its list (MFMA pre-test + exact re-test inside the band, chunk table, two-pass cull: the production kernel) and EVERY row equals the oracle's as a set, every
    count exactly (ref/neighbor.cpp:126-191; half lists: what crosses the C-ABI is the reference's own partition, k_build<3>)."""
    import numpy as np
    from test_gpu_parity import handle_from_oracle
    from oracle_lib import Oracle
    o = Oracle(["-s", size, "-n", 20, "--half_neigh", half], precision=prec)
    o.initial(); o.run()
    h = handle_from_oracle(o, precision=prec)
    h.neighbor_build()
    assert h.counter("tiles_ready") == 1
    assert h.neighbor_info()["total"] == int(o.numneigh().sum())
    nb, nn = h.neighbor_download()
    onn = o.numneigh()
    np.testing.assert_array_equal(nn, onn)
    width = int(onn.max())
    mine = _rows_sorted(nb, nn, width)
    del nb
    theirs = _rows_sorted(o.neighbors(), onn, width)
    assert mine.shape == theirs.shape == (size ** 3 * 4, width)
    bad = np.nonzero((mine != theirs).any(axis=1))[0]
    assert len(bad) == 0, (len(bad), bad[:5])
    h.close(); o.close()


# ---- half-list tile kernel: a share that does not fit the packed accumulator (single precision) ---------------------------------------------------
@pytest.mark.parametrize("prec", ["sp", "dp"])
def test_half_list_pair_far_up_the_repulsive_wall(prec):
    """k_lj_half_tile adds a pair's share to the partner in LDS; in single precision x and y travel in ONE 64-bit fixed-point atomic whose fields hold |share| < 16
    (force / c_out). A pair closer than ~0.8 sigma exceeds that: it is marked inside the trip and leaves the chip behind it, straight to the partner's force
    (round 6: the marked pairs are re-evaluated behind the trip instead of branching per pair). One interior atom is pushed to 0.75 sigma of a neighbor (share ~38);
    the device builds its half list on those positions and its forces equal the oracle's ForceLJ::compute_halfneigh (ref/force_lj.cpp:271-357) on the same list."""
    import numpy as np
    from test_gpu_parity import mm
    from oracle_lib import Oracle
    o = Oracle(["-s", 6, "-n", 20, "--half_neigh", 1, "-gn", 0], precision=prec)
    o.initial(); o.run()
    nl, x, typ = o.nlocal(), o.x().copy(), o.type()
    nall = len(x)
    box = o.box()
    prd = np.array(box[0:3])
    cn = float(o.param("cutneigh"))
    inner = np.nonzero(((x[:nl] > cn + 1.0) & (x[:nl] < prd - cn - 1.0)).all(axis=1))[0]          # atoms without periodic images: moving them moves no ghost
    a = int(inner[0])
    d = np.linalg.norm(x[:nl] - x[a], axis=1); d[a] = 1e9
    d[[i for i in range(nl) if i not in set(inner.tolist())]] = 1e9
    b = int(np.argmin(d))
    u = (x[a] - x[b]) / np.linalg.norm(x[a] - x[b])
    x[a] = x[b] + 0.75 * u
    h = mm().Handle(prec)
    h.set_box(box[0:3], [box[3], box[5], box[7]], [box[4], box[6], box[8]])
    h.set_mass(o.param("mass"))
    h.upload(x, np.zeros((nl, 3)), typ, o.tag(), nlocal=nl)
    h.neighbor_setup(o.nbins(), cn, 1, 0, o.ntypes())
    h.force_lj_setup(*o.lj_tables())
    h.neighbor_build()
    assert h.neighbor_tile_stats()["tiles"] > 0
    nb, nn = h.neighbor_download()
    fo, _eo, _vo = o.lj_force_half(x, typ, nl, nall, nb, nn, *o.lj_tables(), 0, 0)
    h.force_compute(0)
    f = h.download(halfneigh=True)["f"][:nl]
    fmax = np.abs(fo[:nl]).max()
    assert fmax > 500.0                                         # (the pair is there: 48 x 38)
    tol = 1e-11 if prec == "dp" else 2e-5
    assert np.abs(f - fo[:nl]).max() <= tol * fmax
    h.close(); o.close()
