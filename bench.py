#!/usr/bin/env python3
"""bench.py — Matom-steps/s of the miniMD hot path (LJ, full neighbor lists, double precision) on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
            or as a plain process: bench.py then re-launches itself through torch.distributed.run on a free port)

A "step" is one MD timestep of BASELINE.json configs[1]: in.lj.miniMD, -s 80 per GPU (2,048,000 atoms per GPU,
weak scaling: the global box is 160x80x80 / 160x160x80 / 160^3 unit cells at 2/4/8 GPUs), full neighbor list,
re-neighboring every 20 steps, thermo every 100 — i.e. exactly the loop the reference times
(Integrate::run, ref/ljs.cpp:470-472). Atoms are resident in HBM when the timed region starts.
Before the W warm-up steps the system is equilibrated for --equil steps (default 100, untimed, part of the set-up):
the lattice has melted and the timed window sees the state SURVEY.md §8(d) names for the roofline figure (positions
and neighbor lists "as they exist at step 100"), whatever K and W are; the GPU is then kept busy for --clock-warm-ms (default 400 ms,
force-kernel launches that do not change the state) because the chip's clocks keep ramping for >100 ms after idling. Re-neighboring follows the global step number,
so every 20 timed steps contain exactly one rebuild.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import random
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_atom(kbar, ghost_ratio, real_bytes=8):
    """compulsory HBM bytes of one LJ full-neighbor force launch per owned atom:
    neighbor indices 4*K + numneigh 4 + x_i 3*real + type 4 + f_i 3*real + first touch of ghosts (3*real+4)*G/N"""
    return 4.0 * kbar + 4 + 3 * real_bytes + 4 + 3 * real_bytes + ghost_ratio * (3 * real_bytes + 4)


def cpu_model():
    """host CPU as /proc/cpuinfo names it + sockets x cores (the cpu_baseline figure varies by 2x between boxes: say which box)"""
    try:
        names, phys, cores = set(), set(), set()
        pid = None
        for l in open("/proc/cpuinfo"):
            k, _, v = l.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                names.add(v)
            elif k == "physical id":
                pid = v
                phys.add(v)
            elif k == "core id":
                cores.add((pid, v))
        return "%s, %d socket(s), %d cores, %d hardware threads" % (" / ".join(sorted(names)) or "unknown CPU", max(len(phys), 1), len(cores) or (os.cpu_count() or 0), os.cpu_count() or 0)
    except Exception:  # noqa: BLE001
        return "unknown CPU, %d hardware threads" % (os.cpu_count() or 0)


def cpu_baseline(size, nsteps):
    """rank 0, N=1 only: the UNMODIFIED reference (oracle/_ref/miniMD_ref_dp, built from /root/reference by
    oracle/Makefile) on this box's host cores, bounded sample of the same workload."""
    exe = os.path.join(REPO, "oracle", "_ref", "miniMD_ref_dp")
    # the reference's OpenMP loops stop scaling (and then collapse) far below the 256 hardware threads of the
    # GPU box's host: use one thread per physical core of ONE socket unless told otherwise
    cores = int(os.environ.get("MMD_CPU_THREADS", "0")) or max(1, min(32, (os.cpu_count() or 2) // 2))   # 32 measured best on the 2x64-core host (16: 57, 32: 61, 64: 40, 128: 33 Matom-steps/s)
    data = os.path.join(REPO, "data")
    if os.path.exists(exe):
        cmd = [exe, "-i", "in.lj.miniMD", "-s", str(size), "-n", str(nsteps), "--half_neigh", "0", "-t", str(cores)]
        kind = "reference"
    else:
        exe = os.path.join(REPO, "oracle", "mmd_oracle_dp")
        subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle"), "oracle"], check=False)
        cmd = [exe, "-i", "in.lj.miniMD", "-s", str(size), "-n", str(max(nsteps // 2, 1)), "--half_neigh", "0"]
        kind, cores = "port", 1
    try:
        env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores")
        r = subprocess.run(cmd, cwd=data, capture_output=True, text=True, timeout=900, env=env)
        line = [l for l in r.stdout.splitlines() if "PERF_SUMMARY" in l and not l.startswith("#")][0].split()
        return {"value": float(line[9]) / 1e6, "unit": "Matom-steps/s", "cores": cores, "kind": kind,
                "sample": "%s, in.lj.miniMD -s %d --half_neigh 0 DP, %s steps, t_total %.2f s; host: %s" % (
                    "ref/ MPI-stub + OpenMP -t %d (OMP_PROC_BIND=close; thread count = MMD_CPU_THREADS or min(32, hardware threads / 2))" % cores
                    if kind == "reference" else "oracle C restatement, 1 thread", size, line[2], float(line[4]), cpu_model())}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "Matom-steps/s", "cores": cores, "kind": kind, "sample": "failed: %r; host: %s" % (e, cpu_model())}


class stdout_to_stderr:
    """libraries that talk on the C stdout (gloo's connection report, RCCL's version banner) must not get in front of the ONE JSON
    line: while this is active file descriptor 1 points at stderr"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:                                   # (the C library's own stdout buffer — a pipe is fully buffered — would otherwise be flushed at exit, behind the JSON line)
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def perf_summary_cold(size, nsteps=None):
    """what a user of the drop-in executable sees: `miniMD_dp -i in.lj.miniMD -s <size> --half_neigh 0` as a fresh process (cold GPU clocks,
    no equilibration, no warm-up, the deck's 100 steps — or `-n nsteps`: the reference's own published logs are 1000- and 10 000-step runs,
    tests/reference_output/864k.lj:133) and the value of its own PERF_SUMMARY line = natoms*ntimes/t_total of
    Integrate::run (ref/ljs.cpp:470-495). Reported beside the steady-state `value`, never instead of it."""
    exe = os.path.join(REPO, "minimd_amd", "bin", "miniMD_dp")
    cmd = [exe, "-i", "in.lj.miniMD", "-s", str(size), "--half_neigh", "0"] + (["-n", str(nsteps)] if nsteps else [])
    try:
        r = subprocess.run(cmd, cwd=os.path.join(REPO, "data"), capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if "PERF_SUMMARY" in l and not l.startswith("#")][0].split()
        rows = [l.split() for l in r.stdout.splitlines() if l[:1].isdigit() and len(l.split()) == 5]
        return {"value": float(line[9]) / 1e6, "unit": "Matom-steps/s", "steps": int(line[2]), "natoms": int(line[3]), "t_total_s": float(line[4]),
                "t_force_s": float(line[5]), "t_neigh_s": float(line[6]), "t_comm_s": float(line[7]),
                "last_row": " ".join(rows[-1][:4]) if rows else None, "cmd": "minimd_amd/bin/miniMD_dp " + " ".join(cmd[1:])}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "Matom-steps/s", "cmd": " ".join(cmd), "error": repr(e)}


def rank_path_loopback_child(size, steps):
    """(child process of rank_path_loopback: prints one JSON object)"""
    import minimd_amd
    s = minimd_amd.Sim(["-s", size, "--half_neigh", 0, "-n", steps], precision="dp", quiet=True)
    h = s.handle
    with stdout_to_stderr():          # (ncclCommInitRank prints RCCL's version banner)
        h.init_rccl(h.unique_id(), 0, 1)
    h.set_option("force_transport", 1)
    s.initial()
    s.run_steps(40)
    t_w = time.perf_counter()
    while (time.perf_counter() - t_w) < 0.3:
        h.profile_kernel(0, 50)
    best = min(s.run_steps(steps) for _ in range(2))
    st = h.run_stats()
    nat = s.natoms()
    choice = h.counter("overlap_choice")
    trial = {"without_overlap_ms_per_step": h.counter("overlap_trial_off_ns") * 1e-6, "with_overlap_ms_per_step": h.counter("overlap_trial_on_ns") * 1e-6}
    in_x, direct = h.counter("halo_in_x_steps"), h.counter("borders_direct")
    # the form the automatic choice did NOT take, on the same state (explicit option), for the record
    other = 1 - choice if choice in (0, 1) else 1
    h.set_option("overlap", other)
    s.run_steps(40)
    best_other = min(s.run_steps(steps) for _ in range(2))
    s.close()
    vals = {choice if choice in (0, 1) else 0: nat * steps / best / 1e6, other: nat * steps / best_other / 1e6}
    print(json.dumps({"value": nat * steps / best / 1e6, "unit": "Matom-steps/s", "ms_per_step": best * 1e3 / steps, "steps": steps,
                      "halo_bytes_per_step": st["bytes_sent"] / steps,
                      "overlap": {"chosen": choice, "trial": trial, "value_without_overlap": vals.get(0), "value_with_overlap": vals.get(1)},
                      "direct_borders": direct, "steps_without_unpack_kernel": in_x,
                      "note": "one rank, self swaps through RCCL loop-back on one GPU (no xGMI transfer); halo overlap chosen by the library's own timed trial (option overlap -1)"}), flush=True)


def rank_path_loopback(size, steps):
    """What the multi-GPU step costs a rank apart from the wire, measured on THIS one GPU: the same workload on one rank whose periodic self swaps are
    routed through RCCL (option force_transport: direct halo packed, sent to itself with ncclSend/ncclRecv, unpacked; borders and exchange through
    their fixed-size messages; halo on the communication stream under the interior tiles) — the code path of a rank inside `--gpus 8`, xGMI transfer
    time excepted. A diagnostic next to `value` (which is the production one-rank path), never instead of it. Runs as a process of its own with a
    time limit: whatever happens in there, the bench line is printed."""
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--loopback-child", "--size", str(size), "--steps", str(max(steps, 60))],
                           capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"value": None, "error": (r.stderr or r.stdout)[-300:]}
        return json.loads(line[-1])
    except Exception as e:  # noqa: BLE001
        return {"value": None, "error": repr(e)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=80, help="unit cells per GPU edge (BASELINE configs[1]: 80)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=100)
    ap.add_argument("--windows", type=int, default=3, help="timed windows of --steps steps each, back to back: the FIRST is `value` (the contract's "
                                                          "K steps), all of them are listed in `value_windows` so that a short window shows its spread")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold run of the drop-in executable (perf_summary_cold)")
    ap.add_argument("--sustained-steps", type=int, default=2000, help="ONE further timed window of this many steps behind the --windows short ones (0: none): what ~0.5 s of "
                                                                      "this load runs at, next to the K-step `value`")
    ap.add_argument("--no-loopback", action="store_true", help="skip the one-GPU measurement of the multi-rank code path (rank_path_loopback)")
    ap.add_argument("--loopback-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--equil", type=int, default=100, help="untimed equilibration steps before the warm-up (set-up, see the module docstring)")
    ap.add_argument("--clock-warm-ms", type=float, default=400.0,
                    help="set-up: keep the GPU busy this long (force-kernel launches that leave the state untouched) so that the warm-up "
                         "and the timed steps run at settled clocks whatever W is (the chip ramps for >100 ms after idling)")
    args = ap.parse_args()
    if args.loopback_child:
        rank_path_loopback_child(args.size, args.steps)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher — one rank per GPU through torch.distributed.run on a free
        # loop-back port; the ranks' stdout (rank 0's single JSON line) passes straight through
        # (a port below the kernel's ephemeral range: one handed out by bind(0) can become the source port of somebody's outgoing connection before the rendezvous listens on it)
        port = 0
        for _try in range(64):
            cand = random.randint(20000, 29999)
            with socket.socket() as so:
                try:
                    so.bind(("127.0.0.1", cand))
                    port = cand
                    break
                except OSError:
                    pass
        if not port:
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.run(cmd, env=env).returncode)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; running with the %d rank(s) that exist" % (args.gpus, world, world), file=sys.stderr)

    import torch
    import minimd_amd

    dist = None
    invalid_reason = None          # set when the line is NOT a figure of the production data path (debug transport): "valid": false
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # control plane only; a short timeout so that a rank that died does not park the others for the default half hour
        # (gloo announces its connections on the C++ stdout: keep stdout for the ONE JSON line, send that chatter to stderr)
        with stdout_to_stderr():
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
            dist.barrier()
        ndev = max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank % ndev)
        if ndev < world and "MMD_BENCH_TRANSPORT" not in os.environ:
            # fewer GPUs than ranks: RCCL refuses two ranks on one device, so the halos fall back to the host-staged
            # transport (the JSON line says so: "transport": "host")
            os.environ["MMD_BENCH_TRANSPORT"] = "gloo"
            invalid_reason = "%d ranks share %d GPU(s): halos are staged through host memory over gloo (debug transport), not RCCL over xGMI" % (world, ndev)
            if rank == 0:
                print("bench.py: %d ranks on %d GPU(s): host-staged halos instead of RCCL" % (world, ndev), file=sys.stderr)
        if os.environ.get("MMD_BENCH_TRANSPORT") == "gloo" and invalid_reason is None:
            invalid_reason = "MMD_BENCH_TRANSPORT=gloo: halos are staged through host memory (debug transport), not RCCL over xGMI"
        if os.environ.get("MMD_BENCH_TRANSPORT") == "gloo":
            # debugging aid (e.g. two ranks sharing one GPU): host-staged halos over gloo instead of RCCL
            from minimd_amd import api
            from minimd_amd.transport import GlooTransport
            _tr = GlooTransport()
            api.sim_set_host_transport(_tr.sendrecv, _tr.allreduce, "dp")
        # data plane: the library's own RCCL communicator (ncclSend/ncclRecv halos over xGMI)
        L = minimd_amd.load_library("dp")
        if os.environ.get("MMD_BENCH_TRANSPORT") != "gloo":
            obj = [None]
            if rank == 0:
                import ctypes
                buf = ctypes.create_string_buffer(128)
                assert L.mmd_comm_unique_id(buf) == 0, L.mmd_last_error()
                obj = [buf.raw]
            dist.broadcast_object_list(obj, src=0)
            L.mmd_sim_set_unique_id(obj[0])

    # weak scaling: per-GPU sub-box stays size^3 unit cells; Comm::setup factorises the ranks 1/2/4/8 ->
    # 1x1x1 / 2x1x1 / 2x2x1 / 2x2x2 for these boxes (min surface, ref/comm.cpp:80-126)
    dims = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world)
    if dims is None:
        dims = (world, 1, 1)
    nx, ny, nz = (args.size * d for d in dims)
    sim_args = ["-i", "in.lj.miniMD", "-nx", nx, "-ny", ny, "-nz", nz, "--half_neigh", "0", "-n", args.steps]
    sim, err = None, None
    try:
        with stdout_to_stderr():          # (ncclCommInitRank prints RCCL's version banner)
            sim = minimd_amd.Sim(sim_args, precision="dp", quiet=True)
    except Exception as e:  # noqa: BLE001
        if dist is None or os.environ.get("MMD_BENCH_TRANSPORT") == "gloo":
            raise
        err = e
    if dist is not None and os.environ.get("MMD_BENCH_TRANSPORT") != "gloo":
        # the RCCL communicator may fail on some ranks only (e.g. several ranks on one device): the ranks agree over the gloo control
        # plane, and either ALL keep RCCL or ALL switch to the host-staged transport (the JSON line says which: "transport")
        ok = torch.tensor([0 if sim is None else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if rank == 0 or err is not None:
                print("bench.py rank %d: RCCL path failed on some rank (%s); every rank falls back to host-staged halos" % (rank, err), file=sys.stderr)
            if sim is not None:
                sim.close()
            # a GPU per rank was there and RCCL still did not come up: whatever follows is a debug-transport number, and the line says so
            invalid_reason = "RCCL communicator failed on some rank although %d GPU(s) serve %d ranks (%s): host-staged halos, NOT a scaling figure" % (
                max(torch.cuda.device_count(), 1), world, str(err)[:200] if err is not None else "failure on another rank")
            os.environ["MMD_BENCH_TRANSPORT"] = "gloo"
            from minimd_amd import api
            from minimd_amd.transport import GlooTransport
            _tr = GlooTransport()
            api.sim_set_host_transport(_tr.sendrecv, _tr.allreduce, "dp")
            sim = minimd_amd.Sim(sim_args, precision="dp", quiet=True)
    natoms = sim.natoms()
    for kv in filter(None, os.environ.get("MMD_BENCH_OPTIONS", "").split(",")):     # A/B knobs, e.g. "build_waves=1,fuse=1"
        k, v = kv.split("=")
        sim.handle.set_option(k, int(v))
    # the force kernel's clock: every 7th launch (library default; a launch that carries the event pair costs 11 us of gaps around it: 20 steps are timed by launches 3, 10, 17), every launch of a region shorter than 14 steps
    timed_every = 7 if args.steps >= 14 else 1
    sim.handle.set_option("time_force_sample", timed_every)
    # torch's own HIP context comes up at the first torch.cuda call: have that happen HERE, not inside the first fence() in front of the timed region
    # (the GPU would idle for the time it takes and start the first window with its clocks on the way down)
    if torch.cuda.is_available() and not os.environ.get("MMD_BENCH_LATE_TORCH"):       # (the variable: A/B of this very line)
        torch.cuda.synchronize()
    sim.initial()
    if args.equil > 0:
        sim.run_steps(args.equil)
    t_w = time.perf_counter()
    while args.clock_warm_ms > 0 and (time.perf_counter() - t_w) * 1e3 < args.clock_warm_ms:
        sim.handle.profile_kernel(0, 100)            # Force::compute only (no integrator): positions / velocities unchanged
    if args.warmup > 0:
        sim.run_steps(args.warmup)

    def fence():
        if dist is not None:
            dist.barrier()
        sim.handle.sync()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    sim.run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    tm = sim.handle.timers()
    rs = sim.handle.run_stats()
    # the device-clock stamps of the FIRST window's force launches (the window `value` is): read before the next run resets them
    clk_n = sim.handle.counter("force_clock_launches")
    k_ms_all = sim.handle.counter("force_clock_ns") * 1e-6 / clk_n if clk_n > 0 else None
    clk_ns_s, clk_n_s = sim.handle.counter("force_clock_sampled_ns"), sim.handle.counter("force_clock_sampled_launches")
    gaps = sim.handle.counter("force_clock_gaps")
    overhead_ms = sim.handle.counter("force_clock_gap_ns") * 1e-6 / gaps if gaps > 0 else None
    # further windows of the same length, back to back (each fenced like the first; `value` is the first one alone)
    window_s = [dt]
    for _ in range(max(args.windows, 1) - 1):
        fence()
        tw0 = time.perf_counter()
        sim.run_steps(args.steps)
        fence()
        tw = time.perf_counter() - tw0
        if dist is not None:
            t = torch.tensor([tw], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tw = float(t.item())
        window_s.append(tw)
    # sustained: ONE window of --sustained-steps steps (default 2000 = ~0.45 s at -s 80: 100 re-neighborings, 20 thermo rows — the reference's loop as its own
    # 1000 / 10 000-step logs run it), fenced like the others; the force kernel's device-clock span at its start, middle and end shows what the chip's clocks do
    sustained = None
    if invalid_reason is not None:
        args.sustained_steps = min(args.sustained_steps, 200)          # (debug transport, ranks sharing GPUs: tens of ms per step — the long window proves nothing there)
    if args.sustained_steps > 0:
        fence()
        ts0 = time.perf_counter()
        sim.run_steps(args.sustained_steps)
        fence()
        ts = time.perf_counter() - ts0
        if dist is not None:
            t = torch.tensor([ts], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ts = float(t.item())
        tms = sim.handle.timers()
        sustained = {"steps": args.sustained_steps, "seconds": ts, "ms_per_step": ts / args.sustained_steps * 1e3, "value": natoms * args.sustained_steps / ts / 1e6,
                     "unit": "Matom-steps/s", "vs_value": (natoms * args.sustained_steps / ts) / (natoms * args.steps / dt),
                     "phases_s": {k: tms[k] for k in ("total", "comm", "force", "neigh", "extra")},
                     # device-clock span of the force launches (first workgroup's start to last workgroup's end): mean of the window's first 100 launches, median of
                     # the kept ones (its first and last 128), mean of its last 100
                     "force_kernel_span_ms": {"first_100": sim.handle.counter("force_clock_first_ns") * 1e-6, "median_kept": sim.handle.counter("force_clock_median_ns") * 1e-6,
                                              "last_100": sim.handle.counter("force_clock_last_ns") * 1e-6}}
    nlocal, nghost, _ = sim.handle.counts()
    # per-rank view of the timed region (max over ranks of every phase, rank 0's own next to it): with these a SCALE line is
    # diagnosable from the record alone — where the time went, how often the host stalled the GPU, how many bytes the halos moved
    mine = {"rank": rank, "phases_s": {k: tm[k] for k in ("total", "comm", "force", "neigh", "extra")}, "host_syncs": rs["host_syncs"],
            "transport_syncs": rs["transport_syncs"], "bytes_sent": rs["bytes_sent"], "nlocal": nlocal, "nghost": nghost,
            # since the handle was created: fall-backs of the fixed-size message paths, the overlap trial's verdict, the RCCL bring-up check (tools/rehearsal_report.py)
            "counters": {c: sim.handle.counter(c) for c in ("exchange_fast", "exchange_overflows", "borders_direct", "borders_general", "overlap_choice", "overlap_trial_off_ns",
                                                           "overlap_trial_on_ns", "rccl_check_partners", "rccl_check_us", "halo_in_x_steps")}}
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    ninfo = sim.handle.neighbor_info()
    kbar = ninfo["total"] / max(nlocal, 1)
    bpa = algorithmic_bytes_per_atom(kbar, nghost / max(nlocal, 1))
    # the force kernel's time: EVERY launch of the timed region stamps the device clock itself (first workgroup's start, last workgroups' end: no event
    # packets on the stream) — `frac` is their average, the launch behind the neighbor build and the last one of the slice included; the event pairs on
    # every 7th launch (rocprof's notion of a kernel's duration: dispatch to completion signal) are reported next to it as the sampled figure
    k_ms_sampled = tm["force_kernel_ms"] / max(tm["force_launches"], 1)
    # a launch's device-clock span (first workgroup's start to last workgroup's end) is shorter than the duration rocprofv3 reports for it (dispatch to
    # completion): by the completion of the launch + the dispatch of its successor, i.e. by the idle time between two launches that follow each other directly —
    # which the same stamps give (mean over the back-to-back pairs of the timed region). `frac` prices every launch at span + that gap: the all-launch figure in
    # the profiler's terms (cross-checked against a trace of the same process: profiles/r05_frac_crosscheck.txt). The event pairs of the sampled clock add
    # ~9 us of marker packets to the launches that carry them: `frac_sampled` is the lower bound they give.
    k_ms_span_sampled = clk_ns_s * 1e-6 / clk_n_s if clk_n_s > 0 else None
    k_ms = (k_ms_all + overhead_ms) if (k_ms_all and overhead_ms is not None) else (k_ms_all or k_ms_sampled)
    achieved = bpa * nlocal / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
    achieved_sampled = bpa * nlocal / (k_ms_sampled * 1e-3) / 1e9 if k_ms_sampled > 0 else None
    traffic, traffic_source = None, None
    for tname in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        tpath = os.path.join(REPO, "profiles", tname)
        if world == 1 and args.size == 80 and os.path.exists(tpath):
            # HBM bytes per launch of the same kernel on the same workload from the committed rocprofv3 --pmc passes
            # (counters cannot be collected from inside the timed process); see profiles/README.md
            traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
            traffic_source = "profiles/%s (separate rocprofv3 --pmc passes of this kernel on this workload, FETCH_SIZE x2 + WRITE_SIZE; not measured in this run)" % tname
            break
    # SURVEY §8(d) kernel-only figure: 20 back-to-back launches of the force kernel alone (evflag 0, no fused integrator) on the
    # state the timed region left behind, hipEvents on the compute stream
    k_only_ms = sim.handle.profile_kernel(0, 20) if nlocal else None
    tinfo = sim.handle.transport_info()
    copy_gbs = None
    if rank == 0:
        # measured device-copy bandwidth of this GPU (read + write bytes of a 1 GiB device-to-device copy), the practical
        # ceiling next to the 8 TB/s datasheet peak
        try:
            a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
            b = torch.empty_like(a)
            b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del a, b
        except Exception:
            copy_gbs = None
    out = {
        "metric": "Matom-steps/sec (LJ, full-neigh)", "value": natoms * args.steps / dt / 1e6, "unit": "Matom-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        # false = the halos did not travel over RCCL (ranks sharing a GPU, forced debug transport, or RCCL failed): not a scaling figure
        "valid": invalid_reason is None, "reason": invalid_reason,
        # consecutive timed windows of `steps` steps each (max over ranks); value == value_windows[0]
        "value_windows": [natoms * args.steps / w / 1e6 for w in window_s],
        # one long window behind them (see --sustained-steps); `value` stays the contract's K-step window
        "sustained": sustained,
        "config": {"workload": "in.lj.miniMD -s %d per GPU (global %dx%dx%d cells, %d atoms), full neighbor list, DP, "
                               "reneigh 20, thermo 100; set-up: %d untimed equilibration steps + %.0f ms of clock warm-up before the warm-up steps" % (args.size, nx, ny, nz, natoms, args.equil, args.clock_warm_ms),
                   "parallelism": "spatial %dx%dx%d, %s" % (dims + ({"rccl": "RCCL p2p halos over xGMI", "host": "host-staged halos (debug transport)",
                                                                      "none": "single rank"}[tinfo["kind"]],)),
                   "transport": tinfo["kind"], "transport_ranks": tinfo["nranks"]},
        # the launched kernel is the LJ force WITH the velocity-Verlet update fused in (it also reads/writes v and writes the
        # new x: +80 B/atom, and skips the f store: -24 B); `achieved` still counts only SURVEY §8d's force-kernel bytes
        "roofline": {"bound": "hbm", "kernel": "k_lj_full_tile (ForceLJ::compute_fullneigh + fused Integrate)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
                     # `frac` is the in-run figure over ALL launches of the timed region (these launches also carry the integrator); `frac_sampled` the event-pair figure of
                     # every 7th launch (every launch when --steps < 14); `frac_kernel_only` = SURVEY 8(d)'s force kernel alone on the same thermalised state, 20 launches
                     "frac_in_run": (achieved / HBM_PEAK_GBS) if achieved else None, "launches_timed_every": timed_every,
                     "frac_source": ("every force launch of the timed region (%d): device-clock span of the launch + the mean idle time between launches that follow each other directly (%d pairs)" % (clk_n, gaps))
                     if (k_ms_all and overhead_ms is not None) else "event pairs on the sampled launches",
                     "kernel_span_ms_all_launches": k_ms_all, "launches_all": clk_n, "kernel_span_ms_sampled_launches": k_ms_span_sampled,
                     "dispatch_and_completion_overhead_ms": overhead_ms, "event_pair_minus_span_ms": (k_ms_sampled - k_ms_span_sampled) if k_ms_span_sampled else None,
                     "frac_span_only": (bpa * nlocal / (k_ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms_all else None,
                     "kernel_ms_sampled": k_ms_sampled, "launches_sampled": tm["force_launches"],
                     "frac_sampled": (achieved_sampled / HBM_PEAK_GBS) if achieved_sampled else None,
                     "kernel_only_ms": k_only_ms,
                     "frac_kernel_only": (bpa * nlocal / (k_only_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_only_ms else None,
                     "measured_copy_GBs": copy_gbs,
                     "kernel_ms": k_ms, "launches": clk_n if k_ms_all else tm["force_launches"], "bytes_per_atom": bpa, "atoms_per_launch": nlocal,
                     # informational: the same launches with the fused integrator's own compulsory bytes counted too
                     # (+ v read/write 48 B, + new x 32 B, - the f store it skips 24 B)
                     "achieved_incl_fused_integrator": ((bpa + 56.0) * nlocal / (k_ms * 1e-3) / 1e9) if k_ms > 0 else None,
                     "kbar": kbar, "ghost_ratio": nghost / max(nlocal, 1)},
        "phases_s": {k: tm[k] for k in ("total", "comm", "force", "neigh", "extra")},
        # max over the ranks (the slowest rank sets the step time); rebuild steps = steps/20
        "phases_s_max": {k: max(r["phases_s"][k] for r in per_rank) for k in ("total", "comm", "force", "neigh", "extra")},
        "host_syncs_per_step": max(r["host_syncs"] for r in per_rank) / max(args.steps, 1),
        "host_syncs_per_rebuild": max(r["host_syncs"] for r in per_rank) / max(args.steps // 20, 1),
        # Force::compute launches of re-neighboring steps that were enqueued behind the build, before its result words reached the host
        # (whole life of the handle; "noop" = the build's verdict on the device cancelled them and the step loop launched again)
        "force_launched_behind_build": {"launches": sim.handle.counter("spec_runs"), "noop": sim.handle.counter("spec_fails")},
        # waits of the host-staged test transport (ranks sharing a GPU): staging of messages through host memory, absent with RCCL
        "host_transport_syncs_per_step": max(r["transport_syncs"] for r in per_rank) / max(args.steps, 1),
        "halo_bytes_per_step": {"sum_over_ranks": sum(r["bytes_sent"] for r in per_rank) / max(args.steps, 1),
                                "max_rank": max(r["bytes_sent"] for r in per_rank) / max(args.steps, 1)},
        "atoms_per_rank": {"owned_min": min(r["nlocal"] for r in per_rank), "owned_max": max(r["nlocal"] for r in per_rank),
                           "ghost_max": max(r["nghost"] for r in per_rank)},
        # every rank's own view of the timed window (multi-rank runs: tools/rehearsal_report.py holds it against DESIGN.md §5.5's expectations line by line)
        "per_rank": per_rank if world > 1 else None,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.size, args.cpu_steps)
        else:
            out["cpu_baseline"] = None
    sim.close()
    if rank == 0:
        out["rank_path_loopback"] = rank_path_loopback(args.size, args.steps) if (world == 1 and not args.no_loopback and not args.no_cold) else None
        out["perf_summary_cold"] = perf_summary_cold(args.size) if (world == 1 and not args.no_cold) else None
        out["perf_summary_cold_1000_steps"] = perf_summary_cold(args.size, 1000) if (world == 1 and not args.no_cold) else None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
