"""The process-level contract of a multi-rank run, on CPU: the reference's harness starts `${MPISTART} -np N ./miniMD ...` (ref/run_one_test:50) and the
program asks MPI who it is (ref/ljs.cpp:63-68). The drop-in reads the launcher's environment instead (csrc/launch.cpp) and the ranks meet on a TCP mesh
that also serves as the debug transport for ranks sharing a GPU. Here: plain processes under each launcher's variables, and under the image's own
`mpiexec` (MPICH hydra), push Comm::setup's swap pattern through that mesh."""
import json
import os
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(REPO, "tests", "mesh_worker.py")
LAUNCH_VARS = ["RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK",
               "PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID", "PMIX_RANK", "SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID", "MASTER_ADDR", "MASTER_PORT",
               "SLURM_JOB_ID", "SLURM_STEP_ID", "SLURM_STEP_NUM_TASKS", "SLURM_STEP_TASKS_PER_NODE", "PMIX_SIZE", "MMD_LAUNCHER", "MMD_NRANKS"]


def clean_env():
    return {k: v for k, v in os.environ.items() if k not in LAUNCH_VARS}


def collect(outdir, world):
    res = [json.load(open(os.path.join(outdir, "rank%d.json" % r))) for r in range(world)]
    for r, x in enumerate(res):
        assert x["rank"] == r and x["world"] == world and x["errors"] == [], x
    return res


def mpiexec():
    for cand in (shutil.which("mpiexec"), "/opt/conda/bin/mpiexec"):
        if cand and os.path.exists(cand):
            return cand
    return None


@pytest.mark.parametrize("launcher,world", [("openmpi", 3), ("pmi", 2), ("slurm", 4), ("torchrun", 8)])
def test_plain_processes_under_a_launchers_environment(launcher, world, tmp_path, port):
    """N plain processes, each with only the variables one launcher would export; MASTER_* set for torchrun only — the others meet on the port derived
    from the job id (Slurm) or the common parent's pid."""
    names = {"openmpi": ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK"), "pmi": ("PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID"),
             "slurm": ("SLURM_PROCID", "SLURM_STEP_NUM_TASKS", "SLURM_LOCALID"), "torchrun": ("RANK", "WORLD_SIZE", "LOCAL_RANK")}[launcher]
    procs = []
    for r in range(world):
        env = clean_env()
        env[names[0]], env[names[1]], env[names[2]] = str(r), str(world), str(r)
        if launcher == "torchrun":
            env["MASTER_ADDR"], env["MASTER_PORT"] = "127.0.0.1", str(port)
        if launcher == "slurm":
            env["SLURM_JOB_ID"], env["SLURM_STEP_ID"] = str(100000 + port), "0"
        procs.append(subprocess.Popen([sys.executable, WORKER, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
    res = collect(str(tmp_path), world)
    assert all(x["launcher"] == launcher for x in res)
    assert len({x["port"] for x in res}) == 1
    if launcher == "torchrun":
        assert res[0]["port"] == port


@pytest.mark.skipif(mpiexec() is None, reason="no mpiexec in this image")
def test_under_mpiexec(tmp_path):
    """`mpiexec -np 3 <plain program>`: the launcher of ref/run_one_test:50 on this image (MPICH hydra exports PMI_RANK / PMI_SIZE / MPI_LOCALRANKID); no
    MASTER_* anywhere — the ranks meet on the port derived from their common parent, the hydra proxy."""
    r = subprocess.run([mpiexec(), "-np", "3", sys.executable, WORKER, str(tmp_path)], env=clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    res = collect(str(tmp_path), 3)
    assert all(x["launcher"] == "pmi" for x in res) and [x["local_rank"] for x in res] == [0, 1, 2]


def _launch_env(extra):
    """mmd_launch_env as a fresh process sees it under `extra` (the library caches nothing, but the environment must be the process's own)"""
    code = "import sys, json; sys.path.insert(0, %r); from minimd_amd import api; print(json.dumps(api.launch_env()))" % REPO
    r = subprocess.run([sys.executable, "-c", code], env=dict(clean_env(), **extra), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_a_batch_shell_is_a_singleton():
    """`sbatch -n 4` exports SLURM_PROCID=0 / SLURM_NTASKS=4 to the batch script itself: a bare ./miniMD started there is ONE rank for MPI (MPI_Init without
    mpiexec: a singleton, ref/ljs.cpp:63-68) and for the drop-in — the Slurm variables count only inside a job step (numeric SLURM_STEP_ID, size from
    SLURM_STEP_NUM_TASKS). Round-5 advisor finding."""
    batch = {"SLURM_PROCID": "0", "SLURM_NTASKS": "4", "SLURM_LOCALID": "0", "SLURM_JOB_ID": "4711"}
    le = _launch_env(batch)
    assert (le["rank"], le["nranks"], le["launcher"]) == (0, 1, "single"), le
    le = _launch_env(dict(batch, SLURM_STEP_ID="batch"))                       # (the batch step's id is not a number)
    assert le["nranks"] == 1, le
    le = _launch_env(dict(batch, SLURM_STEP_ID="0", SLURM_STEP_NUM_TASKS="2", SLURM_PROCID="1"))      # srun -n 2 inside the 4-task allocation
    assert (le["rank"], le["nranks"], le["launcher"]) == (1, 2, "slurm"), le


def test_explicit_override_makes_a_singleton():
    """a stale RANK / WORLD_SIZE (a shell that once ran under torchrun) would make a bare start wait for ranks that never come: MMD_LAUNCHER=none or
    MMD_NRANKS=1 says so; MMD_LAUNCHER=<name> believes only that launcher's variables."""
    stale = {"RANK": "0", "WORLD_SIZE": "8", "LOCAL_RANK": "0"}
    assert _launch_env(stale)["nranks"] == 8
    assert _launch_env(dict(stale, MMD_LAUNCHER="none"))["nranks"] == 1
    assert _launch_env(dict(stale, MMD_NRANKS="1"))["launcher"] == "single"
    le = _launch_env(dict(stale, MMD_LAUNCHER="openmpi", OMPI_COMM_WORLD_RANK="2", OMPI_COMM_WORLD_SIZE="3"))
    assert (le["rank"], le["nranks"], le["launcher"]) == (2, 3, "openmpi"), le


MESH_CODE = ("import sys; sys.path.insert(0, %r); from minimd_amd import api\n"
             "le = api.launch_env(); a, p = api.launch_rendezvous()\n"
             "m = api.Mesh(le['rank'], le['nranks'], a, p); print('mesh up', le['rank']); m.close()\n") % REPO


@pytest.mark.parametrize("rank", [0, 1])
def test_rendezvous_names_the_variables_it_believed(rank, port):
    """a rank of 2 whose partner never comes (a stale WORLD_SIZE, a batch shell): the error says where rank and size came from and how to run a single rank,
    instead of a bare time-out."""
    env = dict(clean_env(), RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MMD_MESH_PATIENCE="2")
    r = subprocess.run([sys.executable, "-c", MESH_CODE], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in r.stderr and "MMD_LAUNCHER=none" in r.stderr, r.stderr[-1500:]


def test_a_silent_stranger_does_not_stall_the_rendezvous(port):
    """A stranger that connects to rank 0's port and says nothing (a port scanner, another job) is dropped after a few seconds; the job's own rank still
    gets in. Round-5 advisor finding."""
    import socket
    import time
    code = MESH_CODE
    env = dict(clean_env(), OMPI_COMM_WORLD_RANK="0", OMPI_COMM_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p0 = subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    # a silent stranger on the rendezvous port (+17, launch.cpp), then the real rank 1
    silent = None
    for _ in range(200):
        try:
            silent = socket.create_connection(("127.0.0.1", port + 17), timeout=1)
            break
        except OSError:
            time.sleep(0.05)
    assert silent is not None
    time.sleep(0.3)
    p1 = subprocess.Popen([sys.executable, "-c", code], env=dict(env, OMPI_COMM_WORLD_RANK="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    o0, e0 = p0.communicate(timeout=120)
    o1, e1 = p1.communicate(timeout=120)
    silent.close()
    assert p0.returncode == 0 and p1.returncode == 0 and "mesh up 0" in o0 and "mesh up 1" in o1, (e0[-1500:], e1[-1500:])


def test_rehearsal_report_reads_a_multi_rank_bench_line():
    """tools/rehearsal_report.py on the bench line of the 8-rank dress rehearsal kept under profiles/ (one GPU, debug transport): the geometry lines of every rank
    PASS (owned atoms, ghosts, halo bytes per step, synchronisations per re-neighboring, no overflow, direct borders, buckets partitioning the wall clock), the
    transport / timing lines say LOOK — the page a first multi-GPU lease is read with."""
    import importlib.util
    import io
    spec = importlib.util.spec_from_file_location("rehearsal_report", os.path.join(REPO, "tools", "rehearsal_report.py"))
    rr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rr)
    out = io.StringIO()
    ok = rr.report(os.path.join(REPO, "profiles", "r06_rehearsal_bench_n8.json"), out)
    text = out.getvalue()
    assert ok is False and text.count(" rank ") == 8
    for what in ("owned atoms", "ghost atoms", "halo bytes per step", "host syncs per re-neighboring   ", "exchange: fixed-size messages overflowed", "borders as one exchange", "buckets partition the wall clock"):
        rows = [l for l in text.splitlines() if l.strip().startswith(what)]
        assert len(rows) == 8 and all(l.rstrip().endswith("PASS") for l in rows), (what, rows[:2])
    assert "halos over RCCL (valid)" in text and [l for l in text.splitlines() if "halos over RCCL" in l][0].rstrip().endswith("LOOK")
