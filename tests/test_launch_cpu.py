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
               "SLURM_JOB_ID", "SLURM_STEP_ID"]


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
             "slurm": ("SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID"), "torchrun": ("RANK", "WORLD_SIZE", "LOCAL_RANK")}[launcher]
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
