#!/usr/bin/env python3
"""tools/run_one_test.py — the reference's validation harness (ref/run_one_test, ref/run_tests) for the drop-in executable.

    python tools/run_one_test.py <exe> <np> <nt> <size> <nsteps> <neighlist> <ghostcomm> <input>
    python tools/run_one_test.py --scope <0..8> [--input lj|eam] [--halfneigh 0|1] [--exe minimd_amd/bin/miniMD_dp]

<exe> is run exactly as ref/run_one_test:50 runs it (`-t nt -s size -n nsteps --half_neigh neighlist -gn ghostcomm --yaml_output 0 -dm
-i in.<input>.miniMD`), the thermo block between "# Timestep T U P Time" and "# Performance Summary" is cut out of its stdout, paired
line by line with the reference log of the same system size (tests/golden/reference_output.json = the rows of the reference tree's
tests/reference_output/*.lj|*.eam) and judged by the reference's own statistical rule (ref/run_one_test:121-138): a row's |dT|, |dU|, |dP|
count as a miss when they exceed  stddev/sqrt(natoms) * sqrt(2)*(0.5 + atan2(step - d*floatsize, 50)/3.1415) + add;  the run PASSES while
misses <= 3*0.38*rows. Output lines follow the reference's ("Testfile: ...", natoms, "   PASSED (T: ..; E: ..; P: ..; Expected <=0.38)").
np > 1 is started the way ref/run_one_test:50 starts it, `${MPISTART:-mpiexec} -np N ${MPIOPTIONS} <exe> ...` (plain processes with Open MPI's
variables where no launcher exists); the executable takes its rank from the launcher's environment and says in its banner which transport the ranks
agreed on (RCCL with a GPU each, the built-in TCP mesh when they share GPUs). Exit status 0 = every run passed. --scope N runs the list of
ref/run_tests:42-151 for that scope."""
import argparse
import json
import math
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden", "reference_output.json")


def pass_rule(rows_ref, rows_test, natoms, floatsize, eam):
    """ref/run_one_test:121-138. rows: (step, T, U, P); paired line by line like `pr -m` does."""
    s_t, s_e, s_p = (13, 1300, 300) if eam else (0.4, 0.575, 3)
    d = 1000 if eam else 175
    add_t, add_e, add_p = (2e-3, 1, 0.3) if eam else (1e-5, 1e-5, 1e-5)
    rn = math.sqrt(natoms)
    t = e = p = total = 0
    for a, b in zip(rows_ref, rows_test):
        x = math.sqrt(2) * (0.5 + math.atan2(a[0] - d * floatsize, 50) / 3.1415)
        t += abs(a[1] - b[1]) > s_t / rn * x + add_t
        e += abs(a[2] - b[2]) > s_e / rn * x + add_e
        p += abs(a[3] - b[3]) > s_p / rn * x + add_p
        total += 1
    return total > 0 and (t + e + p) <= 3 * 0.38 * total, (t, e, p, total)


def thermo_block(stdout):
    """lines between '# Timestep T U P Time' and '# Performance Summary' (ref/run_one_test:67-93)"""
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


def mpistart():
    """the launcher of ref/run_one_test:50 (${MPISTART}, default mpiexec / mpirun on PATH or the image's /opt/conda/bin/mpiexec); [] when there is none"""
    import shutil
    want = os.environ.get("MPISTART", "").split()
    if want:
        return want
    for cand in (shutil.which("mpiexec"), shutil.which("mpirun"), "/opt/conda/bin/mpiexec"):
        if cand and os.path.exists(cand):
            return [cand]
    return []


def run_one(exe, nprocs, nt, size, nsteps, neighlist, ghostcomm, inp, quiet=False):
    print(" ")
    print("running miniMD test: exe=%s np=%d nt=%d size=%d nsteps=%d neighlist=%d ghostcomm=%d input=%s" % (exe, nprocs, nt, size, nsteps, neighlist, ghostcomm, inp))
    exe = os.path.abspath(exe)
    if not (os.path.isfile(exe) and os.access(exe, os.X_OK)):
        print("Did not find the miniMD executable. Aborting.")
        return False
    golden = json.load(open(GOLDEN))
    suffix = inp.split("-")[0]                       # lj-data / eam-data logs are not shipped as fixtures (1m.data is user-supplied)
    ref = [(k, v) for k, v in golden.items() if k.endswith("." + suffix) and v["size"][0] == size]
    if not ref:
        print("no reference output for size %d input %s" % (size, inp))
        return False
    name, ref = ref[0]
    print("Testfile: %s" % ref["source"])
    argv = [exe, "-t", str(nt), "-s", str(size), "-n", str(nsteps), "--half_neigh", str(neighlist), "-gn", str(ghostcomm), "--yaml_output", "0", "-dm",
            "-i", "in.%s.miniMD" % inp]
    cwd = os.path.join(REPO, "data")
    # every variable csrc/launch.cpp reads (mmd_launch_env, mmd_launch_rendezvous): the np = 1 entries must not inherit somebody's RANK / SLURM_* / PMIX_*
    launch_vars = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE",
                   "OMPI_COMM_WORLD_LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_SIZE", "PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID", "MPI_LOCALNRANKS", "PMIX_RANK", "PMIX_SIZE",
                   "PMIX_LOCAL_RANK", "PMIX_LOCAL_SIZE", "PMIX_NAMESPACE", "PMI_JOBID", "SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID", "SLURM_NTASKS_PER_NODE",
                   "SLURM_STEP_ID", "SLURM_STEP_NUM_TASKS", "SLURM_STEP_TASKS_PER_NODE", "SLURM_JOB_ID", "MMD_LAUNCHER", "MMD_NRANKS", "MMD_TRANSPORT")
    base_env = {k: v for k, v in os.environ.items() if k not in launch_vars}
    base_env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if nprocs == 1:
        outs = [subprocess.run(argv, cwd=cwd, env=base_env, capture_output=True, text=True)]
    elif mpistart():
        # exactly ref/run_one_test:50: ${MPISTART} -np $2 ${MPIOPTIONS} ./exe ... — the executable reads the launcher's environment (PMI_RANK / OMPI_COMM_WORLD_RANK
        # ...), the ranks meet on a port derived from the job, and use RCCL with a GPU each or the built-in TCP mesh when they share GPUs (named in the banner)
        cmd = mpistart() + ["-np", str(nprocs)] + os.environ.get("MPIOPTIONS", "").split() + argv
        outs = [subprocess.run(cmd, cwd=cwd, env=base_env, capture_output=True, text=True)]
    else:
        # no MPI launcher on this box: N plain processes with the variables Open MPI's mpirun would export
        procs = []
        for r in range(nprocs):
            env = dict(base_env, OMPI_COMM_WORLD_RANK=str(r), OMPI_COMM_WORLD_SIZE=str(nprocs), OMPI_COMM_WORLD_LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(argv, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = []
        for pr in procs:
            o, e = pr.communicate()
            outs.append(subprocess.CompletedProcess(argv, pr.returncode, o, e))
    out = outs[0]
    if any(o.returncode != 0 for o in outs):
        print("   FAILED (exit status %s)\n%s" % ([o.returncode for o in outs], "\n".join(o.stderr[-500:] for o in outs if o.returncode)))
        return False
    rows = thermo_block(out.stdout)
    if not rows:
        print("FAILED")
        return False
    tr = [l.strip() for l in out.stdout.splitlines() if l.strip().startswith("# Transport:")]
    if tr:
        print("   " + tr[0])
    fs = [l.split()[4] for l in out.stdout.splitlines() if l.startswith("# Size of float")]
    floatsize = int(fs[0]) if fs else 8
    print(ref["natoms"])
    ok, (t, e, p, total) = pass_rule([tuple(r) for r in ref["rows"]], rows, ref["natoms"], floatsize, suffix == "eam")
    if ok:
        print("   PASSED (T: %g ; E: %g ; P: %g ; Expected <=0.38)" % (t / total, e / total, p / total))
    else:
        print("   Failed (%g ; %g ; %g ; Expected 0.32+-0.06)" % (t / max(total, 1), e / max(total, 1), p / max(total, 1)))
    if not quiet and not ok:
        for a, b in zip(ref["rows"], rows):
            print("     ref %s   test %s" % (a, b))
    return ok


def scope_runs(scope):
    """(np, size) list and nsteps / threads of ref/run_tests:42-151"""
    nsteps, threads, mpi_size = {0: (100, 1, 0), 1: (1000, 1, 1), 2: (100, 1, 2), 3: (1000, 1, 2), 4: (10000, 4, 2), 5: (1000, 4, 1),
                                 6: (100, 4, 2), 7: (1000, 4, 2), 8: (10000, 4, 2)}[scope]
    sizes_more = [16, 20, 30, 40, 60]
    runs = [(1, 10)] + ([(1, s) for s in sizes_more] if mpi_size > 1 else [])
    for nprocs in (3, 8):
        if mpi_size > 0:
            runs.append((nprocs, 10))
        if mpi_size > 1:
            runs += [(nprocs, s) for s in sizes_more]
    return nsteps, threads, runs


def main():
    if len(sys.argv) == 9 and not sys.argv[1].startswith("--"):
        a = sys.argv
        sys.exit(0 if run_one(a[1], int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), int(a[7]), a[8]) else 1)
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scope", type=int, default=0)
    ap.add_argument("--input", default="lj")
    ap.add_argument("--halfneigh", type=int, default=0)
    ap.add_argument("--exe", default=os.path.join(REPO, "minimd_amd", "bin", "miniMD_dp"))
    args = ap.parse_args()
    print(" ")
    print("running miniMD tests scope=%d input=%s halfneigh=%d" % (args.scope, args.input, args.halfneigh))
    nsteps, threads, runs = scope_runs(args.scope)
    bad = 0
    for nprocs, size in runs:
        bad += 0 if run_one(args.exe, nprocs, threads, size, nsteps, args.halfneigh, 0, args.input) else 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
