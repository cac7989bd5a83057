#!/bin/bash
# tools/gpu_rehearsal.sh [gpu counts, default "1 2 4 8"] [size, default 80]
# The multi-GPU evidence of BASELINE.json configs[3] (in.lj.miniMD -s 80 per rank, weak-scaled over the GPUs of one node, RCCL halos over xGMI) in ONE call on
# whatever box it is given:
#   1. bench.py --gpus N for every N of the list (N > visible GPUs: the ranks share GPUs over the debug transport and the line says "valid": false) ->
#      gpurun_out/rehearsal/bench_nN.json + one summary line each: value, phases_s_max, host_syncs_per_rebuild, halo_bytes_per_step, transport, valid;
#   2. the RCCL parity tests that wait for two visible GPUs (tests/test_gpu_more.py -k rccl_two_gpus; skipped, and reported as skipped, on one GPU);
#   3. a kernel trace of a 2-rank run of the drop-in executable (one rocprofv3 per rank) -> per rank the kernels of a few plain steps and of one
#      re-neighboring window (tools/rocpd_steps.py, tools/rocpd_timeline.py) in gpurun_out/rehearsal/timeline_rank*.txt.
# What the numbers should be on real GPUs is in DESIGN.md §5 ("first run on more than one device").
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIST=${1:-"1 2 4 8"}; S=${2:-80}
O=gpurun_out/rehearsal
rm -rf $O; mkdir -p $O
NDEV=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null)
echo "visible GPUs: $NDEV" | tee $O/summary.txt
for N in $LIST; do
  extra="--no-cold --no-loopback"; [ $N -gt 1 ] && extra="$extra --no-cpu-baseline"
  timeout -k 5 900 python bench.py --gpus $N --size $S --steps 100 --warmup 20 $extra > $O/bench_n$N.json 2> $O/bench_n$N.err
  rc=$?
  python - $O/bench_n$N.json $N $rc <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("N=%s value %.1f Matom-steps/s  %.4f ms/step  valid %s  transport %s  phases_s_max %s  host_syncs_per_rebuild %.1f  halo B/step max rank %.0f  atoms/rank %s%s" % (
        d["n_gpus"], d["value"], d["ms_per_step"], d["valid"], d["config"]["transport"], {k: round(v, 5) for k, v in d["phases_s_max"].items()}, d["host_syncs_per_rebuild"],
        d["halo_bytes_per_step"]["max_rank"], d["atoms_per_rank"], ("  REASON: " + d["reason"]) if d.get("reason") else ""))
except Exception as e:
    print("N=%s FAILED (exit status %s): %r" % (sys.argv[2], sys.argv[3], e))
PY
done
echo "--- RCCL parity between two devices" | tee -a $O/summary.txt
timeout -k 5 1200 python -m pytest tests/test_gpu_more.py -m gpu -q -k "rccl_two_gpus" -rs 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -6 | tee -a $O/summary.txt
echo "--- kernel trace of a 2-rank run of the drop-in executable (one rocprofv3 per rank, plain processes started the way mpirun starts them)" | tee -a $O/summary.txt
# (each rank under its own profiler with its own output directory: two ranks writing ONE rocpd database block each other — seen as a hang)
EXE=$PWD/minimd_amd/bin/miniMD_dp
PORT=$((23000 + RANDOM % 2000))
for r in 0 1; do
  (cd data && OMPI_COMM_WORLD_RANK=$r OMPI_COMM_WORLD_SIZE=2 OMPI_COMM_WORLD_LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
     timeout -k 5 300 rocprofv3 --kernel-trace -d $OLDPWD/$O/kt_rank$r -o t -- $EXE -i in.lj.miniMD -nx $((2 * S)) -ny $S -nz $S --half_neigh 0 -n 100 > $OLDPWD/$O/trace_rank$r.log 2>&1) &
done
wait
grep -h "Transport\|PERF_SUMMARY" $O/trace_rank0.log | grep -v "^#.*MPI_proc" | tee -a $O/summary.txt
for r in 0 1; do
  db=$(find $O/kt_rank$r -name "*.db" | head -1)
  if [ -n "$db" ]; then
    { echo "# rank $r of 2: miniMD_dp -nx $((2 * S)) -ny $S -nz $S --half_neigh 0 under rocprofv3 --kernel-trace"; echo "## plain steps"; python tools/rocpd_steps.py $db 1 | head -16
      echo "## one re-neighboring"; python tools/rocpd_timeline.py $db; } > $O/timeline_rank$r.txt 2>&1
    echo "timeline_rank$r.txt: $(grep 're-neighboring window' $O/timeline_rank$r.txt)" | tee -a $O/summary.txt
  else echo "rank $r: no trace database (see $O/trace_rank$r.log)" | tee -a $O/summary.txt; fi
done
find $O -name "*.db" -delete
echo "summary: $O/summary.txt"
