#!/bin/bash
# tools/gpu_rehearsal.sh [gpu counts, default "1 2 4 8"] [size, default 80]
# The multi-GPU evidence of BASELINE.json configs[3] (in.lj.miniMD -s 80 per rank, weak-scaled over the GPUs of one node, RCCL halos over xGMI) in ONE call on
# whatever box it is given:
#   1. bench.py --gpus N for every N of the list (N > visible GPUs: the ranks share GPUs over the debug transport and the line says "valid": false) ->
#      gpurun_out/rehearsal/bench_nN.json + tools/rehearsal_report.py: per rank, DESIGN.md §5.5's expectation next to the observed value, PASS / LOOK per line;
#      then the driver's own `bench.py --gpus 8 --steps 20 --warmup 5` and config D through the drop-in executable (8 plain ranks, 2x2x2, -s 80 per rank);
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
  [ $rc -ne 0 ] && echo "N=$N: bench.py exit status $rc: $(tail -2 $O/bench_n$N.err | cut -c1-300)" | tee -a $O/summary.txt
  # every rank's observed values next to DESIGN.md §5.5's expectations, PASS / LOOK per line
  python tools/rehearsal_report.py $O/bench_n$N.json | tee -a $O/summary.txt
done
echo "--- the driver's own scaling command (python3 bench.py --gpus 8 --steps 20 --warmup 5)" | tee -a $O/summary.txt
timeout -k 5 900 python3 bench.py --gpus 8 --steps 20 --warmup 5 > $O/bench_driver_n8.json 2> $O/bench_driver_n8.err
echo "exit status $?, $(grep -c '^{' $O/bench_driver_n8.json) JSON line(s): $(python -c "import json,sys; d=json.loads([l for l in open('$O/bench_driver_n8.json') if l.startswith('{')][-1]); print('value %.1f  valid %s  transport %s x %d  %s' % (d['value'], d['valid'], d['config']['transport'], d['config']['transport_ranks'], d.get('reason') or ''))" 2>&1 | cut -c1-400)" | tee -a $O/summary.txt
echo "--- config D through the drop-in executable: 8 plain ranks, 2x2x2, -s $S per rank, the deck's 100 steps (PERF_SUMMARY + last thermo row; reference row at -s 80 per rank: 100 6.918446e-01 -5.655620e+00 7.666032e-01)" | tee -a $O/summary.txt
EXE=$PWD/minimd_amd/bin/miniMD_dp
PORT=$((25000 + RANDOM % 2000))
for r in 0 1 2 3 4 5 6 7; do
  (cd data && OMPI_COMM_WORLD_RANK=$r OMPI_COMM_WORLD_SIZE=8 OMPI_COMM_WORLD_LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
     timeout -k 5 900 $EXE -i in.lj.miniMD -nx $((2 * S)) -ny $((2 * S)) -nz $((2 * S)) --half_neigh 0 > $OLDPWD/$O/exe8_rank$r.log 2>&1) &
done
wait
grep -h "Transport\|MPI processes\|PERF_SUMMARY\|^100 " $O/exe8_rank0.log | grep -v "^#.*MPI_proc" | tee -a $O/summary.txt
for r in 1 2 3 4 5 6 7; do grep -h "ERROR\|FAILED" $O/exe8_rank$r.log | sed "s/^/rank $r: /" | tee -a $O/summary.txt; done
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
