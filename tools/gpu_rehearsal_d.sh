#!/bin/bash
# tools/gpu_rehearsal_d.sh [size per rank, default 80] — BASELINE configs[3] at its REAL geometry on whatever box this is:
#   1. the drop-in executable as 8 plain ranks (2x2x2, -s S per rank, full lists, DP, the deck's 100 steps), started the way mpirun starts them;
#      on fewer than 8 GPUs the ranks share devices over the TCP mesh (banner: DEBUG transport). Last thermo row against the reference's own row
#      for that box (tests/golden/ref_runs.json: lj_s160_half_n100 — mode independence, tests/reference_output/README:3-5).
#   2. the driver's exact multi-GPU command `python3 bench.py --gpus 8 --steps 20 --warmup 5` -> one JSON line (valid: false + reason on < 8 GPUs).
# Output: gpurun_out/rehearsal_d/{exe_rank*.log,bench_n8.json,summary.txt}
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
S=${1:-80}
O=$PWD/gpurun_out/rehearsal_d
rm -rf $O; mkdir -p $O
EXE=$PWD/minimd_amd/bin/miniMD_dp
PORT=$((23000 + RANDOM % 2000))
t0=$(date +%s.%N)
for r in 0 1 2 3 4 5 6 7; do
  (cd data && OMPI_COMM_WORLD_RANK=$r OMPI_COMM_WORLD_SIZE=8 OMPI_COMM_WORLD_LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
     timeout -k 5 1200 $EXE -i in.lj.miniMD -nx $((2 * S)) -ny $((2 * S)) -nz $((2 * S)) --half_neigh 0 > $O/exe_rank$r.log 2>&1) &
done
wait
t1=$(date +%s.%N)
{ echo "# 8 ranks of miniMD_dp -nx $((2*S)) -ny $((2*S)) -nz $((2*S)) --half_neigh 0: wall $(echo "$t1 - $t0" | bc) s"
  grep -h "Transport\|MPI processes\|PERF_SUMMARY\|^[0-9]* [0-9.e+-]* [0-9.e+-]* [0-9.e+-]*" $O/exe_rank0.log | grep -v "^#.*MPI_proc"
  for r in 1 2 3 4 5 6 7; do tail -2 $O/exe_rank$r.log | sed "s/^/rank $r: /"; done; } | tee $O/summary.txt
t0=$(date +%s.%N)
timeout -k 5 1500 python3 bench.py --gpus 8 --steps 20 --warmup 5 --size $S > $O/bench_n8.json 2> $O/bench_n8.err
rc=$?
t1=$(date +%s.%N)
{ echo "# python3 bench.py --gpus 8 --steps 20 --warmup 5: exit $rc, wall $(echo "$t1 - $t0" | bc) s, $(grep -c '^{' $O/bench_n8.json) JSON line(s)"
  cut -c1-3000 $O/bench_n8.json; tail -5 $O/bench_n8.err; } | tee -a $O/summary.txt
