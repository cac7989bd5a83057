#!/bin/bash
# One gpurun call that produces the evidence kept under profiles/ (round tag = $1, default r02):
#   bench line, kernel statistics of every BASELINE configuration, PMC passes of the hot kernels, configuration runs.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r05}
O=gpurun_out/prof_$R
mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -c 600 $O/bench_n1.json
python tools/run_configs.py > $O/configs.txt 2>&1
cat $O/configs.txt
stats() {   # name, command...
  name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt_$name -o t -- "$@" > $GRAFT_REPO_ROOT/$O/kt_$name.log 2>&1)
  python tools/rocpd_stats.py $(find $O/kt_$name -name "*.db" | head -1) $O/kernel_stats_$name.md > /dev/null
  head -8 $O/kernel_stats_$name.md
}
stats bench python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-loopback
for c in A Bh C Ch E; do stats $c python $GRAFT_REPO_ROOT/tools/run_one.py $c; done
# PMC passes (each counter group its own rocprofv3 run, --kernel-trace only)
bash tools/pmc_force.sh $O/pmc_lj_full  k_lj_full_tile   tools/prof_force.py --kernels 0 --reps 5 > $O/pmc_lj_full.txt 2>&1
bash tools/pmc_force.sh $O/pmc_lj_half  k_lj_half_tile   tools/prof_force.py --half 1 --kernels 0 --reps 5 > $O/pmc_lj_half.txt 2>&1
bash tools/pmc_force.sh $O/pmc_eam      "k_eam_.*_tile"  tools/prof_force.py --deck in.eam.miniMD --size 64 --kernels 0 --reps 5 > $O/pmc_eam.txt 2>&1
bash tools/pmc_force.sh $O/pmc_build    k_build_rows     tools/prof_force.py --steps 20 --kernels 1 > $O/pmc_build.txt 2>&1
tail -30 $O/pmc_lj_full.txt
# kernel timelines of one re-neighboring
bash tools/gpu_timeline.sh 80 > $O/timeline_reneighboring_s80.txt 2>&1
bash tools/gpu_timeline.sh 32 > $O/timeline_reneighboring_s32.txt 2>&1
rm -rf gpurun_out/tl80 gpurun_out/tl32
# the multi-rank code path on this one GPU (periodic self swaps through RCCL loop-back): kernels of plain steps and of one re-neighboring, halo form chosen by the library (overlap -1)
bash tools/gpu_loopback_timeline.sh 80 -1 tllb > /dev/null 2>&1
cp gpurun_out/tllb/timeline.txt $O/timeline_rank_path_loopback_s80.txt
rm -rf gpurun_out/tllb
timeout 400 python tools/loopback_probe.py 80 "overlap=-1" "overlap=0" "overlap=1" "overlap=0,halo_recv=1" "overlap=0,direct_borders=0,halo_recv=1" 2>&1 | grep "^-s" > $O/rank_path_loopback.txt
# host-side pricing of two proposals on real tile lists; Force::compute on rows re-ordered by distance
{ timeout 300 python tools/window_probe.py in.lj.miniMD 80 1; timeout 300 python tools/window_probe.py in.eam.miniMD 64 0; } 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostn\|^Libr" > $O/window_probe.txt
{ timeout 300 python tools/sorted_rows_probe.py in.eam.miniMD 64; timeout 300 python tools/sorted_rows_probe.py in.lj.miniMD 80; } 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostn\|^Libr" > $O/sorted_rows_probe.txt
# what the driver runs, three times, and one such slice under the profiler
for r in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null > $O/bench_driver_shape_$r.json; done
bash tools/gpu_slice.sh > $O/slice_driver_shape.txt 2>&1
# the driver's command UNDER the profiler: its own line (roofline.frac) next to rocprofv3's kernel averages of the same process
bash tools/gpu_stats_bench.sh prof_$R/stats_driver > $O/stats_driver.txt 2>&1
cp $O/stats_driver/kernel_stats_bench.md $O/kernel_stats_bench_driver_shape.md; cp $O/stats_driver/bench.json $O/bench_driver_shape_profiled.json
# keep the merge small: drop the raw rocprof trees, keep logs + summaries
find $O -name "*.db" -delete; find $O -type d -name "kt_*" -exec rm -rf {} + 2>/dev/null; find $O -mindepth 1 -maxdepth 1 -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
ls -la $O
