#!/bin/bash
# kernel statistics (rocprofv3 --kernel-trace --stats) of ONE configuration of tools/run_one.py: tools/gpu_stats_one.sh <config> [outdir]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
c=$1; O=${2:-gpurun_out/stats_$c}
mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o t -- python $GRAFT_REPO_ROOT/tools/run_one.py $c > $GRAFT_REPO_ROOT/$O/run.log 2>&1)
tail -1 $O/run.log
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) $O/kernel_stats_$c.md > /dev/null
head -${3:-12} $O/kernel_stats_$c.md | cut -c1-150
find $O -name "*.db" -delete
