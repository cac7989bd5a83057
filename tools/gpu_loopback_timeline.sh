#!/bin/bash
# tools/gpu_loopback_timeline.sh <size> <overlap> [outname]: kernel timeline (plain steps + one re-neighboring) of the multi-rank code path in RCCL loop-back on one GPU
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=${1:-80}; OV=${2:-0}; O=gpurun_out/${3:-tllb}
rm -rf $O; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O -o t -- python $GRAFT_REPO_ROOT/tools/loopback_trace.py $S $OV > /dev/null 2>&1)
DB=$(find $O -name "*.db" | head -1)
{ echo "# tools/loopback_trace.py $S $OV under rocprofv3 --kernel-trace: one rank of in.lj.miniMD -s $S whose self swaps go through RCCL (force_transport), overlap $OV"
  echo "## plain steps"; python tools/rocpd_steps.py $DB 1 | head -14; echo "## one re-neighboring"; python tools/rocpd_timeline.py $DB; } > $O/timeline.txt 2>&1
cat $O/timeline.txt
