#!/bin/bash
# kernel timeline of one re-neighboring at -s $1 (default 32)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=${1:-32}
mkdir -p gpurun_out/tl$S
(cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl$S -o t -- python $GRAFT_REPO_ROOT/bench.py --size $S --steps 60 --warmup 20 --equil 40 --no-cpu-baseline --no-loopback > $GRAFT_REPO_ROOT/gpurun_out/tl$S/bench.log 2>&1)
tail -1 gpurun_out/tl$S/bench.log | cut -c1-200
python tools/rocpd_timeline.py $(find gpurun_out/tl$S -name "*.db" | head -1)
