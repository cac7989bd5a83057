cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/tlc; mkdir -p gpurun_out/tlc
(cd $GRAFT_REPO_ROOT/data && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tlc -o t -- $GRAFT_REPO_ROOT/minimd_amd/bin/miniMD_dp -i in.lj.miniMD -s 80 --half_neigh 0 > $GRAFT_REPO_ROOT/gpurun_out/tlc/run.log 2>&1)
grep PERF_SUMMARY gpurun_out/tlc/run.log | tail -1
DB=$(find gpurun_out/tlc -name "*.db" | head -1)
for w in 1 2; do echo "== rebuild $w"; python tools/rocpd_timeline.py $DB $w | awk '$3=="+gap" && ($4>3.0) || /window/' | cut -c1-150; done
rm -rf gpurun_out/tlc
