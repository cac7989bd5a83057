cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/tls; mkdir -p gpurun_out/tls
(cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tls -o t -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-loopback > $GRAFT_REPO_ROOT/gpurun_out/tls/bench.log 2>&1)
tail -1 gpurun_out/tls/bench.log | cut -c1-160
python tools/rocpd_slice.py $(find gpurun_out/tls -name "*.db" | head -1) 3
rm -rf gpurun_out/tls
