cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for ev in 7 1; do      # force-kernel clock on every 7th / on every launch (option time_force_sample)
for S in 80 32; do
rm -rf gpurun_out/tlp; mkdir -p gpurun_out/tlp
(cd /tmp && MMD_SIM_OPTIONS=time_force_sample=$ev rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tlp -o t -- python $GRAFT_REPO_ROOT/tools/run_configs.py "$( [ $S = 80 ] && echo 'B ' || echo 'A ')" > /dev/null 2>&1)
echo "== time_force_sample $ev size $S"; python tools/rocpd_steps.py $(find gpurun_out/tlp -name "*.db" | head -1) 1 | head -8
done; done
rm -rf gpurun_out/tlp
