cd $GRAFT_REPO_ROOT
{
echo "# tools/run_one_test.py --scope 4 (the reference validation procedure, ref/run_tests scope 4: 10000 steps, sizes 10/16/20/30/40/60, one rank; multi-rank entries skipped: one GPU)"
echo "# one MI355X, minimd_amd/bin/miniMD_dp (round 4, final kernels of the round: + second candidate list of EAM for every tile, kernel clock on every 7th launch), thermo block against tests/golden/reference_output.json, pass rule of ref/run_one_test:121-138"
for inp in lj eam; do for hn in 0 1; do echo; echo "## ${inp}_half${hn}"; timeout 600 python tools/run_one_test.py --scope 4 --input $inp --halfneigh $hn 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; done; done
} > gpurun_out/harness_scope4.txt 2>&1
echo "PASSED $(grep -c PASSED gpurun_out/harness_scope4.txt) FAILED $(grep -c FAILED gpurun_out/harness_scope4.txt)"
