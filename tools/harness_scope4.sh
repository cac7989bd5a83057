# (on the GPU box) the ONE-RANK entries of ref/run_tests scope 4 — 10000 steps at sizes 10/16/20/30/40/60 for LJ full / LJ half / EAM full / EAM half — through tools/run_one_test.py
# (the np = 3 / 8 entries of that scope are 10000-step runs of up to 864 k atoms; on one GPU the ranks would share it over the host-staged TCP mesh: hours. np > 1 is covered at
# scope 1 by tools/harness_scope1.sh). -> gpurun_out/harness_scope4.txt
cd $GRAFT_REPO_ROOT
{
echo "# ref/run_tests scope 4, one-rank entries (10000 steps, sizes 10/16/20/30/40/60): tools/run_one_test.py <exe> 1 4 <size> 10000 <halfneigh> 0 <input>"
echo "# one MI355X, minimd_amd/bin/miniMD_dp (round 6, final library), thermo block against tests/golden/reference_output.json, pass rule of ref/run_one_test:121-138"
for inp in lj eam; do for hn in 0 1; do echo; echo "## ${inp}_half${hn}"
  for s in 10 16 20 30 40 60; do timeout 120 python -u tools/run_one_test.py minimd_amd/bin/miniMD_dp 1 4 $s 10000 $hn 0 $inp 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; done
done; done
} > gpurun_out/harness_scope4.txt 2>&1
echo "PASSED $(grep -c PASSED gpurun_out/harness_scope4.txt) FAILED $(grep -c FAILED gpurun_out/harness_scope4.txt)"
