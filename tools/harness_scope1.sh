#!/bin/bash
# (on the GPU box) ref/run_tests scope 1 — size 10 at np 1, 3 and 8, 1000 steps — for LJ full lists, LJ half lists and EAM through tools/run_one_test.py, which starts
# np > 1 the way ref/run_one_test:50 does (mpiexec -np N ./miniMD ...): on one GPU the ranks share it over the executable's TCP mesh. -> gpurun_out/harness_scope1.txt
cd $GRAFT_REPO_ROOT
{ for cfg in "lj 0" "lj 1" "eam 0"; do set -- $cfg
    python tools/run_one_test.py --scope 1 --input $1 --halfneigh $2 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl"; echo "exit status $?"
  done; } > gpurun_out/harness_scope1.txt 2>&1
grep -c PASSED gpurun_out/harness_scope1.txt; grep -i "fail\|Transport" gpurun_out/harness_scope1.txt | sort | uniq -c
