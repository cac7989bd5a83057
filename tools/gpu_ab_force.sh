# same-box A/B of Force::compute (mmd_profile_kernel(0), -s 80 DP full lists): product library against variants/<name>..., three rounds, every run under timeout 60
cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do for v in base "$@"; do
  if [ $v = base ]; then unset MMD_LIB_DIR; else export MMD_LIB_DIR=variants/$v; fi
  timeout 60 python tools/prof_force.py --steps 40 --kernels 0 2>&1 | grep "^force" | sed "s/^/$v: /"
done; done
