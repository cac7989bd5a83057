# same-box A/B of the neighbor build (mmd_profile_kernel(1): build + binning, repeated on resident data): product library against variants/<name>...
cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do for v in base "$@"; do
  if [ $v = base ]; then unset MMD_LIB_DIR; else export MMD_LIB_DIR=variants/$v; fi
  timeout 40 python tools/prof_force.py --steps 40 --kernels 1 2>&1 | grep "neighbor_build" | sed "s/^/$v: /"
done; done
