# phase split of k_build_rows by ablation (profiling libraries variants/profile, variants/profile_nomfma): tools/gpu_build_ablate.sh
cd $GRAFT_REPO_ROOT
for v in ${@:-profile profile_nomfma}; do
MMD_LIB_DIR=variants/$v python - <<PY
import sys; sys.path.insert(0, '.')
import minimd_amd
sim = minimd_amd.Sim(["-s", 80, "--half_neigh", 0, "-n", 40]); sim.initial(); sim.run_steps(40)
h = sim.handle
for ab in (0, 1, 2, 3, 16, 8, 32, 0):
    h.set_option("ablate", ab)
    print("$v ablate=%2d  neighbor build + binning %.4f ms" % (ab, h.profile_kernel(1, 12)))
PY
done
