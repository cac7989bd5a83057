#!/usr/bin/env python3
"""A/B of the EAM tile sweeps (config C): python tools/eam_ab.py [eam_mlo ...] — run under MMD_LIB_DIR=variants/<name> for the tuning builds
of tools/build_variant.sh (EAM_FSTRIDE, EAM_FU, EAM_FWAVES, EAM_LDS_BUDGET). Prints Matom-steps/s, force time per call, LDS bytes and workgroups per CU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minimd_amd

for mlo in [int(a) for a in sys.argv[1:]] or [-1]:
    s = minimd_amd.Sim(["-i", "in.eam.miniMD", "-s", 64, "--half_neigh", 0, "-n", 100])
    s.handle.set_option("eam_mlo", mlo)
    s.initial()
    s.run_steps(20)
    sec = s.run_steps(100)
    tm = s.handle.timers()
    h = s.handle
    print("%-10s mlo %4d: %.1f Matom-steps/s  force %.4f ms/call  LDS %d / %d B  workgroups per CU %d / %d  cmax %d  last row %s" % (
        os.environ.get("MMD_LIB_DIR", "base").split("/")[-1], mlo, s.natoms() * 100 / sec / 1e6, tm["force_kernel_ms"] / max(tm["force_launches"], 1),
        h.counter("eam_lds_density"), h.counter("eam_lds_force"), h.counter("eam_wg_density"), h.counter("eam_wg_force"), h.counter("tile_cmax"), s.rows()[-1]))
    s.close()
