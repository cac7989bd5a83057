#!/usr/bin/env python3
"""Copy the evidence one `tools/gpu_profiles.sh <tag>` call left under gpurun_out/prof_<tag>/ into profiles/ (tracked) and write
profiles/<tag>_roofline.md: per-kernel average duration (rocprofv3 --kernel-trace --stats), SURVEY.md §8(d) algorithmic bytes, fraction of
the 8 TB/s HBM peak, and the PMC traffic (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md's gfx950 read correction).
usage: python tools/collect_profiles.py r02"""
import json
import os
import re
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles")
names = {"bench_n1.json": "bench_n1.json", "configs.txt": "configs.txt", "kernel_stats_bench.md": "kernel_stats_bench_s80_full_dp.md",
         "kernel_stats_A.md": "kernel_stats_A_s32_full_dp.md", "kernel_stats_Bh.md": "kernel_stats_Bhalf_s80_half_dp.md",
         "kernel_stats_C.md": "kernel_stats_C_eam_s64_full_dp.md", "kernel_stats_Ch.md": "kernel_stats_Chalf_eam_s64_half_dp.md",
         "kernel_stats_E.md": "kernel_stats_E_s160_half_sp.md", "pmc_lj_full.txt": "pmc_k_lj_full_tile.txt", "pmc_lj_half.txt": "pmc_k_lj_half_tile.txt",
         "pmc_eam.txt": "pmc_k_eam_tile.txt", "pmc_build.txt": "pmc_k_build_rows.txt",
         "timeline_reneighboring_s80.txt": "timeline_reneighboring_s80.txt", "timeline_reneighboring_s32.txt": "timeline_reneighboring_s32.txt",
         "timeline_rank_path_loopback_s80.txt": "timeline_rank_path_loopback_s80.txt", "rank_path_loopback.txt": "rank_path_loopback.txt", "window_probe.txt": "window_probe.txt", "kernel_stats_bench_driver_shape.md": "kernel_stats_bench_driver_shape.md", "bench_driver_shape_profiled.json": "bench_driver_shape_profiled.json", "sorted_rows_probe.txt": "sorted_rows_probe.txt",
         # what the driver runs (bench.py --gpus 1 --steps 20 --warmup 5), three times, and one such slice under the profiler
         "bench_driver_shape_1.json": "bench_driver_shape_1.json", "bench_driver_shape_2.json": "bench_driver_shape_2.json",
         "bench_driver_shape_3.json": "bench_driver_shape_3.json", "slice_driver_shape.txt": "slice_driver_shape.txt"}
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (tag, b)))


def stats(fn):
    out = {}
    p = os.path.join(src, fn)
    if not os.path.exists(p):
        return out
    for line in open(p):
        m = re.match(r"\| `(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?).*?` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            out[m.group(1)] = (int(m.group(2)), float(m.group(4)))          # calls, avg us
    return out


def pmc(fn, kernel_prefix, counter):
    p = os.path.join(src, fn)
    best = None
    if os.path.exists(p):
        for line in open(p):
            if kernel_prefix in line and (" " + counter + " ") in line:
                v = float(re.search(r"avg/dispatch ([\d.e+]+)", line).group(1))
                n = int(re.search(r"dispatches=(\d+)", line).group(1))
                if best is None or n > best[1]:
                    best = (v, n)
    return best[0] if best else None


bench = json.loads([l for l in open(os.path.join(src, "bench_n1.json")) if l.startswith("{")][-1])
rf = bench["roofline"]
PEAK = 8000.0
rows = []


def row(label, kernel, stats_file, natoms, bytes_per_atom, pmc_file=None, pmc_kernel=None, note=""):
    st = stats(stats_file)
    hit = [(k, v) for k, v in st.items() if k.startswith(kernel)]
    if not hit:
        return
    k, (calls, avg_us) = max(hit, key=lambda kv: kv[1][0])
    alg = bytes_per_atom * natoms
    gbs = alg / (avg_us * 1e-6) / 1e9
    traffic = None
    if pmc_file:
        f_, w_ = pmc(pmc_file, pmc_kernel or kernel, "FETCH_SIZE"), pmc(pmc_file, pmc_kernel or kernel, "WRITE_SIZE")
        if f_ is not None and w_ is not None:
            traffic = (2 * f_ + w_) * 1024          # counters are in KiB
    rows.append("| %s | `%s` | %d | %.1f | %.0f | %.0f | %.0f | %.1f %% | %s | %s |" % (
        label, k, calls, avg_us, bytes_per_atom, alg / 1e6, gbs, 100 * gbs / PEAK, ("%.0f MB" % (traffic / 1e6)) if traffic else "-", note))


N80, N64, N160, N32 = 2048000, 1048576, 16384000, 131072
row("B LJ full -s 80 DP (fused integrator)", "k_lj_full_tile<0, 1>", "kernel_stats_bench.md", N80, rf["bytes_per_atom"], "pmc_lj_full.txt", "k_lj_full_tile<0, 1>", "the roofline kernel of bench.py")
row("B LJ full -s 80 DP (force only)", "k_lj_full_tile<0, 0>", "kernel_stats_bench.md", N80, rf["bytes_per_atom"], "pmc_lj_full.txt", "k_lj_full_tile<0, 0>", "kernel-only launches (mmd_profile_kernel)")
row("A LJ full -s 32 DP", "k_lj_full_tile<0, 1>", "kernel_stats_A.md", N32, 4 * 76.3 + 60 + 0.365 * 28, note="2 081 pencil tiles: less than one full wave of workgroups")
row("B' LJ half -s 80 DP", "k_lj_half_tile<0, 1", "kernel_stats_Bh.md", N80, 269, "pmc_lj_half.txt", "k_lj_half_tile<0, 1", "LDS + L2 atomics, not HBM, bound it")
row("C EAM -s 64 DP: density sweep", "k_eam_density_tile<0, 0", "kernel_stats_C.md", N64, 4 * 60.25 + 4 + 28 + 8 + 0.167 * 28, "pmc_eam.txt", "k_eam_density_tile<0, 0", "")
row("C EAM -s 64 DP: force sweep (fused integrator)", "k_eam_force_tile<0, 1, 0", "kernel_stats_C.md", N64, 4 * 60.25 + 4 + 28 + 8 + 24 + 0.167 * 8, "pmc_eam.txt", "k_eam_force_tile<0, 1, 0", "both sweeps + fp halo: 592 B/atom; rows in two parts, core part on 18 of 20 steps")
row("E LJ half -s 160 SP", "k_lj_half_tile<0, 1", "kernel_stats_E.md", N160, 211, note="")
row("Neighbor build -s 80 DP (per rebuild)", "k_build_rows<0, 0>", "kernel_stats_bench.md", N80, 347, "pmc_build.txt", "k_build_rows<0, 0>", "instruction-issue bound (%s VALU + %s SALU + %s LDS wave-instructions per launch)" % tuple(
        ("%.2e" % v) if v else "?" for v in (pmc("pmc_build.txt", "k_build_rows<0, 0>", "SQ_INSTS_VALU"), pmc("pmc_build.txt", "k_build_rows<0, 0>", "SQ_INSTS_SALU"),
                                             pmc("pmc_build.txt", "k_build_rows<0, 0>", "SQ_INSTS_LDS"))))
row("Neighbor build -s 160 SP half (per rebuild)", "k_build_rows<2, 0>", "kernel_stats_E.md", N160, 190, note="")
hdr = ["# %s — per-kernel roofline table (8 TB/s HBM3E peak)" % tag, "",
       "Average durations: `rocprofv3 --kernel-trace --stats` of the un-modified commands (`profiles/%s_kernel_stats_*.md`); algorithmic bytes: SURVEY.md §8(d);" % tag,
       "traffic: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (`profiles/%s_pmc_*.txt`), 2 x FETCH_SIZE + WRITE_SIZE (KiB)." % tag, "",
       "| configuration | kernel | calls | avg us | B/atom | algorithmic MB | GB/s | of peak | HBM traffic (PMC) | note |", "|---|---|---:|---:|---:|---:|---:|---:|---:|---|"]
open(os.path.join(dst, tag + "_roofline.md"), "w").write("\n".join(hdr + rows) + "\n")
print("\n".join(hdr + rows))
f_, w_ = pmc("pmc_lj_full.txt", "k_lj_full_tile<0, 1>", "FETCH_SIZE"), pmc("pmc_lj_full.txt", "k_lj_full_tile<0, 1>", "WRITE_SIZE")
if f_ and w_:
    json.dump({"kernel": "k_lj_full_tile<0,1> (LJ full-neighbor force + fused integrator), in.lj.miniMD -s 80, DP",
               "FETCH_SIZE_KiB_per_launch": f_, "WRITE_SIZE_KiB_per_launch": w_,
               "hbm_bytes_per_launch": (2 * f_ + w_) * 1024,
               "note": "separate rocprofv3 --pmc passes (tools/pmc_force.sh); reads doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes), an upper bound for the gathers"},
              open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
