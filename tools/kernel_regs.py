#!/usr/bin/env python3
"""Registers and LDS of every device kernel in the built objects (minimd_amd/build/<prec>/*.o): the code object is taken out of the host object's .hip_fatbin section
(llvm-objcopy, clang-offload-bundler) and its AMDGPU metadata notes are read (llvm-readelf --notes). On gfx950 a wavefront's VGPR budget decides the occupancy in steps of 8:
512 / 96 = 5 wavefronts per SIMD, 512 / 104 = 4 — a kernel that creeps from 96 to 100 VGPRs loses a fifth of its latency hiding (round 5: -2.4 % on the LJ step).
usage: tools/kernel_regs.py [dp|sp] [name filter]   (also imported by tests/test_host_cpu.py)"""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return [o.split("(")[0] for o in out]
    except Exception:  # noqa: BLE001
        return names


def kernels(prec="dp"):
    """{demangled kernel name: {"vgpr": n, "sgpr": n, "lds": bytes, "unit": file}}"""
    res = {}
    d = os.path.join(REPO, "minimd_amd", "build", prec)
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(os.listdir(d)):
            if not f.endswith(".o"):
                continue
            fat, co = os.path.join(tmp, f + ".fatbin"), os.path.join(tmp, f + ".co")
            r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, os.path.join(d, f)], capture_output=True)
            if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"],
                               capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            # the keys of a kernel's metadata map are sorted: .agpr_count .args ... .group_segment_fixed_size ... .sgpr_count .symbol ... .vgpr_count
            # (.name lines inside .args are argument names) — a record opens at .agpr_count (or the first key seen) and closes at .vgpr_count
            cur = {}
            rows = []
            for line in notes.splitlines():
                m = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2)
                if k in ("agpr_count", "group_segment_fixed_size", "sgpr_count", "vgpr_count"):
                    cur[k] = v
                elif k == "symbol" and v.endswith(".kd"):
                    cur["symbol"] = v[:-3]
                if k == "vgpr_count":
                    if "symbol" in cur:
                        rows.append(cur)
                    cur = {}
            names = demangle([r_["symbol"] for r_ in rows])
            for r_, n in zip(rows, names):
                res[n] = {"vgpr": int(r_["vgpr_count"]), "agpr": int(r_.get("agpr_count", 0)), "sgpr": int(r_.get("sgpr_count", 0)),
                          "lds": int(r_.get("group_segment_fixed_size", 0)), "unit": f[:-2]}
    return res


if __name__ == "__main__":
    prec = sys.argv[1] if len(sys.argv) > 1 else "dp"
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    ks = kernels(prec)
    print("# kernel registers / static LDS of minimd_amd/build/%s (gfx950): waves per SIMD = 512 // (VGPRs rounded up to 8), at most 8" % prec)
    print("%-100s %5s %5s %7s %5s  %s" % ("kernel", "VGPR", "SGPR", "LDS B", "w/SIMD", "unit"))
    for n in sorted(ks, key=lambda k: (ks[k]["unit"], k)):
        if flt in n:
            k = ks[n]
            print("%-100s %5d %5d %7d %5d  %s" % (n[:100], k["vgpr"], k["sgpr"], k["lds"], min(8, 512 // max(8, (k["vgpr"] + 7) // 8 * 8)), k["unit"]))
