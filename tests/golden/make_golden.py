#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run ONLY in the build container, where
/root/reference exists; the GPU box and the test-suite only ever read the committed outputs).

Three sources, all *data* (inputs / expected outputs), no reference source text:

1. reference_output.json  — the thermo tables (step, T, U, P) and natoms of the reference's own
   known-good logs  /root/reference/tests/reference_output/{4k,...,864k}.{lj,eam}.
2. ref_runs.json          — thermo rows printed by the UNMODIFIED reference built by
   `make -C oracle ref` (oracle/_ref/miniMD_ref_{dp,sp}) for the BASELINE.json configurations that
   have no published log (-s 32, -s 80, EAM -s 64, SP -s 32) plus small cases, incl. the structural
   counts of its YAML report (nghost, total neighbors).
3. arrays_*.npz           — per-atom arrays (x, v, f, type, neighbor rows, fp ...) dumped by
   oracle/_ref/ref_dump_{dp,sp}: our driver (oracle/ref_dump.cpp) linked against the reference
   objects; small systems (-s 4 … 6).

usage:  python tests/golden/make_golden.py [--big]     (--big also runs -s 80 / EAM -s 64, ~2 min)
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
REFBIN = os.path.join(REPO, "oracle", "_ref")
DATA = os.path.join(REPO, "data")


def parse_thermo(text):
    """rows between '# Timestep T U P Time' and '# Performance Summary' -> [[step,T,U,P],...]"""
    rows, on = [], False
    for line in text.splitlines():
        if line.startswith("# Timestep"):
            on = True
            continue
        if line.startswith("# Performance Summary"):
            break
        if on:
            p = line.split()
            if len(p) >= 4:
                try:
                    rows.append([int(p[0]), float(p[1]), float(p[2]), float(p[3])])
                except ValueError:
                    pass
    return rows


def published_logs():
    out = {}
    d = os.path.join(REF, "tests", "reference_output")
    for fn in sorted(os.listdir(d)):
        if not re.match(r"^\d+k\.(lj|eam)$", fn):
            continue
        text = open(os.path.join(d, fn)).read()
        natoms = int(re.search(r"# Atoms: (\d+)", text).group(1))
        m = re.search(r"unit cells: (\d+) (\d+) (\d+)", text)
        half = int(re.search(r"# Half neighborlists: (-?\d+)", text).group(1))
        perf = [l for l in text.splitlines() if "PERF_SUMMARY" in l and not l.startswith("#")]
        out[fn] = {
            "source": "tests/reference_output/" + fn,
            "natoms": natoms,
            "size": [int(m.group(i)) for i in (1, 2, 3)],
            "half_neigh": half,
            "floatsize": int(re.search(r"# Size of float: (\d+)", text).group(1)),
            "rows": parse_thermo(text),
            "perf_summary": perf[0].split() if perf else None,
        }
    return out


def run_ref(prec, args, cwd):
    exe = os.path.join(REFBIN, "miniMD_ref_" + prec)
    r = subprocess.run([exe] + args, cwd=cwd, capture_output=True, text=True, check=True)
    return r.stdout


def ref_runs(big):
    cases = [
        # name, prec, args
        ("lj_s10_full_n1000", "dp", ["-i", "in.lj.miniMD", "-s", "10", "-n", "1000", "--half_neigh", "0"]),
        ("lj_s10_half_gn1_n1000", "dp", ["-i", "in.lj.miniMD", "-s", "10", "-n", "1000", "--half_neigh", "1", "-gn", "1"]),
        ("lj_s10_half_gn0_n1000", "dp", ["-i", "in.lj.miniMD", "-s", "10", "-n", "1000", "--half_neigh", "1", "-gn", "0"]),
        ("lj_s16_full_n300", "dp", ["-i", "in.lj.miniMD", "-s", "16", "-n", "300", "--half_neigh", "0"]),
        ("lj_s20_full_n200", "dp", ["-i", "in.lj.miniMD", "-s", "20", "-n", "200", "--half_neigh", "0"]),
        ("lj_nx12_ny8_nz10_full_n200", "dp", ["-i", "in.lj.miniMD", "-nx", "12", "-ny", "8", "-nz", "10", "-n", "200", "--half_neigh", "0"]),
        ("lj_s32_full_n100", "dp", ["-i", "in.lj.miniMD", "-s", "32", "-n", "100", "--half_neigh", "0", "-t", "8"]),
        ("lj_s32_half_n100", "dp", ["-i", "in.lj.miniMD", "-s", "32", "-n", "100", "--half_neigh", "1", "-t", "8"]),
        ("lj_s32_full_n20", "dp", ["-i", "in.lj.miniMD", "-s", "32", "-n", "20", "--half_neigh", "0", "-t", "8"]),
        ("eam_s10_full_n1000", "dp", ["-i", "in.eam.miniMD", "-s", "10", "-n", "1000", "--half_neigh", "0"]),
        ("eam_s10_half_n300", "dp", ["-i", "in.eam.miniMD", "-s", "10", "-n", "300", "--half_neigh", "1"]),
        ("eam_s16_full_n200", "dp", ["-i", "in.eam.miniMD", "-s", "16", "-n", "200", "--half_neigh", "0", "-t", "8"]),
        ("lj_s32_full_n100_sp", "sp", ["-i", "in.lj.miniMD", "-s", "32", "-n", "100", "--half_neigh", "0", "-t", "8"]),
        ("lj_s32_half_n100_sp", "sp", ["-i", "in.lj.miniMD", "-s", "32", "-n", "100", "--half_neigh", "1", "-t", "8"]),
        ("lj_s10_full_n1000_sp", "sp", ["-i", "in.lj.miniMD", "-s", "10", "-n", "1000", "--half_neigh", "0"]),
        # boxes thinner than the neighbor cutoff: need = 2 ghost layers per dimension (ref/comm.cpp:150-152), chained self swaps
        ("lj_s1_full_n60", "dp", ["-i", "in.lj.miniMD", "-s", "1", "-n", "60", "--half_neigh", "0"]),
        ("lj_1x3x2_half_n60", "dp", ["-i", "in.lj.miniMD", "-nx", "1", "-ny", "3", "-nz", "2", "-n", "60", "--half_neigh", "1"]),
        ("lj_1x3x2_full_n60", "dp", ["-i", "in.lj.miniMD", "-nx", "1", "-ny", "3", "-nz", "2", "-n", "60", "--half_neigh", "0"]),
        ("eam_2x1x3_full_n60", "dp", ["-i", "in.eam.miniMD", "-nx", "2", "-ny", "1", "-nz", "3", "-n", "60", "--half_neigh", "0"]),
        # the reference's CoMD-parameter decks (data/in.*.miniMD_comd carry the values of ref/in.*.miniMD_comd): the only LJ deck
        # with epsilon, sigma != 1 (0.167 / 2.315, cutoff 4.59, dt 5e-5) and an EAM deck at another density with skin 0.5, thermo 10
        ("lj_comd_s10_full_n1000", "dp", ["-i", "in.lj.miniMD_comd", "-s", "10", "-n", "1000", "--half_neigh", "0"]),
        ("lj_comd_s10_half_n1000", "dp", ["-i", "in.lj.miniMD_comd", "-s", "10", "-n", "1000", "--half_neigh", "1"]),
        ("lj_comd_s10_full_n300_sp", "sp", ["-i", "in.lj.miniMD_comd", "-s", "10", "-n", "300", "--half_neigh", "0"]),
        ("eam_comd_s10_full_n300", "dp", ["-i", "in.eam.miniMD_comd", "-s", "10", "-n", "300", "--half_neigh", "0"]),
        ("eam_comd_s10_half_n300", "dp", ["-i", "in.eam.miniMD_comd", "-s", "10", "-n", "300", "--half_neigh", "1"]),
    ]
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    if big:
        cases += [
            ("lj_s80_full_n100", "dp", ["-i", "in.lj.miniMD", "-s", "80", "-n", "100", "--half_neigh", "0", "-t", "8"]),
            ("lj_s80_half_n100", "dp", ["-i", "in.lj.miniMD", "-s", "80", "-n", "100", "--half_neigh", "1", "-t", "8"]),
            ("eam_s64_full_n100", "dp", ["-i", "in.eam.miniMD", "-s", "64", "-n", "100", "--half_neigh", "0", "-t", "8"]),
            # BASELINE configs[4] at its real size in DOUBLE precision (16.4 M atoms, half lists; ~5 min on 8 threads): the row the
            # device's DP run of config E is compared with digit for digit (the reference's SP sums are useless at this size)
            ("lj_s160_half_n100", "dp", ["-i", "in.lj.miniMD", "-s", "160", "-n", "100", "--half_neigh", "1", "-t", "8"]),
            # BASELINE configs[2] weak-scaled over 8 ranks (2x2x2 of -s 64 = 128^3 cells, 8.4 M atoms, 40 steps): the row the 8-rank EAM dress rehearsal is held against
            ("eam_s128_full_n40", "dp", ["-i", "in.eam.miniMD", "-s", "128", "-n", "40", "--half_neigh", "0", "-t", "8"]),
        ]
        # (ref_runs.json also holds "lj_s144_full_n100", 11.9 M atoms: its rows were taken from a separate 9-minute run of
        #  oracle/_ref/miniMD_ref_dp -i in.lj.miniMD -s 144 -n 100 --half_neigh 0 -t 8 and are kept when this script rewrites the file)
    if only:
        cases = [c for c in cases if c[0] in only[0]]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("in.lj.miniMD", "in.eam.miniMD", "in.lj.miniMD_comd", "in.eam.miniMD_comd", "Cu_u6.eam"):
            os.symlink(os.path.join(DATA, f), os.path.join(tmp, f))
        for name, prec, args in cases:
            print("ref run", name, flush=True)
            # '-o 1 --yaml_screen' prints the reference's YAML report with the structural counts
            text = run_ref(prec, args + ["-o", "1", "--yaml_screen"], tmp)
            for f in os.listdir(tmp):
                if f.endswith(".yaml"):
                    os.remove(os.path.join(tmp, f))
            natoms = int(re.search(r"# Atoms: (\d+)", text).group(1))
            ent = {"args": args, "precision": prec, "natoms": natoms, "rows": parse_thermo(text)}
            m = re.search(r"# Neighbor bins: (\d+) (\d+) (\d+)", text)
            ent["nbin"] = [int(m.group(i)) for i in (1, 2, 3)]
            for key, pat in (("nghost", r"# Nghost:\s+([-+.\deE]+) ave"),
                             ("nlocal", r"# Nlocal:\s+([-+.\deE]+) ave"),
                             ("neigh_total", r"# Total # of neighbors = ([-+.\deE]+)")):
                mm = re.search(pat, text)
                ent[key] = float(mm.group(1)) if mm else None
            out[name] = ent
    return out


def read_dump(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            hdr = f.read(24)
            if len(hdr) < 24:
                break
            name = hdr[:16].split(b"\0")[0].decode()
            dtype = chr(hdr[16])
            (count,) = struct.unpack("<q", f.read(8))
            npdt = {"i": np.int32, "d": np.float64, "f": np.float32}[dtype]
            out[name] = np.frombuffer(f.read(count * np.dtype(npdt).itemsize), dtype=npdt).copy()
    return out


def array_fixtures():
    cases = [
        # name, prec, deck, size, half, gn, nsteps, ntypes, nbins
        ("arrays_lj_s4_full", "dp", "in.lj.miniMD", 4, 0, 1, 20, 4, -1),
        ("arrays_lj_s4_half_gn1", "dp", "in.lj.miniMD", 4, 1, 1, 20, 4, -1),
        ("arrays_lj_s4_half_gn0", "dp", "in.lj.miniMD", 4, 1, 0, 20, 4, -1),
        ("arrays_lj_s6_full", "dp", "in.lj.miniMD", 6, 0, 1, 40, 4, -1),
        ("arrays_lj_s5_full_b7", "dp", "in.lj.miniMD", 5, 0, 1, 20, 1, 7),
        ("arrays_eam_s4_full", "dp", "in.eam.miniMD", 4, 0, 0, 20, 4, -1),
        ("arrays_eam_s4_half", "dp", "in.eam.miniMD", 4, 1, 0, 20, 4, -1),
        ("arrays_lj_s4_full_sp", "sp", "in.lj.miniMD", 4, 0, 1, 20, 4, -1),
        ("arrays_lj_s4_half_gn1_sp", "sp", "in.lj.miniMD", 4, 1, 1, 20, 4, -1),
    ]
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("in.lj.miniMD", "in.eam.miniMD", "Cu_u6.eam"):
            os.symlink(os.path.join(DATA, f), os.path.join(tmp, f))
        for name, prec, deck, size, half, gn, nsteps, ntypes, nbins in cases:
            print("ref dump", name, flush=True)
            out = os.path.join(tmp, name + ".bin")
            subprocess.run([os.path.join(REFBIN, "ref_dump_" + prec), deck, str(size), str(half), str(gn),
                            str(nsteps), out, str(ntypes), str(nbins)], cwd=tmp, check=True,
                           stdout=subprocess.DEVNULL)
            d = read_dump(out)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)


def main():
    big = "--big" in sys.argv
    if not os.path.isdir(REF):
        sys.exit("this script needs /root/reference (build container only)")
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "ref"], check=True, stdout=subprocess.DEVNULL)
    partial = any(a.startswith("--only=") for a in sys.argv)       # --only=name,name: add / refresh just those ref_runs entries
    if not partial:
        json.dump(published_logs(), open(os.path.join(HERE, "reference_output.json"), "w"), indent=0)
    path = os.path.join(HERE, "ref_runs.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update(ref_runs(big))
    json.dump(old, open(path, "w"), indent=0)
    if not partial:
        array_fixtures()


if __name__ == "__main__":
    main()
