#!/usr/bin/env python3
"""Golden thermo rows for the LAMMPS data-file path: runs the UNMODIFIED reference (oracle/_ref/miniMD_ref_{dp,sp},
built by `make -C oracle ref`) on the data files of tests/datafile_fixture.py and stores the rows it prints, the
banner's neighbor-bin line and the files' sha256 in datafile_runs.json. Build container only.

usage: python tests/golden/make_datafile_golden.py"""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "tests"))
import datafile_fixture as fx  # noqa: E402
from make_golden import parse_thermo  # noqa: E402

RUNS = [
    ("lj_5x6x7", "dp", ["-i", "in.lj.miniMD", "-n", "200", "--half_neigh", "0"]),
    ("lj_5x6x7", "dp", ["-i", "in.lj.miniMD", "-n", "200", "--half_neigh", "1"]),
    ("lj_5x6x7", "dp", ["-i", "in.lj.miniMD", "-n", "100", "--half_neigh", "0", "-b", "4"]),
    ("lj_5x6x7", "sp", ["-i", "in.lj.miniMD", "-n", "200", "--half_neigh", "0"]),
    ("eam_4x5x5", "dp", ["-i", "in.eam.miniMD", "-n", "100", "--half_neigh", "0"]),
    ("lj_thin_1x6x7", "dp", ["-i", "in.lj.miniMD", "-n", "100", "--half_neigh", "0"]),
    ("lj_thin_1x6x7", "dp", ["-i", "in.lj.miniMD", "-n", "100", "--half_neigh", "1"]),
]


def main():
    out = {"files": {}, "runs": []}
    with tempfile.TemporaryDirectory() as tmp:
        paths = {}
        for name in fx.CASES:
            paths[name] = os.path.join(tmp, name + ".data")
            x, v, prd, mass, sha = fx.write_case(name, paths[name])
            out["files"][name] = {"sha256": sha, "natoms": len(x), "prd": [float(p) for p in prd], "mass": mass}
        for name, prec, args in RUNS:
            exe = os.path.join(REPO, "oracle", "_ref", "miniMD_ref_" + prec)
            cmd = [exe] + args + ["-f", paths[name], "-t", "1"]
            r = subprocess.run(cmd, cwd=os.path.join(REPO, "data"), capture_output=True, text=True, timeout=600)
            rows = parse_thermo(r.stdout)
            bins = re.search(r"# Neighbor bins: (\d+) (\d+) (\d+)", r.stdout)
            natoms = re.search(r"# Atoms: (\d+)", r.stdout)
            dens = re.search(r"# Density: ([0-9.eE+-]+)", r.stdout)
            assert rows and bins, r.stdout[-2000:] + r.stderr[-2000:]
            out["runs"].append({"case": name, "precision": prec, "args": args, "rows": rows, "bins": [int(b) for b in bins.groups()],
                                "natoms": int(natoms.group(1)), "density": dens.group(1)})
            print(name, prec, " ".join(args), "bins", bins.groups(), "rows", rows[-1])
    with open(os.path.join(HERE, "datafile_runs.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
