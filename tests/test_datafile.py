"""LAMMPS data-file path (`-f / --data_file`, read_lammps_data ref/setup.cpp:55-301).

CPU tests: the parser restatement against the generating arrays (bit-exact), section/header grammar and error
behaviour, sub-box selection. GPU tests: whole runs from the data files against rows the UNMODIFIED reference printed
for the same bytes (tests/golden/datafile_runs.json, made by tests/golden/make_datafile_golden.py)."""
import json
import os
import subprocess

import numpy as np
import pytest

import datafile_fixture as fx
import minimd_amd
from minimd_amd import api

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(REPO, "tests", "golden", "datafile_runs.json")))


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not (os.path.exists(api.lib_path("dp")) and os.path.exists(api.lib_path("sp"))):
        api.build()


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("data")
    out = {}
    for name in fx.CASES:
        p = str(d / (name + ".data"))
        x, v, prd, mass, sha = fx.write_case(name, p)
        assert sha == GOLD["files"][name]["sha256"], "fixture bytes differ from the ones the golden rows were made with"
        out[name] = (p, x, v, prd, mass)
    return out


@pytest.mark.parametrize("prec", ["dp", "sp"])
@pytest.mark.parametrize("name", sorted(fx.CASES))
def test_reader_returns_the_written_arrays(files, name, prec):
    p, x, v, prd, mass = files[name]
    n, prd_r, mass_r, xr, vr = api.lammps_data_read(p, prec)
    real = np.float64 if prec == "dp" else np.float32
    assert n == len(x) == GOLD["files"][name]["natoms"]
    assert np.array_equal(prd_r, (prd - 0.0).astype(real))
    assert mass_r == (None if mass is None else real(mass))
    assert np.array_equal(xr, x.astype(real)) and np.array_equal(vr, v.astype(real))     # indexed by file id - 1


def test_subboxes_partition_the_file_in_id_order(files):
    p, x, v, prd, _ = files["lj_5x6x7"]
    n, prd_r, _, xr, vr = api.lammps_data_read(p)
    seen = []
    for lo0, hi0 in ((0.0, prd[0] / 2), (prd[0] / 2, prd[0])):
        xs, vs, t, tag = api.lammps_data_select(xr, vr, [lo0, 0, 0], [hi0, prd[1], prd[2]], ntypes=4)
        assert np.all(np.diff(tag) > 0)                                 # file id order (ref/setup.cpp:281-286)
        assert np.array_equal(xs, xr[tag - 1]) and np.array_equal(vs, vr[tag - 1])
        assert np.all((xs[:, 0] >= lo0) & (xs[:, 0] < hi0))
        assert t.min() >= 0 and t.max() <= 3
        seen.append(tag)
    assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(1, n + 1))
    # types restart from srand(5413) per call, like each reference rank (ref/ljs.cpp:110, ref/atom.cpp:97)
    _, _, t_all, _ = api.lammps_data_select(xr, vr, [0, 0, 0], prd, ntypes=4)
    _, _, t_left, _ = api.lammps_data_select(xr, vr, [0, 0, 0], [prd[0] / 2, prd[1], prd[2]], ntypes=4)
    assert np.array_equal(t_left, t_all[:len(t_left)])


def _write(tmp_path, text):
    p = tmp_path / "f.data"
    p.write_text(text)
    return str(p)


HEAD = "title\n\n2 atoms\n1 atom types\n0 5 xlo xhi\n0 6 ylo yhi\n0 7 zlo zhi\n\n"


def test_grammar_sections_in_any_order_and_comments(tmp_path):
    p = _write(tmp_path, "t\n# c\n2 atoms # two\n\n0 5.5 xlo xhi\n0 6 ylo yhi # y\n0 7 zlo zhi\n\nAtoms\n\n2 1 1 2 3\n1 1 0.5 0.25 0.125\n\n"
                         "Masses\n\n1 2.5\n\nVelocities\n\n1 -1 -2 -3\n2 4 5 6\n")
    n, prd, mass, x, v = api.lammps_data_read(p)
    assert n == 2 and list(prd) == [5.5, 6, 7] and mass == 2.5
    assert x.tolist() == [[0.5, 0.25, 0.125], [1, 2, 3]] and v.tolist() == [[-1, -2, -3], [4, 5, 6]]
    # no Velocities / Masses section: zeros and "no mass"
    n, prd, mass, x, v = api.lammps_data_read(_write(tmp_path, HEAD + "Atoms\n\n1 1 1 1 1\n2 1 2 2 2\n"))
    assert mass is None and not v.any() and x.tolist() == [[1, 1, 1], [2, 2, 2]]


@pytest.mark.parametrize("text,msg", [
    (HEAD + "Velocities\n\n1 0 0 0\n2 0 0 0\n", "Must read Atoms before Velocities"),
    (HEAD + "Bonds\n\n1 1 1 2\n", "Unknown identifier in data file: Bonds"),
    (HEAD + "Atoms\n\n1 1 1 1 1\n", "unexpected end of file"),
    (HEAD + "Atoms\n\n1 1 1 1 1\n7 1 2 2 2\n", "bad line"),
    ("title\n\n0 5 xlo xhi\n\nAtoms\n\n", "header"),
])
def test_malformed_files_are_errors_not_crashes(tmp_path, text, msg):
    with pytest.raises(api.MMDError, match=msg):
        api.lammps_data_read(_write(tmp_path, text))


def test_missing_file_is_an_error():
    with pytest.raises(api.MMDError, match="Cannot open file"):
        api.lammps_data_read("/nonexistent/x.data")


# ---- whole runs on the GPU against the reference's rows for the same bytes -------------------------------------------------
def rows_close(rows, ref, rel):
    assert [r[0] for r in rows] == [r[0] for r in ref]
    for a, b in zip(rows, ref):
        for k in (1, 2, 3):
            assert abs(a[k] - b[k]) <= rel * max(abs(b[k]), 1e-3) + 6e-7 * abs(b[k]), (a, b)   # golden rows carry 7 digits


@pytest.mark.gpu
@pytest.mark.parametrize("run", GOLD["runs"], ids=lambda r: "%s-%s-%s" % (r["case"], r["precision"], "_".join(r["args"][2:])))
def test_run_from_data_file_matches_reference_rows(files, run):
    p = files[run["case"]][0]
    s = minimd_amd.Sim(run["args"] + ["-f", p], precision=run["precision"])
    assert s.natoms() == run["natoms"]
    if "thin" in run["case"]:
        # a box thinner than half the cutoff: three ghost layers in x (ref/comm.cpp:150-152). A ghost's image code holds at most +-2 box
        # lengths, so the root + image shortcuts (one-kernel ghost update, ghosts staged from their owners) must stay off here
        assert s.handle.comm_info()["need"].tolist() == [3, 1, 1]
    s.initial(); s.run()
    rows_close(s.rows(), run["rows"], 2e-6 if run["precision"] == "dp" else 2e-4)
    nl, ng, _ = s.handle.counts()
    assert nl == run["natoms"]
    s.close()


@pytest.mark.gpu
def test_executable_with_data_file_prints_the_reference_banner(files):
    run = GOLD["runs"][0]
    p = files[run["case"]][0]
    exe = os.path.join(REPO, "minimd_amd", "bin", "miniMD_dp")
    r = subprocess.run([exe] + run["args"] + ["-f", p, "-t", "1"], cwd=os.path.join(REPO, "data"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert "\t# Datafile: %s\n" % p in out
    assert "\t# Atoms: %d\n" % run["natoms"] in out
    assert "\t# Neighbor bins: %d %d %d\n" % tuple(run["bins"]) in out          # density-derived bins (ref/setup.cpp:229-236)
    assert "\t# Density: %s\n" % run["density"] in out
