cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/dbgtl; mkdir -p gpurun_out/dbgtl
(cd /tmp && timeout -k 5 90 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/dbgtl -o t -- python $GRAFT_REPO_ROOT/tools/dbg/dbg_flag.py ${1:-40} > $GRAFT_REPO_ROOT/gpurun_out/dbgtl/log.txt 2>&1)
tail -3 gpurun_out/dbgtl/log.txt
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/dbgtl/**/*.db", recursive=True)[0]
rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_halo_flag_set" in r[0]]
print("flag kernels", len(idx))
i0 = idx[0] if idx else len(rows) - 12
t0 = rows[max(i0 - 6, 0)][1]
for n, s, e in rows[max(i0 - 6, 0): i0 + 8]:
    print("%10.1f us  dur %10.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n.split("(")[0][:60]))
PY
