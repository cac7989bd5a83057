#!/usr/bin/env python3
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
for i, (n, s, e) in enumerate(rows):
    if "k_build_rows" in n:
        prev = " <- ".join("%s %.0f" % (rows[j][0].split("(")[0].replace("void ", "")[:22], (rows[j][2] - rows[j][1]) / 1e3) for j in range(i - 1, max(i - 4, -1), -1))
        nxt = rows[i + 1][0].split("(")[0][:30] if i + 1 < len(rows) else ""
        print("k_build_rows %7.1f us   gap before %5.1f   after: %s   before: %s" % ((e - s) / 1e3, (s - rows[i - 1][2]) / 1e3, nxt, prev))
