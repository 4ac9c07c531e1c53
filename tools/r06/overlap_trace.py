"""Do two kernels of a step overlap?  From a rocprofv3 kernel-trace database: for the last 20 dispatches of kernel A (substring)
print start / end of A and of the next dispatch of kernel B relative to A's start.
    python tools/r06/overlap_trace.py <results.db> plane_spread_kernel rows_only_kernel"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
A, B = sys.argv[2], sys.argv[3]
ia = [i for i, r in enumerate(rows) if A in r[0]][-20:]
for i in ia:
    a = rows[i]
    near = [r for r in rows[max(0, i - 3): i + 4] if B in r[0]]
    if not near:
        continue
    b = min(near, key=lambda r: abs(r[1] - a[1]))
    print(f"{A}: 0.0 .. {(a[2] - a[1]) / 1e3:7.1f} us   {B}: {(b[1] - a[1]) / 1e3:7.1f} .. {(b[2] - a[1]) / 1e3:7.1f} us")
