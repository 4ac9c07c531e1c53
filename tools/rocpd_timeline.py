"""Print the kernel timeline of one step (between two launches of an anchor kernel) in the middle found in a rocprofv3 rocpd database:
start offset, duration and the idle gap before every kernel (microseconds).

    python tools/rocpd_timeline.py <results.db> [first_kernel_substring]
"""
import sqlite3
import sys


def main(path, anchor="distance_forward"):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    starts = [k for k, r in enumerate(rows) if anchor in r[0]]
    if len(starts) < 3:
        print("anchor kernel not found")
        return
    mid = len(starts) // 2
    a, b = starts[mid], starts[mid + 1]  # one full step in the middle of the run
    t0 = rows[a][1]
    prev_end = None
    busy = 0.0
    for name, s, e in rows[a:b]:
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        busy += (e - s) / 1e3
        print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}  {name[:100]}")
        prev_end = e
    print(f"step span {(rows[b][1] - t0) / 1e3:.2f} us, kernels busy {busy:.2f} us, {b - a} kernels")


if __name__ == "__main__":
    main(*sys.argv[1:])
