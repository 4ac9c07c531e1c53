"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table.

    python tools/rocpd_summary.py gpurun_out/prof/xxx_results.db [> profiles/rNN_kernel_stats.txt]
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':<100} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:100]:<100} {a[0]:>7d} {a[1]:>12.1f} {a[1]/a[0]:>10.2f} {a[2]:>10.2f} {a[3]:>10.2f} {100*a[1]/tot:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
