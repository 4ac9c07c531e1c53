"""Summarise a rocprofv3 rocpd sqlite database into per-kernel tables.

    python tools/rocpd_summary.py <results.db> [<results2.db> ...]

Kernel-trace databases give calls / total / avg / min / max duration and the register footprint;
databases collected with --pmc also give the per-dispatch average of every counter.
"""
import sqlite3
import sys


def summarise(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, vgpr_count, sgpr_count, lds_size, scratch_size from kernels").fetchall()
    agg = {}
    for name, s, e, vg, sg, lds, scr in rows:
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0, vg, sg, lds, scr])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1.0
    print(f"# {path}")
    print(f"{'kernel':<90} {'calls':>6} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6} {'vgpr':>5} {'sgpr':>5} {'lds':>6} {'scr':>5}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:90]:<90} {a[0]:>6d} {a[1]:>11.1f} {a[1]/a[0]:>9.2f} {a[2]:>9.2f} {a[3]:>9.2f} {100*a[1]/tot:>6.2f} {a[4]:>5} {a[5]:>5} {a[6]:>6} {a[7]:>5}")
    try:
        pm = c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        print()
        print(f"{'kernel':<90} {'counter':<22} {'avg/dispatch':>16} {'dispatches':>10}")
        for name, cn, v, n in sorted(pm, key=lambda r: (r[0], r[1])):
            print(f"{name[:90]:<90} {cn:<22} {v:>16.1f} {n:>10d}")
    print()


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
