"""Turn the SQ-counter summary of a profiled bench run (tools/profile_final.sh -> <tag>_sq_counters.txt, per shader engine
averages of one rocprofv3 --pmc pass) into profiles/sq_counters.json: what bench.py's roofline_valu.pmc quotes.

    python tools/sq_to_json.py profiles/<tag>_sq_counters.txt profiles/sq_counters.json [workload]

For every co-scheduled spread + pair-sum launch found (the dominant launch of a step): VALUBusy = 4 SQ_ACTIVE_INST_VALU /
(SIMDs per shader engine x SQ_BUSY_CYCLES) (MI355X_MICROARCH.md, SQ counters), vector instructions per launch
(SQ_INSTS_VALU x shader engines), the share of wave cycles spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES), the SQ clock under
the counters.  The kernel NAME is recorded: bench.py refuses to quote counters of a kernel it does not launch."""
import json
import os
import re
import sys

DOMINANT = ("plane_rows_capped_kernel", "plane_rows_kernel", "spread_rows_capped_kernel", "spread_rows_kernel", "frames_plane_rows_kernel", "frames_spread_rows_kernel")
N_SE = 32           # shader engines x XCDs the per-engine averages are over (rocpd_summary.py divides by it)
SIMD_PER_SE = 32    # 8 CUs x 4 SIMDs


def parse(src, dtype_tag="float"):
    stats, ctr = {}, {}
    for line in open(src):
        m = re.match(r"^(.*?)\s+(SQ_[A-Z_]+)\s+([0-9.]+)\s+(\d+)\s*$", line)
        if m:
            ctr.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
            continue
        m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s*$", line)
        if m:
            stats[m.group(1).strip()] = {"calls": int(m.group(2)), "avg_us": float(m.group(4)), "vgpr": int(m.group(8)),
                                         "sgpr": int(m.group(9)), "lds": int(m.group(10)), "scratch": int(m.group(11))}
    best = None
    for name, c in ctr.items():
        if dtype_tag not in name or not any(f"mipme::{k}<" in name for k in DOMINANT):
            continue
        if best is None or stats.get(name, {}).get("calls", 0) > stats.get(best, {}).get("calls", 0):
            best = name
    if best is None:
        raise SystemExit(f"no co-scheduled spread + pair-sum launch in {src}")
    c, s = ctr[best], stats.get(best, {})
    short = re.search(r"mipme::(\w+<[^>]*>)", best).group(1)
    us = s.get("avg_us")
    return {
        "kernel": short,
        "valu_busy": round(4 * c["SQ_ACTIVE_INST_VALU"] / (SIMD_PER_SE * c["SQ_BUSY_CYCLES"]), 4),
        "valu_instructions_per_launch": round(c["SQ_INSTS_VALU"] * N_SE),
        "wait_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
        "sq_clock_GHz": None if not us else round(c["SQ_BUSY_CYCLES"] / (us * 1e3), 3),
        "launch_us_under_counters": us,
        "registers": {k: s.get(k) for k in ("vgpr", "sgpr", "lds", "scratch")},
        "raw_per_shader_engine": {k: c[k] for k in sorted(c)},
        "source": src,
    }


def main(src, dst, workload="water"):
    data = json.load(open(dst)) if os.path.exists(dst) else {}
    data.setdefault("formula", "valu_busy = 4 SQ_ACTIVE_INST_VALU / (32 SIMDs per shader engine x SQ_BUSY_CYCLES); "
                               "wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; one rocprofv3 --pmc pass of bench.py, per-engine averages")
    data.setdefault("workloads", {})[workload] = parse(src, "double" if workload == "ionic" else "float")
    json.dump(data, open(dst, "w"), indent=1)
    print(json.dumps(data["workloads"][workload], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
