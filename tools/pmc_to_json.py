"""Turn the PMC summary written by tools/profile_gpu.sh into profiles/pmc_traffic.json (HBM bytes per launch).

    python tools/pmc_to_json.py gpurun_out/<tag>_pmc.txt profiles/pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
reports half the bytes of wide (16 B/lane) coalesced streaming reads; our stream kernels read 8-byte index pairs +
4-byte reals, and comparing with their known byte counts shows the same factor 1/2 (distance_forward: 49.5 MB raw
vs 95 MB streamed; rspace_backward: 29 MB vs 57 MB), while the row kernels (8-byte entries, gathers) match their
expected bytes un-doubled.  The file therefore records the raw counters, the correction factor applied per kernel
and the corrected bytes.
"""
import json
import re
import sys

NAMES = {
    "distance_forward_packed_kernel": ("pair_distance_forward", 2.0),
    # 8-byte entries read by 16-lane row groups = 128-byte coalesced requests, tallied at 64 B like the wide streams
    # (raw 43.9 MB against a 76.5 MB entry stream that is read exactly once)
    "sr_fused_rows_kernel": ("rspace_forward", 2.0),
    # spread + pair sum in one launch: the pair-entry stream dominates the traffic (same correction)
    "spread_rows_kernel": ("spread+rspace_forward", 2.0),
    "xconv_kernel": ("convolve_xfused_x_stage", 1.0),
    "sr_fused_finalize_kernel": ("forces_finalize", 1.0),
    "distance_forward_kernel": ("pair_distance_forward", 2.0),
    "rspace_backward_kernel": ("rspace_backward", 2.0),
    "distance_backward_rows_kernel": ("pair_distance_backward", 1.0),
    "rspace_rows_kernel": ("rspace_forward", 1.0),
    "spread_brick_kernel": ("spread", 1.0),
    "gather_brick_kernel": ("gather", 1.0),
    "gather_grad_brick_kernel": ("gather_grad", 1.0),
    "apply_filter_kernel": ("apply_filter", 1.0),
}


def main(src, dst):
    raw = {}
    for line in open(src):
        m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+(\d+)\s*$", line)
        if not m:
            continue
        for key, (name, corr) in NAMES.items():
            if f"mipme::{key}<" in m.group(1):
                raw.setdefault(name, {"fetch_correction": corr})[m.group(2)] = float(m.group(3)) * 1024.0
    out = {}
    for name, d in raw.items():
        fetch = d.get("FETCH_SIZE", 0.0)
        write = d.get("WRITE_SIZE", 0.0)
        out[name] = {
            "fetch_bytes_raw": fetch,
            "write_bytes_raw": write,
            "fetch_correction": d["fetch_correction"],
            "hbm_bytes_per_launch": fetch * d["fetch_correction"] + write,
        }
    json.dump({"source": src, "kernels": out}, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
