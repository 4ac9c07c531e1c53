"""Turn the PMC summary written by tools/profile_gpu.sh into profiles/pmc_traffic.json (HBM bytes per launch).

    python tools/pmc_to_json.py gpurun_out/<tag>_pmc.txt profiles/pmc_traffic.json [workload]

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3).  gfx950 correction (MI355X_MICROARCH.md, HBM section: FETCH_SIZE =
TCC_EA0_RDREQ x 64 B although the requests are 128-byte lines), calibrated on this stack with known byte counts in the access
patterns of these kernels (tools/fetch_calib.hip -> profiles/r02_fetch_calib.txt): streaming reads of 1 GiB at 4, 8 and 16
bytes per lane ALL report 0.500 GiB; 64 M random 16-byte gathers from a 768 MiB table report 64 B per gather (one line,
tallied at half its size) and nothing when the table is cache resident; streaming stores report their exact size.  Hence

    hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE        for every kernel.

The file records the raw counters, the factor and the corrected bytes.
"""
import json
import re
import sys

FETCH_CORRECTION = 2.0

NAMES = {
    "spread_rows_kernel": "spread+rspace_forward",
    "spread_rows_capped_kernel": "spread+rspace_forward",
    "plane_rows_kernel": "spread+rspace_forward",
    "plane_rows_capped_kernel": "spread+rspace_forward",
    "gather_tail_kernel": "gather+energy+forces",
    "gather_brick_kernel": "gather",
    "bin_atoms_kernel": "bin_atoms",
    "xconv_kernel": "convolve_xfused_x_stage",
    "yz_planes_kernel": "convolve_xfused_yz_planes",
    "sr_fused_finalize_kernel": "forces_finalize",
    "sr_fused_rows_kernel": "rspace_forward",
    "distance_forward_packed_kernel": "pair_distance_forward",
    "distance_forward_kernel": "pair_distance_forward",
    "rspace_backward_kernel": "rspace_backward",
    "distance_backward_rows_kernel": "pair_distance_backward",
    "rspace_rows_kernel": "rspace_forward_unfused",
    "spread_brick_kernel": "spread",
    "gather_grad_brick_kernel": "gather_grad",
    "apply_filter_kernel": "apply_filter",
}


def main(src, dst, dtype_tag="float"):
    raw = {}
    for line in open(src):
        m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+(\d+)\s*$", line)
        if not m or dtype_tag not in m.group(1):
            continue
        for key, name in NAMES.items():
            if f"mipme::{key}<" in m.group(1):
                # several instantiations of one kernel (e.g. forward / inverse planes): keep the larger dispatch count's entry
                slot = raw.setdefault(name, {})
                if m.group(2) not in slot or int(m.group(4)) > slot[m.group(2)][1]:
                    slot[m.group(2)] = (float(m.group(3)) * 1024.0, int(m.group(4)))
    out = {}
    for name, d in raw.items():
        fetch = d.get("FETCH_SIZE", (0.0, 0))[0]
        write = d.get("WRITE_SIZE", (0.0, 0))[0]
        out[name] = {
            "fetch_bytes_raw": fetch,
            "write_bytes_raw": write,
            "fetch_correction": FETCH_CORRECTION,
            "hbm_bytes_per_launch": fetch * FETCH_CORRECTION + write,
        }
    return out


def write(src, dst, workload=None):
    """workload = None: the headline workload (top level of the file); else a section under "workloads" (e.g. "dispersion":
    python tools/pmc_to_json.py gpurun_out/<tag>_cfg5_pmc.txt profiles/pmc_traffic.json dispersion)."""
    import os

    out = main(src, dst)
    data = json.load(open(dst)) if (workload and os.path.exists(dst)) else {}
    if workload:
        data.setdefault("workloads", {})[workload] = {"source": src, "kernels": out}
    else:
        keep = data.get("workloads") or (json.load(open(dst)).get("workloads") if os.path.exists(dst) else None)
        data = {"source": src, "calibration": "profiles/r02_j_fetch_calib.txt", "kernels": out}
        if keep:
            data["workloads"] = keep
    json.dump(data, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    write(*sys.argv[1:])
