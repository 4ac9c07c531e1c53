"""Round 5: which kernels are held to fewer resident waves by their SCALAR registers than by their vector registers?
gfx950 admits floor(800 / (ceil(sgpr / 16) * 16 + 16)) waves per SIMD (MI355X_MICROARCH.md, "Residency"), min(8, 512 // alloc) by
vector registers (allocation granule 8).  Compiles every csrc/*.hip with -Rpass-analysis=kernel-resource-usage (object files go
to /tmp) and prints the kernels with sgpr-waves < vgpr-waves.

    python tools/r05/sgpr_audit.py [file.hip ...]
"""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "torch-pme_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage".split()


def remarks(path):
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c", path, "-o", "/tmp/sgpr_audit_%s.o" % os.path.basename(path)],
                         capture_output=True, text=True)
    return path, out.stderr


def main(files):
    with ThreadPoolExecutor(len(files)) as ex:
        for path, text in ex.map(remarks, files):
            rows, cur = [], {}
            for line in text.splitlines():
                m = re.search(r"remark:\s+(.*?)(?: \[-Rpass)", line)
                if not m:
                    continue
                t = m.group(1)
                if t.startswith("Function Name:"):
                    cur = {"name": t.split(":", 1)[1].strip()}
                    rows.append(cur)
                else:
                    k, _, v = t.partition(":")
                    cur[k.strip()] = v.strip()
            names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True,
                                   text=True).stdout.splitlines()
            seen = set()
            for r, d in zip(rows, names):
                try:
                    sg, vg = int(r["TotalSGPRs"]), int(r["VGPRs"]) + int(r.get("AGPRs", 0))
                except (KeyError, ValueError):
                    continue
                wv = min(8, 512 // max(8, (vg + 7) // 8 * 8))
                ws = min(8, 800 // (((sg + 15) // 16) * 16 + 16))
                key = d.replace("mipme::", "")[:120]
                if ws < wv and key not in seen:
                    seen.add(key)
                    print(f"{os.path.basename(path):14s} sgpr {sg:3d} -> {ws} waves   vgpr {vg:3d} -> {wv} waves   {key}")


if __name__ == "__main__":
    main(sys.argv[1:] or sorted(glob.glob(os.path.join(SRC, "*.hip"))))
