"""Registers, scratch and LDS of every kernel in a built object file, read from the code object's metadata (no recompilation):
    python tools/r06/kernel_resources.py torch-pme_amd/csrc/bricks.o [substring ...]
Waves per SIMD follow MI355X_MICROARCH.md ("Residency"): min(8, 512 // vgpr granule-of-8) by vector registers,
800 // (ceil(sgpr / 16) * 16 + 16) by scalar registers."""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def resources(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = d + "/fat", d + "/co"
        subprocess.run([LLVM + "llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], check=True)
        subprocess.run([LLVM + "clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    rows = []
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda key: re.search(rf"\.{key}:\s+(\S+)", k)  # noqa: E731
        if g("name"):
            rows.append(dict(name=g("name").group(1), sgpr=int(g("sgpr_count").group(1)), vgpr=int(g("vgpr_count").group(1)),
                             scratch=int(g("private_segment_fixed_size").group(1)), lds=int(g("group_segment_fixed_size").group(1))))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["demangled"] = re.sub(r"\(.*", "", n).replace("void mipme::", "")
    return rows


if __name__ == "__main__":
    pats = sys.argv[2:]
    for r in resources(sys.argv[1]):
        if pats and not any(p in r["demangled"] for p in pats):
            continue
        wv = min(8, 512 // ((r["vgpr"] + 7) // 8 * 8)) if r["vgpr"] else 8
        ws = min(8, 800 // ((r["sgpr"] + 15) // 16 * 16 + 16))
        print(f"{r['demangled'][:100]:100s} sgpr {r['sgpr']:3d} vgpr {r['vgpr']:3d} scratch {r['scratch']:4d} lds {r['lds']:6d} waves v{wv}/s{ws}")
