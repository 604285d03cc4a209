#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in a built object (csrc/<name>.o), read from the code object's metadata notes.

    python tools/kernel_resources.py gemm [filter-substring ...]

Runs without a GPU (llvm-objcopy / clang-offload-bundler / llvm-readelf from /opt/rocm/lib/llvm/bin).  A kernel with
scratch > 0 or spills > 0 is a bug on this path: the tile kernels are sized for 168 (10 waves) or 256 (8 waves) registers."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3dtopia-xl_amd", "csrc")


def resources(obj: str):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", e).group(1))
        name = re.search(r"\.symbol:\s+(\S+?)\.kd", e).group(1)
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out.append(dict(name=re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", "")), agpr=int(re.match(r":\s+(\d+)", e).group(1)), vgpr=g("vgpr_count"),
                        sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"), spill=g("vgpr_spill_count"),
                        lds=g("group_segment_fixed_size")))
    return out


if __name__ == "__main__":
    base = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    bad = 0
    for r in resources(os.path.join(CSRC, base + ".o")):
        if all(f in r["name"] for f in sys.argv[2:]):
            flag = "  <-- SCRATCH" if r["scratch"] or r["spill"] else ""
            bad += bool(flag)
            print(f"{r['name'][:64]:64s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds']:6d} scratch {r['scratch']}{flag}")
    sys.exit(1 if bad else 0)
