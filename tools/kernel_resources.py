#!/usr/bin/env python3
"""Per-kernel register / LDS / spill report of a built object or library (gfx950 code object metadata).

    python tools/kernel_resources.py early_exit_transformer_amd/csrc/libeec.so [filter]

Reads the AMDGPU metadata notes with llvm-readelf after unbundling the offload bundle."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "dev.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={path}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            # a shared library: the fat binary sits in .hip_fatbin
            fb = os.path.join(td, "fatbin")
            subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb], check=True)
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    rows = []
    for blk in notes.split("- .agpr_count:")[1:]:
        def f(key):
            m = re.search(rf"\.{key}:\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = f("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"^void eec::", "", dem).split("(")[0]
        if flt and flt not in dem:
            continue
        rows.append((dem, f("vgpr_count"), f("vgpr_spill_count"), f("sgpr_count"), f("sgpr_spill_count"),
                     f("private_segment_fixed_size"), f("group_segment_fixed_size")))
    print(f"{'kernel':60s} {'vgpr':>5s} {'vspill':>6s} {'sgpr':>5s} {'sspill':>6s} {'scratch':>7s} {'lds':>7s}")
    for r in sorted(rows):
        print(f"{r[0][:60]:60s} {r[1]:>5s} {r[2]:>6s} {r[3]:>5s} {r[4]:>6s} {r[5]:>7s} {r[6]:>7s}")


if __name__ == "__main__":
    main()
