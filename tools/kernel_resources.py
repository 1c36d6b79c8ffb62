#!/usr/bin/env python3
"""Registers / LDS / spills of the kernels in rlvi_amd/librlvi_gfx950.so (from the code objects' notes).

    python tools/kernel_resources.py [substring ...]

What co-residency a cooperating kernel can count on follows from these numbers (MI355X_MICROARCH.md,
"Register files": waves per SIMD = min(8, 512 // alloc), alloc = vgpr + agpr rounded up to 8)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "rlvi_amd", "librlvi_gfx950.so")


def main():
    want = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        lib = os.path.join(tmp, "lib.so")
        with open(LIB, "rb") as f, open(lib, "wb") as g:
            g.write(f.read())
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], cwd=tmp, capture_output=True)
        rows = []
        for name in sorted(os.listdir(tmp)):
            if "hipv4" not in name:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, name)],
                                 capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                def get(key):
                    m = re.search(r"\." + key + r":\s*(\S+)", blk)
                    return m.group(1) if m else "?"
                sym = get("name")
                dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(.*\)$", "", dem).replace("void rlvi::", "")
                agpr = blk.strip().split()[0]
                rows.append((dem, int(get("vgpr_count")), int(agpr), int(get("sgpr_count")),
                             int(get("group_segment_fixed_size")), get("vgpr_spill_count"),
                             get("private_segment_fixed_size")))
    print(f"{'kernel':58s} vgpr agpr sgpr   lds spill scratch waves/SIMD")
    for dem, v, ag, sg, lds, spill, scr in sorted(rows):
        if want and not any(w in dem for w in want):
            continue
        alloc = (v + 7) // 8 * 8
        print(f"{dem[:58]:58s} {v:4d} {ag:4d} {sg:4d} {lds:5d} {spill:>5s} {scr:>7s} {min(8, 512 // max(alloc, 8)):5d}")


if __name__ == "__main__":
    main()
