#!/usr/bin/env python3
"""Resource table of every kernel in libvisomatch.so, from the code objects' own metadata (no GPU needed):
VGPRs, SGPRs, static LDS bytes per workgroup, scratch, maximum workgroup size, and what follows for residency on gfx950
(512 VGPRs per SIMD lane-slot in granules of 8, 8 wave slots per SIMD, 160 KB LDS per compute unit):
waves per SIMD by registers, workgroups per CU by LDS (static part only: kernels with `extern __shared__` arrays are sized at
launch - k_dc2_prepare_lds, k_dc2_merge, k_dc2_support_lds - and say "dyn").
  python tools/kernel_resources.py [libvisomatch.so] > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "opencl-structure-from-motion_amd", "libvisomatch.so")
LLVM = "/opt/rocm/lib/llvm/bin"
rows = []
with tempfile.TemporaryDirectory() as td:
    tmp = os.path.join(td, "lib.so")
    os.symlink(os.path.abspath(lib), tmp)
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", tmp], check=True, stdout=subprocess.DEVNULL, cwd=td)
    for fn in sorted(os.listdir(td)):
        if "amdgcn" not in fn:
            continue
        txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", os.path.join(td, fn)], check=True, capture_output=True, text=True).stdout
        doc = txt[txt.index("---"):]
        doc = doc[:doc.index("\n...")] if "\n..." in doc else doc
        meta = yaml.safe_load(doc)
        for k in meta.get("amdhsa.kernels", []):
            name = subprocess.run(["c++filt", k[".name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            rows.append((name, k[".vgpr_count"], k.get(".agpr_count", 0), k[".sgpr_count"], k[".group_segment_fixed_size"],
                         k[".private_segment_fixed_size"], k[".max_flat_workgroup_size"], bool(k.get(".uses_dynamic_stack"))))
print("%-46s %5s %5s %6s %7s %7s %9s %9s" % ("kernel", "VGPR", "SGPR", "LDS/WG", "scratch", "max WG", "waves/SIMD", "WGs/CU(LDS)"))
for name, v, a, s, lds, scr, wg, dyn in sorted(set(rows)):
    tot = v + a
    gran = (max(tot, 1) + 7) // 8 * 8
    waves = min(8, 512 // gran)
    by_lds = ("%d" % min(32, (160 * 1024) // lds)) if lds else "-"
    print("%-46s %5d %5d %6d %7d %7d %9d %9s" % (name[:46], tot, s, lds, scr, wg, waves, by_lds))
