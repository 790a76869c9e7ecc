#!/usr/bin/env python
"""Register / scratch / LDS footprint of the kernels in libaq3d.so (llvm-objdump --offloading + llvm-readelf --notes).
    python scratch/kernel_regs.py [name-regex]"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "quantization-on-3d-object-detection_amd", "libaq3d.so")
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
tmp = tempfile.mkdtemp()
shutil.copy(so, os.path.join(tmp, "lib.so"))
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", "lib.so"], check=True, capture_output=True, cwd=tmp)
for f in sorted(os.listdir(tmp)):
    if "gfx950" not in f:
        continue
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if not pat.search(name):
            continue
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() if shutil.which("c++filt") else name
        print("%-90s vgpr %3d agpr %3d sgpr %3d scratch %5d lds %6d" % (dem[:90], g("vgpr_count"), int(re.match(r":\s+(\d+)", blk).group(1)), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
shutil.rmtree(tmp)
