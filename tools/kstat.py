#!/usr/bin/env python3
"""Print per-kernel register/scratch metadata from a hipcc -save-temps gfx950 .s file."""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size: *\d+", txt, re.S):
    g = lambda k: re.search(r"\." + k + r": *(\S+)", blk)
    name = g("name").group(1)
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()[:90]
    except Exception:
        pass
    print(f"{name:90s} vgpr={g('vgpr_count').group(1):>4} agpr={g('agpr_count').group(1):>4} sgpr={g('sgpr_count').group(1):>4} "
          f"scratch={g('private_segment_fixed_size').group(1):>5} vspill={g('vgpr_spill_count').group(1):>4} sspill={g('sgpr_spill_count').group(1):>4} lds={g('group_segment_fixed_size').group(1)}")
