#!/usr/bin/env python3
"""Register budget of the kernels of one translation unit: tools/regs.py dlm_sparse16.hip [name-filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, ".."))
from bayesian_dlms_amd import build as b
path = os.path.join(b.CSRC, src)
cmd = [b.HIPCC] + b.FILE_FLAGS.get(src, b.FLAGS) + ["-c", path, "-o", "/tmp/regs_tmp.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
name = None
row = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void dlm::", "")
        row = {}
        continue
    for key in ("VGPRs", "AGPRs", "VGPRs Spill", "SGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "ScratchSize [bytes/lane]"):
        m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
        if m:
            row[key] = int(m.group(1))
    if "LDS Size" in line and name and flt in name:
        print(f"{name:58s} vgpr {row.get('VGPRs', 0):4d} agpr {row.get('AGPRs', 0):3d} spill {row.get('VGPRs Spill', 0):3d} scratch {row.get('ScratchSize [bytes/lane]', 0):4d} occ {row.get('Occupancy [waves/SIMD]', 0)} lds {row.get('LDS Size [bytes/block]', 0)}")
