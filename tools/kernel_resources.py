#!/usr/bin/env python3
"""per-kernel register / spill / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950):
   python tools/kernel_resources.py odam_amd/csrc/conv_gemm.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|VGPRs Spill|LDS Size \[bytes/block\])\s*:\s*(\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'spill':>6} {'scratch':>8} {'occ':>4}  kernel")
for r in rows:
    print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('SGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>6} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>8} {r.get('Occupancy [waves/SIMD]', '?'):>4}  {r['name'][:150]}")
