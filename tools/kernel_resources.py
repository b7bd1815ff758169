#!/usr/bin/env python3
"""Registers / spills / scratch / LDS of every kernel of one translation unit, from hipcc's resource-usage remarks
(cross-compiles for gfx950; no GPU needed):

    python tools/kernel_resources.py nl-vsgg_amd/csrc/kernels_gemm_t16.hip [-DSTTRAN_GEMM_EXPERIMENT ...]
"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
       "-c", os.path.basename(src), "-o", "/dev/null"] + extra
out = subprocess.run(cmd, cwd=os.path.dirname(os.path.abspath(src)), capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", dem).replace("sttran::", "").replace("void ", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':90s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'LDS':>7s} {'occ':>3s}")
for r in rows:
    print(f"{r['name'][:90]:90s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('SGPRs', '?'):>5s} "
          f"{r.get('VGPRs Spill', '?'):>6s} {r.get('SGPRs Spill', '?'):>6s} {r.get('ScratchSize [bytes/lane]', '?'):>7s} "
          f"{r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>3s}")
