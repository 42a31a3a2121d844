#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPRs, AGPRs, scratch bytes, LDS bytes, occupancy) of one HIP source of the library, from
hipcc's -Rpass-analysis=kernel-resource-usage.  Usage: python tools/kernel_resources.py ssak_amd/csrc/gemm_p8.hip [name filter] [-- extra flags]"""
import re
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
src = args[0]
flt = args[1] if len(args) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "-ffp-contract=fast", "-mllvm",
       "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-Rpass-analysis=kernel-resource-usage", "-c",
       "-o", "/dev/null", src] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name)
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" ")[0].replace("TotalSGPRs", "SGPRs")] = int(m.group(2))
print(f"{'kernel':110s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scr':>5s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    if flt and flt not in r["name"]:
        continue
    print(f"{r['name'][:110]:110s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('SGPRs', 0):5d} {r.get('ScratchSize', 0):5d} {r.get('LDS', 0):7d} {r.get('Occupancy', 0):4d}")
