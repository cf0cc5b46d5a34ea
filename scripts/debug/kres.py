#!/usr/bin/env python
"""Kernel resource usage table: python scripts/kres.py [regex] (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = "vidcom2_amd/csrc/vc2_kernels.hip"
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
extra = sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", src,
                      "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +([\w \[\]/]+?): (.*?) \[-Rpass", line)
    if not m:
        m2 = re.search(r"(Function Name|Name): (\S+)", line)
        if m2: cur = m2.group(2); rows[cur] = {}
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ("Function Name", "Name"):
        cur = v; rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in rows.items():
    dem = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    if pat and not pat.search(short): continue
    print(f"{short:60s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('TotalSGPRs', r.get('SGPRs','?')):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>5} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?')}")
