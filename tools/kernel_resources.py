#!/usr/bin/env python3
"""Prints VGPR / scratch / occupancy / LDS per kernel of libcocr_hip (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-mllvm', '-amdgpu-mfma-vgpr-form', '-I', ROOT + '/include',
       ROOT + '/conformer_ocr_amd/csrc/cocr_api.hip', '-o', '/tmp/_res.so', '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, []
for l in out.splitlines():
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = {'name': m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (('V', 'VGPRs'), ('A', 'AGPRs'), ('scr', r'ScratchSize \[bytes/lane\]'), ('occ', r'Occupancy \[waves/SIMD\]'),
                     ('lds', r'LDS Size \[bytes/block\]')):
        m = re.search(r'remark: .*?\b' + pat + r': (\d+)', l)
        if m and cur is not None:
            cur[key] = m.group(1)
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for r in rows:
    n = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    n = re.sub(r'\(.*', '', n)
    if flt in n:
        print(f"{n[:120]:120s} V{r.get('V')} A{r.get('A')} scr{r.get('scr')} occ{r.get('occ')} lds{r.get('lds')}")
