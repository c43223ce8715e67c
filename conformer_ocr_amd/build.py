"""Builds libcocr_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

    python -m conformer_ocr_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB_DIR = os.path.join(PKG, 'lib')
LIB = os.path.join(LIB_DIR, 'libcocr_hip.so')
INCLUDE = os.path.join(os.path.dirname(PKG), 'include')
ARCH = 'gfx950'


def sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    return srcs + [os.path.join(INCLUDE, 'cocr.h')]


def up_to_date() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(s) <= t for s in sources())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libcocr_hip.so')
    os.makedirs(LIB_DIR, exist_ok=True)
    # -amdgpu-mfma-vgpr-form: MFMA results in VGPRs (gfx950 has one unified file).  By default the register allocator parked the
    # attention / GEMM / frontend accumulators in AGPRs and paid a v_accvgpr_read/write per value the VALU touched (attention: 52 per
    # key tile; decoder GEMM 7.1 -> 3.5 us, attention 15.8 -> 14.9 us, fused frontend 144 -> 138 us with the flag).
    cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-shared', '-fgpu-rdc' if False else '-fno-gpu-rdc',
           '-mllvm', '-amdgpu-mfma-vgpr-form',
           '-Wall', '-Wno-unused-function', '-I', INCLUDE, os.path.join(CSRC, 'cocr_api.hip'), '-o', LIB + '.tmp']
    cmd[1:1] = os.environ.get('COCR_HIPCC_FLAGS', '').split()      # dev builds, e.g. -DCOCR_CHAIN_STAMPS_BUILD
    if verbose:
        print(' '.join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
    except subprocess.CalledProcessError:
        # a compiler without that backend option: same code, accumulators where the register allocator puts them
        if '-amdgpu-mfma-vgpr-form' not in cmd:
            raise
        i = cmd.index('-amdgpu-mfma-vgpr-form')
        del cmd[i - 1:i + 1]
        if verbose:
            print('retrying without -amdgpu-mfma-vgpr-form', flush=True)
        subprocess.check_call(cmd)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
