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


def source_hash() -> str:
    """sha256 over the kernel sources and the C header (names + contents): compiled into the library (`cocr_version()`) and
    written beside it, so that a library built from other sources is recognised whatever the file times say (the .so is
    git-ignored and travels with the tree to the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    for s in sources():
        h.update(os.path.basename(s).encode() + b'\0')
        with open(s, 'rb') as fp:
            h.update(fp.read())
    h.update(os.environ.get('COCR_HIPCC_FLAGS', '').encode())
    return h.hexdigest()[:16]


def built_hash() -> str:
    try:
        with open(LIB + '.srchash') as fp:
            return fp.read().strip()
    except OSError:
        return ''


def up_to_date() -> bool:
    return os.path.exists(LIB) and built_hash() == source_hash()


def _compile(args):
    cmd, verbose = args
    if verbose:
        print(' '.join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
    except subprocess.CalledProcessError:
        # a compiler without that backend option: same code, accumulators where the register allocator puts them
        if '-amdgpu-mfma-vgpr-form' not in cmd:
            raise
        i = cmd.index('-amdgpu-mfma-vgpr-form')
        cmd = cmd[:i - 1] + cmd[i + 1:]
        if verbose:
            print('retrying without -amdgpu-mfma-vgpr-form', flush=True)
        subprocess.check_call(cmd)


def build(force: bool = False, verbose: bool = True, out: str = None) -> str:
    """One object per translation unit (cocr_api.hip and the row-chain instantiation units), compiled in parallel, one link.
    `out`: write the library there instead (dev experiments: COCR_HIPCC_FLAGS=-D... builds loaded through COCR_LIB_PATH)."""
    if out is not None:
        return _build_to(out, verbose)
    if not force and up_to_date():
        return LIB
    return _build_to(LIB, verbose)


def _build_to(LIB: str, verbose: bool) -> str:
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libcocr_hip.so')
    os.makedirs(LIB_DIR, exist_ok=True)
    src_hash = source_hash()
    obj_dir = os.path.join(LIB_DIR, 'obj', src_hash)
    os.makedirs(obj_dir, exist_ok=True)
    # -amdgpu-mfma-vgpr-form: MFMA results in VGPRs (gfx950 has one unified file).  By default the register allocator parked the
    # attention / GEMM / frontend accumulators in AGPRs and paid a v_accvgpr_read/write per value the VALU touched (attention: 52 per
    # key tile; decoder GEMM 7.1 -> 3.5 us, attention 15.8 -> 14.9 us, fused frontend 144 -> 138 us with the flag).
    flags = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-mllvm', '-amdgpu-mfma-vgpr-form',
             '-Wall', '-Wno-unused-function', f'-DCOCR_SRC_HASH="{src_hash}"', '-I', INCLUDE]
    flags[0:0] = os.environ.get('COCR_HIPCC_FLAGS', '').split()      # dev builds, e.g. -DCOCR_CHAIN_STAMPS_BUILD
    units = [f for f in sorted(os.listdir(CSRC)) if f.endswith('.hip')]
    objs = [os.path.join(obj_dir, u[:-4] + '.o') for u in units]
    jobs = [([hipcc] + flags + ['-c', os.path.join(CSRC, u), '-o', o], verbose) for u, o in zip(units, objs)]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
        list(pool.map(_compile, jobs))
    link = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-fno-gpu-rdc'] + objs + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.check_call(link)
    os.replace(LIB + '.tmp', LIB)
    with open(LIB + '.srchash.tmp', 'w') as fp:
        fp.write(src_hash + '\n')
    os.replace(LIB + '.srchash.tmp', LIB + '.srchash')
    for d in os.listdir(os.path.join(LIB_DIR, 'obj')):                # objects of earlier source states
        if d != src_hash:
            shutil.rmtree(os.path.join(LIB_DIR, 'obj', d), ignore_errors=True)
    return LIB


if __name__ == '__main__':
    out = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else None
    print(build(force='--force' in sys.argv, out=out))
