#!/usr/bin/env python3
"""GEMM variant sweep on the GPU (development): python tools/bench_gemm.py"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_ocr_amd import _lib
lib = _lib.load()
fn = lib.cocr_dev_bench_gemm
fn.restype = C.c_int
fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
shapes = {'ffn128': (9600, 128, 256), 'ffn256': (9600, 256, 256), 'ffn512': (9600, 512, 256), 'ffn_up': (9600, 1024, 256), 'ffn2048': (9600, 2048, 256), 'ffn_down': (9600, 256, 1024), 'proj256': (9600, 256, 256), 'qkv': (9600, 768, 256),
          'glu': (9600, 512, 256), 'front_pw': (230400, 256, 256), 'front_out': (9600, 256, 6144)}
plain = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 30, 31, 20, 21]
if len(sys.argv) > 1:
    plain = [int(x) for x in sys.argv[1].split(',')]
ln = [10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20]
if len(sys.argv) > 2:
    shapes = {k: v for k, v in shapes.items() if k in sys.argv[2].split(',')}
for name, (M, N, K) in shapes.items():
    vs = plain + (ln if (K == 256 and len(sys.argv) <= 1 and False) else [])
    row = []
    for v in vs:
        us = C.c_double()
        rc = fn(v, M, N, K, 20, C.byref(us))
        if rc != 0:
            row.append(f'v{v}:ERR({lib.cocr_last_error().decode()[:40]})')
            continue
        tf = 2.0 * M * N * K / us.value / 1e6
        row.append(f'v{v}:{us.value:7.1f}us/{tf:5.0f}TF')
    print(f'{name:10s} M={M} N={N} K={K}\n   ' + '  '.join(row), flush=True)
