#!/usr/bin/env python3
"""Pretty-prints a bench.py JSON line: python tools/show_bench.py gpurun_out/bench.json"""
import json
import sys
r = json.load(open(sys.argv[1]))
print('lines/s', r['value'], 'ms/step', r['ms_per_step'], 'TF', r['achieved_tflops_whole_path'])
print('cpu', r['cpu_baseline'])
print('roofline', r['roofline'])
tot = 0.0
for k, v in r['kernels'].items():
    t = v['avg_ms'] * 1e3 * v['launches_per_step']
    tot += t
    print(f"{k:18s} {v['avg_ms'] * 1e3:8.1f} us x{v['launches_per_step']:3d} = {t:8.1f} us share {v['share']:.3f} {v['achieved']:9.1f} {v['unit']:8s} frac {v['frac']:.4f}")
print('sum of kernel time per step (us):', round(tot, 1))
