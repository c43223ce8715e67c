#!/usr/bin/env python3
"""BASELINE configs[4]: latency of small batches with CTC beam search (beam 16) on one GPU.

    python tools/latency_beam.py [--iters 200]

Per batch size B in {1, 4, 8}: forward (bf16, cfg2, 96x1200 lines resident in HBM) + beam decode + D2H of the label records,
synchronous per call (one request at a time); prints p50 / p99 milliseconds per batch and per line, for beam 16 and greedy."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_ocr_amd import synth  # noqa: E402
from conformer_ocr_amd.engine import HipRecognizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--width', type=int, default=1200)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    hp = synth.hparams('cfg2')
    eng = HipRecognizer(hp, dev, 'bf16')
    eng.load_state(synth.make_state_dict(hp, seed=1236, decoder_gain=8.0))
    eng.finalize()
    eng.set_graph(True)
    out = {}
    for B in (1, 4, 8):
        img, lens = synth.make_lines(B, hp.height, args.width, seed=7 + B)
        x = torch.from_numpy(img[:, 0]).to(dev)
        lens32 = lens.astype(np.int32)
        eng.reserve(B, args.width)
        buf = torch.empty((B, eng.out_len(args.width), hp.num_classes), dtype=torch.float32, device=dev)
        for mode in ('beam16', 'greedy'):
            ts = []
            for i in range(args.iters + 10):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                logits, ol = eng.forward(x, lens32, out=buf)
                recs = eng.ctc_beam(logits, ol, 16) if mode == 'beam16' else eng.ctc_greedy(logits, ol)
                ts.append((time.perf_counter() - t0) * 1e3)
            ts = np.sort(np.array(ts[10:]))
            out[f'B{B}_{mode}'] = {'p50_ms': round(float(ts[len(ts) // 2]), 3), 'p99_ms': round(float(ts[int(len(ts) * 0.99) - 1]), 3),
                                  'p50_ms_per_line': round(float(ts[len(ts) // 2]) / B, 3), 'labels_line0': len(recs[0])}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
