#!/usr/bin/env python3
"""Beam-search kernel alone: device time per call (HIP events) for B lines of T frames, and (dev builds with
COCR_HIPCC_FLAGS=-DCOCR_CHAIN_STAMPS_BUILD + COCR_CHAIN_STAMPS=1) the per-phase cycle totals printed when the engine is destroyed.

    python tools/beam_probe.py [--lines 1] [--frames 300] [--beam 16] [--gain 8]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_ocr_amd import synth  # noqa: E402
from conformer_ocr_amd.engine import HipRecognizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lines', type=int, default=1)
    ap.add_argument('--frames', type=int, default=300)
    ap.add_argument('--beam', type=int, default=16)
    ap.add_argument('--gain', type=float, default=8.0)
    ap.add_argument('--iters', type=int, default=50)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    hp = synth.hparams('tiny')
    eng = HipRecognizer(hp, dev, 'fp32')
    eng.load_state(synth.make_state_dict(hp, seed=3))
    eng.finalize()
    g = torch.Generator().manual_seed(5)
    for ncls in (256, 64):
        logits = (torch.randn((args.lines, args.frames, ncls), generator=g) * args.gain).to(dev)
        ol = np.full((args.lines,), args.frames, dtype=np.int32)
        for _ in range(3):
            recs = eng.ctc_beam(logits, ol, args.beam)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(args.iters):
            e0.record()
            eng.lib.cocr_ctc_beam  # (the call below goes through the same entry)
            recs = eng.ctc_beam(logits, ol, args.beam)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f'ncls {ncls}: beam {args.beam}, {args.lines} x {args.frames} frames: p50 {np.median(ts) * 1e3:.0f} us per call '
              f'({np.median(ts) * 1e3 / args.frames:.2f} us per frame), labels {len(recs[0])}')
    del eng


if __name__ == '__main__':
    main()
