#!/usr/bin/env python3
"""Lines/s through the user-facing loop `conformer_ocr_amd.evaluate.recognize` (collate, pinned upload, forward, greedy decode, codec)
on the cfg2 text fixture's lines repeated, by the number of batches it keeps in flight:   python tools/recognize_rate.py [--repeat 40]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (load_text_fixture)
from conformer_ocr_amd.codec import ascii_codec  # noqa: E402
from conformer_ocr_amd.evaluate import recognize  # noqa: E402
from conformer_ocr_amd.pred import PytorchRecognitionModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--repeat', type=int, default=80)
    ap.add_argument('--mixed', action='store_true', help='lines of widths 400..2400 (step 8) instead of the fixture lines: many batch shapes')
    args = ap.parse_args()
    fix = bench.load_text_fixture('cfg2_text')
    hp = fix['hp']
    net = PytorchRecognitionModel(**hp.as_dict(), input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1, conv_dropout_p=0.1,
                                  codec=ascii_codec(hp.num_classes), compute_dtype='bf16')
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in fix['state'].items()})
    net = net.to('cuda:0').eval()
    f32 = [np.asarray(ln, dtype=np.float32) for ln in fix['lines']] * args.repeat
    if args.mixed:
        rng = np.random.default_rng(11)
        f32 = [np.ascontiguousarray(f32[i % len(f32)][:, :1200].repeat(2, axis=1)[:, :int(w)]) for i, w in enumerate(rng.integers(50, 301, size=1280) * 8)]
    u8 = [np.rint(ln * 255.0).astype(np.uint8) for ln in f32]
    print(f'{len(f32)} lines, height {f32[0].shape[0]}, widths {min(l.shape[1] for l in f32)}..{max(l.shape[1] for l in f32)}')
    for name, lines in (('float32', f32), ('uint8', u8)):
        for streams in (1, 2, 4):
            t0 = time.perf_counter()
            recognize(net, lines, batch_size=32, edge=200, streams=streams)               # first pass: engines, pinned buffers, graph captures per batch shape
            recognize(net, lines, batch_size=32, edge=200, streams=streams)
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            t0 = time.perf_counter()
            recognize(net, lines, batch_size=32, edge=200, streams=streams)
            dt = time.perf_counter() - t0
            print(f'{name:8s} streams {streams}: {len(lines) / dt:9.0f} lines/s ({dt * 1e3:.0f} ms; the two passes before: {first * 1e3:.0f} ms)', flush=True)


if __name__ == '__main__':
    main()
