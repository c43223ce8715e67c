#!/usr/bin/env python3
"""Per-kernel-family dispatch times of one forward (HIP events on the forward's stream, one batch in flight), for A/B runs of
dev builds:   COCR_LIB_PATH=conformer_ocr_amd/lib/exp1.so python tools/kernel_times.py [--config cfg2] [--batch 32] [--width 1200] [--rows R]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_ocr_amd import synth  # noqa: E402
from conformer_ocr_amd.engine import HipRecognizer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='cfg2')
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--width', type=int, default=1200)
ap.add_argument('--rows', type=int, default=0)
ap.add_argument('--reps', type=int, default=20)
args = ap.parse_args()
hp = synth.hparams(args.config)
dev = torch.device('cuda', 0)
eng = HipRecognizer(hp, dev, 'bf16')
eng.load_state(synth.make_state_dict(hp, seed=1, decoder_gain=1.0, style='text'))
eng.finalize()
eng.set_chain_rows(args.rows)
img, lens, _, _ = synth.make_text_lines(args.batch, hp.height, args.width, seed=3)
x = torch.from_numpy(img[:, 0]).to(dev)
for _ in range(3):
    lg, ol = eng.forward(x, lens)
    eng.ctc_greedy(lg, ol)
torch.cuda.synchronize()
eng.profile(True)
for _ in range(args.reps):
    lg, ol = eng.forward(x, lens)
    eng.ctc_greedy(lg, ol)
prof = eng.profile_read()
eng.profile(False)
ovh = prof.pop('event_pair_overhead', (0.0, 0))[0]
tot = 0.0
out = []
for k, (ms, cnt) in prof.items():
    us = (ms - ovh) * 1e3
    tot += us * cnt / args.reps
    out.append(f'{k}={us:.1f}x{cnt // args.reps}')
print(os.environ.get('COCR_LIB_PATH', 'default'), f'rows={args.rows}', f'sum={tot:.0f}us', ' '.join(out))
