#!/usr/bin/env python3
"""Time of one training step of the whole network (cocr_train_step + cocr_train_adamw; fp32) and a short training run on synthetic text
lines:   python tools/train_bench.py [--config cfg2] [--batch 32] [--width 1200] [--steps 5] [--fit 0]
--fit K: K steps on text lines (conformer_ocr_amd.synth.make_text_lines) from random weights, printing the loss and the greedy CER of
the trained model against the ground truth every few steps (the inference path serves the trained weights after sync)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_ocr_amd import synth  # noqa: E402
from conformer_ocr_amd.codec import ascii_codec  # noqa: E402
from conformer_ocr_amd.evaluate import ErrorRate  # noqa: E402
from conformer_ocr_amd.pred import PytorchRecognitionModel  # noqa: E402
from conformer_ocr_amd.train import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='cfg2')
ap.add_argument('--layers', type=int, default=0)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--width', type=int, default=1200)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--fit', type=int, default=0)
ap.add_argument('--lr', type=float, default=1e-3)
ap.add_argument('--matmul', default='highest', help="'highest' (exact fp32 products) or 'medium' (bf16-rounded operands, the reference's training setting)")
args = ap.parse_args()
kw = {'num_encoder_layers': args.layers} if args.layers else {}
hp = synth.hparams(args.config, **kw)
state = synth.make_state_dict(hp, seed=1, decoder_gain=1.0)
net = PytorchRecognitionModel(**hp.as_dict(), input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1, conv_dropout_p=0.1,
                              codec=ascii_codec(hp.num_classes), compute_dtype='bf16' if hp.encoder_dim in (256, 512) else 'fp32')
net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
net = net.to('cuda:0').eval()
image, lens, texts, _ = synth.make_text_lines(args.batch, hp.height, args.width, seed=3)
batch = {'image': torch.from_numpy(image).cuda(), 'seq_lens': torch.from_numpy(lens), 'target': torch.tensor([c for t in texts for c in t]),
         'target_lens': torch.tensor([len(t) for t in texts])}
tr = Trainer(net, lr=args.lr, weight_decay=1e-2, warmup=10, matmul_precision=args.matmul)
out = {'config': args.config, 'layers': hp.num_encoder_layers, 'batch': args.batch, 'width': args.width}
tr.training_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = tr.training_step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
out.update(ms_per_step=round(dt * 1e3, 2), lines_per_s=round(args.batch / dt, 1), last_loss=loss)
if args.fit:
    hist = []
    for step in range(args.fit):
        loss = tr.training_step(batch)
        if step % max(1, args.fit // 8) == 0 or step == args.fit - 1:
            tr.sync_module()
            pred = net.predict_labels(batch['image'], batch['seq_lens'])
            cer = ErrorRate()
            cer.update([[r[0] for r in line] for line in pred], texts)
            hist.append({'step': step, 'loss': round(loss, 2), 'cer': round(cer.compute(), 4)})
    out['fit'] = hist
print(json.dumps(out))
