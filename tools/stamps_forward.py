#!/usr/bin/env python3
"""Dev: a few cfg2 forwards of 32 x 96x1200 on a COCR_CHAIN_STAMPS_BUILD library (COCR_LIB_PATH, COCR_CHAIN_STAMPS=1): the cycle stamps
of the frontend / attention kernels print when the engine is destroyed."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_ocr_amd import synth  # noqa: E402
from conformer_ocr_amd.engine import HipRecognizer  # noqa: E402

dev = torch.device('cuda', 0)
hp = synth.hparams('cfg2')
eng = HipRecognizer(hp, dev, 'bf16')
eng.load_state(synth.make_state_dict(hp, seed=1236, decoder_gain=8.0))
eng.finalize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
img, lens = synth.make_lines(N, hp.height, 1200, seed=7)
x = torch.from_numpy(img[:, 0]).to(dev)
for _ in range(3):
    eng.forward(x, lens.astype(np.int32))
torch.cuda.synchronize()
del eng
