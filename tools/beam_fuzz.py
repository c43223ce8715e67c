#!/usr/bin/env python3
"""cocr_ctc_beam (rank + walk kernels) against the exhaustive kernel (COCR_BEAM_REF=1) on shapes beyond the test suite's: 2 and 256
classes, beam 1 and 31, one frame, and lines long enough that the back-pointers leave LDS.  Every output field must be identical.

    python tools/beam_fuzz.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_ocr_amd.engine import HipRecognizer  # noqa: E402
from conformer_ocr_amd.spec import HParams  # noqa: E402


def main():
    hp = HParams(num_classes=2, height=16, encoder_dim=16, num_encoder_layers=1, num_attention_heads=1, conv_kernel_size=3, subsampling_conv_channels=8)
    dev = torch.device('cuda', 0)
    fast = HipRecognizer(hp, dev, 'fp32')
    os.environ['COCR_BEAM_REF'] = '1'
    ref = HipRecognizer(hp, dev, 'fp32')
    bad = 0
    for (C, T, beam, scale) in [(256, 50, 16, 2.0), (2, 30, 4, 1.0), (3, 40, 1, 1.0), (200, 60, 31, 1.5), (64, 1, 16, 1.0), (40, 2000, 16, 2.0), (40, 1000, 32, 0.5),
                                (128, 1400, 16, 0.3), (7, 500, 32, 1.0), (256, 300, 32, 3.0)]:
        g = np.random.default_rng(C * 131 + T + beam)
        N = 3
        logits = (g.normal(size=(N, T, C)) * scale).astype(np.float32)
        logits[:, :, 0] += 1.0
        logits[1, :, 1:] = np.round(logits[1, :, 1:] * 2) / 2
        lens = [T, max(1, T - 1), max(1, T // 2)]
        x = torch.from_numpy(logits).to(dev)
        a, b = fast.ctc_beam(x, lens, beam), ref.ctc_beam(x, lens, beam)
        ok = a == b
        bad += not ok
        print(f'C={C:3d} T={T:4d} beam={beam:2d}: {"identical" if ok else "DIFFERENT"} ({[len(r) for r in a]} labels)', flush=True)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
