#!/usr/bin/env python3
"""The FFN products by themselves under the profiler (BASELINE north_star: "MFMA utilisation on FFN GEMMs evidenced by rocprof").

    COCR_FFN_PROBE=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA \
        --output-format csv -d out -- python3 tools/ffn_probe.py

With COCR_FFN_PROBE=1 every forward launches, beside its own kernels, `rowchain_kernel<256, MT, 0, ST_FFN, -1, -1, -1>`: one feed-forward
module (LayerNorm'd operand tile in LDS, 4 hidden chunks of W1 / SiLU / W2 with the hidden chunk on chip, residual + LayerNorm epilogue) on
the metric batch's 9600 rows, results discarded.  This script runs a few forwards in the 96-row and in the 48-row form and prints the
event-timed duration of that launch; the counters come from the profiler's CSV (tools/summarize_rocprof.py sq)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_ocr_amd import synth  # noqa: E402
from conformer_ocr_amd.engine import HipRecognizer  # noqa: E402

os.environ.setdefault('COCR_FFN_PROBE', '1')
dev = torch.device('cuda', 0)
hp = synth.hparams('cfg2')
eng = HipRecognizer(hp, dev, 'bf16')
eng.load_state(synth.make_state_dict(hp, seed=1236, decoder_gain=8.0))
eng.finalize()
img, lens = synth.make_lines(32, hp.height, 1200, seed=7)
x = torch.from_numpy(img[:, 0]).to(dev)
out = {}
for rows in (0, 48):
    eng.set_chain_rows(rows)
    for _ in range(3):
        eng.forward(x, lens.astype(np.int32))
    eng.profile(True)
    for _ in range(10):
        eng.forward(x, lens.astype(np.int32))
    prof = eng.profile_read()
    eng.profile(False)
    ovh = prof.get('event_pair_overhead', (0.0, 0))[0]
    ms = prof['ffn_probe'][0] - ovh
    flop = 4.0 * 9600 * hp.encoder_dim * hp.encoder_dim * hp.feed_forward_expansion_factor
    out['rows96' if rows == 0 else 'rows48'] = {'avg_us': round(ms * 1e3, 2), 'tflops': round(flop / ms / 1e9, 1), 'frac_of_chip_peak': round(flop / ms / 1e9 / 2500.0, 4),
                                               'workgroups': -(-9600 // (96 if rows == 0 else 48))}
print(json.dumps(out))
