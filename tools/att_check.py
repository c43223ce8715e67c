#!/usr/bin/env python3
"""Dev: the LDS-resident attention kernel against the tiled one (COCR_ATT_TILED=1) and against itself (run-to-run), ctx taps of block 0."""
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n, w):
    from conformer_ocr_amd import synth
    from tests.hip_util import hip_tap, make_engine
    hp = synth.hparams('cfg2', num_encoder_layers=2)
    state = synth.make_state_dict(hp, seed=3, decoder_gain=1.0, style='text')
    image, lens = synth.make_lines(n, hp.height, w, seed=5, widths=[max(40, w - 37 * i) for i in range(n)])
    x = torch.from_numpy(image[:, 0]).cuda()
    eng = make_engine(hp, state, 'bf16')
    eng.set_debug(True)
    outs = []
    for _ in range(3):
        lg, _ = eng.forward(x, lens)
        torch.cuda.synchronize()
        outs.append((eng.tap('l0.ctx').copy(), eng.tap('l1.ctx').copy(), lg.cpu().numpy().copy()))
    return outs


if __name__ == '__main__':
    if len(sys.argv) > 1:
        n, w = int(sys.argv[1]), int(sys.argv[2])
        outs = run(n, w)
        np.savez(sys.argv[3], c0=outs[0][0], c1=outs[0][1], lg=outs[0][2])
        for i in (1, 2):
            print('  run 0 vs run %d: ctx0 %.3e ctx1 %.3e logits %.3e' % (i, np.abs(outs[0][0] - outs[i][0]).max(), np.abs(outs[0][1] - outs[i][1]).max(),
                                                                           np.abs(outs[0][2] - outs[i][2]).max()))
    else:
        for n, w in ((17, 1200), (5, 640), (2, 300), (2, 500)):
            print('case', n, w)
            for tiled in ('0', '1'):
                env = dict(os.environ, COCR_ATT_TILED=tiled)
                r = subprocess.run([sys.executable, __file__, str(n), str(w), f'/tmp/att_{tiled}.npz'], env=env, capture_output=True, text=True)
                print(' tiled=' + tiled, r.stdout.strip(), r.stderr[-300:] if r.returncode else '')
            a, b = np.load('/tmp/att_0.npz'), np.load('/tmp/att_1.npz')
            d = np.abs(a['c0'] - b['c0'])
            print('  resident vs tiled: ctx0 max %.3e (at %s) ctx1 %.3e logits %.3e' % (d.max(), np.unravel_index(d.argmax(), d.shape) if d.ndim else 0, np.abs(a['c1'] - b['c1']).max(),
                                                                                    np.abs(a['lg'] - b['lg']).max()))
