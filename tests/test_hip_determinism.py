"""GPU: run-to-run determinism of the bf16 forward (no atomics, fixed reduction orders: two forwards of the same batch must give the same
bits; a difference is a hazard or a race -- round 3 found two that way: matrix results read too early by inline assembly, and load-count
waits that no longer counted the loads the compiler had kept).  Random ragged batches through the metric model's, the wide model's and the
reference's default model's kernels, both attention kernels; each batch runs plain, captured and replayed on one engine, then as the first
forward of a fresh engine behind a forward of other data."""
import numpy as np
import pytest
import torch

from conformer_ocr_amd import synth
from tests.hip_util import make_engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('config,layers,resident_min', [('cfg2', 3, '192'), ('cfg2', 2, '1'), ('cfg1', 2, '1'), ('cfg4', 2, '192')])
def test_forward_is_bit_reproducible(config, layers, resident_min, monkeypatch):
    monkeypatch.setenv('COCR_ATT_RESIDENT_MIN', resident_min)
    g = np.random.default_rng(len(config) * 1000 + layers)
    hp = synth.hparams(config, num_encoder_layers=layers)
    state = synth.make_state_dict(hp, seed=5, decoder_gain=4.0, style='text')
    eng = make_engine(hp, state, 'bf16')
    eng.set_graph(True)
    for k in range(8):
        n = int(g.integers(1, 9)) if k % 3 else int(g.integers(24, 41))
        w = int(g.integers(16, 2400)) if k % 4 else int(g.integers(900, 1281))
        image, lens = synth.make_lines(n, hp.height, w, seed=100 + k, widths=sorted((int(x) for x in g.integers(9, w + 1, size=n)), reverse=True))
        x = torch.from_numpy(image[:, 0]).cuda()
        outs = []
        for _ in range(3):
            lg, _ = eng.forward(x, lens)
            torch.cuda.synchronize()
            outs.append(lg.cpu().numpy().copy())
        assert np.isfinite(outs[0]).all(), (k, n, w)
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), (k, n, w)
        if k % 2 == 0:
            fresh = make_engine(hp, state, 'bf16').forward(x, lens)[0].cpu().numpy()
            np.testing.assert_array_equal(fresh, outs[0], err_msg=f'first forward of a fresh engine, case {k} (n={n}, w={w})')
