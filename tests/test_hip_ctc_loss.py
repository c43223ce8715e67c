"""GPU: the CTC loss kernel (cocr_ctc_loss) against the CPU restatement and against the vectors torch's CTCLoss produced for the
reference's `_step` lines (tests/golden/ctc_loss.npz).  fp32 log-domain arithmetic: torch's own float32 result differs from its
float64 result by up to 6e-4 in the gradient at 300 frames (values reach 1e3, one float32 ulp there is 1e-4), so the bars are
loss: 2e-6 relative + 1e-3 absolute per line; gradient: 2e-3 absolute (entries lie in [-1, 1])."""
import os

import numpy as np
import pytest
import torch

from oracle import ctc_loss_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ctc_loss.npz')


def _engine():
    from conformer_ocr_amd.ctc_decoder import _scratch_engine
    return _scratch_engine(torch.device('cuda', 0))


def _run(probits, targets, out_lens, label_lens, with_grad=True):
    nll, grad = _engine().ctc_loss(torch.from_numpy(np.ascontiguousarray(probits, dtype=np.float32)).cuda(), out_lens, targets, label_lens, with_grad)
    torch.cuda.synchronize()
    return nll.cpu().numpy(), (grad.cpu().numpy() if grad is not None else None)


@pytest.mark.parametrize('name', ['mixed', 'wide', 'long'])
def test_matches_the_reference_criterion(name):
    d = np.load(GOLD)
    nll, grad = _run(d[name + '.probits'], d[name + '.targets'], d[name + '.out_lens'], d[name + '.label_lens'])
    np.testing.assert_allclose(nll, d[name + '.per_line64'], rtol=2e-6, atol=1e-3)
    assert abs(float(nll.astype(np.float64).sum()) - float(d[name + '.loss64'])) < 2e-6 * float(d[name + '.loss64']) + 1e-3
    np.testing.assert_allclose(grad, d[name + '.grad64'], atol=2e-3)
    lens = d[name + '.out_lens']
    for n in range(len(lens)):
        assert not grad[n, lens[n]:].any()                       # frames beyond the line: exactly zero
    only_loss, none = _run(d[name + '.probits'], d[name + '.targets'], d[name + '.out_lens'], d[name + '.label_lens'], with_grad=False)
    assert none is None
    np.testing.assert_array_equal(only_loss, nll)


@pytest.mark.parametrize('N,T,C,max_l', [(1, 1, 2, 1), (5, 33, 7, 16), (3, 130, 65, 64), (2, 260, 257, 127), (2, 520, 40, 255), (33, 300, 100, 120)])
def test_against_the_restatement_on_random_batches(N, T, C, max_l):
    g = np.random.default_rng(N * 1000 + T)
    probits = (g.standard_normal((N, T, C)) * 2.0).astype(np.float32)
    out_lens = g.integers(max(1, T // 2), T + 1, size=N)
    out_lens[0] = T
    label_lens = np.array([int(g.integers(0, min(max_l, l // 2) + 1)) for l in out_lens])
    label_lens[0] = min(max_l, T // 2)                              # the longest line carries the most labels (states per lane = template)
    targets = np.concatenate([g.integers(1, C, size=l) for l in label_lens] + [np.zeros(0, np.int64)])
    want_nll, want_grad = R.ctc_loss(probits, targets, out_lens, label_lens)
    nll, grad = _run(probits, targets, out_lens, label_lens)
    np.testing.assert_allclose(nll, want_nll, rtol=2e-6, atol=1e-3)
    np.testing.assert_allclose(grad, want_grad, atol=2e-3)
    # posteriors of a frame sum to one: each valid frame's gradient sums to softmax.sum() - 1 = 0
    for n in range(N):
        if want_nll[n] > 0:
            assert np.abs(grad[n, :out_lens[n]].sum(axis=-1)).max() < 2e-3


def test_edge_lines():
    C = 5
    probits = np.random.default_rng(1).standard_normal((5, 6, C)).astype(np.float32)
    #            empty target   no frames+labels  no frames, no labels   'aa' in 2 frames (infeasible)   'aa' in 3 frames (one path)
    out_lens = [6, 0, 0, 2, 3]
    label_lens = [0, 2, 0, 2, 2]
    targets = [3, 4, 1, 1, 2, 2]
    nll, grad = _run(probits, targets, out_lens, label_lens)
    want_nll, want_grad = R.ctc_loss(probits, targets, out_lens, label_lens)
    np.testing.assert_allclose(nll, want_nll, rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(grad, want_grad, atol=1e-5)
    assert nll[1] == 0 and nll[2] == 0 and nll[3] == 0 and not grad[1].any() and not grad[3].any()


def test_runs_are_bitwise_reproducible():
    d = np.load(GOLD)
    a = _run(d['long.probits'], d['long.targets'], d['long.out_lens'], d['long.label_lens'])
    b = _run(d['long.probits'], d['long.targets'], d['long.out_lens'], d['long.label_lens'])
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_argument_errors():
    eng = _engine()
    p = torch.zeros((1, 4, 3), device='cuda')
    with pytest.raises(ValueError):
        eng.ctc_loss(p, [4], [3], [1])              # label outside [1, ncls)
    with pytest.raises(ValueError):
        eng.ctc_loss(p, [4], [0], [1])              # blank as a label
    with pytest.raises(ValueError):
        eng.ctc_loss(p, [5], [1], [1])              # more valid frames than frames
    with pytest.raises(ValueError):
        eng.ctc_loss(p, [4], [1, 2], [1])           # targets / label_lens disagree
