"""CPU: the CTC-loss restatement (oracle/ctc_loss_ref.py) against vectors made by torch's own CTCLoss -- the implementation the
reference calls (model.py:119,136-142) -- through tests/golden/make_ctc_loss_golden.py."""
import os

import numpy as np
import pytest

from oracle import ctc_loss_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ctc_loss.npz')


@pytest.mark.parametrize('name', ['mixed', 'wide', 'long'])
def test_restatement_matches_the_reference_criterion(name):
    d = np.load(GOLD)
    nll, grad = R.ctc_loss(d[name + '.probits'], d[name + '.targets'], d[name + '.out_lens'], d[name + '.label_lens'])
    np.testing.assert_allclose(nll, d[name + '.per_line64'], rtol=1e-12, atol=1e-10)
    assert abs(nll.sum() - float(d[name + '.loss64'])) < 1e-9 * max(1.0, float(d[name + '.loss64']))
    np.testing.assert_allclose(grad, d[name + '.grad64'], atol=1e-6)          # stored as float32
    # float32 torch agrees with float64 torch only to the precision of log-domain float32 arithmetic: the bar for the fp32 kernel
    np.testing.assert_allclose(d[name + '.per_line32'], d[name + '.per_line64'], rtol=2e-6, atol=1e-4)


def test_known_answers():
    # one frame, one label: nll = -log softmax[label]
    p = np.array([[[0.0, 1.0, 2.0]]])
    nll, grad = R.ctc_loss(p, [2], [1], [1])
    sm = np.exp(p[0, 0]) / np.exp(p[0, 0]).sum()
    assert abs(nll[0] + np.log(sm[2])) < 1e-12
    np.testing.assert_allclose(grad[0, 0], sm - np.array([0, 0, 1.0]), atol=1e-12)
    # empty target: every frame blank
    p = np.random.default_rng(0).standard_normal((1, 5, 4))
    nll, grad = R.ctc_loss(p, [], [5], [0])
    lp = R.log_softmax(p[0])
    assert abs(nll[0] + lp[:, 0].sum()) < 1e-12
    # a repeated label needs a blank in between: 'aa' does not fit 2 frames -> infinite -> 0 with zero gradient
    nll, grad = R.ctc_loss(np.zeros((1, 2, 3)), [1, 1], [2], [2])
    assert nll[0] == 0.0 and not grad.any()
    nll, _ = R.ctc_loss(np.zeros((1, 3, 3)), [1, 1], [3], [2])
    assert abs(nll[0] - 3 * np.log(3.0)) < 1e-12                                 # exactly one path: a - a


def test_gradient_is_the_derivative():
    g = np.random.default_rng(5)
    p = g.standard_normal((1, 9, 6))
    tgt = [2, 2, 5]
    nll, grad = R.ctc_loss(p, tgt, [9], [3])
    eps = 1e-6
    for (t, c) in [(0, 0), (3, 2), (8, 5), (4, 1)]:
        q = p.copy()
        q[0, t, c] += eps
        num = (R.ctc_loss(q, tgt, [9], [3])[0][0] - nll[0]) / eps
        assert abs(num - grad[0, t, c]) < 1e-5
