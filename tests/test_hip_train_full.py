"""GPU: the training step of the WHOLE network (cocr_train_*: train-mode forward, CTC criterion, backward through decoder and encoder,
AdamW) against torch autograd through the oracle's train mode in float64 -- the oracle that tests/test_oracle.py pins on the reference's
own training step (tests/golden/tiny_train.npz: loss, probits, every gradient, running statistics)."""
import numpy as np
import pytest
import torch

from conformer_ocr_amd import synth
from conformer_ocr_amd.engine import HipRecognizer
from conformer_ocr_amd.spec import model_state_spec
from tests.test_oracle import oracle_train_grads

pytestmark = pytest.mark.gpu


def _engine(hp, state):
    eng = HipRecognizer(hp, torch.device('cuda', 0), 'fp32')
    eng.load_state(state)
    eng.train_begin()
    return eng


CASES = {
    # the reference's own fixture configuration (tiny_train.npz), ragged widths
    'tiny': dict(hp=lambda: synth.hparams('tiny'), seed=4321, n=3, W=64, widths=[64, 37, 50], targets=[[3, 1, 4], [1, 5], [9, 2, 6, 5]], gain=1.0),
    # subsampling factor 8 (one more depthwise / pointwise frontend stage)
    'tiny8': dict(hp=lambda: synth.hparams('tiny', subsampling_factor=8, height=32), seed=77, n=2, W=96, widths=[96, 61], targets=[[2, 7], [4]], gain=1.0),
    # the metric model's shapes (D=256, 4 heads of 64, 256 conv channels, kernel 31), two blocks, short lines
    'cfg2x2': dict(hp=lambda: synth.hparams('cfg2', num_encoder_layers=2), seed=5, n=2, W=120, widths=[120, 77], targets=[[5, 9, 9, 3], [17]], gain=1.0),
}


@pytest.mark.parametrize('name', ['tiny', 'tiny8', 'cfg2x2'])
def test_every_gradient_against_autograd_of_the_oracle(name):
    c = CASES[name]
    hp = c['hp']()
    state = synth.make_state_dict(hp, seed=c['seed'], decoder_gain=c['gain'])
    image, lens = synth.make_lines(c['n'], hp.height, c['W'], seed=c['seed'], widths=c['widths'])
    loss64, probits64, grads64, bn = oracle_train_grads(hp, state, image, lens, c['targets'])
    eng = _engine(hp, state)
    tg = [x for s in c['targets'] for x in s]
    loss = eng.train_step(torch.from_numpy(image[:, 0]).cuda(), lens, tg, [len(s) for s in c['targets']])
    assert abs(loss - loss64) <= 2e-4 * abs(loss64), (loss, loss64)
    # every parameter: |got - ref| <= 2e-3 max|ref| + 1e-5 (the key projection's bias has an exactly zero gradient -- a constant added to
    # every key shifts all scores of a query alike, softmax does not see it -- so a purely relative measure would divide noise by noise)
    bad, nparams = {}, 0
    for k, (shape, kind) in model_state_spec(hp).items():
        if kind != 'param':
            continue
        nparams += 1
        got, ref = eng.train_grad(k), grads64[k].reshape(shape)
        err = float(np.abs(got - ref).max())
        if not err <= 2e-3 * float(np.abs(ref).max()) + 1e-5:
            bad[k] = (err, float(np.abs(ref).max()))
    assert nparams == len(grads64) and not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1][0])[:12])
    # BatchNorm running statistics after the step (momentum 0.1, unbiased batch variance)
    M = probits64.shape[0] * probits64.shape[1]
    for l, (mu, var) in bn.items():
        p = f'encoder.layers.{l}.sequential.2.module.sequential.5.'
        rm = 0.9 * state[p + 'running_mean'].astype(np.float64) + 0.1 * mu.numpy()
        rv = 0.9 * state[p + 'running_var'].astype(np.float64) + 0.1 * var.numpy() * M / (M - 1)
        assert np.abs(eng.train_value(p + 'running_mean') - rm).max() <= 1e-5
        assert np.abs(eng.train_value(p + 'running_var') - rv).max() <= 1e-5


def test_adamw_steps_follow_torch_and_lower_the_loss():
    """Three optimizer steps on a fixed batch: after every step the parameters equal torch.optim.AdamW (fp32) fed with the SAME gradients
    (read back from the device: the gradients' own parity is the test above; Adam divides a gradient by its own magnitude, so feeding it
    the float64 gradients instead would compare rounding noise on the parameters whose exact gradient is zero -- the key projection's
    bias, the constant direction of the positional table).  The loss decreases; train_end + finalize serves the trained weights."""
    c = CASES['tiny']
    hp = c['hp']()
    state = synth.make_state_dict(hp, seed=c['seed'], decoder_gain=1.0)
    image, lens = synth.make_lines(c['n'], hp.height, c['W'], seed=c['seed'], widths=c['widths'])
    tg, tl = [x for s in c['targets'] for x in s], [len(s) for s in c['targets']]
    eng = _engine(hp, state)
    x = torch.from_numpy(image[:, 0]).cuda()
    names = [k for k, (_, kind) in model_state_spec(hp).items() if kind == 'param']
    tparams = {k: torch.tensor(np.asarray(state[k], dtype=np.float32), requires_grad=True) for k in names}
    opt = torch.optim.AdamW(list(tparams.values()), lr=1e-3, weight_decay=1e-2)
    losses = []
    for step in range(3):
        losses.append(eng.train_step(x, lens, tg, tl))
        for k in names:
            tparams[k].grad = torch.from_numpy(eng.train_grad(k).reshape(tparams[k].shape).copy())
        opt.step()
        eng.train_adamw(1e-3, weight_decay=1e-2)
        for k in names:
            assert np.abs(eng.train_value(k) - tparams[k].detach().numpy()).max() <= 2e-6, (step, k)
    assert losses[2] < losses[1] < losses[0]
    eng.train_end()
    eng.finalize()
    lg, _ = eng.forward(x, lens)
    assert bool(torch.isfinite(lg).all())


def test_dropout_is_reproducible_and_changes_the_step():
    """Dropout masks come from (seed, site, index): the same seed gives the same loss and gradients bit for bit, another seed a different
    step; the forward mask and the backward mask agree (a finite-difference check of one bias gradient with the masks held fixed)."""
    c = CASES['tiny']
    hp = c['hp']()
    state = synth.make_state_dict(hp, seed=c['seed'], decoder_gain=1.0)
    image, lens = synth.make_lines(c['n'], hp.height, c['W'], seed=c['seed'], widths=c['widths'])
    tg, tl = [x for s in c['targets'] for x in s], [len(s) for s in c['targets']]
    x = torch.from_numpy(image[:, 0]).cuda()
    p = (0.1, 0.1, 0.1, 0.1)
    name = 'encoder.layers.0.sequential.0.module.sequential.4.linear.bias'
    eng = _engine(hp, state)
    l1 = eng.train_step(x, lens, tg, tl, dropout=p, seed=11)
    g1 = eng.train_grad(name).copy()
    l2 = eng.train_step(x, lens, tg, tl, dropout=p, seed=11)
    assert l1 == l2 and np.array_equal(g1, eng.train_grad(name))
    l3 = eng.train_step(x, lens, tg, tl, dropout=p, seed=12)
    assert l3 != l1 and abs(l3 - eng.train_step(x, lens, tg, tl)) > 1e-4        # (dropout 0: yet another loss)
    # finite differences with the masks of seed 11 held fixed
    eps = 1e-2
    for j in (0, 3):
        ls = []
        for sgn in (+1, -1):
            st = dict(state)
            b = state[name].copy()
            b[j] += sgn * eps
            st[name] = b
            e2 = _engine(hp, st)
            ls.append(e2.train_step(x, lens, tg, tl, dropout=p, seed=11))
        fd = (ls[0] - ls[1]) / (2 * eps)
        assert abs(fd - g1[j]) <= 5e-2 * max(1.0, abs(g1[j])), (j, fd, g1[j])
