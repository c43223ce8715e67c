"""GPU: the training step behind torch autograd (conformer_ocr_amd/autograd.py; reference model.py:129-152,283-289: `loss.backward()`
+ any torch optimizer over `nn.parameters()`).

* `.grad` on every parameter of `net.nn` after `net.training_step(batch).backward()` equals torch autograd through the oracle's train
  mode in float64 (the oracle pinned on the reference's own training step, tests/golden/tiny_train.npz);
* three steps of `torch.optim.AdamW(net.nn.parameters())` equal three steps of the library's own `cocr_train_adamw` (2e-6), and the
  BatchNorm running statistics in the module follow;
* gradient accumulation (two backward calls without zero_grad) adds, as autograd promises; SGD with momentum drives the same step."""
import numpy as np
import pytest
import torch

from conformer_ocr_amd import synth
from conformer_ocr_amd.codec import ascii_codec
from conformer_ocr_amd.engine import HipRecognizer
from conformer_ocr_amd.pred import PytorchRecognitionModel
from conformer_ocr_amd.spec import model_state_spec
from tests.test_oracle import oracle_train_grads

pytestmark = pytest.mark.gpu
CASE = dict(seed=4321, n=3, W=64, widths=[64, 37, 50], targets=[[3, 1, 4], [1, 5], [9, 2, 6, 5]])


def _setup(drop=0.0):
    hp = synth.hparams('tiny')
    state = synth.make_state_dict(hp, seed=CASE['seed'], decoder_gain=1.0)
    image, lens = synth.make_lines(CASE['n'], hp.height, CASE['W'], seed=CASE['seed'], widths=CASE['widths'])
    net = PytorchRecognitionModel(**hp.as_dict(), input_dropout_p=drop, feed_forward_dropout_p=drop, attention_dropout_p=drop,
                                  conv_dropout_p=drop, codec=ascii_codec(hp.num_classes), compute_dtype='fp32')
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    net = net.to('cuda:0').train()
    batch = {'image': torch.from_numpy(image).cuda(), 'seq_lens': torch.from_numpy(lens),
             'target': torch.tensor([x for s in CASE['targets'] for x in s]), 'target_lens': torch.tensor([len(s) for s in CASE['targets']])}
    return hp, state, image, lens, net, batch


def test_frozen_holders_say_so():
    hp, state, image, lens, net, batch = _setup()
    with pytest.raises(RuntimeError, match='requires_grad_'):
        net.training_step(batch)


def test_grad_on_every_parameter_equals_autograd_of_the_oracle():
    hp, state, image, lens, net, batch = _setup()
    net.nn.requires_grad_(True)
    loss64, probits64, grads64, bn = oracle_train_grads(hp, state, image, lens, CASE['targets'])
    loss = net.training_step(batch)
    assert loss.requires_grad and loss.dim() == 0 and loss.is_cuda
    assert abs(float(loss.detach()) - loss64) <= 2e-4 * abs(loss64)
    assert all(p.grad is None for p in net.nn.parameters())
    loss.backward()
    params = dict(net.nn.named_parameters())
    bad, n = {}, 0
    for k, (shape, kind) in model_state_spec(hp).items():
        if kind != 'param':
            continue
        n += 1
        got, ref = params[k].grad.cpu().numpy(), grads64[k].reshape(shape)
        err = float(np.abs(got - ref).max())
        if not err <= 2e-3 * float(np.abs(ref).max()) + 1e-5:
            bad[k] = (err, float(np.abs(ref).max()))
    assert n == len(grads64) == len(params) and not bad, bad
    # the module's BatchNorm buffers moved like nn.BatchNorm1d's in train mode (momentum 0.1, unbiased batch variance)
    bufs = dict(net.nn.named_buffers())
    M = probits64.shape[0] * probits64.shape[1]
    for l, (mu, var) in bn.items():
        p = f'encoder.layers.{l}.sequential.2.module.sequential.5.'
        rm = 0.9 * state[p + 'running_mean'].astype(np.float64) + 0.1 * mu.numpy()
        rv = 0.9 * state[p + 'running_var'].astype(np.float64) + 0.1 * var.numpy() * M / (M - 1)
        assert np.abs(bufs[p + 'running_mean'].cpu().numpy() - rm).max() <= 1e-5
        assert np.abs(bufs[p + 'running_var'].cpu().numpy() - rv).max() <= 1e-5
        assert int(bufs[p + 'num_batches_tracked']) == 1
    # accumulation: a second backward without zero_grad adds the second step's gradients (same batch, moved running statistics do
    # not enter a train-mode step: the same gradients again)
    g1 = {k: p.grad.clone() for k, p in params.items()}
    net.training_step(batch).backward()
    for k, p in params.items():
        assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-5, atol=1e-7), k
    # a scaled loss scales the gradients (the chain rule through the op)
    for p in params.values():
        p.grad = None
    (0.25 * net.training_step(batch)).backward()
    for k, p in params.items():
        assert torch.allclose(p.grad, 0.25 * g1[k], rtol=1e-5, atol=1e-8), k


# Parameters with a direction of EXACTLY zero gradient: the key projection's bias (a constant added to every key shifts all scores of a
# query alike: softmax does not see it) and the positional projection's weight (its product with the sinusoid table's constant
# direction).  What reaches Adam there is rounding noise, which Adam normalises to +-lr per step -- and two runs' noise differs as soon
# as their parameters differ in the last bit.  The Adam comparison leaves these tensors out; the SGD comparison below (linear in the
# gradient: noise stays noise) covers every parameter.
ZERO_GRAD_DIRECTIONS = ('key_proj.linear.bias', 'pos_proj.linear.weight')


def _twin(hp, state):
    eng = HipRecognizer(hp, torch.device('cuda', 0), 'fp32')
    eng.load_state(state)
    eng.train_begin()
    return eng


def test_torch_adamw_on_the_module_equals_the_library_optimizer():
    hp, state, image, lens, net, batch = _setup()
    net.nn.requires_grad_(True)
    eng = _twin(hp, state)
    x = torch.from_numpy(image[:, 0]).cuda()
    tg, tl = batch['target'].tolist(), batch['target_lens'].tolist()
    opt = torch.optim.AdamW(net.nn.parameters(), lr=1e-3, weight_decay=1e-2)
    losses = []
    names = [k for k, (_, kind) in model_state_spec(hp).items() if kind == 'param']
    for step in range(3):
        want_loss = eng.train_step(x, lens, tg, tl)
        eng.train_adamw(1e-3, weight_decay=1e-2)
        opt.zero_grad()
        loss = net.training_step(batch)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        assert abs(losses[-1] - want_loss) <= 1e-5 * abs(want_loss), (step, losses[-1], want_loss)
        params = dict(net.nn.named_parameters())
        for k in names:
            if k.endswith(ZERO_GRAD_DIRECTIONS):
                continue
            assert np.abs(eng.train_value(k) - params[k].detach().cpu().numpy()).max() <= 2e-6, (step, k)
        for k, b in net.nn.named_buffers():
            if k.endswith('running_mean') or k.endswith('running_var'):
                assert np.abs(eng.train_value(k) - b.cpu().numpy()).max() <= 1e-6, (step, k)
    assert losses[2] < losses[1] < losses[0]
    # the trained module serves: the inference path re-packs the new weights by itself
    net.eval()
    lg, _ = net.forward(batch['image'], batch['seq_lens'])
    assert bool(torch.isfinite(lg).all())


def test_torch_sgd_on_the_module_equals_plain_updates_with_the_library_gradients():
    """Every parameter, three steps: `torch.optim.SGD(net.nn.parameters())` through the autograd op against p -= lr * g with the
    gradients a second engine computes for the same parameter values."""
    hp, state, image, lens, net, batch = _setup()
    net.nn.requires_grad_(True)
    x = torch.from_numpy(image[:, 0]).cuda()
    tg, tl = batch['target'].tolist(), batch['target_lens'].tolist()
    names = [k for k, (_, kind) in model_state_spec(hp).items() if kind == 'param']
    cur = {k: np.asarray(v, dtype=np.float32).copy() for k, v in state.items()}
    opt = torch.optim.SGD(net.nn.parameters(), lr=1e-3)
    for step in range(3):
        eng = _twin(hp, cur)
        want_loss = eng.train_step(x, lens, tg, tl)
        for k in names:
            cur[k] = cur[k] - np.float32(1e-3) * eng.train_grad(k).reshape(cur[k].shape)
        for k in cur:
            if k.endswith('running_mean') or k.endswith('running_var'):
                cur[k] = eng.train_value(k).reshape(cur[k].shape)
        opt.zero_grad()
        loss = net.training_step(batch)
        loss.backward()
        opt.step()
        assert abs(float(loss.detach()) - want_loss) <= 1e-5 * abs(want_loss)
        params = dict(net.nn.named_parameters())
        for k in names:
            assert np.abs(cur[k] - params[k].detach().cpu().numpy()).max() <= 2e-6, (step, k)


def test_sgd_with_momentum_and_dropout_drive_the_same_op():
    """Another optimizer of the reference's list (model.py:285-289) and the constructor's dropout probabilities: runs, lowers the loss,
    reproducible per `dropout_seed`."""
    out = []
    for rep in range(2):
        hp, state, image, lens, net, batch = _setup(drop=0.1)
        net.nn.requires_grad_(True)
        net.dropout_seed = 5
        opt = torch.optim.SGD(net.nn.parameters(), lr=1e-4, momentum=0.9)
        ls = []
        for _ in range(6):
            opt.zero_grad()
            loss = net.training_step(batch)
            loss.backward()
            opt.step()
            ls.append(float(loss.detach()))
        out.append(ls)
    assert out[0] == out[1]
    assert out[0][-1] < out[0][0]
