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
    # subsampling factor 2 (conv.0 + ReLU only in front of the output linear)
    'tiny2': dict(hp=lambda: synth.hparams('tiny', subsampling_factor=2), seed=78, n=2, W=40, widths=[40, 27], targets=[[2, 7, 1], [4]], gain=1.0),
    # subsampling factor 8 (one more depthwise / pointwise frontend stage)
    'tiny8': dict(hp=lambda: synth.hparams('tiny', subsampling_factor=8, height=32), seed=77, n=2, W=96, widths=[96, 61], targets=[[2, 7], [4]], gain=1.0),
    # the metric model's shapes (D=256, 4 heads of 64, 256 conv channels, kernel 31), two blocks, short lines
    'cfg2x2': dict(hp=lambda: synth.hparams('cfg2', num_encoder_layers=2), seed=5, n=2, W=120, widths=[120, 77], targets=[[5, 9, 9, 3], [17]], gain=1.0),
}


@pytest.mark.parametrize('name', ['tiny', 'tiny2', 'tiny8', 'cfg2x2'])
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


def test_training_step_against_the_reference_at_the_metric_models_shapes():
    """tests/golden/cfg2x2_train.npz (round 4): the HIP training step against the REFERENCE's own float64 step at the metric model's shapes
    (two blocks), without the oracle in between: loss, probits, and for every parameter the 64 sampled gradient entries and the gradient's
    L2 norm / largest magnitude the fixture holds."""
    import os
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'cfg2x2_train.npz'))
    c = CASES['cfg2x2']
    hp = c['hp']()
    state = synth.make_state_dict(hp, seed=c['seed'], decoder_gain=c['gain'])
    image, lens = synth.make_lines(c['n'], hp.height, c['W'], seed=c['seed'], widths=c['widths'])
    eng = _engine(hp, state)
    tg = [x for s in c['targets'] for x in s]
    assert tg == g['target'].tolist()
    loss = eng.train_step(torch.from_numpy(image[:, 0]).cuda(), lens, tg, [len(s) for s in c['targets']])
    assert abs(loss - float(g['loss'])) <= 2e-4 * abs(float(g['loss'])), (loss, float(g['loss']))
    bad, n = {}, 0
    for k, (shape, kind) in model_state_spec(hp).items():
        if kind != 'param':
            continue
        n += 1
        got, ref, nrm = eng.train_grad(k).reshape(-1), g['gs:' + k], g['gn:' + k]
        err = float(np.abs(got[g['gi:' + k]] - ref).max())
        l2 = float(np.sqrt((got.astype(np.float64) ** 2).sum()))
        if not (err <= 2e-3 * nrm[3] + 1e-5 and abs(l2 - nrm[2]) <= 2e-3 * nrm[2] + 1e-5 and abs(float(np.abs(got).max()) - nrm[3]) <= 2e-3 * nrm[3] + 1e-5):
            bad[k] = (err, l2, nrm.tolist())
    assert n == sum(1 for f in g.files if f.startswith('gi:')) and not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1][0])[:8])


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


@pytest.mark.parametrize('n,w', [(3, 232), (1, 70), (4, 520)])
def test_attention_as_batched_products_equals_the_row_kernels(n, w, monkeypatch):
    """The training step's attention runs as batched exact-fp32 MFMA products when d_head is a multiple of 32 (cfg2: 4 heads of 64),
    else (and with COCR_TRAIN_ATTN_NAIVE=1) one wave per query / key row.  Same dropout masks (a function of seed, site and element):
    loss and EVERY gradient of the two forms agree to fp32 summation order, with all four dropout sites active."""
    c = CASES['cfg2x2']
    hp = c['hp']()
    state = synth.make_state_dict(hp, seed=c['seed'], decoder_gain=1.0)
    widths = [w, max(9, w * 3 // 5), max(9, w - 31), w][:n]
    image, lens = synth.make_lines(n, hp.height, w, seed=9, widths=widths)
    tl = [4, 1, 3, 2][:n]
    tg = [5, 9, 9, 3, 17, 2, 2, 40, 7, 7][:sum(tl)]
    x = torch.from_numpy(image[:, 0]).cuda()
    p = (0.1, 0.1, 0.1, 0.1)
    a = _engine(hp, state)
    la = a.train_step(x, lens, tg, tl, dropout=p, seed=21)
    ga = {k: a.train_grad(k).copy() for k, (shape, kind) in model_state_spec(hp).items() if kind == 'param'}
    monkeypatch.setenv('COCR_TRAIN_ATTN_NAIVE', '1')
    b = _engine(hp, state)
    lb = b.train_step(x, lens, tg, tl, dropout=p, seed=21)
    assert abs(la - lb) <= 1e-5 * abs(lb), (la, lb)
    bad = {}
    for k, g in ga.items():
        ref = b.train_grad(k)
        err = float(np.abs(g - ref).max())
        # (the key projection's bias: an exactly zero gradient, rounding noise only; v_bias sums ~10^5 products per entry in two different
        # orders: 0.6e-4 ... 1.2e-4 of its largest entry measured, depending on the rounding of the values upstream)
        if not err <= 2.5e-4 * float(np.abs(ref).max()) + 1e-5:
            bad[k] = (err, float(np.abs(ref).max()))
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1][0])[:12])


def test_medium_matmul_precision_stays_close_to_the_exact_step():
    """matmul_precision 'medium' (the reference's torch.set_float32_matmul_precision('medium')): the Linear / pointwise-conv products of
    the step on bf16-rounded operands with fp32 accumulation.  Loss within 1e-3 relative of the exact-fp32 step, every gradient within
    10 % of the largest entry of its tensor (bf16 operand rounding through two blocks, forward and backward); bit-reproducible."""
    c = CASES['cfg2x2']
    hp = c['hp']()
    state = synth.make_state_dict(hp, seed=c['seed'], decoder_gain=1.0)
    image, lens = synth.make_lines(3, hp.height, 232, seed=9, widths=[232, 137, 200])
    tg, tl = [5, 9, 9, 3, 17, 2, 2, 40], [4, 1, 3]
    x = torch.from_numpy(image[:, 0]).cuda()
    a = _engine(hp, state)
    la = a.train_step(x, lens, tg, tl)
    b = HipRecognizer(hp, torch.device('cuda', 0), 'fp32')
    b.load_state(state)
    b.train_begin('medium')
    lb = b.train_step(x, lens, tg, tl)
    assert lb != la and abs(la - lb) <= 1e-3 * abs(la), (la, lb)
    names = [k for k, (shape, kind) in model_state_spec(hp).items() if kind == 'param']
    scale = max(float(np.abs(a.train_grad(k)).max()) for k in names)        # (the key projection's bias has an exactly zero gradient: absolute floor)
    worst = {}
    for k in names:
        ref, got = a.train_grad(k), b.train_grad(k)
        err = float(np.abs(got - ref).max())
        if err > 0.1 * float(np.abs(ref).max()) + 1e-3 * scale:
            worst[k] = (err, float(np.abs(ref).max()))
    assert not worst, dict(sorted(worst.items(), key=lambda kv: -kv[1][0])[:8])
    g1 = b.train_grad('decoder.weight').copy()
    assert b.train_step(x, lens, tg, tl) == lb and np.array_equal(g1, b.train_grad('decoder.weight'))


def test_weight_gradients_from_k_major_operands_equal_those_from_transposed_copies(monkeypatch):
    """'medium' precision, round 4: dW = dY^T X straight from the row-major bf16 copies of dY and X (gemm_tn_kernel: ds_read_b64_tr_b16
    fragments, no transposed copies) against the same product on transposed copies (COCR_TRAIN_NO_TN=1: the form before).  The same bf16
    operands, the same split of the rows over workgroups; only the order of the 32 products inside a k-chunk differs: every gradient within
    1e-5 of its tensor's largest entry, the loss bit-equal (the forward is the same code)."""
    c = CASES['cfg2x2']
    hp = c['hp']()
    state = synth.make_state_dict(hp, seed=c['seed'], decoder_gain=1.0)
    image, lens = synth.make_lines(5, hp.height, 400, seed=19, widths=[400, 137, 333, 200, 64])
    tg, tl = [5, 9, 9, 3, 17, 2, 2, 40, 7, 7, 1], [4, 1, 3, 2, 1]
    x = torch.from_numpy(image[:, 0]).cuda()

    def run():
        e = HipRecognizer(hp, torch.device('cuda', 0), 'fp32')
        e.load_state(state)
        e.train_begin('medium')
        return e, e.train_step(x, lens, tg, tl)
    monkeypatch.setenv('COCR_TRAIN_NO_TN', '1')
    a, la = run()
    monkeypatch.delenv('COCR_TRAIN_NO_TN')
    b, lb = run()
    assert la == lb
    names = [k for k, (shape, kind) in model_state_spec(hp).items() if kind == 'param']
    scale = max(float(np.abs(a.train_grad(k)).max()) for k in names)        # (the key projection's bias has an exactly zero gradient: absolute floor)
    bad = {}
    for k in names:
        ref, got = a.train_grad(k), b.train_grad(k)
        err, top = float(np.abs(got - ref).max()), float(np.abs(ref).max())
        if not err <= 1e-5 * top + 1e-6 * scale:
            bad[k] = (err, top)
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1][0])[:8])


def test_trainer_follows_the_reference_training_loop():
    """conformer_ocr_amd.train.Trainer = training_step + configure_optimizers + optimizer_step / lr_scheduler_step of the reference
    (model.py:147-152,238-321): warm-up exactly as the reference applies it, epoch-wise schedules equal to torch's schedulers, the loss
    of a fixed batch falls, sync_module hands the trained values (and moved BatchNorm statistics) to the drop-in class."""
    from conformer_ocr_amd.codec import ascii_codec
    from conformer_ocr_amd.pred import PytorchRecognitionModel
    from conformer_ocr_amd.train import Trainer
    c = CASES['tiny']
    hp = c['hp']()
    state = synth.make_state_dict(hp, seed=c['seed'], decoder_gain=1.0)
    image, lens = synth.make_lines(c['n'], hp.height, c['W'], seed=c['seed'], widths=c['widths'])
    net = PytorchRecognitionModel(**hp.as_dict(), input_dropout_p=0.0, feed_forward_dropout_p=0.1, attention_dropout_p=0.1, conv_dropout_p=0.1,
                                  codec=ascii_codec(hp.num_classes), compute_dtype='fp32')
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    net = net.to('cuda:0').eval()
    before = net.predict_labels(torch.from_numpy(image).cuda(), torch.from_numpy(lens))
    w0 = net.nn.state_dict()['decoder.weight'].clone()
    rm0 = net.nn.state_dict()['encoder.layers.0.sequential.2.module.sequential.5.running_mean'].clone()
    batch = {'image': torch.from_numpy(image), 'seq_lens': torch.from_numpy(lens), 'target': torch.tensor([x for s in c['targets'] for x in s]),
             'target_lens': torch.tensor([len(s) for s in c['targets']])}
    tr = Trainer(net, lr=2e-3, weight_decay=1e-2, warmup=3, schedule='cosine', cos_t_max=4, cos_min_lr=1e-4, seed=5)
    lrs, losses = [], []
    for _ in range(12):
        lrs.append(tr.lr)
        losses.append(tr.training_step(batch))
    np.testing.assert_allclose(lrs[:5], [2e-3, 2e-3 / 3, 4e-3 / 3, 2e-3, 2e-3], rtol=1e-12)         # model.py:246-252
    assert losses[-1] < 0.7 * losses[0]
    opt = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=2e-3)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 4, 1e-4)
    for _ in range(6):
        opt.step(); sch.step()
        assert abs(tr.end_epoch() - sch.get_last_lr()[0]) <= 1e-12
    tr.sync_module()
    sd = net.nn.state_dict()
    assert float((sd['decoder.weight'] - w0).abs().max()) > 1e-4
    assert float((sd['encoder.layers.0.sequential.2.module.sequential.5.running_mean'] - rm0).abs().max()) > 1e-4
    after = net.predict_labels(torch.from_numpy(image).cuda(), torch.from_numpy(lens))      # the inference path re-packs the trained weights
    assert isinstance(after, list) and len(after) == len(before)
    with pytest.raises(ValueError):
        Trainer(net, schedule='1cycle')                                                     # model.py:309 rejects it too


@pytest.mark.timeout(300)
def test_training_from_random_weights_learns_to_read_the_text_lines():
    """End to end: a 2-block model of the metric's shapes, RANDOM weights, 300 AdamW steps (dropout 0.1, warm-up 10) on 16 synthetic text
    lines with their ground truth (conformer_ocr_amd.synth.make_text_lines).  The CTC loss falls by two orders of magnitude and the trained
    model -- served by the bf16 inference path after sync_module -- reads the lines: CER <= 0.02 against the ground truth (1.0 before)."""
    from conformer_ocr_amd.codec import ascii_codec
    from conformer_ocr_amd.evaluate import ErrorRate
    from conformer_ocr_amd.pred import PytorchRecognitionModel
    from conformer_ocr_amd.train import Trainer
    hp = synth.hparams('cfg2', num_encoder_layers=2)
    state = synth.make_state_dict(hp, seed=1, decoder_gain=1.0)
    net = PytorchRecognitionModel(**hp.as_dict(), input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1, conv_dropout_p=0.1,
                                  codec=ascii_codec(hp.num_classes), compute_dtype='bf16')
    net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    net = net.to('cuda:0').eval()
    image, lens, texts, _ = synth.make_text_lines(16, hp.height, 600, seed=3)
    batch = {'image': torch.from_numpy(image).cuda(), 'seq_lens': torch.from_numpy(lens), 'target': torch.tensor([c for t in texts for c in t]),
             'target_lens': torch.tensor([len(t) for t in texts])}

    def cer():
        pred = net.predict_labels(batch['image'], batch['seq_lens'])
        e = ErrorRate()
        e.update([[r[0] for r in line] for line in pred], texts)
        return e.compute()
    assert cer() >= 0.9
    tr = Trainer(net, lr=1e-3, weight_decay=1e-2, warmup=10)
    losses = [tr.training_step(batch) for _ in range(300)]
    assert losses[-1] < 0.01 * losses[0]
    tr.sync_module()
    assert cer() <= 0.02
