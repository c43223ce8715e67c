"""The CPU oracle (oracle/conformer_ref.py) against the golden vectors made by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conformer_ocr_amd.spec import HParams, encoder_state_spec, flops_per_line, out_len
from oracle.conformer_ref import Oracle, out_len as oracle_out_len

TOL = 1e-4   # fp32 restatement vs fp32 reference, abs on logits of magnitude ~10 (measured <= 3e-5)


@pytest.mark.parametrize('name', ['tiny', 'tiny2', 'tiny8', 'cfg1'])
def test_oracle_matches_reference_logits(case, name):
    hp, state, image, lens, g = case(name)
    o = Oracle(hp, state, torch.float32)
    logits, olens = o.forward(torch.from_numpy(image), torch.from_numpy(lens))
    assert olens.dtype == torch.int32 and olens.tolist() == g['out_lens'].tolist()
    assert np.abs(logits.numpy() - g['logits']).max() <= TOL
    np.testing.assert_array_equal(logits.argmax(-1).numpy()[g['margins'] > 1e-3], g['labels'][g['margins'] > 1e-3])


VARIANT_NAMES = ['v_k15', 'v_k7', 'v_h8', 'v_nohalf', 'v_f8', 'v_ff2', 'v_d512k7', 'v_cfg1w']


@pytest.mark.parametrize('name', VARIANT_NAMES)
def test_oracle_matches_reference_on_hyper_parameter_variants(case, name):
    """Round 4: the restatement against the reference itself on other hyper-parameters of the measured model (make_golden.py VARIANTS:
    depthwise kernels 15 / 7, 8 heads of 32, the full-step residual with 97 classes, subsampling factor 8, feed-forward expansion 2, the
    wide model with kernel 7, the reference's default model at 1200-pixel lines)."""
    hp, state, image, lens, g = case(name)
    logits, olens = Oracle(hp, state, torch.float32).forward(torch.from_numpy(image), torch.from_numpy(lens))
    assert olens.tolist() == g['out_lens'].tolist()
    assert np.abs(logits.numpy() - g['logits']).max() <= TOL
    np.testing.assert_array_equal(logits.argmax(-1).numpy()[g['margins'] > 1e-3], g['labels'][g['margins'] > 1e-3])


def test_oracle_stage_taps_match_reference(case):
    hp, state, image, lens, g = case('tiny')
    taps = {}
    Oracle(hp, state, torch.float32).forward(torch.from_numpy(image), torch.from_numpy(lens), taps)
    names = [k[4:] for k in g.files if k.startswith('tap:')]
    assert len(names) == 3 + 5 * hp.num_encoder_layers
    for n in names:
        assert np.abs(taps[n].numpy() - g['tap:' + n]).max() <= 5e-6, n


def test_oracle_fp64_agrees(case):
    hp, state, image, lens, g = case('tiny')
    l64, _ = Oracle(hp, state, torch.float64).forward(torch.from_numpy(image), torch.from_numpy(lens))
    assert np.abs(l64.numpy() - g['logits']).max() <= TOL


def test_oracle_cfg2_two_lines(case):
    """Metric configuration: lines are independent (no cross-sample op), so the first two lines of the
    32-line golden batch are reproduced from a 2-line batch of the same width."""
    hp, state, image, lens, g = case('cfg2')
    logits, olens = Oracle(hp, state).forward(torch.from_numpy(image[:2]), torch.from_numpy(lens[:2]))
    assert olens.tolist() == [300, 300]
    assert np.abs(logits.numpy() - g['logits_head'][:2]).max() <= TOL


def test_padding_leak_is_reproduced(case):
    """a line's logits depend on the padded batch width -- the oracle must not mask."""
    hp, state, image, lens, g = case('tiny')
    o = Oracle(hp, state)
    alone, _ = o.forward(torch.from_numpy(image[1:2, :, :, :40]), torch.tensor([37]))
    padded = g['logits'][1, :alone.shape[1]]
    assert np.abs(alone.numpy()[0] - padded).max() > 1e-3


def test_state_dict_key_map_matches_reference(golden_meta):
    for name in ('tiny', 'tiny2', 'tiny8'):
        m = golden_meta[name]
        spec = encoder_state_spec(HParams(**m['hparams']))
        assert [(k, list(v[0])) for k, v in spec.items()] == [(k, s) for k, s, _ in m['encoder_state_keys']]


def test_cfg1_param_count(golden_meta):
    hp = HParams(**golden_meta['cfg1']['hparams'])
    n = sum(int(np.prod(s)) for s, kind in encoder_state_spec(hp).values() if kind == 'param')
    assert n == golden_meta['cfg1']['n_params_encoder'] == 8217904


def test_out_len_matches_reference_float_form():
    # calc_length (convolution.py:240-247) as the reference computes it: float32 div/floor
    for l in list(range(1, 70)) + [300, 400, 512, 1199, 1200, 2400, 19999]:
        x = torch.tensor(float(l))
        for _ in range(2):
            x = torch.floor(torch.div(x + (2 - 3), 2) + 1.0)
        assert out_len(l, 2) == int(x) == int(oracle_out_len(torch.tensor([l]), 2)[0])
    assert [out_len(w) for w in (1200, 512, 300)] == [300, 128, 75]


def test_flop_formula():
    from conformer_ocr_amd import synth
    assert abs(flops_per_line(synth.hparams('cfg2'), 1200) / 1e9 - 14.64) < 0.01
    assert abs(flops_per_line(synth.hparams('cfg1'), 512) / 1e9 - 2.25) < 0.01


def _greedy(labels):
    out, prev = [], -1
    for v in labels:
        if v != prev and v != 0:
            out.append(int(v))
        prev = v
    return out


@pytest.mark.parametrize('name,lines', [('cfg2_text', [0, 1, 17]), ('cfg4_text', None)])
def test_text_fixture_pins_both_oracle_modes(text_case, name, lines):
    """The "text" fixtures (a ground truth, a decoder fitted on the reference's encoder output: >= 99.9 % of frames with a top-2
    margin > 1).  fp32 restatement: logits of the stored lines within 1e-4, every frame label and greedy string equal to the
    reference's.  bf16-operand mode (the checker of the library's bf16 kernels): labels equal on every frame with margin > 1
    (a constant, about 4x the measured logit deviation), greedy strings identical on every line."""
    tc = text_case(name)
    # cfg4_text: one small batch of the bucketed set (the whole set takes minutes on the CPU); cfg2_text: 3 of the 32 lines
    if lines is None:
        b = min(range(len(tc.batches)), key=lambda k: tc.batches[k][0] * len(tc.batches[k][1]))
        image, lens, idx = tc.batch(b)
    else:
        image, lens, idx = tc.batch(0)
        image, lens, idx = image[lines], lens[lines], [idx[i] for i in lines]
    x, l = torch.from_numpy(image), torch.from_numpy(lens)
    lg32, ol = Oracle(tc.hp, tc.state).forward(x, l)
    lg16, _ = Oracle(tc.hp, tc.state, bf16_operands=True).forward(x, l)
    checked = total = 0
    for k, i in enumerate(idx):
        T = int(tc.out_lens[i])
        assert int(ol[k]) == T
        if f'logits_line{i}' in tc.g.files:
            assert np.abs(lg32[k].numpy() - tc.g[f'logits_line{i}']).max() <= TOL
        a32, a16 = lg32[k, :T].argmax(-1).numpy(), lg16[k, :T].argmax(-1).numpy()
        np.testing.assert_array_equal(a32, tc.labels[i])
        sel = tc.margins[i] > 1.0
        checked += int(sel.sum()); total += T
        np.testing.assert_array_equal(a16[sel], tc.labels[i][sel])
        assert _greedy(a32) == tc.ref_strings[i] == tc.texts[i]
        assert _greedy(a16) == tc.ref_strings[i]
    assert checked >= 0.95 * total
    dev = float((lg16 - lg32).abs().max())
    assert 1e-3 < dev < 0.5, dev                     # the two modes differ by bf16 rounding, not by an indexing error


def oracle_train_grads(hp, state, image, lens, targets, dtype=torch.float64):
    """Loss, probits and d loss / d parameter of the reference's training step through the oracle's train mode + torch autograd."""
    o = Oracle(hp, state, dtype)
    params = {k: v for k, v in o.w.items() if v.is_floating_point() and 'running_' not in k}
    for v in params.values():
        v.requires_grad_(True)
    probits, ol = o.forward_train(torch.from_numpy(image).to(dtype), torch.from_numpy(np.asarray(lens)))
    target = torch.tensor([c for s in targets for c in s], dtype=torch.long)
    tl = torch.tensor([len(s) for s in targets], dtype=torch.long)
    loss = torch.nn.functional.ctc_loss(torch.nn.functional.log_softmax(probits, -1).transpose(0, 1), target, ol.long(), tl,
                                        reduction='sum', zero_infinity=True)                 # model.py:119,136-142
    loss.backward()
    return float(loss.detach()), probits.detach().numpy(), {k: v.grad.numpy() for k, v in params.items()}, o.bn_batch_stats


def test_oracle_train_mode_matches_the_reference_training_step():
    """tests/golden/tiny_train.npz: the reference encoder in train mode (batch-statistics BatchNorm, dropout 0) + decoder + CTC loss,
    `loss.backward()` in float64.  The oracle's train mode reproduces the loss, the probits, EVERY parameter gradient and the
    BatchNorm running statistics the step leaves (momentum 0.1, unbiased batch variance)."""
    import os
    from conformer_ocr_amd import synth
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'tiny_train.npz'))
    hp = synth.hparams('tiny')
    state = synth.make_state_dict(hp, seed=4321, decoder_gain=1.0)
    image, lens = synth.make_lines(3, hp.height, 64, seed=4321, widths=[64, 37, 50])
    loss, probits, grads, bn = oracle_train_grads(hp, state, image, lens, [[3, 1, 4], [1, 5], [9, 2, 6, 5]])
    assert abs(loss - float(g['loss'])) <= 1e-9 * abs(float(g['loss']))
    assert np.abs(probits - g['probits']).max() <= 1e-10
    names = [k[5:] for k in g.files if k.startswith('grad:')]
    assert sorted(names) == sorted(grads)
    for k in names:
        ref = g['grad:' + k]
        assert np.abs(grads[k].reshape(ref.shape) - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), k
    M = probits.shape[0] * probits.shape[1]
    for l, (mu, var) in bn.items():
        p = f'encoder.layers.{l}.sequential.2.module.sequential.5.'
        rm = 0.9 * state[p + 'running_mean'].astype(np.float64) + 0.1 * mu.numpy()
        rv = 0.9 * state[p + 'running_var'].astype(np.float64) + 0.1 * var.numpy() * M / (M - 1)
        assert np.abs(rm - g['buf:' + p + 'running_mean']).max() <= 1e-9 and np.abs(rv - g['buf:' + p + 'running_var']).max() <= 1e-9


def test_oracle_train_mode_matches_the_reference_at_the_metric_models_shapes():
    """tests/golden/cfg2x2_train.npz (round 4): the reference's training step in float64 at D = 256, 4 heads of 64, 256 conv channels, kernel
    31, two blocks -- loss, probits, and per parameter 64 sampled gradient entries + sum / sum of magnitudes / L2 norm / largest magnitude."""
    import os
    from conformer_ocr_amd import synth
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'cfg2x2_train.npz'))
    hp = synth.hparams('cfg2', num_encoder_layers=2)
    state = synth.make_state_dict(hp, seed=5, decoder_gain=1.0)
    image, lens = synth.make_lines(2, hp.height, 120, seed=5, widths=[120, 77])
    loss, probits, grads, bn = oracle_train_grads(hp, state, image, lens, [[5, 9, 9, 3], [17]])
    assert abs(loss - float(g['loss'])) <= 1e-9 * abs(float(g['loss']))
    assert np.abs(probits - g['probits']).max() <= 1e-5
    names = [k[3:] for k in g.files if k.startswith('gi:')]
    assert sorted(names) == sorted(grads)
    for k in names:
        flat, n = grads[k].reshape(-1), g['gn:' + k]
        assert np.abs(flat[g['gi:' + k]] - g['gs:' + k]).max() <= 1e-9 * max(1.0, n[3]), k
        got = np.array([flat.sum(), np.abs(flat).sum(), np.sqrt((flat ** 2).sum()), np.abs(flat).max()])
        assert np.abs(got - n).max() <= 1e-8 * max(1.0, n[1]), k

