"""GPU parity of the MEASURED bf16 path (fused frontend, split-K output linear, 96-/32-row chain kernels, rel-pos attention):

* stage by stage against the oracle's bf16-operand mode (oracle/conformer_ref.py: same arithmetic, operands rounded where the
  kernels round them) -- the taps come from the TAPS instantiation of the chain kernels themselves, not from fallback kernels;
* the "text" fixtures (tests/golden/make_golden.py run_text_case: a ground truth, a decoder fitted on the reference's own
  encoder output, >= 99.9 % of frames with a top-2 margin > 1): frame labels equal to the reference's on every frame with
  margin > 1 -- a CONSTANT filter --, greedy strings identical on every line, CER against the ground truth 0;
* A/B: the chain / fused-frontend path against the one-kernel-per-product path of the same library on the same batch.
"""
import json
import os

import numpy as np
import pytest
import torch

from conformer_ocr_amd import synth
from conformer_ocr_amd.codec import ascii_codec
from conformer_ocr_amd.evaluate import ErrorRate, recognize
from conformer_ocr_amd.pred import PytorchRecognitionModel
from tests.hip_util import hip_tap, make_engine, run_hip

pytestmark = pytest.mark.gpu

MARGIN = 1.0            # constant label filter (text fixtures: >= 99.9 % of frames pass it)
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')


def _log(name, rec):
    os.makedirs(LOG, exist_ok=True)
    with open(os.path.join(LOG, 'parity.jsonl'), 'a') as fp:
        fp.write(json.dumps({'test': name, **rec}) + '\n')


def _greedy(labels):
    out, prev = [], -1
    for v in labels:
        if v != prev and v != 0:
            out.append(int(v))
        prev = v
    return out


def _oracle_taps(hp, state, image, lens, bf16):
    from oracle.conformer_ref import Oracle
    taps = {}
    lg, ol = Oracle(hp, state, bf16_operands=bf16).forward(torch.from_numpy(image), torch.from_numpy(lens), taps)
    return lg.numpy(), {k: v.numpy() for k, v in taps.items()}


def _ulp_ok(got, ref):
    """bf16-stored values: equal up to one bf16 rounding step of the reference value (a borderline rounding may flip) plus 5e-3:
    the stage's own LayerNorm output is rounded to bf16 before the product, and a flipped rounding there (0.4 % of an operand
    of magnitude 1..3, times a weight of ~0.1, a handful of elements per million) moves a projection by 1-3e-3."""
    return np.abs(got - ref) <= np.abs(ref) * 2.0 ** -7 + 5e-3


@pytest.mark.parametrize('n,w,ksz', [(3, 232, 31), (17, 1200, 31), (3, 232, 15), (17, 1200, 7)])
def test_chain_kernel_stages_in_isolation_against_bf16_oracle(n, w, ksz):
    """Two blocks of the metric's model on ragged lines; (3, 232) runs the 32-row workgroups, (17, 1200) = 5100 rows the 96-row
    form.  The taps come from the TAPS instantiation of the chain kernels themselves.  Every stage is checked IN ISOLATION: the
    bf16-operand oracle recomputes the stage from the HIP path's own input to it (the previous tap), so the only differences
    left are that one stage's accumulation order, its exp2 / rcp approximations and borderline bf16 roundings --
    fp32 stream taps within 6e-3 absolute, bf16-stored operands within one bf16 rounding step.  (Chained through all stages
    the same comparison reaches 2e-2 after two blocks: each flipped rounding is re-amplified by the following LayerNorms.)"""
    from oracle.conformer_ref import Oracle
    # conv_kernel_size 15 / 7: the stand-alone depthwise kernel + the chain shapes WITHOUT the depthwise prologue -- what those models
    # run in production; their taps, too, come from the chain kernels (cocr_api.hip no longer leaves the chain path in debug mode)
    hp = synth.hparams('cfg2', num_encoder_layers=2, conv_kernel_size=ksz)
    state = synth.make_state_dict(hp, seed=31, decoder_gain=1.0, style='text')
    widths = [max(40, w - 37 * i) for i in range(n)]
    image, lens = synth.make_lines(n, hp.height, w, seed=77, widths=widths)
    eng, logits, _ = run_hip(hp, state, image, lens, 'bf16', debug=True)
    N, T = n, logits.shape[1]
    o = Oracle(hp, state, bf16_operands=True)
    tap = lambda nm: torch.from_numpy(np.ascontiguousarray(hip_tap(eng, nm, hp, N, T)))
    worst, bad = {}, {}

    def stream(nm, got, ref, tol=6e-3):
        worst[nm] = float((got - ref).abs().max())
        if not worst[nm] <= tol:
            bad[nm] = worst[nm]

    def stored(nm, got, ref):
        ok = _ulp_ok(got.numpy(), ref.numpy())
        worst[nm] = float((got - ref).abs().max())
        if not ok.all():
            bad[nm] = (int((~ok).sum()), worst[nm])

    with torch.no_grad():
        x = torch.from_numpy(image).squeeze(1)
        z3 = o.front_pw(o.front_conv12(x), 3)
        stored('front.z3', tap('front.z3'), z3)
        stream('front.y', tap('front.y'), o.front_out(tap('front.z3')))
        y = tap('front.y')
        for l in range(hp.num_encoder_layers):
            stream(f'l{l}.ffn1', tap(f'l{l}.ffn1'), o.ffn(y, l, 0))
            y = tap(f'l{l}.ffn1')
            t = {}
            ref = o.mhsa(y, l, t)
            for nm in ('q', 'k', 'v'):
                stored(f'l{l}.{nm}', tap(f'l{l}.{nm}'), t[f'l{l}.{nm}'])
            stored(f'l{l}.ctx', tap(f'l{l}.ctx'), t[f'l{l}.ctx'])
            stream(f'l{l}.mhsa', tap(f'l{l}.mhsa'), ref)
            y = tap(f'l{l}.mhsa')
            t = {}
            o.convmod(y, l, t)
            stored(f'l{l}.glu', tap(f'l{l}.glu'), t[f'l{l}.glu'])
            t = {}
            ref = o.convmod(y, l, t, glu=tap(f'l{l}.glu'))             # the depthwise stage from the HIP path's own GLU output: one flipped
            stored(f'l{l}.dw', tap(f'l{l}.dw'), t[f'l{l}.dw'])          # bf16 rounding of a large GLU value times a tap is not this stage's error
            stream(f'l{l}.conv', tap(f'l{l}.conv'), ref)
            y = tap(f'l{l}.conv')
            stream(f'l{l}.ffn2', tap(f'l{l}.ffn2'), o.ffn(y, l, 3))
            y = tap(f'l{l}.ffn2')
            ref = o._ln(y, f'encoder.layers.{l}.sequential.4')
            if l + 1 < hp.num_encoder_layers:
                stream(f'l{l}.out', tap(f'l{l}.out'), ref)
            else:
                stored(f'l{l}.out', tap(f'l{l}.out'), o.r(ref))         # last block: exists only as the bf16 decoder operand
            y = tap(f'l{l}.out')
        stream('logits', torch.from_numpy(logits), y @ o.w['decoder.weight'].t() + o.w['decoder.bias'])
    # and chained: the whole path against both oracle modes
    lg16, _ = _oracle_taps(hp, state, image, lens, True)
    lg32, _ = _oracle_taps(hp, state, image, lens, False)
    worst['chained logits vs bf16 oracle'] = float(np.abs(logits - lg16).max())
    worst['chained logits vs fp32 oracle'] = float(np.abs(logits - lg32).max())
    _log(f'chain_stages_n{n}_w{w}_k{ksz}', worst)
    assert not bad, bad
    assert worst['chained logits vs bf16 oracle'] <= 0.05 and worst['chained logits vs fp32 oracle'] <= 0.15


def test_rows_per_workgroup_forms_are_bit_identical():
    """cocr_set_chain_rows: 32-, 48- and 96-row workgroups of the row-chain kernels compute every row with the same arithmetic in the same
    order (explicit fused multiply-adds in the LayerNorm statistics: no instantiation-dependent contraction) -- bit-identical logits."""
    hp = synth.hparams('cfg2', num_encoder_layers=2)
    state = synth.make_state_dict(hp, seed=3, decoder_gain=1.0, style='text')
    image, lens, _, _ = synth.make_text_lines(17, hp.height, 1200, seed=5)
    x = torch.from_numpy(image[:, 0]).cuda()
    eng = make_engine(hp, state, 'bf16')
    out = {}
    for rows in (0, 96, 48, 32):
        eng.set_chain_rows(rows)
        lg, _ = eng.forward(x, lens)
        torch.cuda.synchronize()
        out[rows] = lg.cpu().numpy().copy()
    for rows in (96, 48, 32):
        np.testing.assert_array_equal(out[0], out[rows])


def _run_text_fixture(tc, dtype, env=None):
    """All batches of a text fixture through the C ABI; returns per-line (labels over the line's own frames, logits of lines 0/1)."""
    eng = make_engine(tc.hp, tc.state, dtype)
    labels, heads = {}, {}
    for b in range(len(tc.batches)):
        image, lens, idx = tc.batch(b)
        lg, ol = eng.forward(torch.from_numpy(image[:, 0]).cuda(), lens)
        torch.cuda.synchronize()
        lg = lg.cpu().numpy()
        for k, i in enumerate(idx):
            assert int(ol[k]) == int(tc.out_lens[i])
            labels[i] = lg[k, :int(ol[k])].argmax(-1)
            if f'logits_line{i}' in tc.g.files:
                heads[i] = lg[k]
    return labels, heads


@pytest.mark.parametrize('name', ['cfg1_text', 'cfg2_text', 'cfg4_text'])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_text_fixture_labels_and_strings(text_case, name, dtype):
    """cfg2_text = BASELINE configs[1]'s model and batch (32 x 96x1200); cfg4_text = configs[3] (D=512, L=16, widths 400..2400 in
    200-px buckets); cfg1_text = the reference's default model (D=144: in bf16 the zero-padded 256-wide layout) on the metric's batch.  fp32: logits within 1e-3, every frame label equal.  bf16: labels equal on every frame with margin > 1
    (>= 95 % of frames, a constant filter), greedy strings identical to the reference's on EVERY line, = the ground truth."""
    tc = text_case(name)
    labels, heads = _run_text_fixture(tc, dtype)
    dev = max(float(np.abs(heads[i] - tc.g[f'logits_line{i}']).max()) for i in heads)
    checked = total = mism = mism_all = 0
    strings_equal = 0
    for i in range(tc.n):
        sel = tc.margins[i] > MARGIN
        checked += int(sel.sum()); total += sel.size
        mism += int((labels[i][sel] != tc.labels[i][sel]).sum())
        mism_all += int((labels[i] != tc.labels[i]).sum())
        strings_equal += _greedy(labels[i]) == tc.ref_strings[i]
    cer = ErrorRate()
    cer.update([_greedy(labels[i]) for i in range(tc.n)], tc.texts)
    _log(f'{name}_{dtype}', {'max_abs_logit_dev': dev, 'frames': total, 'frames_checked': checked, 'label_mismatch_checked': mism,
                             'label_mismatch_all_frames': mism_all, 'strings_identical': strings_equal, 'lines': tc.n,
                             'cer_vs_truth': cer.compute()})
    assert checked >= 0.95 * total
    assert mism == 0
    assert strings_equal == tc.n
    assert cer.compute() == 0.0
    if dtype == 'fp32':
        assert dev <= 1e-3 and mism_all == 0
    else:
        assert dev <= 0.35           # measured 0.207 - 0.227 on logits of +-22 .. 28
        assert mism_all <= 3         # measured 0 / 0 / 1: the frames under the margin filter agree as well


def test_cfg2_text_logits_against_bf16_oracle(text_case):
    """The full 12-block measured path against the bf16-operand oracle on two lines of the metric batch."""
    tc = text_case('cfg2_text')
    image, lens, idx = tc.batch(0)
    eng = make_engine(tc.hp, tc.state, 'bf16')
    lg, _ = eng.forward(torch.from_numpy(image[:, 0]).cuda(), lens)       # the full 32-line batch: the 96-row kernels
    torch.cuda.synchronize()
    lg16, _ = _oracle_taps(tc.hp, tc.state, image[:2], lens[:2], True)
    lg32, _ = _oracle_taps(tc.hp, tc.state, image[:2], lens[:2], False)
    d16 = float(np.abs(lg[:2].cpu().numpy() - lg16).max())
    d32 = float(np.abs(lg[:2].cpu().numpy() - lg32).max())
    _log('cfg2_text_vs_oracles', {'vs_bf16_oracle': d16, 'vs_fp32_oracle': d32, 'logit_absmax': float(np.abs(lg32).max())})
    # 12 blocks on logits of +-30: every borderline bf16 rounding that falls the other way is re-amplified by the LayerNorms that
    # follow, so the chained comparison is only a little tighter than the one against the fp32 arithmetic (measured 0.18 vs 0.23);
    # the stage-level evidence is test_chain_kernel_stages_in_isolation_against_bf16_oracle
    assert d16 <= 0.3, d16
    assert d16 < d32


@pytest.mark.parametrize('flag', ['COCR_NO_CHAIN', 'COCR_NO_FRONT96', 'COCR_NO_DW_FUSE', 'COCR_NO_FRONT_CHAIN'])
def test_fast_path_against_per_product_kernels(text_case, flag, monkeypatch):
    """A/B inside the library: the chain kernels / the fused frontend / the fused depthwise prologue / the frontend's output linear as the
    first chain stage (COCR_NO_FRONT_CHAIN: as a split-K GEMM + reduction) against the one-kernel-per-product forms (same operands, same rounding points, other accumulation orders) on the metric batch."""
    tc = text_case('cfg2_text')
    image, lens, idx = tc.batch(0)
    x = torch.from_numpy(image[:, 0]).cuda()
    a, _ = make_engine(tc.hp, tc.state, 'bf16').forward(x, lens)
    monkeypatch.setenv(flag, '1')
    b, _ = make_engine(tc.hp, tc.state, 'bf16').forward(x, lens)
    torch.cuda.synchronize()
    a, b = a.cpu().numpy(), b.cpu().numpy()
    d = float(np.abs(a - b).max())
    la, lb = a.argmax(-1), b.argmax(-1)
    _log(f'ab_{flag}', {'max_abs_logit_diff': d, 'label_diff_frames': int((la != lb).sum())})
    if flag == 'COCR_NO_DW_FUSE':        # the stand-alone depthwise kernel accumulates in the prologue's order and the first product now starts
        assert d == 0.0, d               # from zero in both forms: bit-identical (that the flag selects another kernel shows in the profile)
    else:
        assert d > 0.0                   # the flag did select other kernels
    assert d <= 0.35, d                  # logits of +-30 after 12 blocks; other accumulation orders flip borderline bf16 roundings
    assert int((la != lb).sum()) == 0
    assert all(_greedy(la[n]) == _greedy(lb[n]) == tc.ref_strings[idx[n]] for n in range(len(idx)))


def test_narrow_model_as_a_zero_padded_wide_one(case, monkeypatch):
    """The reference's default model (encoder_dim 144, 4 heads of 36, feed-forward 576) runs in bf16 as a zero-padded 256 / 768-wide model
    through the row-chain kernels (cocr_api.hip: set_engine_dims).  A/B against the model's own layout on the per-product kernels
    (COCR_NO_PAD=1): the padding itself adds nothing (zeros), so the difference is what any other accumulation order costs; against
    the reference's fp32 logits both stay inside the bf16 bound and agree on every frame label outside the margin filter."""
    hp, state, image, lens, g = case('cfg1')
    x = torch.from_numpy(image[:, 0]).cuda()
    a, _ = make_engine(hp, state, 'bf16').forward(x, lens)
    monkeypatch.setenv('COCR_NO_PAD', '1')
    b, _ = make_engine(hp, state, 'bf16').forward(x, lens)
    torch.cuda.synchronize()
    a, b, ref = a.cpu().numpy(), b.cpu().numpy(), g['logits']
    d = float(np.abs(a - b).max())
    da, db = float(np.abs(a - ref).max()), float(np.abs(b - ref).max())
    _log('ab_COCR_NO_PAD', {'max_abs_logit_diff': d, 'padded_vs_reference': da, 'own_layout_vs_reference': db})
    assert d > 0.0                       # the flag did select the other layout
    assert d <= 0.45 and da <= 0.35 and db <= 0.35, (d, da, db)
    sel = g['margins'] > 0.7
    assert (a.argmax(-1)[sel] == g['labels'][sel]).all() and (b.argmax(-1)[sel] == g['labels'][sel]).all()


@pytest.mark.parametrize('D,heads,ksz,C', [(160, 4, 31, 32), (192, 8, 31, 64), (208, 4, 15, 32), (144, 2, 31, 32), (240, 8, 7, 256),
                                          (384, 8, 31, 64), (320, 4, 31, 32), (448, 8, 15, 256)])
def test_zero_padded_layouts_of_other_narrow_models(D, heads, ksz, C, monkeypatch):
    """Other widths (to 256 and to 512), head counts (slots of 32, 64 and 128 columns), depthwise kernel sizes (31: fused prologue; others:
    the separate kernel) and frontends under the zero-padded layout: against the fp32 oracle and against the model's own layout (COCR_NO_PAD=1)."""
    from tests.hip_util import oracle_taps
    hp = synth.hparams('cfg1', encoder_dim=D, num_attention_heads=heads, conv_kernel_size=ksz, subsampling_conv_channels=C, num_encoder_layers=2,
                       num_classes=40)
    state = synth.make_state_dict(hp, seed=D + heads, decoder_gain=4.0)
    image, lens = synth.make_lines(3, hp.height, 332, seed=D, widths=[332, 201, 97])
    ref, ref_lens, _ = oracle_taps(hp, state, image, lens)
    x = torch.from_numpy(image[:, 0]).cuda()
    a, ol = make_engine(hp, state, 'bf16').forward(x, lens)
    monkeypatch.setenv('COCR_NO_PAD', '1')
    b, _ = make_engine(hp, state, 'bf16').forward(x, lens)
    torch.cuda.synchronize()
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert ol.tolist() == ref_lens.tolist()
    da, db, d = float(np.abs(a - ref).max()), float(np.abs(b - ref).max()), float(np.abs(a - b).max())
    _log(f'pad_{D}_{heads}_{ksz}_{C}', {'padded_vs_oracle': da, 'own_layout_vs_oracle': db, 'padded_vs_own_layout': d})
    assert d > 0.0 and np.isfinite(a).all()
    assert da <= 0.15 and db <= 0.15 and d <= 0.15, (da, db, d)


def test_bucketed_loop_on_the_wide_model(text_case):
    """BASELINE configs[3] through the drop-in class: `evaluate.recognize` (fixed 200-px buckets, batches of 8, pipelined upload)
    on the D=512 / L=16 model, mixed widths 400..2400; strings equal to the reference's greedy strings of the same padded
    batches (fp32 and bf16), hence to the ground truth."""
    tc = text_case('cfg4_text')
    codec = ascii_codec(tc.hp.num_classes)
    want = [''.join(x[0] for x in codec.decode([(l, 0, 0, 0.0) for l in s])) for s in tc.ref_strings]
    for dtype in ('fp32', 'bf16'):
        net = PytorchRecognitionModel(**tc.hp.as_dict(), input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1,
                                      conv_dropout_p=0.1, codec=codec, compute_dtype=dtype)
        net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in tc.state.items()})
        net = net.to('cuda:0').eval()
        got = recognize(net, tc.lines, batch_size=tc.batch_size, edge=tc.edge)
        assert [got[i] for i in range(tc.n)] == want, dtype
    # the report of the reference's test loop (cli/test.py:194-224) against the ground truth: no error, no confusion
    from conformer_ocr_amd.evaluate import evaluate
    truth = [''.join(x[0] for x in codec.decode([(l, 0, 0, 0.0) for l in s])) for s in tc.texts]
    rep = evaluate(net, tc.lines, truth, report=True, batch_size=tc.batch_size, edge=tc.edge)
    assert rep['cer'] == 0.0 and rep['wer'] == 0.0 and rep['errors'] == 0 and rep['confusions'] == {} and rep['lines'] == tc.n
    assert f"{rep['chars']}\tCharacters" in rep['report']


@pytest.mark.parametrize('n,w', [(3, 40), (3, 68), (3, 132), (2, 300), (5, 640), (4, 648), (3, 1000), (17, 1200), (2, 1277), (32, 1200), (64, 300),
                                 (3, 1300), (2, 2400), (2, 2810), (1, 5000), (12, 2400)])
def test_lds_resident_attention_equals_the_tiled_kernel_bit_for_bit(n, w, monkeypatch):
    """Lines of at most 320 output frames, in batches of at least 192 workgroups (24 lines x 4 heads x 2), run `relpos_attention_full_kernel` (K, V and the positional band of a (line, head) resident in LDS,
    2 - 3 query tiles per wave, no barriers in the key loop); COCR_ATT_TILED=1 keeps the tiled kernel.  Same products, same shift, the same
    lazy-rescaling decisions per query tile: the logits of two blocks must be IDENTICAL, on ragged batches from 10 to 320 frames, and
    identical from run to run (the first version read matrix results in inline assembly too early: last-bit noise from run to run).
    Round 4: two waves per SIMD with at most two query tiles each, and lines of MORE than 320 frames in key passes (325 frames = 2 x 192
    keys, 600 = 2 x 320, 703 = 3 x 256, 1250 = 4 x 320: K / V / band restaged between two barriers, the softmax state carried over; behind
    COCR_ATT_RESIDENT_LONG=1 -- at those lengths the tiled kernel is the faster one and stays the default)."""
    hp = synth.hparams('cfg2', num_encoder_layers=2)
    state = synth.make_state_dict(hp, seed=3, decoder_gain=1.0, style='text')
    image, lens = synth.make_lines(n, hp.height, w, seed=5, widths=[max(33, w - 37 * i) for i in range(n)])
    x = torch.from_numpy(image[:, 0]).cuda()
    if n < 32:
        monkeypatch.setenv('COCR_ATT_RESIDENT_MIN', '1')       # (by itself the library picks the resident kernel from 192 workgroups on: the last two of the first row)
    monkeypatch.setenv('COCR_ATT_RESIDENT_LONG', '1')         # (and for lines of at most 320 frames: the second row's cases are an A/B switch)       # (by itself the library picks the resident kernel from 192 workgroups on: the last two cases)
    eng = make_engine(hp, state, 'bf16')
    runs = [eng.forward(x, lens)[0].cpu().numpy().copy() for _ in range(3)]
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])
    monkeypatch.setenv('COCR_ATT_TILED', '1')
    tiled = make_engine(hp, state, 'bf16').forward(x, lens)[0].cpu().numpy()
    np.testing.assert_array_equal(runs[0], tiled)


@pytest.mark.parametrize('layers,n,w', [(1, 9, 500), (2, 3, 232), (3, 2, 96), (2, 32, 1200)])
def test_skipped_zero_k_steps_of_the_padded_default_model_change_no_bit(layers, n, w, monkeypatch):
    """The reference's default model (encoder_dim 144, feed-forward 576) runs zero-padded to 256 / 768; its row-chain instantiation skips
    the k-steps that multiply zero columns (5 of 8 per K = D product, 2 of 8 in the FFN's last hidden chunk; NOT in the attention
    out-projection, whose operand has real columns in every head slot).  A skipped product is an exact zero: COCR_NO_KSKIP=1 (every k-step
    multiplied) must give the same bits.  Each comparison is the FIRST forward of a fresh engine behind a forward of other data: the first
    version's load-count waits (`s_waitcnt vmcnt(16)`: 16 ring loads assumed behind a DMA, 10 issued once the compiler dropped the dead ones)
    read LDS too early, which showed only where the LDS did not already hold the same rows from the launch before."""
    hp = synth.hparams('cfg1', num_encoder_layers=layers)
    state = synth.make_state_dict(hp, seed=7, decoder_gain=4.0)
    image, lens = synth.make_lines(n, hp.height, w, seed=11, widths=[max(40, w - 29 * i) for i in range(n)])
    x = torch.from_numpy(image[:, 0]).cuda()
    monkeypatch.setenv('COCR_NO_KSKIP', '1')
    ref = make_engine(hp, state, 'bf16').forward(x, lens)[0].cpu().numpy()
    monkeypatch.delenv('COCR_NO_KSKIP')
    assert np.isfinite(ref).all() and np.abs(ref).max() > 1.0
    for rep in range(3):
        other, olens = synth.make_lines(5 + rep, hp.height, 300 + 64 * rep, seed=50 + rep)
        scr = make_engine(hp, state, 'bf16')
        scr.forward(torch.from_numpy(other[:, 0]).cuda(), olens)                 # other rows through the same kernels' LDS
        torch.cuda.synchronize()
        got = make_engine(hp, state, 'bf16').forward(x, lens)[0].cpu().numpy()   # first forward of a fresh engine
        np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize('n,w,widths,height', [(3, 232, [232, 137, 200], 96), (2, 1200, [1200, 1111], 96), (5, 61, None, 96), (1, 2400, None, 96),
                                              (3, 300, [300, 77, 201], 48), (2, 200, None, 128)])
def test_fused_frontend_of_the_default_model_equals_the_separate_kernels(n, w, widths, height, monkeypatch):
    """32 conv channels (the reference's default model, default_specs.py:48-61): conv.0 + ReLU + depthwise conv.2 + pointwise conv.3 + ReLU
    run as ONE launch (conv.hip.h: frontend_conv12pw32_kernel; Z2 stays in LDS, the pointwise conv is one matrix instruction per 16
    positions x 16 channels).  A/B against the separate kernels (COCR_NO_FRONT32=1: VALU kernel -> Z2 in memory -> GEMM): the same
    rounding points (Z2 and Z3 are bf16 either way); the bias enters the accumulation at another place, which flips the bf16 rounding of
    about one Z3 element per line -- lines without a flip come out bit-identical, the others within the band every A/B of this file has
    (0.1 measured on logits of +-12; fused vs unfused 256-channel frontend: 0.25); u8 lines too; frame counts that are no multiple of 16."""
    hp = synth.hparams('cfg1', num_encoder_layers=2, height=height)          # (line heights 48 / 128: 12 / 32 feature rows per frame)
    state = synth.make_state_dict(hp, seed=9, decoder_gain=8.0)
    image, lens = synth.make_lines(n, hp.height, w, seed=13, widths=widths)
    for as_u8 in (False, True):
        x = torch.from_numpy((image[:, 0] * 255).round().astype(np.uint8) if as_u8 else image[:, 0]).cuda()
        monkeypatch.setenv('COCR_NO_FRONT32', '1')
        ref = make_engine(hp, state, 'bf16').forward(x, lens)[0].cpu().numpy()
        monkeypatch.delenv('COCR_NO_FRONT32')
        got = make_engine(hp, state, 'bf16').forward(x, lens)[0].cpu().numpy()
        assert np.isfinite(got).all() and np.abs(ref).max() > 1.0
        assert np.abs(got - ref).max() <= 0.25, float(np.abs(got - ref).max())
        assert (got.argmax(-1) != ref.argmax(-1)).mean() <= 0.02
