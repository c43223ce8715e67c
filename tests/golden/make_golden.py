#!/usr/bin/env python3
"""
Generates the golden fixtures in this directory by running THE REFERENCE ITSELF
(`/root/reference/conformer_ocr/conformer/encoder.py::ConformerEncoder`, torch
CPU fp32, eval mode) on seeded synthetic weights and line batches.

Run in the authoring container only (the reference does not travel to the GPU
box):   python tests/golden/make_golden.py

What is stored is data only: inputs are regenerated from seeds by
`conformer_ocr_amd.synth`, expected outputs (logits, per-stage activations,
lengths, state-dict key list) are stored as .npz.  `conformer_ocr.pred` /
`.model` are not importable (SyntaxError at pred.py:213-217; lightning/kraken
absent), so the 3 lines of `PytorchRecognitionModel.forward` (pred.py:119-121:
squeeze/transpose, encoder, `nn.Linear` decoder) are applied here around the
imported encoder.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

from conformer_ocr.conformer.encoder import ConformerEncoder  # noqa: E402  (the reference)

from conformer_ocr_amd import synth  # noqa: E402
from conformer_ocr_amd.spec import HParams  # noqa: E402


def build_reference(hp: HParams, state):
    enc = ConformerEncoder(in_channels=1, input_dim=hp.height, encoder_dim=hp.encoder_dim,
                           num_layers=hp.num_encoder_layers, num_attention_heads=hp.num_attention_heads,
                           feed_forward_expansion_factor=hp.feed_forward_expansion_factor,
                           conv_expansion_factor=hp.conv_expansion_factor,
                           conv_kernel_size=hp.conv_kernel_size, half_step_residual=hp.half_step_residual,
                           subsampling_conv_channels=hp.subsampling_conv_channels,
                           subsampling_factor=hp.subsampling_factor)
    dec = torch.nn.Linear(hp.encoder_dim, hp.num_classes, bias=True)                 # pred.py:90
    enc_sd = {k[len('encoder.'):]: torch.from_numpy(np.asarray(v)) for k, v in state.items() if k.startswith('encoder.')}
    missing = enc.load_state_dict(enc_sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    dec.load_state_dict({'weight': torch.from_numpy(state['decoder.weight']), 'bias': torch.from_numpy(state['decoder.bias'])})
    return enc.eval(), dec.eval()


def reference_forward(enc, dec, image, lens, taps=None):
    """pred.py:118-122 around the reference encoder; optional forward hooks collect stage outputs."""
    hooks = []
    if taps is not None:
        def tap(name, fn=lambda t: t):
            def hook(_m, _i, o):
                taps[name] = fn(o.detach().clone())
            return hook
        cs = enc.conv_subsample
        hooks.append(cs.conv[2].register_forward_hook(tap('front.z2', lambda t: t.permute(0, 2, 3, 1))))   # (B,C,T,F)->(B,T,F,C)
        hooks.append(cs.conv[4].register_forward_hook(tap('front.z3', lambda t: t.permute(0, 2, 3, 1))))
        hooks.append(cs.out.register_forward_hook(tap('front.y')))
        for l, layer in enumerate(enc.layers):
            for i, nm in enumerate(('ffn1', 'mhsa', 'conv', 'ffn2', 'out')):
                hooks.append(layer.sequential[i].register_forward_hook(tap(f'l{l}.{nm}')))
    with torch.no_grad():
        line = torch.from_numpy(image).squeeze(1).transpose(1, 2)                   # pred.py:119
        eo, el = enc(line, torch.from_numpy(lens))                                  # pred.py:120
        logits = dec(eo)                                                            # pred.py:121
    for h in hooks:
        h.remove()
    return logits.numpy(), el.numpy()


def calibrate_decoder_bias(enc, state, image, lens, blank_share=0.35):
    """With random weights every frame decodes to the same label (the frame-independent part of the
    encoder output dominates).  Centre the decoder on the reference's own mean encoder output and
    lift the blank so that it wins about `blank_share` of the frames: greedy strings then have
    runs, repeats and blanks.  The resulting bias vector is stored in the fixture."""
    with torch.no_grad():
        eo, _ = enc(torch.from_numpy(image).squeeze(1).transpose(1, 2), torch.from_numpy(lens))
    w = torch.from_numpy(state['decoder.weight'])
    mu = eo.reshape(-1, eo.shape[-1]).mean(0)
    bias = -(w @ mu)
    lg = eo @ w.t() + bias
    gap = (lg[..., 1:].max(-1).values - lg[..., 0]).flatten()
    bias[0] += torch.quantile(gap, blank_share)
    return bias.numpy().astype(np.float32)


def margins(logits):
    srt = np.sort(logits, axis=-1)
    return (srt[..., -1] - srt[..., -2]).astype(np.float32)


def run_case(meta, name, hp, seed, n, W, widths=None, with_taps=False, head=None, gain=8.0,
             line_seed=None, reuse=None, style='plain'):
    """One fixture: seeded weights (+ calibrated decoder bias) and lines -> reference outputs.
    style='text': the encoder draws re-scaled as in the text fixtures (synth.make_state_dict: a random-weight encoder that is local instead
    of 90 % frame-independent), the decoder still random (gain) with the calibrated bias -- logits with a usable top-2 margin on most frames."""
    line_seed = seed if line_seed is None else line_seed
    image, lens = synth.make_lines(n, hp.height, W, seed=line_seed, widths=widths)
    if reuse is None:
        state = synth.make_state_dict(hp, seed=seed, decoder_gain=gain, style=style)
        enc, _ = build_reference(hp, state)
        state['decoder.bias'] = calibrate_decoder_bias(enc, state, image, lens)
        enc, dec = build_reference(hp, state)
    else:
        state, enc, dec = reuse
    taps = {} if with_taps else None
    logits, olens = reference_forward(enc, dec, image, lens, taps)
    out = {'out_lens': olens, 'decoder_bias': state['decoder.bias'],
           'labels': np.argmax(logits, -1).astype(np.int16), 'margins': margins(logits).astype(np.float16)}
    if head is None:
        out['logits'] = logits
    else:
        out['logits_head'] = logits[:head]
    if with_taps:
        for k, v in taps.items():
            out['tap:' + k] = v.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    lab = out['labels']
    mg = margins(logits)
    meta[name] = {'hparams': hp.as_dict(), 'seed': seed, 'line_seed': line_seed, 'decoder_gain': gain, 'weight_style': style, 'N': n, 'W': W,
                  'margin_share_gt_0p7': float((mg > 0.7).mean()), 'margin_median': float(np.median(mg)),
                  'widths': [int(x) for x in lens], 'sha256_logits': hashlib.sha256(logits.tobytes()).hexdigest(),
                  'distinct_labels': int(len(np.unique(lab))), 'blank_share': float((lab == 0).mean()),
                  'margin_q01': float(np.quantile(margins(logits), 0.01))}
    print(name, {k: meta[name][k] for k in ('distinct_labels', 'blank_share', 'margin_q01')})
    return state, enc, dec


# ------------------------------------------------------------------------------------------------ "text" (peaked) fixtures
def frame_targets(T, texts, spans, edge=2.0):
    """Per-frame training target of a text line: output frame t sees pixels around 4t + 1.5 (two stride-2 stages); the glyph's
    label where that pixel lies inside the glyph (`edge` px from its ends), blank (0) everywhere else."""
    px = 4.0 * np.arange(T) + 1.5
    tgt = np.zeros((T,), dtype=np.int64)
    for a, (x0, x1) in zip(texts, spans):
        tgt[(px >= x0 + edge) & (px < x1 - edge)] = a
    return tgt


def fit_text_decoder(frames, targets, ncls, used, iters=2000, wd=1e-4, seed=0):
    """Multinomial logistic regression (Adam, full batch) of the per-frame targets on the REFERENCE's encoder output `frames`
    (M, D): the decoder a training run would have produced for this synthetic text, fitted on this very batch.  Classes >= `used`
    get zero weights and bias -30 (never win).  Returns decoder.weight (ncls, D), decoder.bias (ncls) as float32 numpy."""
    torch.manual_seed(seed)
    X = torch.from_numpy(frames)
    tg = torch.from_numpy(targets)
    mu = X.mean(0)
    Xc = X - mu
    W = torch.zeros(ncls, X.shape[1], requires_grad=True)
    b = torch.zeros(ncls, requires_grad=True)
    live = torch.zeros(ncls, dtype=torch.bool)
    live[:used] = True
    opt = torch.optim.Adam([W, b], lr=0.02)
    for _ in range(iters):
        opt.zero_grad()
        lg = (Xc @ W.t() + b).masked_fill(~live, -30.0)
        loss = torch.nn.functional.cross_entropy(lg, tg) + wd * (W ** 2).sum()
        loss.backward()
        opt.step()
    with torch.no_grad():
        Wf = W.detach().clone()
        Wf[~live] = 0.0
        bf = (b - Wf @ mu).detach().clone()
        bf[~live] = -30.0
    return Wf.numpy().astype(np.float32), bf.numpy().astype(np.float32)


def collapse(labels):
    """CTC greedy collapse of a frame-label array: merge repeats, drop blanks."""
    out, prev = [], -1
    for v in labels:
        v = int(v)
        if v != prev and v != 0:
            out.append(v)
        prev = v
    return out


def ragged(seqs, dtype=np.int16):
    return (np.asarray([x for s in seqs for x in s], dtype=dtype), np.asarray([len(s) for s in seqs], dtype=np.int32))


def run_text_case(meta, name, hp, seed, widths, batch_size, edge, alphabet=24, head=2):
    """A fixture with a ground truth: `len(widths)` text lines (conformer_ocr_amd.synth.make_text_lines, one seed per line),
    grouped into fixed-edge width buckets exactly like conformer_ocr_amd.evaluate.recognize does (edge 0: one batch, no
    bucketing), 'text'-style weights, the decoder fitted on the reference's own encoder output; stores the fitted decoder, the
    reference's per-frame labels / top-2 margins / greedy label strings per line and the ground-truth strings."""
    from conformer_ocr_amd.evaluate import collate, make_batches
    n = len(widths)
    lines, texts, spans = [], [], []
    for i, w in enumerate(widths):
        im, _, tx, sp = synth.make_text_lines(1, hp.height, int(w), seed=seed + 1000 + i, alphabet=alphabet, alphabet_seed=seed)
        lines.append(im[0, 0]); texts.append(tx[0]); spans.append(sp[0])
    batches = make_batches(list(widths), batch_size, edge) if edge else [(int(max(widths)), list(range(n)))]
    state = synth.make_state_dict(hp, seed=seed, decoder_gain=1.0, style='text')
    enc, _ = build_reference(hp, state)
    eos, olens = {}, {}
    with torch.no_grad():
        for bw, idx in batches:
            im, lens = collate(lines, idx, bw)
            eo, el = enc(im.squeeze(1).transpose(1, 2), lens)                     # pred.py:119-120
            for k, i in enumerate(idx):
                eos[i] = eo[k].numpy()
                olens[i] = int(el[k])
    # decoder fitted on the frames of every line inside its own length
    fr = np.concatenate([eos[i][:olens[i]] for i in range(n)])
    tg = np.concatenate([frame_targets(olens[i], texts[i], spans[i]) for i in range(n)])
    W, b = fit_text_decoder(fr, tg, hp.num_classes, alphabet + 1)
    state['decoder.weight'], state['decoder.bias'] = W, b
    dec = torch.nn.Linear(hp.encoder_dim, hp.num_classes)
    dec.load_state_dict({'weight': torch.from_numpy(W), 'bias': torch.from_numpy(b)})
    labels, margs, strings, heads = [], [], [], []
    with torch.no_grad():
        for i in range(n):
            lg = dec(torch.from_numpy(eos[i])).numpy()                            # pred.py:121 (all frames of the padded row)
            v = lg[:olens[i]]
            labels.append(np.argmax(v, -1)); margs.append(margins(v)); strings.append(collapse(labels[-1]))
            if i < head:
                heads.append(lg)
    lab_flat, lab_len = ragged(labels)
    txt_flat, txt_len = ragged(texts)
    str_flat, str_len = ragged(strings)
    out = {'decoder_weight': W, 'decoder_bias': b, 'out_lens': np.asarray([olens[i] for i in range(n)], dtype=np.int32),
           'labels': lab_flat, 'margins': np.concatenate(margs).astype(np.float16), 'texts': txt_flat, 'text_lens': txt_len,
           'ref_strings': str_flat, 'ref_string_lens': str_len}
    for i, h in enumerate(heads):
        out[f'logits_line{i}'] = h.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    allm = np.concatenate(margs)
    exact = sum(strings[i] == texts[i] for i in range(n))
    from conformer_ocr_amd.evaluate import edit_distance
    cer = sum(edit_distance(strings[i], texts[i]) for i in range(n)) / max(1, sum(len(t) for t in texts))
    meta[name] = {'hparams': hp.as_dict(), 'seed': seed, 'style': 'text', 'alphabet': alphabet, 'widths': [int(w) for w in widths],
                  'batch_size': batch_size, 'edge': edge, 'frames': int(allm.size), 'frac_margin_gt_1': float((allm > 1.0).mean()),
                  'margin_q01': float(np.quantile(allm, 0.01)), 'reference_lines_equal_to_truth': int(exact),
                  'reference_cer_vs_truth': float(cer), 'blank_share': float((lab_flat == 0).mean())}
    print(name, {k: meta[name][k] for k in ('frames', 'frac_margin_gt_1', 'margin_q01', 'reference_lines_equal_to_truth', 'reference_cer_vs_truth')})
    return state


def text_case_cfg1(meta):
    # cfg1_text: the reference's default model (default_specs.py: D=144, 16 blocks, 4 heads, C=32) on the metric's batch shape
    run_text_case(meta, 'cfg1_text', synth.hparams('cfg1'), 2011, [1200] * 32, 32, 0)


def text_cases(meta):
    text_case_cfg1(meta)
    # cfg2_text: the metric's configuration and batch (32 lines of 96x1200), text lines + fitted decoder
    run_text_case(meta, 'cfg2_text', synth.hparams('cfg2'), 2001, [1200] * 32, 32, 0)
    # cfg4_text: BASELINE configs[3] -- wide conformer, widths U{400..2400} step 8, bucket edge 200, batches of <= 8
    g = np.random.Generator(np.random.PCG64(4004))
    widths = (400 + 8 * g.integers(0, 251, 24)).tolist()
    run_text_case(meta, 'cfg4_text', synth.hparams('cfg4'), 2004, widths, 8, 200)


def tiny2_case(meta):
    """tiny2: subsampling_factor 2 (conv.0 + ReLU, then the output linear: convolution.py:182-215 with one sampling stage)."""
    hp2 = synth.hparams('tiny', subsampling_factor=2)
    _, enc2, _ = run_case(meta, 'tiny2', hp2, 1239, 3, 48, widths=[48, 31, 40])
    meta['tiny2']['encoder_state_keys'] = [(k, list(v.shape), str(v.dtype)) for k, v in enc2.state_dict().items()]


def long_case(meta):
    """tiny_long: one line of 5150 frames -- longer than RelPositionalEncoding(max_len=5000): the reference rebuilds its table
    (embedding.py:35-41).  Stored: the logits of the first and last 256 frames, the labels of all."""
    hp = synth.hparams('tiny')
    image, lens = synth.make_lines(1, hp.height, 20600, seed=1240)
    state = synth.make_state_dict(hp, seed=1240, decoder_gain=8.0)
    enc, dec = build_reference(hp, state)
    logits, olens = reference_forward(enc, dec, image, lens)
    np.savez_compressed(os.path.join(HERE, 'tiny_long.npz'), out_lens=olens, decoder_bias=state['decoder.bias'],
                        labels=np.argmax(logits, -1).astype(np.int16), margins=margins(logits).astype(np.float16),
                        logits_first=logits[:, :256], logits_last=logits[:, -256:])
    meta['tiny_long'] = {'hparams': hp.as_dict(), 'seed': 1240, 'line_seed': 1240, 'decoder_gain': 8.0, 'N': 1, 'W': 20600,
                         'widths': [int(x) for x in lens], 'sha256_logits': hashlib.sha256(logits.tobytes()).hexdigest(), 'frames': int(olens[0])}
    print('tiny_long', meta['tiny_long']['frames'], 'frames')


def cfg2_cases(meta):
    """cfg2: the metric's configuration (D=256, L=12, h=4, C=256), batch 32 x 96x1200: full logits of lines 0..3, argmax labels + top-2
    margins of all 32 lines.  Round 4: 'text'-style encoder draws under the random decoder -- with the plain draws 9 - 12 %
    of the frames had a top-2 margin above the bf16 band (the label comparison of the bf16 mode was close to vacuous and 7 % of all frame
    labels flipped); the ratio of bf16 noise to margin does not depend on the decoder gain, only on how frame-dependent the encoder output is."""
    hp2 = synth.hparams('cfg2')
    reuse = run_case(meta, 'cfg2', hp2, 1236, 32, 1200, head=4, style='text')
    # cfg2 ragged: 6 lines of mixed widths right-padded to 1200 (padding-leak case at full size), same weights
    run_case(meta, 'cfg2_ragged', hp2, 1236, 6, 1200, widths=[1200, 1111, 903, 640, 417, 1200], head=2,
             line_seed=1237, reuse=reuse, style='text')


# Round 4: hyper-parameter VARIANTS of the measured model, each through the reference itself -- other instantiations of the same kernels
# (depthwise kernel sizes, head sizes, the eighth-rate frontend, the full-step residual, an odd class count, the wide model with a short
# depthwise kernel, the reference's default model at the metric's line width).  Small batches: the fixtures hold full logits.
VARIANTS = {
    #  name: (base config, overrides, seed, lines, width, widths)
    'v_k15':    ('cfg2', dict(num_encoder_layers=2, conv_kernel_size=15), 3101, 3, 400, [400, 333, 250]),
    'v_k7':     ('cfg2', dict(num_encoder_layers=2, conv_kernel_size=7), 3102, 2, 264, [264, 199]),
    'v_h8':     ('cfg2', dict(num_encoder_layers=2, num_attention_heads=8), 3103, 3, 400, [400, 287, 350]),
    'v_nohalf': ('cfg2', dict(num_encoder_layers=2, half_step_residual=False, num_classes=97), 3104, 2, 328, [328, 240]),
    'v_f8':     ('cfg2', dict(num_encoder_layers=2, subsampling_factor=8), 3105, 2, 480, [480, 391]),
    'v_ff2':    ('cfg2', dict(num_encoder_layers=2, feed_forward_expansion_factor=2), 3106, 2, 296, [296, 180]),
    'v_d512k7': ('cfg4', dict(num_encoder_layers=1, conv_kernel_size=7), 3107, 2, 360, [360, 299]),
    'v_cfg1w':  ('cfg1', dict(), 3108, 2, 1200, [1200, 1040]),
}


def variant_cases(meta):
    for name, (base, over, seed, n, W, widths) in VARIANTS.items():
        hp = synth.hparams(base, **over)
        run_case(meta, name, hp, seed, n, W, widths=widths, style='text')


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == 'text':       # only the text fixtures, merged into the existing meta.json
        with open(os.path.join(HERE, 'meta.json')) as fp:
            meta = json.load(fp)
        text_cases(meta)
        with open(os.path.join(HERE, 'meta.json'), 'w') as fp:
            json.dump(meta, fp, indent=1)
        return
    if len(sys.argv) > 1 and sys.argv[1] in ('tiny2', 'long', 'text1', 'cfg2', 'variants'):      # only that fixture, merged into the existing meta.json
        with open(os.path.join(HERE, 'meta.json')) as fp:
            meta = json.load(fp)
        {'tiny2': tiny2_case, 'long': long_case, 'text1': text_case_cfg1, 'cfg2': cfg2_cases, 'variants': variant_cases}[sys.argv[1]](meta)
        with open(os.path.join(HERE, 'meta.json'), 'w') as fp:
            json.dump(meta, fp, indent=1)
        return
    meta = {}
    # tiny: full per-stage taps, padding leak exercised (widths 64/37/50 padded to 64)
    hp = synth.hparams('tiny')
    _, enc, _ = run_case(meta, 'tiny', hp, 1234, 3, 64, widths=[64, 37, 50], with_taps=True)
    meta['tiny']['encoder_state_keys'] = [(k, list(v.shape), str(v.dtype)) for k, v in enc.state_dict().items()]
    # tiny8: subsampling_factor 8 (extra depthwise/pointwise stage)
    hp8 = synth.hparams('tiny', subsampling_factor=8, height=32)
    _, enc8, _ = run_case(meta, 'tiny8', hp8, 1235, 2, 96, widths=[96, 61])
    meta['tiny8']['encoder_state_keys'] = [(k, list(v.shape), str(v.dtype)) for k, v in enc8.state_dict().items()]
    tiny2_case(meta)
    long_case(meta)
    # cfg1: default_specs.py verbatim, BASELINE configs[0]: 4 lines 96x512 (lens 512,400,300,512)
    hp1 = synth.hparams('cfg1')
    _, enc1, _ = run_case(meta, 'cfg1', hp1, 1235, 4, 512, widths=[512, 400, 300, 512])
    meta['cfg1']['n_params_encoder'] = int(sum(p.numel() for p in enc1.parameters()))
    cfg2_cases(meta)
    # cfg4 (wide conformer D=512, L=16, h=8): 3 lines of bucketed widths padded to 1400
    hp4 = synth.hparams('cfg4')
    run_case(meta, 'cfg4', hp4, 1238, 3, 1400, widths=[1400, 1256, 1208], head=1)
    text_cases(meta)
    variant_cases(meta)
    with open(os.path.join(HERE, 'meta.json'), 'w') as fp:
        json.dump(meta, fp, indent=1)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == '__main__':
    main()
