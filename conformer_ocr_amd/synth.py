"""
Deterministic synthetic weights and line batches.

Every tensor is drawn from its own ``numpy.random.Generator(PCG64)`` whose seed
is derived from (global seed, tensor name), so the GPU box regenerates exactly
the weights the golden fixtures were made with, without any file of the
reference travelling.  Distributions: Linear / conv weights U(-a, a) with the
fan-in bound sqrt(3/fan_in), biases N(0, 0.02) (non-zero so that the padded image region
leaks into the activations like it does in the reference), LayerNorm /
BatchNorm gamma 1 + N(0, 0.05), beta N(0, 0.05), BN running_mean N(0, 0.1),
running_var U(0.5, 1.5), u_bias / v_bias U(-a, a) likewise.
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np

from .spec import HParams, model_state_spec

# named configurations of BASELINE.json (`configs[i]`) plus the tiny fixture config
CONFIGS: Dict[str, Dict] = {
    # tiny fixture config: awkward sizes on purpose
    'tiny': dict(num_classes=11, height=16, encoder_dim=32, num_encoder_layers=2, num_attention_heads=4,
                 conv_kernel_size=7, subsampling_conv_channels=8, subsampling_factor=4),
    # configs[0]: default_specs.py verbatim (reference default_specs.py:48-61)
    'cfg1': dict(num_classes=128, height=96, encoder_dim=144, num_encoder_layers=16, num_attention_heads=4,
                 conv_kernel_size=31, subsampling_conv_channels=32, subsampling_factor=4),
    # configs[1..2,4]: the "default conformer" the metric is quoted on
    'cfg2': dict(num_classes=128, height=96, encoder_dim=256, num_encoder_layers=12, num_attention_heads=4,
                 conv_kernel_size=31, subsampling_conv_channels=256, subsampling_factor=4),
    # configs[3]: wide conformer
    'cfg4': dict(num_classes=128, height=96, encoder_dim=512, num_encoder_layers=16, num_attention_heads=8,
                 conv_kernel_size=31, subsampling_conv_channels=256, subsampling_factor=4),
}


def hparams(name: str, **override) -> HParams:
    kw = dict(CONFIGS[name])
    kw.update(override)
    return HParams(**kw)


def _rng(seed: int, name: str) -> np.random.Generator:
    h = hashlib.sha256(f'{seed}:{name}'.encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(h[:8], 'little')))


def _bound(shape: Tuple[int, ...]) -> float:
    """Variance-preserving uniform bound sqrt(3 / fan_in): keeps the input-dependent part of the
    signal alive through 12-16 random blocks (a xavier bound on the (C,1,3,3) frontend taps and the
    depthwise taps attenuates it until every frame decodes to the same label)."""
    if len(shape) < 2:
        return 0.1
    recept = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    return math.sqrt(3.0 / (shape[1] * recept))


TEXT_OUT_PROJ_GAIN = 0.3       # style='text': attention out_proj scale
TEXT_DW_SIGMA = 1.5            # style='text': Gaussian window (frames) on the conv module's depthwise taps


def make_state_dict(hp: HParams, seed: int = 1234, decoder_gain: float = 1.0, style: str = 'plain') -> 'OrderedDict[str, np.ndarray]':
    """Synthetic `nn.state_dict()` (keys `encoder.*`, `decoder.*`) as float32 numpy arrays.

    `decoder_gain` scales decoder.weight so that random-weight logits get usable
    top-2 margins.

    style='text' (the "peaked" fixtures): the same draws, then two re-scalings that make a random-weight encoder behave like
    a trained recogniser -- LOCAL: a random attention layer is a near-uniform average over the whole line, and 12-16 of them
    leave an encoder output that is 90 % frame-independent (top-2 margins of any decoder are then a few bf16 roundings wide).
    `out_proj.weight` x 0.3 keeps the attention path alive at a third of its weight, and a Gaussian window (sigma 1.5 frames,
    energy preserved) on the depthwise taps makes the conv module local.  On lines from `make_text_lines` the encoder output
    then clusters by glyph (same-glyph distance 3.5 vs 7.6 between glyph means, measured on the reference itself) and a linear
    decoder fitted on it (tests/golden/make_golden.py) reads the synthetic text back."""
    out: 'OrderedDict[str, np.ndarray]' = OrderedDict()
    for name, (shape, kind) in model_state_spec(hp).items():
        g = _rng(seed, name)
        leaf = name.rsplit('.', 1)[-1]
        if kind == 'counter':
            out[name] = np.zeros((), dtype=np.int64)
        elif leaf == 'running_mean':
            out[name] = g.normal(0.0, 0.1, shape).astype(np.float32)
        elif leaf == 'running_var':
            out[name] = g.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf in ('u_bias', 'v_bias'):
            a = _bound(shape)
            out[name] = g.uniform(-a, a, shape).astype(np.float32)
        elif len(shape) == 1 and leaf == 'weight':          # LayerNorm / BatchNorm gamma
            out[name] = (1.0 + g.normal(0.0, 0.05, shape)).astype(np.float32)
        elif len(shape) == 1 and leaf == 'bias':
            is_norm = name.replace('.bias', '.weight') in out and out[name.replace('.bias', '.weight')].ndim == 1
            out[name] = g.normal(0.0, 0.05 if is_norm else 0.02, shape).astype(np.float32)
        else:
            a = _bound(shape)
            w = g.uniform(-a, a, shape).astype(np.float32)
            if name == 'decoder.weight':
                w = (w * np.float32(decoder_gain)).astype(np.float32)
            out[name] = w
    if style == 'text':
        for name in out:
            if name.endswith('attention.out_proj.linear.weight'):
                out[name] = (out[name] * np.float32(TEXT_OUT_PROJ_GAIN)).astype(np.float32)
            elif name.endswith('module.sequential.4.conv.weight'):                  # conv module depthwise taps (D, 1, k)
                k = out[name].shape[-1]
                win = np.exp(-0.5 * ((np.arange(k, dtype=np.float64) - k // 2) / TEXT_DW_SIGMA) ** 2)
                win = (win * np.sqrt(k / (win ** 2).sum())).astype(np.float32)
                out[name] = (out[name] * win).astype(np.float32)
    elif style != 'plain':
        raise ValueError(f'unknown weight style {style!r}')
    return out


def text_alphabet(seed: int, alphabet: int = 24):
    """`alphabet` distinct glyphs: (6 x 4 on/off pattern, width in pixels 24..40), drawn once per seed."""
    g = _rng(seed, f'alphabet:{alphabet}')
    pats = []
    while len(pats) < alphabet:
        p = g.uniform(0.0, 1.0, (6, 4)) > 0.5
        if p.sum() < 6 or any((p == q[0]).all() for q in pats):
            continue
        pats.append((p, int(g.integers(24, 41))))
    return [(p.astype(np.float32), w) for p, w in pats]


def make_text_lines(n: int, height: int, width: int, seed: int = 1234, alphabet: int = 24,
                    widths: Optional[List[int]] = None, alphabet_seed: Optional[int] = None):
    """Synthetic TEXT lines with a ground truth: each line is a random string over `alphabet` glyphs (labels 1..alphabet, fixed
    pattern and width per glyph, intensity 0.75-1.0), 12-20 blank pixels between glyphs, pixel noise sigma 0.03, u8-quantised
    like `make_lines`.  Returns (image (N,1,H,W) float32, seq_lens (N,), texts: label list per line, spans: (x0, x1) pixel
    span of every glyph).  `widths`: per-line widths (glyphs stop 8 px before a line's own end; the rest is zero padding); `alphabet_seed`: the seed the
    glyph shapes are drawn from (default: `seed`) -- lines made by separate calls share their alphabet through it."""
    pats = text_alphabet(seed if alphabet_seed is None else alphabet_seed, alphabet)
    g = _rng(seed, f'text:{n}:{height}:{width}:{alphabet}')
    img = np.zeros((n, height, width), dtype=np.float32)
    lens = np.full((n,), width, dtype=np.int64) if widths is None else np.asarray(widths, dtype=np.int64)
    assert lens.shape == (n,) and lens.max() <= width
    cell_h = -(-height // 6)
    texts, spans = [], []
    for i in range(n):
        x = int(g.integers(8, 20))
        txt, sp = [], []
        while True:
            a = int(g.integers(0, alphabet))
            p, w = pats[a]
            if x + w + 8 > lens[i]:
                break
            img[i, :, x:x + w] = np.kron(p, np.ones((cell_h, -(-w // 4)), dtype=np.float32))[:height, :w] * np.float32(g.uniform(0.75, 1.0))
            txt.append(a + 1)
            sp.append((x, x + w))
            x += w + int(g.integers(12, 21))
        texts.append(txt)
        spans.append(sp)
    img = np.clip(img + g.normal(0.0, 0.03, img.shape).astype(np.float32), 0.0, 1.0)
    u8 = np.rint(img * 255.0).astype(np.uint8)[:, None, :, :]
    for i, w in enumerate(lens):
        u8[i, :, :, int(w):] = 0
    return (u8.astype(np.float32) / np.float32(255.0)), lens, texts, spans


def make_lines(n: int, height: int, width: int, seed: int = 1234,
               widths: Optional[List[int]] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Synthetic line batch in the layout the reference's collate hands over
    (cli/test.py:186-189): image (N,1,H,W) float32 in [0,1], right-zero-padded to
    the batch width; seq_lens (N,) pixel widths.  Each line is a run of block
    "glyphs" (random 6x3 on/off patterns, 10-39 px wide) plus pixel noise, so that
    neighbouring frames differ; pixels are u8-quantised (round(255 x)/255) so that
    u8 and f32 ingest see the same values."""
    g = _rng(seed, f'lines:{n}:{height}:{width}')
    img = np.zeros((n, height, width), dtype=np.float32)
    cell_h = max(1, -(-height // 6))
    for i in range(n):
        x = 0
        while x < width:
            w = int(g.integers(10, 40))
            pat = (g.uniform(0.0, 1.0, (6, 3)) > 0.5).astype(np.float32) * np.float32(g.uniform(0.5, 1.0))
            up = np.kron(pat, np.ones((cell_h, -(-w // 3)), dtype=np.float32))[:height, :w]
            ww = min(w, width - x)
            img[i, :, x:x + ww] = up[:, :ww]
            x += w
    img = np.clip(img + g.normal(0.0, 0.05, img.shape).astype(np.float32), 0.0, 1.0)
    u8 = np.rint(img * 255.0).astype(np.uint8)[:, None, :, :]
    lens = np.full((n,), width, dtype=np.int64) if widths is None else np.asarray(widths, dtype=np.int64)
    assert lens.shape == (n,) and lens.max() <= width
    for i, w in enumerate(lens):
        u8[i, :, :, int(w):] = 0
    return (u8.astype(np.float32) / np.float32(255.0)), lens


def lines_u8(image: np.ndarray) -> np.ndarray:
    """The exact u8 image a `make_lines` batch was quantised from."""
    return np.rint(image * 255.0).astype(np.uint8)
