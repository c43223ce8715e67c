"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 1c): CPU restatement of the line pre-processing step in
front of the recognition path -- SURVEY.md section 8(f).3: kraken `ImageInputTransforms(1, 96, 0, 1, (16, 0), valid_norm=False)`
(reference call sites `dataset.py:89`, `cli/test.py:156`): grayscale, scale to height 96 keeping the aspect ratio, pad 16 px left
and right, to [0, 1].

kraken is a third-party dependency that is not in the tree, so the CHAIN is restated from the survey's description (**parity
unpinned** against kraken itself: output width = int(w * 96 / h), zero padding, no inversion are this build's reading).  The
expensive and exactly-specified part, the resize, is Pillow's 8-bit LANCZOS resampler (what kraken's `_fixed_resize` calls),
restated integer for integer from Pillow's `Resample.c` and PINNED bit-exactly against Pillow itself (tests/test_preproc_oracle.py;
Pillow is importable in this image, here and on the GPU box):
  * per output pixel: centre = (xx + 0.5) * scale, support = 3 * max(scale, 1), taps [int(centre - support + 0.5),
    int(centre + support + 0.5)) clipped to the image, weight = sinc(x) sinc(x / 3) of (tap + 0.5 - centre) / max(scale, 1),
    normalised by their sequential double sum, converted to 22-bit fixed point with round-half-away;
  * value = clip8((2^21 + sum pixel * coefficient) >> 22); horizontal pass first (intermediate image in u8), then vertical;
  * RGB -> L as Pillow: (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16."""
import math
from typing import List, Sequence, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        if x == 0.0:
            return 1.0
        a = x * math.pi
        b = x / 3.0 * math.pi
        return (math.sin(a) / a) * (math.sin(b) / b)
    return 0.0


def resample_coeffs(in_size: int, out_size: int) -> Tuple[List[Tuple[int, int]], List[List[int]]]:
    """Pillow precompute_coeffs + normalize_coeffs_8bpc: per output index (first tap, tap count) and the fixed-point taps."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 3.0 * fscale
    ss = 1.0 / fscale
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        kk.append([int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w])
        bounds.append((xmin, xmax))
    return bounds, kk


def resize_lanczos_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL.Image.resize((out_w, out_h), LANCZOS) of an 8-bit single-channel image (H, W)."""
    h, w = img.shape
    b, k = resample_coeffs(w, out_w)
    tmp = np.zeros((h, out_w), np.uint8)
    for xx in range(out_w):
        x0, n = b[xx]
        acc = (1 << (PRECISION_BITS - 1)) + (img[:, x0:x0 + n].astype(np.int64) * np.asarray(k[xx], np.int64)).sum(1)
        tmp[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    b, k = resample_coeffs(h, out_h)
    out = np.zeros((out_h, out_w), np.uint8)
    for yy in range(out_h):
        y0, n = b[yy]
        acc = (1 << (PRECISION_BITS - 1)) + (tmp[y0:y0 + n].astype(np.int64) * np.asarray(k[yy], np.int64)[:, None]).sum(0)
        out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def rgb_to_l(img: np.ndarray) -> np.ndarray:
    """(H, W, 3) u8 -> (H, W) u8, Pillow's ITU-R 601-2 luma in 16-bit fixed point."""
    v = img.astype(np.int64)
    return ((v[..., 0] * 19595 + v[..., 1] * 38470 + v[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def scaled_width(h: int, w: int, out_h: int) -> int:
    return max(1, int(w * out_h / h))


def preprocess_line(img: np.ndarray, out_h: int = 96, pad: int = 16) -> np.ndarray:
    """One line image (H, W) or (H, W, 3), u8 -> (out_h, W' + 2 pad) u8 (divide by 255 for the path's [0, 1] floats)."""
    if img.ndim == 3:
        img = rgb_to_l(img)
    h, w = img.shape
    r = resize_lanczos_u8(img, scaled_width(h, w, out_h), out_h)
    return np.pad(r, ((0, 0), (pad, pad)))


def collate(lines: Sequence[np.ndarray], out_h: int = 96, pad: int = 16, width: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Pre-processed lines right-zero-padded into (N, out_h, width) u8 (width 0: the widest line) and their widths
    (padding included: the `seq_lens` of the batch contract)."""
    outs = [preprocess_line(x, out_h, pad) for x in lines]
    lens = np.array([o.shape[1] for o in outs], dtype=np.int32)
    wmax = int(width) if width else int(lens.max())
    if lens.max() > wmax:
        raise ValueError(f'a pre-processed line is {int(lens.max())} px wide, the batch only {wmax}')
    batch = np.zeros((len(outs), out_h, wmax), np.uint8)
    for i, o in enumerate(outs):
        batch[i, :, :o.shape[1]] = o
    return batch, lens
