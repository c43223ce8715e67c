"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the CTC decoders the
reference calls on each line's logits (reference pred.py:143,162,177 and
model.py:167: `self.ctc_decoder(seq[:, :seq_len])` with
`kraken.lib.ctc_decoder.greedy_decoder`).

PARITY UNPINNED: the algorithm lives in the third-party package `kraken`
(setup.cfg:36 `kraken>=4.3.13`, metadata hints 5.3 at pred.py:236) which is not
in /root/reference and not installed; the reference holds no test or golden
vector for this boundary.  The functions below restate kraken's published
algorithm and are pinned only by the hand-derived
known-answer vectors in `tests/test_ctc_oracle.py`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def greedy_decoder(outputs: np.ndarray) -> List[Tuple[int, int, int, float]]:
    """kraken.lib.ctc_decoder.greedy_decoder on a (C, T) matrix:
    labels = argmax over classes per frame (first index on ties, numpy semantics);
    runs of equal labels are merged; runs of label 0 (blank) are dropped; each
    remaining run yields (label, first frame, last frame, max over the run of
    outputs[label, t]).  The reference feeds raw logits (no softmax: pred.py:121),
    so the 4th field is a logit."""
    outputs = np.asarray(outputs)
    C, T = outputs.shape
    if T == 0:
        return []
    labels = np.argmax(outputs, axis=0)
    vals = outputs[labels, np.arange(T)]
    res: List[Tuple[int, int, int, float]] = []
    start = 0
    for t in range(1, T + 1):
        if t == T or labels[t] != labels[start]:
            lab = int(labels[start])
            if lab != 0:
                res.append((lab, start, t - 1, float(vals[start:t].max())))
            start = t
    return res


def log_softmax(outputs: np.ndarray) -> np.ndarray:
    """Column-wise (over classes) log-softmax of a (C, T) logit matrix, float32 arithmetic."""
    x = np.asarray(outputs, dtype=np.float32)
    m = x.max(axis=0, keepdims=True)
    return (x - m) - np.log(np.exp(x - m).sum(axis=0, keepdims=True, dtype=np.float32))


def _lse(a: np.float32, b: np.float32) -> np.float32:
    """logaddexp in float32 with the fixed form max + log1p(exp(-|a-b|)) (the GPU kernel uses the same form)."""
    if a == -np.inf:
        return np.float32(b)
    if b == -np.inf:
        return np.float32(a)
    m = a if a > b else b
    d = np.float32(-abs(np.float32(a - b)))
    return np.float32(m + np.log1p(np.exp(d, dtype=np.float32), dtype=np.float32))


def beam_decoder(outputs: np.ndarray, beam_size: int = 16) -> List[Tuple[int, int, int, float]]:
    """CTC prefix beam search (the algorithm of kraken.lib.ctc_decoder.beam_decoder) with
    the semantics this build fixes, because the reference never calls it and kraken's version expects
    probabilities while the reference hands over raw logits:

    * input = raw logits (C, T); frame scores lp = log-softmax over classes (float32); blank = 0;
    * a prefix (label tuple) carries (p_b, p_nb) = log prob of its alignments ending / not ending in
      blank; per frame and parent (beam order): the prefix itself gets p_b (+)= total + lp[0] and, if
      non-empty, p_nb (+)= p_nb + lp[last]; every extension by s >= 1 gets
      p_nb (+)= (p_b if s == last else total) + lp[s]; (+) is logaddexp in the fixed form `_lse`;
    * after each frame the `beam_size` prefixes with the largest logaddexp(p_b, p_nb) survive; ties
      keep creation order (parents in beam order; per parent: itself, then labels ascending);
    * output for the best prefix: (label, start, end, conf); start = frame at which the label was
      appended on the chain of first creators; end = start extended while the next frame (before the
      next label's start) scores the label above blank; conf = max softmax(label) over [start, end]."""
    lp = log_softmax(outputs)
    C, T = lp.shape
    NEG = np.float32(-np.inf)
    beam = [((), np.float32(0.0), NEG, ())]               # (labels, p_b, p_nb, start frames)
    for t in range(T):
        cand = {}                                          # insertion-ordered: creation order
        for key, p_b, p_nb, starts in beam:
            tot = _lse(p_b, p_nb)
            last = key[-1] if key else None
            c = cand.setdefault(key, [NEG, NEG, starts])
            c[0] = _lse(c[0], np.float32(tot + lp[0, t]))
            if last is not None:
                c[1] = _lse(c[1], np.float32(p_nb + lp[last, t]))
            for s in range(1, C):
                add = np.float32((p_b if s == last else tot) + lp[s, t])
                if add == NEG:
                    continue
                c = cand.setdefault(key + (s,), [NEG, NEG, starts + (t,)])
                c[1] = _lse(c[1], add)
        ranked = sorted(enumerate(cand.items()), key=lambda x: (-float(_lse(x[1][1][0], x[1][1][1])), x[0]))
        beam = [(k, v[0], v[1], v[2]) for _, (k, v) in ranked[:beam_size]]
    labels, _, _, starts = beam[0]
    res = []
    for i, (c, s) in enumerate(zip(labels, starts)):
        limit = starts[i + 1] if i + 1 < len(starts) else T
        e = s
        while e + 1 < limit and lp[c, e + 1] > lp[0, e + 1]:
            e += 1
        res.append((int(c), int(s), int(e), float(np.exp(lp[c, s:e + 1].max()))))
    return res
