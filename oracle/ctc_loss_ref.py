"""CPU restatement of the loss of the reference's `RecognitionModel._step` -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench
cpu_baseline may import this; the product path never does).

reference conformer_ocr/model.py:119      criterion = nn.CTCLoss(reduction='sum', zero_infinity=True)      (blank = 0)
reference conformer_ocr/model.py:136-142  logits = log_softmax(probits, -1); loss = criterion(logits.transpose(0, 1), target, encoder_lens, label_lens)

i.e. per line n: nll_n = -log p(target_n | probits_n[:len_n]) by the CTC forward recursion over the blank-extended label sequence,
an infinite nll (no valid alignment: len_n < |target_n| + repeats) replaced by 0 with a zero gradient, and the batch loss is the plain
sum of the nll_n (no division by target length or batch size).  The gradient with respect to the PRE-softmax `probits` is, for frames
t < len_n,   softmax(probits)[t, c] - sum_{s : l'_s = c} alpha_t(s) beta_t(s) / (y_t(c) p(target))   and 0 for t >= len_n.

The algorithm lives in torch (aten LossCTC.cpp), the reference's own dependency, importable here: the restatement is pinned against
`torch.nn.functional.ctc_loss` + autograd on CPU through tests/golden/ctc_loss.npz (made by tests/golden/make_ctc_loss_golden.py, which
calls exactly the two reference lines above).  float64 throughout."""
from __future__ import annotations

import numpy as np


def _lse(*xs):
    m = max(xs)
    if m == -np.inf:
        return -np.inf
    return m + np.log(sum(np.exp(x - m) for x in xs))


def log_softmax(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    m = x.max(axis=-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))


def ctc_line(probits: np.ndarray, target, blank: int = 0):
    """(nll, d nll / d probits) of one line: probits (len, C) -- only the valid frames --, target a sequence of class indices."""
    lp = log_softmax(probits)
    T, C = lp.shape
    L = len(target)
    ext = [blank] * (2 * L + 1)
    ext[1::2] = list(target)
    S = len(ext)
    grad = np.zeros((T, C))
    if T == 0:
        return (0.0 if L == 0 else np.inf), grad
    ext = np.asarray(ext)
    skip = np.zeros(S, dtype=bool)                                    # state s also reachable from s-2
    skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
    ninf = np.full(2, -np.inf)
    alpha = np.full((T, S), -np.inf)
    alpha[0, 0] = lp[0, blank]
    if S > 1:
        alpha[0, 1] = lp[0, ext[1]]
    for t in range(1, T):                                             # vectorised over the states
        p = np.concatenate([ninf, alpha[t - 1]])
        acc = np.logaddexp(p[2:], p[1:-1])
        acc = np.where(skip, np.logaddexp(acc, p[:-2]), acc)
        alpha[t] = acc + lp[t, ext]
    ll = _lse(alpha[T - 1, S - 1], alpha[T - 1, S - 2] if S > 1 else -np.inf)
    if ll == -np.inf:
        return np.inf, grad
    skip_up = np.zeros(S, dtype=bool)                                 # state s also reachable (backwards) from s+2
    skip_up[:-2] = skip[2:]
    beta = np.full((T, S), -np.inf)
    beta[T - 1, S - 1] = lp[T - 1, blank]
    if S > 1:
        beta[T - 1, S - 2] = lp[T - 1, ext[S - 2]]
    for t in range(T - 2, -1, -1):
        p = np.concatenate([beta[t + 1], ninf])
        acc = np.logaddexp(p[:-2], p[1:-1])
        acc = np.where(skip_up, np.logaddexp(acc, p[2:]), acc)
        beta[t] = acc + lp[t, ext]
    occ = np.zeros((T, C))
    for s in range(S):
        occ[:, ext[s]] += np.exp(alpha[:, s] + beta[:, s] - lp[:, ext[s]] - ll)
    grad = np.exp(lp) - occ
    return -ll, grad


def ctc_loss(probits: np.ndarray, targets, out_lens, label_lens, zero_infinity: bool = True):
    """probits (N, T, C); targets: the concatenated 1-D label vector of the reference's batches (dataset collate), label_lens its
    per-line lengths.  Returns (per-line nll (N), gradient (N, T, C)); the reference's loss is nll.sum()."""
    probits = np.asarray(probits, dtype=np.float64)
    N, T, C = probits.shape
    nll = np.zeros(N)
    grad = np.zeros((N, T, C))
    off = 0
    for n in range(N):
        tgt = [int(v) for v in targets[off:off + int(label_lens[n])]]
        off += int(label_lens[n])
        ln = int(out_lens[n])
        v, g = ctc_line(probits[n, :ln], tgt)
        if np.isinf(v):
            if not zero_infinity:
                nll[n] = np.inf
            continue
        nll[n] = v
        grad[n, :ln] = g
    return nll, grad
