"""Golden vectors for the CTC loss of the training / validation step: runs the two lines of the reference's `RecognitionModel._step`
(reference conformer_ocr/model.py:119,136-142) -- `log_softmax` then `nn.CTCLoss(reduction='sum', zero_infinity=True)` on (T, N, C) --
with torch on the CPU in float64 and float32, and autograd for d loss / d probits.  Per-line values come from reduction='none' with
the same flags.  Run here (torch is the reference's own dependency); the output tests/golden/ctc_loss.npz is committed.

    python tests/golden/make_ctc_loss_golden.py
"""
import os

import numpy as np
import torch


def case(seed, N, T, C, max_l, kinds):
    g = np.random.default_rng(seed)
    probits = (g.standard_normal((N, T, C)) * 3.0).astype(np.float32)
    out_lens, label_lens, targets = [], [], []
    for n in range(N):
        kind = kinds[n % len(kinds)]
        ln = T if kind == 'full' else int(g.integers(1, T + 1))
        L = int(g.integers(0, min(max_l, ln) + 1))
        if kind == 'empty':
            L = 0
        if kind == 'repeats':
            L = min(max_l, max(ln // 2, 1))
            lab = np.repeat(g.integers(1, C, size=(L + 1) // 2), 2)[:L]        # aa bb cc: every pair needs a separating blank
        else:
            lab = g.integers(1, C, size=L)
        if kind == 'infeasible':
            ln = max(1, min(ln, 4))
            L = ln + 2
            lab = g.integers(1, C, size=L)
        if kind == 'one':
            ln = 1
            L = 1
            lab = g.integers(1, C, size=1)
        out_lens.append(ln)
        label_lens.append(L)
        targets.extend(int(v) for v in lab)
    return probits, np.array(targets, np.int64), np.array(out_lens, np.int64), np.array(label_lens, np.int64)


def reference_step(probits, targets, out_lens, label_lens, dtype):
    p = torch.tensor(probits, dtype=dtype, requires_grad=True)
    logits = torch.nn.functional.log_softmax(p, dim=-1)                                   # model.py:136
    crit = torch.nn.CTCLoss(reduction='sum', zero_infinity=True)                          # model.py:119
    loss = crit(logits.transpose(0, 1), torch.tensor(targets), torch.tensor(out_lens), torch.tensor(label_lens))   # model.py:139-142
    loss.backward()
    per_line = torch.nn.CTCLoss(reduction='none', zero_infinity=True)(logits.detach().transpose(0, 1), torch.tensor(targets),
                                                                      torch.tensor(out_lens), torch.tensor(label_lens))
    return loss.item(), per_line.numpy(), p.grad.numpy()


def main():
    cases = {
        'mixed': case(1, 6, 40, 12, 14, ['rand', 'full', 'repeats', 'empty', 'infeasible', 'one']),
        'wide': case(2, 4, 75, 120, 40, ['rand', 'full', 'repeats', 'rand']),
        'long': case(3, 3, 300, 50, 140, ['full', 'repeats', 'rand']),
    }
    out = {}
    for name, (probits, targets, out_lens, label_lens) in cases.items():
        l64, pl64, g64 = reference_step(probits, targets, out_lens, label_lens, torch.float64)
        l32, pl32, g32 = reference_step(probits, targets, out_lens, label_lens, torch.float32)
        out.update({f'{name}.probits': probits, f'{name}.targets': targets, f'{name}.out_lens': out_lens, f'{name}.label_lens': label_lens,
                    f'{name}.loss64': np.float64(l64), f'{name}.per_line64': pl64, f'{name}.grad64': g64.astype(np.float32),
                    f'{name}.loss32': np.float64(l32), f'{name}.per_line32': pl32})
        print(name, probits.shape, 'loss', l64, l32, 'max |grad64 - grad32|', float(np.abs(g64 - g32).max()))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ctc_loss.npz'), **out)


if __name__ == '__main__':
    main()
