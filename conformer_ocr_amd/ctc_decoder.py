"""Device CTC decoders with the call signature of `kraken.lib.ctc_decoder` functions, which is what
the reference stores in `PytorchRecognitionModel.ctc_decoder` (pred.py:42,68,93) and calls per line as
`self.ctc_decoder(seq[:, :seq_len])` on a (C, T) float matrix (pred.py:143,162,177).

The model class recognises these objects and decodes the whole batch in one kernel launch without
moving the logits to the host; called directly with a numpy matrix they decode that one line on the
GPU as well.  Any other callable put into `ctc_decoder` is honoured through the reference's own
per-line host loop."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np
import torch

from . import _lib


class _DeviceDecoder:
    kind = 'greedy'
    beam_size = 0

    def _run(self, outputs: np.ndarray, device=None):
        lib = _lib.load()
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        m = np.ascontiguousarray(np.asarray(outputs, dtype=np.float32).T)           # (T, C)
        T, ncls = m.shape
        if T == 0:
            return []
        from .engine import HipRecognizer
        from .spec import HParams
        # a decoder call needs only a device context: borrow a minimal model handle for scratch space
        eng = _scratch_engine(dev)
        lg = torch.from_numpy(m).to(dev).unsqueeze(0)
        if self.kind == 'greedy':
            return eng.ctc_greedy(lg, [T])[0]
        return eng.ctc_beam(lg, [T], self.beam_size)[0]

    def __call__(self, outputs: np.ndarray) -> List[Tuple[int, int, int, float]]:
        return self._run(outputs)


class GreedyDecoder(_DeviceDecoder):
    """kraken.lib.ctc_decoder.greedy_decoder: argmax, merge runs, drop blank 0."""
    kind = 'greedy'

    def __repr__(self):
        return 'greedy_decoder<hip>'


class BeamDecoder(_DeviceDecoder):
    """CTC prefix beam search over log-softmax(logits); semantics in include/cocr.h (cocr_ctc_beam)."""
    kind = 'beam'

    def __init__(self, beam_size: int = 16):
        self.beam_size = int(beam_size)

    def __repr__(self):
        return f'beam_decoder<hip,{self.beam_size}>'


_scratch = {}


def _scratch_engine(dev: torch.device):
    from .engine import HipRecognizer
    from .spec import HParams
    key = (dev.type, dev.index)
    if key not in _scratch:
        hp = HParams(num_classes=2, height=16, encoder_dim=16, num_encoder_layers=1, num_attention_heads=1,
                     conv_kernel_size=3, subsampling_conv_channels=8)
        _scratch[key] = HipRecognizer(hp, dev, 'fp32')
    return _scratch[key]


greedy_decoder = GreedyDecoder()


def beam_decoder(outputs: np.ndarray, beam_size: int = 16):
    return BeamDecoder(beam_size)(outputs)
