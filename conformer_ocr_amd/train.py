"""Training step of the output layer on a frozen backbone -- the first slice of the reference's training loop
(`RecognitionModel.training_step` / `configure_optimizers`, reference conformer_ocr/model.py:147-152,238-250,267-290) on the GPU:

    forward (eval-mode encoder) -> CTC criterion + d loss / d probits (cocr_ctc_loss) -> decoder backward (cocr_decoder_backward)
    -> [all-reduce of the two gradients across ranks] -> AdamW (cocr_decoder_adamw)

This is what the reference does while `freeze_backbone` samples remain (cli/train.py:154-155: "keep the backbone (everything but
the last layer) frozen") and when a model is adapted to a new alphabet.  The encoder's backward is not implemented in this library;
`DecoderTrainer` says so instead of silently training less than asked.

Data-parallel training: one process per GPU; each rank computes its batch's gradients, `torch.distributed.all_reduce` (backend
"nccl" = RCCL over xGMI; two tensors, (ncls x D + ncls) x 4 bytes -- ~100 KB) combines them, every rank applies the same update: the
replicas stay bit-identical without a weight broadcast.  The gradients are AVERAGED over the ranks: the reference trains through
Lightning's `Trainer(devices=...)`, i.e. torch DDP, which divides the all-reduced gradients by the world size whatever the loss's
own reduction is (each rank's loss is the sum over ITS lines, `reduction='sum'`)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .pred import PytorchRecognitionModel


class DecoderTrainer:
    """AdamW training of `net.nn['decoder']` with the rest of `net` frozen.

    Hyper-parameter names and defaults follow the reference (`lrate`, `weight_decay`: model.py:48-50; AdamW betas / eps are torch's
    defaults, as the reference passes none)."""

    def __init__(self, net: PytorchRecognitionModel, lrate: float = 1e-3, weight_decay: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, distributed: Optional[bool] = None):
        self.net = net
        self.lrate, self.weight_decay, self.betas, self.eps = float(lrate), float(weight_decay), tuple(betas), float(eps)
        self.process_group = process_group
        if distributed is None:
            distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size(process_group) > 1
        self.distributed = bool(distributed)
        self.global_step = 0

    def training_step(self, batch: Dict) -> torch.Tensor:
        """One optimizer step on `batch` (the reference's batch dict: image, seq_lens, target, target_lens); returns the batch loss
        (0-dim device tensor, this rank's lines only, like the reference logs `train_loss`)."""
        o = self.net.step(batch, with_grad=True)
        eng = self.net._engine
        gw, gb, _ = eng.decoder_backward(o['grad_probits'])
        if self.distributed:
            reduce_gradients((gw, gb), self.process_group)
        eng.decoder_adamw(gw, gb, self.lrate, self.betas, self.eps, self.weight_decay)
        self.global_step += 1
        return o['loss']

    def sync_module(self) -> None:
        """Copies the trained output layer back into `net.nn['decoder']` (so `save_safetensors` / `state_dict()` see it)."""
        st = self.net._engine.decoder_state()
        dec = self.net.nn['decoder']
        with torch.no_grad():
            dec.weight.copy_(torch.from_numpy(st['decoder.weight']))
            dec.bias.copy_(torch.from_numpy(st['decoder.bias']))
        # the module now equals the engine's weights: keep the engine (and its optimizer state) instead of re-packing on the next forward
        self.net._engine_sig = self.net._signature(self.net._engine.device)


def reduce_gradients(tensors, group=None) -> None:
    """Mean of each gradient tensor over the ranks (torch DDP's semantics), in place: ONE collective on a flat bucket (the tensors
    are small; two launches of a ring all-reduce would cost two latencies over xGMI).  Sum + division: gloo has no AVG op."""
    flat = torch.cat([t.reshape(-1) for t in tensors])
    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=group)
    flat.div_(torch.distributed.get_world_size(group))
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
