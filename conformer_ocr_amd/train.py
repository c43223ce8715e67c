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


class Trainer:
    """Training of the WHOLE network on the GPU: the reference's `training_step` + `configure_optimizers` + `optimizer_step` /
    `lr_scheduler_step` (reference conformer_ocr/model.py:147-152,238-321) behind one object.

        trainer = Trainer(net, lr=1e-3, weight_decay=1e-2, warmup=100, schedule='cosine', cos_t_max=50)
        for batch in loader: loss = trainer.training_step(batch)      # batch: image (N,1,H,W), seq_lens, target, target_lens
        trainer.end_epoch(val_accuracy)                               # epoch-wise schedulers
        trainer.sync_module()                                         # trained values into net.nn (state_dict / save_safetensors / predict_*)

    Forward in train mode (BatchNorm batch statistics + running-statistics update, dropout with the probabilities the model was
    constructed with), summed CTC loss, backward through decoder and encoder, AdamW -- all in libcocr_hip.so (include/cocr.h
    cocr_train_*; fp32).  Optimizer: AdamW / Adam-style arguments as the reference passes them (`torch.optim.AdamW(params, lr=lr,
    weight_decay=weight_decay)`, betas / eps torch's defaults); the reference's other optimizers (SGD, RMSprop with momentum) are not
    built.  Learning rate: linear warm-up over `warmup` steps exactly as `optimizer_step` applies it (the step itself runs at the rate
    set by the previous one; after step g the rate becomes min(1, (g + 1) / warmup) lr while g < warmup), then the epoch-wise schedule.
    Data parallel: one process per GPU, the flat gradient vector is averaged by ONE all-reduce per step (torch DDP's semantics)."""

    SCHEDULES = ('constant', 'exponential', 'cosine', 'step', 'reduceonplateau')

    def __init__(self, net: PytorchRecognitionModel, lr: float = 1e-3, weight_decay: float = 1e-3, optimizer: str = 'AdamW', warmup: int = 0,
                 schedule: str = 'constant', gamma: float = 0.1, cos_t_max: int = 50, cos_min_lr: float = 1e-4, step_size: int = 10,
                 rop_factor: float = 0.1, rop_patience: int = 5, completed_epochs: int = 0, seed: int = 0, process_group=None,
                 distributed: Optional[bool] = None, matmul_precision: str = 'highest'):
        if optimizer not in ('AdamW',):
            raise NotImplementedError(f'optimizer {optimizer}: only AdamW is built (the reference\'s default)')
        if schedule not in self.SCHEDULES:
            raise ValueError(f'Unsupported learning rate scheduler {schedule}.')                      # model.py:309
        self.net = net
        self.base_lr = self.lr = float(lr)
        self.weight_decay, self.warmup, self.schedule = float(weight_decay), int(warmup), schedule
        self.gamma, self.cos_t_max, self.cos_min_lr, self.step_size = float(gamma), int(cos_t_max), float(cos_min_lr), int(step_size)
        self.rop_factor, self.rop_patience = float(rop_factor), int(rop_patience)
        self.epoch = int(completed_epochs)
        self._sched_lr = self._lr_at_epoch(self.epoch)
        self.lr = self._sched_lr
        self._best, self._bad = None, 0
        self.global_step = 0
        self.seed = int(seed)
        self.process_group = process_group
        if distributed is None:
            distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size(process_group) > 1
        self.distributed = bool(distributed)
        dev = next(net.nn.parameters()).device
        from .engine import HipRecognizer
        self.engine = HipRecognizer(net.hparams_record, dev, 'fp32')
        self.engine.load_state({k: v for k, v in net.nn.state_dict().items()}, strict=True)
        # 'medium' = torch.set_float32_matmul_precision('medium') of the reference's cli/train.py:252 (bf16-rounded matmul operands);
        # 'highest' (default) = exact fp32 products, what the gradient-parity tests are stated for
        self.engine.train_begin(matmul_precision)

    def _lr_at_epoch(self, e: int) -> float:
        import math
        if self.schedule == 'exponential':
            return self.base_lr * self.gamma ** e                                                     # lr_scheduler.ExponentialLR
        if self.schedule == 'cosine':                                                                 # CosineAnnealingLR, closed form
            return self.cos_min_lr + (self.base_lr - self.cos_min_lr) * (1 + math.cos(math.pi * e / self.cos_t_max)) / 2
        if self.schedule == 'step':
            return self.base_lr * self.gamma ** (e // self.step_size)                                 # StepLR
        return self.base_lr

    def training_step(self, batch: Dict) -> float:
        """One optimizer step on `batch`; returns the batch's summed CTC loss (this rank's lines)."""
        image = batch['image']
        if image.dim() != 4 or image.shape[1] != 1:
            raise ValueError(f'expected a (N,1,H,W) line batch, got {tuple(image.shape)}')
        x = image.squeeze(1).to(self.engine.device)
        loss = self.engine.train_step(x, torch.as_tensor(batch['seq_lens']).cpu().numpy(), torch.as_tensor(batch['target']).cpu().numpy(),
                                      torch.as_tensor(batch['target_lens']).cpu().numpy(), dropout=self.net.dropout_p,
                                      seed=self.seed * 1000003 + self.global_step)
        if self.distributed:
            g = self.engine.train_grad_buffer()
            torch.distributed.all_reduce(g, op=torch.distributed.ReduceOp.SUM, group=self.process_group)
            g.div_(torch.distributed.get_world_size(self.process_group))
        self.engine.train_adamw(self.lr, weight_decay=self.weight_decay)
        if self.warmup and self.global_step < self.warmup:                                            # model.py:246-252
            self.lr = min(1.0, float(self.global_step + 1) / self.warmup) * self.base_lr
        elif self.warmup and self.global_step == self.warmup:
            self.lr = self._sched_lr
        self.global_step += 1
        return loss

    def end_epoch(self, metric: Optional[float] = None) -> float:
        """Epoch-wise scheduler step (model.py:254-265; not during warm-up); `metric` (validation accuracy, larger is better) drives
        'reduceonplateau'.  Returns the learning rate of the next epoch."""
        self.epoch += 1
        if self.warmup and self.global_step < self.warmup:
            return self.lr
        if self.schedule == 'reduceonplateau':
            if metric is not None:
                if self._best is None or metric > self._best:
                    self._best, self._bad = metric, 0
                else:
                    self._bad += 1
                    if self._bad > self.rop_patience:
                        self._sched_lr *= self.rop_factor
                        self._bad = 0
        else:
            self._sched_lr = self._lr_at_epoch(self.epoch)
        self.lr = self._sched_lr
        return self.lr

    def sync_module(self) -> None:
        """Copies every trained parameter and BatchNorm running statistic into `net.nn` (so `state_dict()`, `save_safetensors` and the
        inference path -- re-packed on its next call -- see them).  Training can continue afterwards."""
        sd = self.net.nn.state_dict()
        with torch.no_grad():
            for k, v in sd.items():
                if k.endswith('num_batches_tracked'):
                    v.add_(self.global_step - int(v))
                    continue
                v.copy_(torch.from_numpy(self.engine.train_value(k)).to(v.device).reshape(v.shape))
