#!/usr/bin/env python3
"""
Golden fixture of the TRAINING step (tests/golden/tiny_train.npz), made by running THE REFERENCE ITSELF in train mode:
`/root/reference/conformer_ocr/conformer/encoder.py::ConformerEncoder` with `.train()` (BatchNorm batch statistics; all dropout
probabilities 0 so that the step is deterministic), the `nn.Linear` decoder and the two loss lines of `RecognitionModel._step`
(model.py:119,136-142), float64, followed by `loss.backward()`.  Stored: loss, probits, the gradient of every parameter, the
BatchNorm running statistics after the step.  Run in the authoring container only:   python tests/golden/make_train_golden.py
"""
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

TARGETS = {'tiny_train': dict(config='tiny', seed=4321, n=3, W=64, widths=[64, 37, 50], targets=[[3, 1, 4], [1, 5], [9, 2, 6, 5]]),
           # round 4: the metric model's shapes (D = 256, 4 heads of 64, 256 conv channels, kernel 31), two blocks, short ragged lines.  4.8 M
           # parameters: per tensor the fixture holds 64 SAMPLED gradient entries (seeded indices) and its sum / sum of magnitudes / L2 norm /
           # largest magnitude instead of the whole gradient.
           'cfg2x2_train': dict(config='cfg2', over=dict(num_encoder_layers=2), seed=5, n=2, W=120, widths=[120, 77], targets=[[5, 9, 9, 3], [17]], sampled=64)}


def main():
    torch.manual_seed(0)
    for name, t in TARGETS.items():
        hp = synth.hparams(t['config'], **t.get('over', {}))
        state = synth.make_state_dict(hp, seed=t['seed'], decoder_gain=1.0)
        image, lens = synth.make_lines(t['n'], hp.height, t['W'], seed=t['seed'], widths=t['widths'])
        enc = ConformerEncoder(in_channels=1, input_dim=hp.height, encoder_dim=hp.encoder_dim, num_layers=hp.num_encoder_layers,
                               num_attention_heads=hp.num_attention_heads, feed_forward_expansion_factor=hp.feed_forward_expansion_factor,
                               conv_expansion_factor=hp.conv_expansion_factor, input_dropout_p=0.0, feed_forward_dropout_p=0.0,
                               attention_dropout_p=0.0, conv_dropout_p=0.0, conv_kernel_size=hp.conv_kernel_size,
                               half_step_residual=hp.half_step_residual, subsampling_conv_channels=hp.subsampling_conv_channels,
                               subsampling_factor=hp.subsampling_factor).double()
        dec = torch.nn.Linear(hp.encoder_dim, hp.num_classes, bias=True).double()
        enc.load_state_dict({k[len('encoder.'):]: torch.from_numpy(np.asarray(v)).double() if np.asarray(v).dtype.kind == 'f' else torch.from_numpy(np.asarray(v))
                             for k, v in state.items() if k.startswith('encoder.')})
        dec.load_state_dict({'weight': torch.from_numpy(state['decoder.weight']).double(), 'bias': torch.from_numpy(state['decoder.bias']).double()})
        enc.train()
        dec.train()
        x = torch.from_numpy(image).double().squeeze(1).transpose(1, 2)             # model.py:132
        eo, el = enc(x, torch.from_numpy(lens))                                     # model.py:134
        probits = dec(eo)
        logits = torch.nn.functional.log_softmax(probits, dim=-1)                   # model.py:136
        target = torch.tensor([c for s in t['targets'] for c in s], dtype=torch.long)
        tl = torch.tensor([len(s) for s in t['targets']], dtype=torch.long)
        crit = torch.nn.CTCLoss(reduction='sum', zero_infinity=True)                # model.py:119
        loss = crit(logits.transpose(0, 1), target, el.long(), tl)                  # model.py:139-142
        loss.backward()
        out = {'loss': np.float64(loss.item()), 'probits': probits.detach().numpy(), 'out_lens': el.numpy(), 'target': target.numpy(),
               'target_lens': tl.numpy()}
        named = [('encoder.' + k, p) for k, p in enc.named_parameters()] + [('decoder.' + k, p) for k, p in dec.named_parameters()]
        if t.get('sampled'):
            out['probits'] = out['probits'].astype(np.float32)
            rng = np.random.default_rng(t['seed'])
            for k, p in named:
                gflat = p.grad.numpy().reshape(-1)
                idx = np.sort(rng.choice(gflat.size, size=min(t['sampled'], gflat.size), replace=False)).astype(np.int64)
                out['gi:' + k], out['gs:' + k] = idx, gflat[idx]
                out['gn:' + k] = np.array([gflat.sum(), np.abs(gflat).sum(), np.sqrt((gflat ** 2).sum()), np.abs(gflat).max()])
        else:
            for k, p in named:
                out['grad:' + k] = p.grad.numpy()
        for k, b in enc.named_buffers():
            if 'running_' in k or 'num_batches' in k:
                out['buf:encoder.' + k] = b.detach().numpy()
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        print(name, 'loss', loss.item(), 'entries', len(out), os.path.getsize(os.path.join(HERE, name + '.npz')))


if __name__ == '__main__':
    main()
