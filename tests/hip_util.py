"""Helpers shared by the GPU parity tests: run the HIP path through the C ABI and compare."""
import numpy as np
import torch

from conformer_ocr_amd.engine import HipRecognizer


def make_engine(hp, state, dtype):
    eng = HipRecognizer(hp, torch.device('cuda', 0), dtype)
    eng.load_state(state)
    eng.finalize()
    return eng


def run_hip(hp, state, image, lens, dtype, debug=False, as_u8=False):
    eng = make_engine(hp, state, dtype)
    eng.set_debug(debug)
    x = torch.from_numpy(image[:, 0])
    if as_u8:
        x = torch.from_numpy(np.rint(image[:, 0] * 255.0).astype(np.uint8))
    logits, out_lens = eng.forward(x.cuda(), lens)
    torch.cuda.synchronize()
    return eng, logits.cpu().numpy(), out_lens


def oracle_taps(hp, state, image, lens):
    from oracle.conformer_ref import Oracle
    taps = {}
    lg, ol = Oracle(hp, state, torch.float32).forward(torch.from_numpy(image), torch.from_numpy(lens), taps)
    out = {k: v.numpy() for k, v in taps.items()}
    # the attention operand layouts of the HIP path
    return lg.numpy(), ol.numpy(), out


def hip_tap(eng, name, hp, N, T):
    """A debug tap in the oracle's layout."""
    a = eng.tap(name)
    h, dh = hp.num_attention_heads, hp.d_head
    dhp, Tp = -(-dh // 32) * 32, -(-T // 64) * 64
    kind = name.split('.')[-1]
    if kind in ('q', 'k', 'v'):
        return a[:N * h * Tp * dhp].reshape(N, h, Tp, dhp)[:, :, :T, :dh].transpose(0, 2, 1, 3)
    if kind in ('z2', 'z3'):
        return a.reshape(N, T, -1, hp.subsampling_conv_channels)
    return a.reshape(N, T, -1)
