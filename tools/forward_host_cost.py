#!/usr/bin/env python3
"""Dev: host cost of one `HipRecognizer.forward` call (32 x 96x1200, graph replay): the Python prelude, the C call, the rest."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_ocr_amd import _lib, synth  # noqa: E402
from conformer_ocr_amd.engine import HipRecognizer, _stream_ptr  # noqa: E402

dev = torch.device('cuda', 0)
hp = synth.hparams('cfg2')
eng = HipRecognizer(hp, dev, 'bf16')
eng.load_state(synth.make_state_dict(hp, seed=1236, decoder_gain=8.0))
eng.finalize()
eng.set_chain_rows(48)
eng.set_graph(True)
img, lens = synth.make_lines(32, hp.height, 1200, seed=7)
x = torch.from_numpy(img[:, 0]).to(dev)
lens32 = lens.astype(np.int32)
out = torch.empty((32, 300, hp.num_classes), dtype=torch.float32, device=dev)
for _ in range(5):
    eng.forward(x, lens32, out=out)
torch.cuda.synchronize()
t_all, t_c = [], []
out_lens = np.zeros(32, dtype=np.int32)
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.forward(x, lens32, out=out)
    t1 = time.perf_counter()
    t_all.append(t1 - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.device(dev):
        eng.lib.cocr_forward(eng._h, C.c_void_p(x.data_ptr()), _lib.F32, 32, 96, 1200, lens32.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(out.data_ptr()),
                             out_lens.ctypes.data_as(C.POINTER(C.c_int32)), _stream_ptr(dev))
    t1 = time.perf_counter()
    t_c.append(t1 - t0)
print('forward() total host %.1f us; the C call alone (incl. ctypes marshalling, torch.cuda.device, stream lookup) %.1f us' % (np.median(t_all) * 1e6, np.median(t_c) * 1e6))
t = []
for _ in range(200):
    t0 = time.perf_counter()
    with torch.cuda.device(dev):
        p = _stream_ptr(dev)
    t.append(time.perf_counter() - t0)
print('torch.cuda.device + current_stream: %.1f us' % (np.median(t) * 1e6))
# fresh tensors (staged replay): what predict_string does
for _ in range(5):
    eng.forward(x.clone(), lens32)
t = []
for _ in range(100):
    xc = x.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.forward(xc, lens32)
    t.append(time.perf_counter() - t0)
print('forward() with a fresh input and output tensor: %.1f us' % (np.median(t) * 1e6))
