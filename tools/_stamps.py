import ctypes as C, sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from conformer_ocr_amd import _lib, synth
from conformer_ocr_amd.engine import HipRecognizer
lib = _lib.load()
buf = (C.c_ulonglong * 64)()
lib.cocr_dev_stamps(buf)
hp = synth.hparams('cfg2')
eng = HipRecognizer(hp, torch.device('cuda', 0), 'bf16')
eng.load_state(synth.make_state_dict(hp, seed=1)); eng.finalize()
img, lens = synth.make_lines(32, 96, 1200, seed=1)
x = torch.from_numpy(img[:, 0]).cuda()
for _ in range(3): eng.forward(x, lens.astype(np.int32))
torch.cuda.synchronize()
lib.cocr_dev_stamps(buf)
for b in range(3):
    t = [buf[b*16+i] for i in range(4)]
    print('wg', b, 'prologue', t[1]-t[0], 'loop', t[2]-t[1], 'epilogue', t[3]-t[2], 'phases[stage,issue,qk+pos,softmax,pv]', [buf[b*16+8+i] for i in range(5)])
