import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_meta():
    with open(os.path.join(GOLDEN, 'meta.json')) as fp:
        return json.load(fp)


def load_case(meta, name):
    """(hp, state, image, lens, golden npz) of one golden fixture; weights and lines are regenerated
    from their seeds, the calibrated decoder bias comes from the fixture."""
    from conformer_ocr_amd import synth
    from conformer_ocr_amd.spec import HParams
    m = meta[name]
    hp = HParams(**m['hparams'])
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    state = synth.make_state_dict(hp, seed=m['seed'], decoder_gain=m['decoder_gain'], style=m.get('weight_style', 'plain'))
    state['decoder.bias'] = g['decoder_bias']
    image, lens = synth.make_lines(m['N'], hp.height, m['W'], seed=m['line_seed'], widths=m['widths'])
    return hp, state, image, lens, g


@pytest.fixture(scope='session')
def case(golden_meta):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_case(golden_meta, name)
        return cache[name]
    return get


class TextCase:
    """A "text" (peaked) fixture: lines with a ground truth, 'text'-style weights, the decoder fitted on the reference's own
    encoder output (tests/golden/make_golden.py run_text_case).  Lines and encoder weights are regenerated from their seeds;
    the fitted decoder, the reference's frame labels / margins / greedy strings and the ground truth come from the npz."""

    def __init__(self, meta, name):
        from conformer_ocr_amd import synth
        from conformer_ocr_amd.evaluate import make_batches
        from conformer_ocr_amd.spec import HParams
        m = meta[name]
        self.name, self.meta = name, m
        self.hp = HParams(**m['hparams'])
        self.g = g = np.load(os.path.join(GOLDEN, name + '.npz'))
        self.state = synth.make_state_dict(self.hp, seed=m['seed'], decoder_gain=1.0, style=m['style'])
        self.state['decoder.weight'], self.state['decoder.bias'] = g['decoder_weight'], g['decoder_bias']
        self.widths = list(m['widths'])
        self.n = len(self.widths)
        self.lines = [synth.make_text_lines(1, self.hp.height, w, seed=m['seed'] + 1000 + i, alphabet=m['alphabet'], alphabet_seed=m['seed'])[0][0, 0]
                      for i, w in enumerate(self.widths)]
        self.batch_size, self.edge = m['batch_size'], m['edge']
        self.batches = make_batches(self.widths, self.batch_size, self.edge) if self.edge else [(max(self.widths), list(range(self.n)))]
        self.out_lens = g['out_lens']
        off = np.concatenate([[0], np.cumsum(self.out_lens)])
        self.labels = [g['labels'][off[i]:off[i + 1]].astype(np.int64) for i in range(self.n)]          # reference argmax, frames < out_len
        self.margins = [g['margins'][off[i]:off[i + 1]].astype(np.float32) for i in range(self.n)]

        def unrag(flat, lens):
            o = np.concatenate([[0], np.cumsum(lens)])
            return [flat[o[i]:o[i + 1]].astype(np.int64).tolist() for i in range(len(lens))]
        self.texts = unrag(g['texts'], g['text_lens'])                      # ground truth label strings
        self.ref_strings = unrag(g['ref_strings'], g['ref_string_lens'])    # the reference's greedy label strings

    def batch(self, b):
        """(image (N,1,H,W) float32, lens, line indices) of batch b -- the batches the reference ran."""
        from conformer_ocr_amd.evaluate import collate
        bw, idx = self.batches[b]
        im, lens = collate(self.lines, idx, bw)
        return im.numpy(), lens.numpy(), idx


@pytest.fixture(scope='session')
def text_case(golden_meta):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = TextCase(golden_meta, name)
        return cache[name]
    return get
