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
    state = synth.make_state_dict(hp, seed=m['seed'], decoder_gain=m['decoder_gain'])
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
