"""Known-answer vectors for the CTC restatement (oracle/ctc_ref.py).  PARITY UNPINNED: kraken is a
third-party dependency absent from the reference tree; these vectors are hand-derived from the
algorithm statement."""
import numpy as np

from oracle.ctc_ref import beam_decoder, greedy_decoder, log_softmax


def onehot_path(path, C=6, hi=5.0, lo=-1.0):
    m = np.full((C, len(path)), lo, dtype=np.float32)
    for t, c in enumerate(path):
        m[c, t] = hi + 0.1 * t
    return m


def test_greedy_known_answer():
    # argmax path [0,3,3,0,3,5,5,0] -> [(3,1,2),(3,4,4),(5,5,6)]
    m = onehot_path([0, 3, 3, 0, 3, 5, 5, 0])
    r = greedy_decoder(m)
    assert [(a, b, c) for a, b, c, _ in r] == [(3, 1, 2), (3, 4, 4), (5, 5, 6)]
    assert np.allclose([x[3] for x in r], [5.2, 5.4, 5.6])          # max over the run of the label's logit


def test_greedy_edges():
    assert greedy_decoder(np.zeros((4, 0), np.float32)) == []
    assert greedy_decoder(onehot_path([0, 0, 0])) == []
    assert [(a, b, c) for a, b, c, _ in greedy_decoder(onehot_path([2]))] == [(2, 0, 0)]
    assert [(a, b, c) for a, b, c, _ in greedy_decoder(onehot_path([1, 1, 1, 1]))] == [(1, 0, 3)]
    # ties: numpy argmax takes the first index -> class 0 (blank) wins a full tie
    assert greedy_decoder(np.zeros((5, 3), np.float32)) == []
    tie = np.zeros((5, 2), np.float32)
    tie[0] = -1.0
    assert [(a, b, c) for a, b, c, _ in greedy_decoder(tie)] == [(1, 0, 1)]


def test_beam_equals_greedy_on_peaked_input():
    path = [0, 3, 3, 0, 3, 5, 5, 0, 1, 0]
    m = onehot_path(path, hi=12.0, lo=-6.0)
    b = beam_decoder(m, 16)
    g = greedy_decoder(m)
    assert [x[0] for x in b] == [x[0] for x in g] == [3, 3, 5, 1]
    assert [(x[1], x[2]) for x in b] == [(x[1], x[2]) for x in g]
    assert all(0.99 < x[3] <= 1.0 for x in b)


def test_beam_sums_paths_where_greedy_does_not():
    # classic case: per frame blank is the single most likely symbol, but label 1 wins once the
    # probability of all alignments of "1" is summed:  p(blank)=0.4, p(1)=0.35, p(2)=0.25 for 2 frames
    p = np.array([[0.4, 0.4], [0.35, 0.35], [0.25, 0.25]], dtype=np.float32)
    m = np.log(p)
    assert greedy_decoder(m) == []
    # "" : 0.16 ; "1": 0.35*0.4*2 + 0.35^2 = 0.4025 ; "2": 0.2625 ...
    assert [x[0] for x in beam_decoder(m, 16)] == [1]
    np.testing.assert_allclose(np.exp(log_softmax(m)), p, rtol=1e-5)
