"""The N > 1 path on CPU: world_size 2 over gloo (lines shard as independent units; the only
collective is the start-up weight broadcast; results gather on rank 0 for reporting)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conformer_ocr_amd import synth
from conformer_ocr_amd.dist import broadcast_state_dict, bucket_width, gather_strings, shard_batches
from conformer_ocr_amd.spec import model_state_spec


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        hp = synth.hparams('tiny')
        spec = {k: v[0] for k, v in model_state_spec(hp).items() if v[1] != 'counter'}
        state = synth.make_state_dict(hp, seed=11) if rank == 0 else None
        got = broadcast_state_dict(state, spec, src=0)
        ref = synth.make_state_dict(hp, seed=11)
        same = all(np.array_equal(got[k], ref[k]) for k in spec)
        # sharding: 7 batches over 2 ranks, every batch exactly once
        mine = shard_batches(7, rank, world)
        strings = [f'line{b}' for b in mine]
        gathered = gather_strings(strings, mine, 7)
        q.put((rank, same, mine, gathered))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world_size_2_broadcast_shard_gather():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(r[1] for r in res)                                   # both ranks hold the root's weights
    assert sorted(res[0][2] + res[1][2]) == list(range(7))          # disjoint cover
    assert res[0][3] == [f'line{i}' for i in range(7)] and res[1][3] is None


def test_bucket_edges_are_rank_independent():
    assert [bucket_width(w) for w in (1, 200, 201, 1199, 1200, 2400)] == [200, 200, 400, 1200, 1200, 2400]
    # the padded width of a line (hence its logits: SURVEY 0.6) depends on the line alone
    widths = [417, 640, 903, 1111]
    assert [bucket_width(w) for w in widths] == [bucket_width(w) for w in reversed(widths)][::-1]


def test_evaluation_helpers():
    from conformer_ocr_amd.evaluate import ErrorRate, collate, edit_distance, make_batches
    assert edit_distance('kitten', 'sitting') == 3 and edit_distance('', 'abc') == 3 and edit_distance('abc', 'abc') == 0
    cer = ErrorRate()
    cer.update(['abcd', 'xy'], ['abed', 'xyz'])
    assert cer.errors == 2 and cer.total == 7 and abs(cer.compute() - 2 / 7) < 1e-12
    wer = ErrorRate(words=True)
    wer.update(['the cat sat'], ['the bat sat down'])
    assert wer.errors == 2 and wer.total == 4
    widths = [417, 640, 903, 1111, 1200, 1150, 130]
    b = make_batches(widths, batch_size=2)
    assert [w for w, _ in b] == [1200, 1200, 1000, 800, 600, 200]
    assert b[0][1] == [4, 5] and b[1][1] == [3]          # widest first inside a bucket
    lines = [np.full((4, w), 0.5, np.float32) for w in (5, 3)]
    im, lens = collate(lines, [0, 1], 8)
    assert im.shape == (2, 1, 4, 8) and lens.tolist() == [5, 3] and float(im[1, 0, 0, 3]) == 0.0


def _grad_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from conformer_ocr_amd.train import reduce_gradients
        g = torch.Generator().manual_seed(100 + rank)
        gw, gb = torch.randn((7, 16), generator=g), torch.randn((7,), generator=g)
        reduce_gradients((gw, gb))
        from conformer_ocr_amd.evaluate import reduce_counts
        counts = reduce_counts([3 + rank, 100 + 10 * rank, 1.5 * (rank + 1)])          # errors, characters, loss sum of each rank's shard
        q.put((rank, gw.numpy(), gb.numpy(), counts))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world_size_2_gradient_all_reduce():
    """Data-parallel output-layer training (conformer_ocr_amd/train.py): the two decoder gradients are AVERAGED over ranks in one
    flat bucket (torch DDP's semantics, what the reference's Lightning Trainer applies), every rank ends with the same tensors."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=100) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want_w = sum(torch.randn((7, 16), generator=torch.Generator().manual_seed(100 + r)) for r in range(2)).numpy()
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    ws = [torch.randn((7, 16), generator=g) for g in gens]
    bs = [torch.randn((7,), generator=g) for g in gens]
    np.testing.assert_allclose(res[0][1], ((ws[0] + ws[1]) / 2).numpy(), rtol=1e-6)
    np.testing.assert_allclose(res[0][2], ((bs[0] + bs[1]) / 2).numpy(), rtol=1e-6)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2]) and want_w.shape == (7, 16)
    assert res[0][3] == res[1][3] == [7.0, 210.0, 4.5]                             # validation counters: summed, identical on both ranks
