"""Evaluation loop of the recognition path -- the counterpart of the reference's `cocr test` inner loop
(reference conformer_ocr/cli/test.py:185-212: batches -> `nn.predict_string` -> `CharErrorRate` / `WordErrorRate`).

Differences that matter on a GPU: lines are grouped into fixed-edge width buckets (a line's logits depend on its padded
width, SURVEY 0.6, so the padding must not depend on batch composition or rank count); batches are double-buffered on
side streams so that host->device copies, the forward and the label read-back overlap (the reference is fully serial);
ranks take batches round-robin with no collective in the loop.  torchmetrics is not a dependency: CER / WER are the same
edit-distance ratios (errors / reference length) it computes."""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple

import os

import numpy as np
import torch

from .dist import bucket_width, shard_batches


def edit_distance(a: Sequence, b: Sequence) -> int:
    """Levenshtein distance (insert / delete / substitute = 1)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


class ErrorRate:
    """torchmetrics.text.CharErrorRate / WordErrorRate semantics: sum(edit distance) / sum(reference length)."""

    def __init__(self, words: bool = False):
        self.words = words
        self.errors = 0
        self.total = 0

    def update(self, preds: Iterable[str], targets: Iterable[str]) -> None:
        for p, t in zip(preds, targets):
            if self.words:
                p, t = p.split(), t.split()
            self.errors += edit_distance(p, t)
            self.total += len(t)

    def compute(self) -> float:
        return self.errors / max(self.total, 1)


def global_align(seq1: Sequence, seq2: Sequence) -> Tuple[int, List, List]:
    """Edit-distance alignment of two sequences (the step `kraken.lib.util... global_align(y, x)` performs in the reference's report,
    cli/test.py:194; kraken is absent here: **parity unpinned**, known-answer tests only): returns (cost, aligned seq1, aligned
    seq2) of equal length, '' marking a gap; cost = insertions + deletions + substitutions (Needleman-Wunsch with unit costs; on equal
    cost a diagonal step is preferred, then a deletion from seq1, then an insertion)."""
    n, m = len(seq1), len(seq2)
    cost = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        cost[i][0] = i
    for j in range(1, m + 1):
        cost[0][j] = j
    for i in range(1, n + 1):
        ci, cp, a = cost[i], cost[i - 1], seq1[i - 1]
        for j in range(1, m + 1):
            ci[j] = min(cp[j - 1] + (a != seq2[j - 1]), cp[j] + 1, ci[j - 1] + 1)
    al1, al2 = [], []
    i, j = n, m
    while i > 0 or j > 0:
        if i > 0 and j > 0 and cost[i][j] == cost[i - 1][j - 1] + (seq1[i - 1] != seq2[j - 1]):
            al1.append(seq1[i - 1]); al2.append(seq2[j - 1]); i -= 1; j -= 1
        elif i > 0 and cost[i][j] == cost[i - 1][j] + 1:
            al1.append(seq1[i - 1]); al2.append(''); i -= 1
        else:
            al1.append(''); al2.append(seq2[j - 1]); j -= 1
    return cost[n][m], al1[::-1], al2[::-1]


def _script(c: str) -> str:
    import unicodedata
    try:
        return unicodedata.name(c).split()[0].title()
    except (ValueError, TypeError):
        return 'Unknown'


def compute_confusions(algn_gt: Sequence[str], algn_pred: Sequence[str]):
    """The tallies of the reference's report (cli/test.py:213, kraken `compute_confusions`; **parity unpinned**): over the aligned
    ground truth / prediction symbols -- `confusions` {(gt, pred): count} of the differing pairs (gaps as ''), `scripts` {script:
    characters of the ground truth}, `ins` {script: inserted characters}, `dels` deleted characters, `subs` {script: substituted}."""
    import collections
    counts, scripts, ins, subs = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
    dels = 0
    for g, p in zip(algn_gt, algn_pred):
        if g != '':
            scripts[_script(g)] += 1
        if g == p:
            continue
        counts[(g, p)] += 1
        if g == '':
            ins[_script(p)] += 1
        elif p == '':
            dels += 1
        else:
            subs[_script(g)] += 1
    return dict(counts.most_common()), dict(scripts), dict(ins), dels, dict(subs)


def render_report(model: str, chars: int, errors: int, char_accuracy: float, word_accuracy: float, confusions, scripts, ins, dels, subs) -> str:
    """Plain-text report with the reference's sections (cli/test.py:214-224: totals, per-script errors, the most frequent confusions)."""
    out = [f'=== report {model} ===', '', f'{chars}\tCharacters', f'{errors}\tErrors',
           f'{char_accuracy * 100:.2f}%\tCharacter Accuracy', f'{word_accuracy * 100:.2f}%\tWord Accuracy', '',
           f'{sum(ins.values())}\tInsertions', f'{dels}\tDeletions', f'{sum(subs.values())}\tSubstitutions', '',
           'Count\tMissed\t%Right']
    for scr, cnt in sorted(scripts.items(), key=lambda kv: -kv[1]):
        miss = subs.get(scr, 0)
        out.append(f'{cnt}\t{miss}\t{100.0 * (cnt - miss) / max(cnt, 1):.2f}%\t{scr}')
    out += ['', 'Errors\tCorrect-Generated']
    for (g, p), cnt in list(confusions.items())[:30]:
        out.append(f'{cnt}\t{{ {g or "<gap>"} }} - {{ {p or "<gap>"} }}')
    return '\n'.join(out)


def make_batches(widths: Sequence[int], batch_size: int, edge: int = 200) -> List[Tuple[int, List[int]]]:
    """[(padded width, [line indices])]: lines sorted into fixed-edge buckets, widest bucket first (like the reference's
    collate, lines inside a batch are ordered by width descending)."""
    buckets: Dict[int, List[int]] = {}
    for i, w in enumerate(widths):
        buckets.setdefault(bucket_width(w, edge), []).append(i)
    out = []
    for bw in sorted(buckets, reverse=True):
        idx = sorted(buckets[bw], key=lambda i: -widths[i])
        out.extend((bw, idx[k:k + batch_size]) for k in range(0, len(idx), batch_size))
    return out


def collate(lines: Sequence[np.ndarray], idx: Sequence[int], width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(N,1,H,W) float32 right-zero-padded batch + pixel widths (the `collate_sequences` contract, cli/test.py:186-189)."""
    h = lines[idx[0]].shape[0]
    batch = np.zeros((len(idx), 1, h, width), dtype=np.float32)
    lens = np.zeros(len(idx), dtype=np.int64)
    for n, i in enumerate(idx):
        w = lines[i].shape[1]
        batch[n, 0, :, :w] = lines[i]
        lens[n] = w
    return torch.from_numpy(batch), torch.from_numpy(lens)


class _PinnedPool:
    """Pinned staging memory, kept across calls: pinning host memory costs milliseconds per allocation (a fresh `pin_memory()` per
    batch held the whole loop at ~2 k lines/s; a set of buffers per batch SHAPE still made the first pass over a mixed-width queue
    -- eleven bucket widths, full and partial batches -- allocate for half a second).  Buffers are flat byte runs in power-of-two
    size classes (at least 1 MiB); a batch of any shape is a view of the first bytes of one.  A class's buffers are allocated
    on demand up to `depth`, handed out through a queue and handed back together with the event of the copy that read them.
    Least recently used classes are dropped beyond `cap` bytes."""

    def __init__(self, cap: int = 2 << 30):
        import queue
        import threading
        self._queue = queue.Queue
        self.lock = threading.Lock()
        self.classes: Dict[int, list] = {}          # bytes per buffer -> [free queue, buffers allocated]
        self.cap = cap

    @staticmethod
    def size_class(nbytes: int) -> int:
        return max(1 << 20, 1 << (max(nbytes, 1) - 1).bit_length())

    def take(self, shape: tuple, dtype: torch.dtype, depth: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(flat pinned buffer, its first bytes viewed as `shape` of `dtype`)."""
        nbytes = int(np.prod(shape)) * (1 if dtype == torch.uint8 else 4)
        key = self.size_class(nbytes)
        with self.lock:
            ent = self.classes.pop(key, None) or [self._queue(), 0]
            self.classes[key] = ent                  # most recently used last
            grow = ent[0].empty() and ent[1] < depth
            if grow:
                ent[1] += 1
                total = sum(e[1] * k for k, e in self.classes.items())
                for k in list(self.classes):
                    if total <= self.cap or k == key:
                        break
                    total -= self.classes.pop(k)[1] * k      # its tensors are freed once their last holders let go
        if not grow:
            try:
                flat, copied = ent[0].get(timeout=2.0)
                if copied is not None:
                    copied.synchronize()
            except Exception:                        # queue.Empty: a consumer that failed kept its buffers -- allocate rather than wait for ever
                with self.lock:
                    ent[1] += 1
                grow = True
        if grow:
            flat = torch.empty(key, dtype=torch.uint8).pin_memory()
        return flat, flat[:nbytes].view(dtype).view(shape)

    def give(self, flat: torch.Tensor, copied) -> None:
        with self.lock:
            ent = self.classes.get(flat.numel())
        if ent is not None:
            ent[0].put((flat, copied))


_PINNED = _PinnedPool()


class _Stager:
    """Producer side of the pipelined loops: a thread collates batch after batch into pinned staging buffers
    (`cocr_collate_lines`: native row copies on a few host threads, outside the GIL) while the caller's thread uploads, launches and
    reads back the batches before.  uint8 lines stay uint8 (the forward ingests them as they are: pixel / 255), anything else becomes
    float32.  Iterating yields (line indices, pinned batch, widths, key); the consumer calls `release(key, batch, copy event)`."""

    def __init__(self, lines: Sequence[np.ndarray], batches: Sequence[Tuple[int, List[int]]], order: Iterable[int], depth: int,
                 threads: int = 8, ahead: int = 2):
        import queue
        import threading
        from . import _lib
        self.lib = _lib.load()
        self.lines, self.batches, self.order = lines, batches, list(order)
        self.depth = max(ahead + 2, depth + ahead + 1)
        self.threads = max(1, min(threads, (os.cpu_count() or 2) // 2))
        self.ready = queue.Queue(maxsize=ahead)
        self.stop = False
        self.thread = threading.Thread(target=self._produce, name='cocr-collate', daemon=True)
        self.thread.start()

    def _collate(self, idx: Sequence[int], width: int):
        import ctypes as C
        lines = self.lines
        u8 = all(lines[i].dtype == np.uint8 for i in idx)
        dtype, npdt = (torch.uint8, np.uint8) if u8 else (torch.float32, np.float32)
        src = [np.ascontiguousarray(lines[i], dtype=npdt) for i in idx]          # no copy for contiguous lines of that type
        h = src[0].shape[0]
        for a in src:
            if a.ndim != 2 or a.shape[0] != h or a.shape[1] > width:
                raise ValueError(f'line of shape {a.shape} in a batch of height {h}, width {width}')
        key, buf = _PINNED.take((len(idx), 1, h, width), dtype, self.depth)      # key: the flat buffer to hand back
        widths = np.array([a.shape[1] for a in src], dtype=np.int32)
        ptrs = (C.c_void_p * len(src))(*[a.__array_interface__['data'][0] for a in src])
        rc = self.lib.cocr_collate_lines(ptrs, widths.ctypes.data_as(C.POINTER(C.c_int32)), len(src), h, 1 if u8 else 4,
                                         C.c_void_p(buf.data_ptr()), width, self.threads)
        if rc:
            raise RuntimeError((self.lib.cocr_last_error() or b'').decode())
        return idx, buf, torch.from_numpy(widths.astype(np.int64)), key

    def _produce(self):
        try:
            for b in self.order:
                if self.stop:
                    return
                width, idx = self.batches[b]
                self.ready.put(self._collate(idx, width))
            self.ready.put(None)
        except BaseException as e:               # handed to the consumer
            self.ready.put(e)

    def __iter__(self):
        while True:
            item = self.ready.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    @staticmethod
    def release(key: torch.Tensor, buf: torch.Tensor, copied) -> None:
        _PINNED.give(key, copied)

    def close(self) -> None:
        self.stop = True
        while self.thread.is_alive():            # unblock a producer waiting on a full queue
            try:
                item = self.ready.get(timeout=0.05)
                if isinstance(item, tuple):
                    _PINNED.give(item[3], None)
            except Exception:
                pass
        self.thread.join()


def recognize(net, lines: Sequence[np.ndarray], batch_size: int = 32, edge: int = 200, rank: int = 0, world: int = 1,
              device: str = 'cuda:0', pipelined: bool = True, streams: int = 1) -> Dict[int, str]:
    """Strings for this rank's share of `lines` (H x w float arrays in [0,1]); keys are line indices.

    pipelined: batch k+1 is collated, copied host->device (pinned staging buffer, side stream) and enqueued while batch k is
    computed; batch k's label records are read back while batch k+1 runs (the reference's loop, cli/test.py:185-212, is
    serial: copy, forward, decode, next).  streams > 1: that many batches in flight, each on a stream and a packed copy of the model
    of its own (`net.engine_pool`): one batch's kernels leave a third of the chip idle, four fill it (bench.py's `value` against
    `value_streams1`).  Same strings every way: a line's logits depend on its padded batch only."""
    batches = make_batches([l.shape[1] for l in lines], batch_size, edge)
    out: Dict[int, str] = {}
    dev = torch.device(device)
    from .ctc_decoder import BeamDecoder, GreedyDecoder
    if pipelined and streams > 1 and isinstance(net.ctc_decoder, (GreedyDecoder, BeamDecoder)):
        engines = net.engine_pool(streams, dev)
        cuda_streams = net.pool_streams(streams, dev)
        beam = net.ctc_decoder.beam_size if isinstance(net.ctc_decoder, BeamDecoder) else 0
        inflight: List[tuple] = []

        lut = net._codec_lut()

        def finish(item):
            idx, k, handle, _keep = item
            if lut is not None:                              # 1:1 codec: one table lookup per line (pred.py predict_string)
                for i, lab in zip(idx, engines[k].collect_labels(handle)):
                    out[i] = ''.join(lut[np.minimum(lab, lut.shape[0] - 1)].tolist())
                return
            for i, locs in zip(idx, engines[k].collect(handle)):
                out[i] = ''.join(x[0] for x in net.codec.decode(locs))
        stager = _Stager(lines, batches, shard_batches(len(batches), rank, world), depth=streams)
        try:
            for j, (idx, staged, lens, key) in enumerate(stager):
                k = j % streams
                eng = engines[k]
                with torch.cuda.stream(cuda_streams[k]):
                    d_im = staged.to(dev, non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record()
                    stager.release(key, staged, copied)
                    lg, ol = eng.forward(d_im.squeeze(1), lens.numpy())
                    handle = (eng._decode_async(eng.lib.cocr_ctc_beam, lg, ol, extra=(int(beam),)) if beam else eng.ctc_greedy_async(lg, ol))
                inflight.append((idx, k, handle, (d_im, lg)))
                if len(inflight) >= streams:
                    finish(inflight.pop(0))
            for item in inflight:
                finish(item)
        finally:
            stager.close()
        return out
    if not pipelined:
        for b in shard_batches(len(batches), rank, world):
            width, idx = batches[b]
            im, lens = collate(lines, idx, width)
            for i, s in zip(idx, net.predict_string(im.to(dev), lens)):
                out[i] = s
        return out
    copy_stream = torch.cuda.Stream(dev)
    stager = _Stager(lines, batches, shard_batches(len(batches), rank, world), depth=2)
    pending = None
    try:
        for idx, staged, lens, key in stager:
            with torch.cuda.stream(copy_stream):
                d_im = staged.to(dev, non_blocking=True)
                arrived = torch.cuda.Event()
                arrived.record(copy_stream)
            stager.release(key, staged, arrived)
            main = torch.cuda.current_stream(dev)
            main.wait_event(arrived)
            handle = net.predict_string_async(d_im, lens)
            d_im.record_stream(main)
            if pending is not None:
                for i, s in zip(pending[0], net.collect_strings(pending[1])):
                    out[i] = s
            pending = (idx, handle)
        if pending is not None:
            for i, s in zip(pending[0], net.collect_strings(pending[1])):
                out[i] = s
    finally:
        stager.close()
    return out


def recognize_crops(net, crops: Sequence[np.ndarray], batch_size: int = 32, edge: int = 200, pad: int = 16, rank: int = 0, world: int = 1,
                    device: str = 'cuda:0') -> Dict[int, str]:
    """The same loop starting one step earlier: `crops` are raw 8-bit line images ((H, W) grayscale or (H, W, 3) RGB, any height).
    Scaling to the model height, padding and collation run on the GPU (`net.transform_lines`, the reference pipeline's
    `ImageInputTransforms`); buckets are formed from the widths the lines will have after scaling, so a line's batch -- hence
    its string -- is the one `recognize` gives for the pre-processed line."""
    from . import _lib
    lib = _lib.load()
    widths = [int(lib.cocr_preproc_width(int(c.shape[0]), int(c.shape[1]), int(net.height), int(pad))) for c in crops]
    batches = make_batches(widths, batch_size, edge)
    out: Dict[int, str] = {}
    pending = None
    for b in shard_batches(len(batches), rank, world):
        width, idx = batches[b]
        im, lens = net.transform_lines([crops[i] for i in idx], pad=pad, bucket_edge=edge, device=device)
        assert im.shape[3] == width, (im.shape, width)
        handle = net.predict_string_async(im, lens)
        if pending is not None:
            for i, s in zip(pending[0], net.collect_strings(pending[1])):
                out[i] = s
        pending = (idx, handle)
    if pending is not None:
        for i, s in zip(pending[0], net.collect_strings(pending[1])):
            out[i] = s
    return out


def evaluate(net, lines: Sequence[np.ndarray], truths: Sequence[str], report: bool = False, model_name: str = 'model', **kw) -> Dict[str, float]:
    """CER / WER of `net` on (lines, truths): the numbers of cli/test.py:211-212; with `report` also the alignment-based tallies and the
    rendered text of cli/test.py:194-224 (`confusions`, `insertions`, `deletions`, `substitutions`, `report`)."""
    pred = recognize(net, lines, **kw)
    cer, wer = ErrorRate(False), ErrorRate(True)
    idx = sorted(pred)
    cer.update([pred[i] for i in idx], [truths[i] for i in idx])
    wer.update([pred[i] for i in idx], [truths[i] for i in idx])
    out = {'cer': cer.compute(), 'wer': wer.compute(), 'chars': cer.total, 'lines': len(idx)}
    if report:
        algn_gt, algn_pred, errors = [], [], 0
        for i in idx:
            c, a1, a2 = global_align(truths[i], pred[i])
            errors += c
            algn_gt.extend(a1)
            algn_pred.extend(a2)
        confusions, scripts, ins, dels, subs = compute_confusions(algn_gt, algn_pred)
        out.update(errors=errors, confusions=confusions, insertions=ins, deletions=dels, substitutions=subs,
                   report=render_report(model_name, cer.total, errors, 1.0 - out['cer'], 1.0 - out['wer'], confusions, scripts, ins, dels, subs))
    return out


def validate(net, lines: Sequence[np.ndarray], truths: Sequence[str], batch_size: int = 32, edge: int = 200, rank: int = 0, world: int = 1,
             device: str = 'cuda:0') -> Dict[str, float]:
    """The reference's validation epoch (model.py:154-193): per batch `_step` (forward + CTC loss), greedy decode of the same
    probits, CER / WER against the targets decoded back through the codec, `val_loss` = mean over batches of the summed loss
    (torchmetrics MeanMetric over `o['loss']`).  The loss, the argmax and the run merging all stay on the device; only label
    records and one float per line come back.  Lines whose text the codec cannot encode completely contribute what it encodes
    (kraken's non-strict codec drops unknown characters)."""
    from .ctc_decoder import GreedyDecoder
    if not isinstance(net.ctc_decoder, GreedyDecoder):
        raise ValueError('validation decodes greedily (model.py:163)')
    batches = make_batches([l.shape[1] for l in lines], batch_size, edge)
    cer, wer = ErrorRate(False), ErrorRate(True)
    losses: List[torch.Tensor] = []
    dev = torch.device(device)
    pending = None

    def finish(p):
        idx, handle, labels = p
        preds = [''.join(x[0] for x in net.codec.decode(locs)) for locs in net._engine.collect(handle)]
        refs = [''.join(x[0] for x in net.codec.decode([(l, 0, 0, 0) for l in lab])) for lab in labels]       # model.py:166-171
        cer.update(preds, refs)
        wer.update(preds, refs)

    for b in shard_batches(len(batches), rank, world):
        width, idx = batches[b]
        im, lens = collate(lines, idx, width)
        labels = [net.codec.encode(truths[i]) for i in idx]
        o = net.step({'image': im.to(dev), 'seq_lens': lens, 'target': np.array([l for lab in labels for l in lab], dtype=np.int64),
                      'target_lens': np.array([len(lab) for lab in labels], dtype=np.int64)})
        handle = net._engine.ctc_greedy_async(o['probits'], o['output_lens'].numpy())
        losses.append(o['loss'])
        if pending is not None:
            finish(pending)
        pending = (idx, handle, labels)
    if pending is not None:
        finish(pending)
    loss_sum = float(torch.stack(losses).sum().item()) if losses else 0.0
    counts = reduce_counts([cer.errors, cer.total, wer.errors, wer.total, loss_sum, len(losses)], dev if world > 1 else None)
    cer.errors, cer.total, wer.errors, wer.total = int(counts[0]), int(counts[1]), int(counts[2]), int(counts[3])
    val_loss = counts[4] / max(counts[5], 1.0)
    return {'cer': cer.compute(), 'wer': wer.compute(), 'val_loss': val_loss, 'val_accuracy': 1.0 - cer.compute(),
            'val_word_accuracy': 1.0 - wer.compute(), 'chars': cer.total, 'batches': int(counts[5])}


def reduce_counts(values: Sequence[float], device=None) -> List[float]:
    """Sum of each entry over the ranks (edit-distance / length / loss counters of a sharded validation epoch: the ratios are formed
    AFTER the sum, like torchmetrics' distributed reduction of CharErrorRate's `errors` / `total` states).  One small all-reduce
    (float64); a no-op without an initialised process group.  `device`: where the collective's buffer lives (the GPU for the nccl =
    RCCL backend, None = CPU for gloo)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [float(v) for v in values]
    use_gpu = device is not None and dist.get_backend() == 'nccl'
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device if use_gpu else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.cpu()]
