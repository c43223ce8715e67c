#!/usr/bin/env python3
"""
bench.py -- throughput of the recognition hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg4]

A "step" is one pass of the hot path over one batch, through the C ABI of libcocr_hip.so:
line batch resident in HBM -> conformer encoder (bf16 operands) -> decoder -> CTC greedy label records.

  --config cfg2 (default; BASELINE.json configs[1], the configuration the metric is quoted on): 32 synthetic 96x1200 lines,
      D=256, 12 blocks.  The batch is the `cfg2_text` fixture's (tests/golden): text lines with a ground truth and a decoder
      fitted on the reference's own encoder output, so that the CER beside the throughput means something.
  --config cfg4 (configs[3]): D=512, 16 blocks, `--lines` lines of widths U{400..2400} step 8 in fixed 200-px buckets,
      batches of <= `--batch` lines padded to their bucket's width; a step is the next batch of that queue.

With N > 1 every rank owns one GPU, receives the packed weights by one RCCL broadcast and processes its own independent
batches (no data-path collective): weak scaling.  Launched either by `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...` (RANK set by the launcher) or plainly as `python bench.py --gpus N ...`: without RANK in the environment this
process starts the N rank processes itself (`launch_ranks`, before anything has touched the GPU) and forwards rank 0's line.
Rank 0 prints ONE JSON line:

  value                 whole-job lines/s, `--streams` (4) batches in flight per GPU, inputs resident in HBM
  value_streams1        the same loop with ONE batch in flight (what a single caller thread without its own streams sees; 48-row workgroups)
  predict_string        `PytorchRecognitionModel.predict_string` (the drop-in surface) called with a FRESH input tensor per call
  ingest                u8 lines from pinned host memory, double-buffered host->device copies inside the loop (PCIe-inclusive)
  cer_vs_reference      CER of the bf16 strings against the reference's fp32 greedy strings (fixture), cer_vs_truth: against the text
  roofline              the dominant kernel: algorithmic FLOP per launch / HIP-event duration measured live on the forward's stream
  cpu_baseline          the CPU oracle (fp32 restatement of the reference) on the host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

# One hardware queue per batch in flight: the HIP runtime multiplexes its streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues,
# and two of this bench's streams sharing a queue serialise "independent" batches (measured: 4 streams on the default 4 queues --
# one is taken by the null stream -- 30.2k lines/s, on 8 queues 38.0k; 3 streams on 4 queues 36.6k).  Read at runtime start-up.
_HWQ_PRESET = os.environ.get('GPU_MAX_HW_QUEUES')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

ROOT = os.path.dirname(os.path.abspath(__file__))


def launch_ranks(n, argv, timeout_s=600.0):
    """`python bench.py --gpus N` without a launcher around it: this process starts N fresh rank processes of this same script --
    one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment, exactly what
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` would give them -- and does no GPU work itself.  (It counts the
    devices with `torch.cuda.device_count()`; should that call bring up the HIP runtime in this process on some build, it stays
    harmless: the ranks are fresh children started with `subprocess.Popen`, never an exec or re-exec of this process.)
    Rank 0's stdout (the ONE JSON line) is drained by a reader thread WHILE the ranks run (a line longer than the pipe's buffer would
    otherwise block rank 0 in its write and the launcher in its wait) and forwarded to this process's stdout; every rank's stderr is
    inherited.  `timeout_s` (`--launch-timeout`, default 600 s): ranks still running then -- e.g. stuck in the RCCL rendezvous -- are
    named on stderr and ended by PID, exit code 124.  Returns the exit code: 0 only if every rank returned 0 and rank 0 printed a line;
    the first failing rank's code otherwise (the other ranks are terminated by PID)."""
    import socket
    import subprocess
    import threading
    if '--dry-run' not in argv:
        import torch
        seen = torch.cuda.device_count()
        if seen < n:
            sys.stderr.write(f'bench.py: --gpus {n} but only {seen} GPU(s) are visible\n')
            return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != '--spawn'], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, cwd=os.getcwd()))
    chunks = []

    def drain():
        for piece in iter(lambda: procs[0].stdout.read(65536), b''):
            chunks.append(piece)
    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    t_end = None if not timeout_s or timeout_s <= 0 else time.time() + timeout_s
    rc, live = 0, set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0:
                    sys.stderr.write(f'bench.py: rank {r} exited with code {c}\n')
                    rc = c if c > 0 else 1
        if live and rc == 0 and t_end is not None and time.time() > t_end:
            sys.stderr.write(f'bench.py: ranks {sorted(live)} still running after {timeout_s:g} s (--launch-timeout): ending them\n')
            rc = 124
        if live and rc == 0:
            time.sleep(0.05)
    for r in live:                                  # a rank failed or the time is up: the others would wait for it at the next barrier
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    reader.join(timeout=20)
    out = b''.join(chunks).decode('utf-8', 'replace')
    lines = [ln for ln in out.splitlines() if ln.strip()]
    if rc == 0 and len(lines) != 1:
        sys.stderr.write(f'bench.py: rank 0 printed {len(lines)} lines, expected one\n')
        rc = 1
    if rc == 0:
        sys.stdout.write(lines[0] + '\n')
        sys.stdout.flush()
    return rc


def _launcher_args(argv):
    """(gpus, spawn, launch timeout) from the command line without importing anything heavy."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--spawn', action='store_true')
    ap.add_argument('--launch-timeout', type=float, default=600.0)
    a, _ = ap.parse_known_args(argv)
    return a.gpus, a.spawn, a.launch_timeout


if __name__ == '__main__' and 'RANK' not in os.environ:
    _n, _spawn, _lt = _launcher_args(sys.argv[1:])
    if _n > 1 or _spawn:
        sys.exit(launch_ranks(_n, sys.argv[1:], timeout_s=_lt))

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)

from conformer_ocr_amd import synth  # noqa: E402
from conformer_ocr_amd.engine import HipRecognizer  # noqa: E402
from conformer_ocr_amd.evaluate import ErrorRate, collate, make_batches  # noqa: E402
from conformer_ocr_amd.spec import HParams, flops_per_line, out_len  # noqa: E402

METRIC = 'text lines/sec (whole node) + CER vs reference, 96×1200 bf16 batch'      # BASELINE.json `metric`, verbatim
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}     # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def kernel_table(hp, n, w, dtype):
    """Algorithmic work per launch of every kernel family for a batch of n lines of width w:
    (bound, flops or bytes per launch).  GEMMs: 2*M*N*K flop.  HBM-bound kernels: one read + one write
    of their operand."""
    es = 2 if dtype == 'bf16' else 4
    D, C, ff = hp.encoder_dim, hp.subsampling_conv_channels, hp.feed_forward_expansion_factor * hp.encoder_dim
    T1, T, F = out_len(w, 1), out_len(w, 2), hp.out_feats
    M = n * T
    ncls = hp.num_classes
    return {
        'frontend_fused': ('mfma', 2.0 * n * (9 * C * T1 * out_len(hp.height, 1) + 9 * C * T * F) + 2.0 * n * T * F * C * C),
        'frontend_conv12': ('valu', 2.0 * n * (9 * C * T1 * out_len(hp.height, 1) + 9 * C * T * F)),
        'gemm_front_pw': ('mfma', 2.0 * n * T * F * C * C),
        'gemm_front_out': ('mfma', 2.0 * M * F * C * D),
        'layernorm': ('hbm', M * D * (4 + es)),
        'gemm_ffn_up': ('mfma', 2.0 * M * D * ff),
        'gemm_ffn_down': ('mfma', 2.0 * M * D * ff),
        'ffn_fused': ('mfma', 4.0 * M * D * ff),
        'ffn_probe': ('mfma', 4.0 * M * D * ff),
        'chain_ffn_qkv': ('mfma', 4.0 * M * D * ff + 6.0 * M * D * D),
        'chain_front_ffn_qkv': ('mfma', 2.0 * M * F * C * D + 4.0 * M * D * ff + 6.0 * M * D * D),
        'chain_attn_out_glu': ('mfma', 6.0 * M * D * D),
        'chain_pw2_ffn_ffn_qkv': ('mfma', 8.0 * M * D * ff + 8.0 * M * D * D),
        'chain_pw2_ffn': ('mfma', 4.0 * M * D * ff + 2.0 * M * D * D),
        'gemm_qkv': ('mfma', 2.0 * M * D * 3 * D),
        'attention': ('mfma', 2.0 * n * 3 * T * T * D),
        'gemm_attn_out': ('mfma', 2.0 * M * D * D),
        'gemm_glu': ('mfma', 2.0 * M * D * 2 * D),
        'dwconv': ('hbm', 2.0 * M * D * es),
        'gemm_pw2': ('mfma', 2.0 * M * D * D),
        'gemm_decoder': ('mfma', 2.0 * M * D * ncls),
        'ctc_greedy': ('hbm', 4.0 * M * ncls),
    }


def engine_dim(hp):
    """Width the bf16 engine runs a model at (cocr_api.hip set_engine_dims): 128 <= encoder_dim < 256 as a zero-padded 256-wide model,
    256 < encoder_dim < 512 as a 512-wide one."""
    d = hp.encoder_dim
    return 256 if 128 <= d <= 256 else 512 if 256 < d <= 512 else d


def chain_block_rows(hp, rows):
    """Rows per workgroup of the row-chain kernels in the throughput form (rowchain.hip.h rowchain_pick_mt)."""
    big = 96 if engine_dim(hp) == 256 else 64
    return big if rows >= big * 50 else 32


def cus_occupied(family, hp, rows):
    """CUs a launch of `family` can occupy (MI355X: 256): the row-block chain kernels run one workgroup per CU."""
    if family.startswith('chain_'):
        return min(256, -(-rows // chain_block_rows(hp, rows)))
    return 256


def profile_kernel_name(family, hp, rows):
    """Substring of the kernel's name in the rocprofv3 summaries."""
    chain = {'chain_pw2_ffn_ffn_qkv': '31, 0, 1, 1, 3', 'chain_attn_out_glu': '0, 0, 2, -1, -1', 'chain_ffn_qkv': '0, 1, 3, -1, -1', 'chain_front_ffn_qkv': '0, 4, 1, 3, -1', 'chain_pw2_ffn': '31, 0, 1, -1, -1'}
    if family in chain:
        return f'rowchain_kernel<{engine_dim(hp)}, {chain_block_rows(hp, rows) // 16}, {chain[family]}'
    return {'ffn_fused': 'ffn_fused_kernel', 'attention': 'relpos_attention_kernel', 'dwconv': 'dwconv_bn_silu', 'frontend_conv12': 'frontend_conv12',
            'frontend_fused': 'frontend96_kernel'}.get(family)


def pmc_traffic(family, live_avg_ms, hp, rows, tag):
    """HBM bytes per launch of a kernel family from the NEWEST committed rocprofv3 PMC summary (PMC counters cannot be read from
    inside the process: they come from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this bench command, condensed
    by tools/summarize_rocprof.py).  Returned with the file it came from and a staleness flag: the kernel's average duration in
    the kernel-trace summary of the same round against the duration measured live in this run."""
    import csv
    import glob
    name = profile_kernel_name(family, hp, rows)
    files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', f'*{tag}_pmc_traffic.csv')) if ('_cfg' in os.path.basename(f)) == bool(tag))
    if name is None or not files:
        return None, None
    f = files[-1]
    traffic = None
    for r in csv.DictReader(open(f)):
        if name in r['kernel']:
            traffic = int(r['hbm_bytes_per_launch(2*fetch+write)'])
            break
    if traffic is None:
        return None, {'file': os.path.relpath(f, ROOT), 'note': f'no row for {name}'}
    info = {'file': os.path.relpath(f, ROOT)}
    stats = f.replace('_pmc_traffic.csv', '_kernel_stats.csv')
    if os.path.exists(stats):
        for r in csv.DictReader(open(stats)):
            if name in r['kernel']:
                prof_ms = float(r['avg_ns']) * 1e-6
                info.update(profile_avg_ms=round(prof_ms, 5), stale=bool(abs(prof_ms - live_avg_ms) > 0.15 * live_avg_ms))
                break
    return traffic, info


def load_text_fixture(name):
    """A "text" fixture of tests/golden (make_golden.py run_text_case): hp, weights (seeded 'text'-style encoder + the fitted
    decoder stored in the npz), lines, ground-truth label strings, the reference's greedy label strings.  None if absent."""
    meta_p, npz_p = os.path.join(GOLDEN, 'meta.json'), os.path.join(GOLDEN, name + '.npz')
    if not (os.path.exists(meta_p) and os.path.exists(npz_p)):
        return None
    m = json.load(open(meta_p)).get(name)
    if m is None:
        return None
    g = np.load(npz_p)
    hp = HParams(**m['hparams'])
    state = synth.make_state_dict(hp, seed=m['seed'], decoder_gain=1.0, style=m['style'])
    state['decoder.weight'], state['decoder.bias'] = g['decoder_weight'], g['decoder_bias']
    lines = [synth.make_text_lines(1, hp.height, w, seed=m['seed'] + 1000 + i, alphabet=m['alphabet'], alphabet_seed=m['seed'])[0][0, 0]
             for i, w in enumerate(m['widths'])]

    def unrag(flat, lens):
        o = np.concatenate([[0], np.cumsum(lens)])
        return [flat[o[i]:o[i + 1]].astype(np.int64).tolist() for i in range(len(lens))]
    return {'hp': hp, 'state': state, 'lines': lines, 'texts': unrag(g['texts'], g['text_lens']), 'meta': m,
            'ref_strings': unrag(g['ref_strings'], g['ref_string_lens'])}


def cpu_baseline(hp, state, width, sample_lines):
    """The CPU oracle (fp32 restatement of the reference forward + greedy decode) on the host cores."""
    from oracle.conformer_ref import Oracle
    from oracle.ctc_ref import greedy_decoder
    # the box's CPU share, not the host's core count: oversubscribing 256 threads on a 16-core share is 200x slower
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    o = Oracle(hp, state)
    img, lens = synth.make_lines(sample_lines, hp.height, width, seed=99)
    x, l = torch.from_numpy(img), torch.from_numpy(lens)
    o.forward(x[:1], l[:1])                                   # warm-up (thread pool, pos tables)
    t0 = time.perf_counter()
    reps = 0
    while True:
        logits, ol = o.forward(x, l)
        for n in range(sample_lines):
            greedy_decoder(logits[n, :int(ol[n])].numpy().T)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or reps >= 8:
            break
    return {'value': round(sample_lines * reps / dt, 3), 'unit': 'lines/s', 'cores': cores, 'kind': 'port',
            'sample': f'{reps} x {sample_lines} lines of 96x{width}, fp32 torch-CPU oracle forward + greedy decode, {dt:.1f} s'}


def strings_of(records):
    return [[int(r[0]) for r in line] for line in records]


def dry_run(args, json_fd):
    """The launcher's rehearsal (tests/test_bench_launcher.py): every rank joins a gloo group, the ranks count themselves by an
    all-reduce of ones, rank 0 prints one line that says it measured nothing."""
    import torch.distributed as dist
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if rank == args.dry_run_fail_rank:
        sys.stderr.write(f'bench.py: dry-run rank {rank} fails on request\n')
        sys.exit(3)
    if rank == args.dry_run_hang_rank:
        sys.stderr.write(f'bench.py: dry-run rank {rank} hangs on request\n')
        time.sleep(3600)
    if world != args.gpus:
        sys.exit(2)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ones = torch.ones(1)
    dist.all_reduce(ones)
    dist.barrier()
    if rank == 0:
        line = (json.dumps({'metric': METRIC, 'value': None, 'unit': 'lines/s', 'n_gpus': world, 'dry_run': True,
                            'config': {'ranks_seen': int(ones.item()), 'backend': 'gloo'},
                            'pad': 'x' * int(os.environ.get('COCR_BENCH_DRYRUN_PAD', '0'))}) + '\n').encode()
        while line:                                  # (a line longer than the pipe's buffer: os.write may take it in pieces)
            line = line[os.write(json_fd, line):]
    else:
        print(f'rank {rank}: this text must not reach the launcher\'s stdout', flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--width', type=int, default=1200)
    ap.add_argument('--lines', type=int, default=320, help='cfg4: lines in the mixed-width queue of one rank')
    ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg4', 'cfg1'])
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-legs', action='store_true', help='skip the streams1 / predict_string / ingest legs (profiling runs)')
    ap.add_argument('--profile-steps', type=int, default=3)
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel from the host instead of replaying a captured hipGraph')
    ap.add_argument('--stagger-us', type=float, default=0.0, help='host-side offset between the first batches of a timed run (they would otherwise start in lockstep: all frontends, then all attention kernels, ... at the same time)')
    ap.add_argument('--queue-depth', type=int, default=2, help='steps queued per stream before the host waits for the oldest (1: a stream\'s next step is launched when its previous one has been collected)')
    ap.add_argument('--private-weights', action='store_true', help='every packed model copy holds its own weights (the form before cocr_share_weights)')
    ap.add_argument('--streams', type=int, default=4, help='independent batches in flight (one packed model + HIP stream each)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the bounded cfg4 / cfg1 legs of the default line')
    ap.add_argument('--prewarm-ms', type=float, default=500.0, help='untimed run of the same loop before the warm-up steps (clock ramp of an idle chip)')
    ap.add_argument('--spawn', action='store_true', help='start the rank process(es) as fresh children of this process even for --gpus 1 (the path --gpus N > 1 always takes when no launcher set RANK)')
    ap.add_argument('--dry-run', action='store_true', help='launcher rehearsal without a GPU: gloo rendezvous, barrier, all-reduce, a line with value null')
    ap.add_argument('--dry-run-fail-rank', type=int, default=-1, help='(dry run) this rank exits with code 3 before the rendezvous')
    ap.add_argument('--dry-run-hang-rank', type=int, default=-1, help='(dry run) this rank sleeps instead of joining the rendezvous: what a rank stuck in RCCL init looks like to the launcher')
    ap.add_argument('--launch-timeout', type=float, default=600.0, help='(launcher) seconds after which ranks that are still running are named on stderr and ended by PID, exit code 124; 0 = never')
    ap.add_argument('--profile-batch-only', action='store_true', help='(rocprofv3 runs of a mixed-width queue) the queue is reduced to its heaviest batch, the one the roofline leg prices: the profile\'s per-kernel averages are then that shape\'s')
    ap.add_argument('--no-latency-leg', action='store_true', help='skip the bounded small-batch latency leg (BASELINE configs[4])')
    args = ap.parse_args()
    # stdout carries exactly ONE JSON line: libraries that print banners to fd 1 (RCCL / gloo at communicator creation) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.dry_run:
        return dry_run(args, json_fd)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_dist = 'RANK' in os.environ          # launched by torch.distributed.run (any world size, also 1)
    if world != args.gpus:                   # a line for N GPUs is only printed by a job that runs on N ranks
        sys.stderr.write(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}\n')
        sys.exit(2)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    ranks_seen = 1
    if use_dist:                             # the ranks count themselves over RCCL: the line says how many took part
        ones = torch.ones(1, device=torch.device('cuda', local_rank))
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    # ---- model: the text fixture's weights when the fixture is there (same FLOPs as any other weights; the CER then has a meaning)
    fix = load_text_fixture({'cfg1': 'cfg1_text', 'cfg2': 'cfg2_text', 'cfg4': 'cfg4_text'}.get(args.config, ''))
    hp = fix['hp'] if fix else synth.hparams(args.config)
    S = max(1, args.streams)
    engines = [HipRecognizer(hp, dev, args.dtype) for _ in range(S)]
    eng = engines[0]
    state = None
    if rank == 0:
        state = fix['state'] if fix else synth.make_state_dict(hp, seed=1236, decoder_gain=8.0)
        eng.load_state(state)
        eng.finalize()
    else:
        eng.finalize_empty()
    if use_dist:
        from conformer_ocr_amd.dist import broadcast_weights
        broadcast_weights(eng, src=0)
    if not args.private_weights:
        for e in engines[1:]:
            e.share_weights(eng)          # one set of weights and tables per GPU: the packed copies differ in workspace and captured launches only
    else:                                 # (every copy its own weights: 4 x ~100 MB no longer fit the Infinity Cache)
        blob = eng.export_blob()
        for e in engines[1:]:
            e.finalize_empty()
            e.import_blob(blob)
        torch.cuda.synchronize(dev)

    # ---- per-rank independent batches, resident in HBM (float32 (N,H,W): what the reference's loader hands over, cli/test.py:189)
    if args.config == 'cfg4':
        g = np.random.Generator(np.random.PCG64(5000 + rank))
        widths = (400 + 8 * g.integers(0, 251, args.lines)).tolist()
        lines = [synth.make_text_lines(1, hp.height, int(w), seed=9000 + 1000 * rank + i, alphabet_seed=fix['meta']['seed'] if fix else 1)[0][0, 0]
                 for i, w in enumerate(widths)]
        plan = make_batches(widths, args.batch, 200)
        workload = f'cfg4: conformer D={hp.encoder_dim} L={hp.num_encoder_layers} heads={hp.num_attention_heads}, {args.lines} lines per GPU of widths ' \
                   f'U{{400..2400}} step 8 in 200-px buckets, batches of <= {args.batch} padded to the bucket width, forward + CTC greedy'
    else:
        if fix and rank == 0 and args.batch == len(fix['lines']) and args.width == 1200:
            lines = fix['lines']
        else:
            lines = [synth.make_text_lines(1, hp.height, args.width, seed=7000 + 100 * rank + i, alphabet_seed=fix['meta']['seed'] if fix else 1)[0][0, 0]
                     for i in range(args.batch)]
        widths = [args.width] * args.batch
        plan = [(args.width, list(range(args.batch)))]
        workload = f'{args.config}: conformer D={hp.encoder_dim} L={hp.num_encoder_layers} heads={hp.num_attention_heads} ' \
                   f'sub_ch={hp.subsampling_conv_channels}, batch {args.batch} x 96x{args.width} per GPU, forward + CTC greedy'
    batches = []
    for bw, idx in plan:
        im, lens = collate(lines, idx, bw)
        batches.append({'x': im[:, 0].contiguous().to(dev), 'lens': lens.numpy().astype(np.int32), 'idx': idx, 'w': bw, 'n': len(idx)})
    if args.profile_batch_only:
        batches = [max(batches, key=lambda b: b['n'] * b['w'])]
    NB = len(batches)
    lines_per_cycle = sum(b['n'] for b in batches)
    max_n, max_w = max(b['n'] for b in batches), max(b['w'] for b in batches)
    for e in engines:
        e.reserve(max_n, max_w)
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    outs = [[torch.empty((b['n'], eng.out_len(b['w']), hp.num_classes), dtype=torch.float32, device=dev) for b in batches] for _ in range(S)]
    use_graph = not args.no_graph and NB <= 4            # the library keeps 16 captured graphs per model (main loop + 2 staging buffers each)
    for e in engines:
        e.set_graph(use_graph)

    _times = []
    def run_steps(n, nstreams, first=0, depth=None):
        """n steps; step i runs batch (first + i) % NB on stream i % nstreams (forward + greedy decode; the label records land in
        pinned host memory) and is collected nstreams steps later, so nstreams independent batches overlap on the GPU.
        Returns (lines processed, records of the last collected step)."""
        pending, recs, done = [], None, 0
        for i in range(n):
            k, b = i % nstreams, (first + i) % NB
            e = engines[(i % S) if os.environ.get('COCR_BENCH_ALT') else k]      # dev: COCR_BENCH_ALT=1 cycles the packed model copies on however few streams
            with torch.cuda.stream(streams[k]):
                logits, out_lens = e.forward(batches[b]['x'], batches[b]['lens'], out=outs[k][b])
                pending.append((e, e.ctc_greedy_async(logits, out_lens), b))
            if i < nstreams - 1 and args.stagger_us > 0:
                t_end = time.perf_counter() + args.stagger_us * 1e-6
                while time.perf_counter() < t_end:
                    pass
            if len(pending) >= nstreams * (depth or args.queue_depth):
                pe, h, pb = pending.pop(0)
                recs = pe.collect(h)
                _times.append(time.perf_counter())
                done += batches[pb]['n']
        for pe, h, pb in pending:
            recs = pe.collect(h)
            _times.append(time.perf_counter())
            done += batches[pb]['n']
        return done, recs

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # pre-warm: every (engine, batch) pair is called three times -- plain (one-time attribute / workspace work), captured into a
    # hipGraph, replayed -- so that no capture or instantiation lands in the timed region whatever --steps / --warmup are
    for k in range(S):
        with torch.cuda.stream(streams[k]):
            for b in range(NB):
                for _ in range(3):
                    lg, ol = engines[k].forward(batches[b]['x'], batches[b]['lens'], out=outs[k][b])
                    engines[k].collect(engines[k].ctc_greedy_async(lg, ol))
    torch.cuda.synchronize(dev)
    # ... and the chip is brought to its steady state (clocks, caches, the runtime's launch pipeline) by `--prewarm-ms` of the same
    # loop before the W warm-up steps: with --steps 20 the first rounds after an idle chip ran 10 % slower than the steady ones
    # (DESIGN.md "The timed region's length"), which says nothing about the kernels
    t_pw = time.perf_counter()
    while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
        run_steps(4 * S, S)

    run_steps(args.warmup, S)
    fence()
    t0 = time.perf_counter()
    done, recs = run_steps(args.steps, S)
    fence()
    dt = time.perf_counter() - t0
    if os.environ.get('COCR_BENCH_TIMES'):
        sys.stderr.write('completion times (ms after t0): ' + ' '.join(f'{(x - t0) * 1e3:.2f}' for x in _times[-args.steps:]) + f' | end {dt * 1e3:.2f}\n')
    if use_dist:
        t = torch.tensor([dt, float(done)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, lines_total = float(tmax[0].item()), float(t[1].item())
    else:
        lines_total = float(done)
    value = lines_total / dt

    extra = {}
    if rank == 0 and not args.no_extra_legs:
        # ---- one batch in flight (a caller without streams of its own): 48-row workgroups of the row-chain kernels (every CU gets one
        # at 32 x 300 frames; what PytorchRecognitionModel sets by default), and the throughput form (96 rows) for comparison
        leg_steps = max(args.steps, 100)             # the side legs are not the timed region of `value`: long enough to be stable at --steps 20
        for rows, key in ((48, 'value_streams1'), (0, 'value_streams1_rows96')):
            eng.set_chain_rows(rows)
            torch.cuda.synchronize(dev)
            run_steps(max(3 * NB, 6), 1, depth=1)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            d1, _ = run_steps(leg_steps, 1, depth=1)
            torch.cuda.synchronize(dev)
            extra[key] = round(d1 / (time.perf_counter() - t1), 2)
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                for b in range(NB):
                    for _ in range(3):              # (the graphs were dropped by the switch: capture them again for the legs below)
                        lg, ol = engines[k].forward(batches[b]['x'], batches[b]['lens'], out=outs[k][b])
                        engines[k].collect(engines[k].ctc_greedy_async(lg, ol))
        torch.cuda.synchronize(dev)

        # ---- the drop-in surface: PytorchRecognitionModel.predict_string, a fresh input tensor per call, no graph replay
        from conformer_ocr_amd.codec import ascii_codec
        from conformer_ocr_amd.pred import PytorchRecognitionModel
        net = PytorchRecognitionModel(**hp.as_dict(), input_dropout_p=0.1, feed_forward_dropout_p=0.1, attention_dropout_p=0.1,
                                      conv_dropout_p=0.1, codec=ascii_codec(hp.num_classes), compute_dtype=args.dtype)
        net.nn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        net = net.to(dev).eval()
        b0 = batches[0]
        lens_t = torch.from_numpy(b0['lens'].astype(np.int64))
        net.predict_string(b0['x'].unsqueeze(1).clone(), lens_t)
        torch.cuda.synchronize(dev)
        reps = 100
        t2 = time.perf_counter()
        for _ in range(reps):
            strs = net.predict_string(b0['x'].unsqueeze(1).clone(), lens_t)       # .clone(): a new device buffer per call, like a loader's batch
        torch.cuda.synchronize(dev)
        extra['predict_string'] = {'value': round(reps * b0['n'] / (time.perf_counter() - t2), 2), 'unit': 'lines/s', 'calls': reps,
                                   'what': 'net.predict_string(fresh (N,1,H,W) device tensor, lens) one call at a time: forward + greedy decode + '
                                           'read-back + codec, no hipGraph, no caller-side streams'}
        del net

        # ---- ingest: u8 lines from pinned host memory, double-buffered host->device copy inside the loop (PCIe-inclusive; never `value`)
        hosts = [torch.from_numpy(np.rint(b['x'].cpu().numpy() * 255.0).astype(np.uint8)).pin_memory() for b in batches]
        stage = [[torch.empty_like(hosts[b], device=dev) for b in range(NB)] for _ in range(2 * S)]
        copy_streams = [torch.cuda.Stream(dev) for _ in range(S)]

        def ingest_steps(n):
            pending, cnt = [], 0
            for i in range(n):
                k, b, slot = i % S, i % NB, i % (2 * S)
                with torch.cuda.stream(copy_streams[k]):
                    stage[slot][b].copy_(hosts[b], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_streams[k])
                with torch.cuda.stream(streams[k]):
                    streams[k].wait_event(ev)
                    lg, ol = engines[k].forward(stage[slot][b], batches[b]['lens'], out=outs[k][b])
                    pending.append((engines[k], engines[k].ctc_greedy_async(lg, ol), b))
                if len(pending) >= S:
                    pe, h, pb = pending.pop(0)
                    pe.collect(h)
                    cnt += batches[pb]['n']
            for pe, h, pb in pending:
                pe.collect(h)
                cnt += batches[pb]['n']
            return cnt
        ingest_steps(8 * S * min(NB, 3))
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        d3 = ingest_steps(leg_steps)
        torch.cuda.synchronize(dev)
        extra['ingest'] = {'value': round(d3 / (time.perf_counter() - t3), 2), 'unit': 'lines/s',
                           'what': f'u8 (N,H,W) batches copied from pinned host memory inside the loop ({S} copy streams, 2 staging buffers per '
                                   'stream), forward ingests u8 directly'}

    # ---- small-batch latency (BASELINE configs[4]; reference call site pred.py:148-164): B in {1, 4, 8} lines of the batch, forward + CTC decode
    # (greedy, beam 16) + read-back of the label records, ONE synchronous call at a time on a replayed hipGraph; p50 / p99 per call.
    # Bounded: <= 0.4 s per (B, decoder) pair
    latency = None
    if rank == 0 and args.config != 'cfg4' and not args.no_extra_legs and not args.no_latency_leg:
        latency = {'what': 'forward + CTC decode + read-back of the label records, one synchronous call at a time (host wall-clock per call, ms); '
                           f'lines of 96x{args.width} resident in HBM, hipGraph replay, beam = 16 (softmax inside the decoder)', 'unit': 'ms per call'}
        b0 = batches[0]
        eng.set_graph(True)
        for B in (1, 4, 8):
            if B > b0['n']:
                continue
            xb, lb = b0['x'][:B].contiguous(), b0['lens'][:B].copy()
            buf = torch.empty((B, eng.out_len(b0['w']), hp.num_classes), dtype=torch.float32, device=dev)
            for mode in ('greedy', 'beam16'):
                def call():
                    lg, ol = eng.forward(xb, lb, out=buf)
                    return eng.ctc_beam(lg, ol, 16) if mode == 'beam16' else eng.ctc_greedy(lg, ol)
                for _ in range(8):                       # plain call, capture, replays
                    call()
                ts, t_leg = [], time.perf_counter()
                while len(ts) < 200 and time.perf_counter() - t_leg < 0.4:
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    r_ = call()
                    ts.append((time.perf_counter() - t1) * 1e3)
                ts = np.sort(np.array(ts))
                latency[f'B{B}_{mode}'] = {'p50': round(float(ts[len(ts) // 2]), 4), 'p99': round(float(ts[min(len(ts) - 1, int(np.ceil(len(ts) * 0.99)) - 1)]), 4),
                                           'p50_per_line': round(float(ts[len(ts) // 2]) / B, 4), 'calls': int(len(ts)), 'labels_emitted': int(sum(len(x) for x in r_))}
        eng.set_graph(use_graph)

    # ---- CER (outside the timed region): the compute dtype's greedy strings on the fixture against the reference's fp32 greedy
    # strings of the same padded batches (tests/golden) and against the ground-truth text
    cer = None
    if rank == 0 and fix:
        fb = [(max(fix['meta']['widths']), list(range(len(fix['lines']))))] if not fix['meta']['edge'] else \
            make_batches(fix['meta']['widths'], fix['meta']['batch_size'], fix['meta']['edge'])
        got = {}
        for bw, idx in fb:
            im, lens = collate(fix['lines'], idx, bw)
            lg, ol = eng.forward(im[:, 0].contiguous().to(dev), lens.numpy())
            for i, r in zip(idx, strings_of(eng.ctc_greedy(lg, ol))):
                got[i] = r
        n = len(fix['lines'])
        c_ref, c_truth = ErrorRate(), ErrorRate()
        c_ref.update([got[i] for i in range(n)], fix['ref_strings'])
        c_truth.update([got[i] for i in range(n)], fix['texts'])
        cer = {'cer_vs_reference': c_ref.compute(), 'cer_vs_truth': c_truth.compute(), 'lines': n, 'characters': c_truth.total,
               'lines_identical_to_reference': sum(got[i] == fix['ref_strings'][i] for i in range(n)),
               'fixture': f"tests/golden/{args.config}_text.npz (reference fp32 greedy strings of the same padded batches)"}

    # ---- roofline of the dominant kernel: HIP events around every launch, on the forward's stream
    roof, kernels = None, {}
    if rank == 0 and args.profile_steps > 0:
        pb = max(range(NB), key=lambda b: batches[b]['n'] * batches[b]['w'])          # the heaviest batch of the queue
        torch.cuda.synchronize(dev)
        eng.profile(True)
        for _ in range(args.profile_steps):
            lg, ol = eng.forward(batches[pb]['x'], batches[pb]['lens'])
            eng.ctc_greedy(lg, ol)
        prof = eng.profile_read()
        eng.profile(False)
        table = kernel_table(hp, batches[pb]['n'], batches[pb]['w'], args.dtype)
        # an empty event pair (one per forward) measures what a bracket costs by itself; it is subtracted from every average so
        # that these numbers are dispatch durations, comparable with `rocprofv3 --kernel-trace --stats` (profiles/)
        ovh = prof.pop('event_pair_overhead', (0.0, 0))[0]
        prof = {k: (max(ms - ovh, 1e-6), cnt) for k, (ms, cnt) in prof.items()}
        total_ms = sum(ms * cnt for ms, cnt in prof.values())
        for name, (ms, cnt) in prof.items():
            bound, work = table.get(name, ('hbm', 0.0))
            rec = {'avg_ms': round(ms, 5), 'launches_per_step': cnt // args.profile_steps, 'share': round(ms * cnt / total_ms, 4)}
            if bound == 'hbm':
                rec.update(bound='hbm', achieved=round(work / (ms * 1e-3) / 1e9, 2), peak=HBM_PEAK_GBS, unit='GB/s')
            else:
                rec.update(bound=bound, achieved=round(work / (ms * 1e-3) / 1e12, 3), unit='TFLOP/s',
                           peak=MFMA_PEAK_TFLOPS[args.dtype] if bound == 'mfma' else 157.3)
            rec['frac'] = round(rec['achieved'] / rec['peak'], 5)
            kernels[name] = rec
        dom = max(kernels, key=lambda k: kernels[k]['share'])
        d = kernels[dom]
        rows = batches[pb]['n'] * eng.out_len(batches[pb]['w'])
        traffic, tinfo = pmc_traffic(dom, d['avg_ms'], hp, rows, '' if args.config == 'cfg2' else '_' + args.config)
        roof = {'kernel': dom, 'bound': 'hbm' if d['bound'] == 'hbm' else 'mfma', 'achieved': d['achieved'], 'peak': d['peak'],
                'unit': d['unit'], 'frac': d['frac'], 'traffic': traffic, 'traffic_profile': tinfo, 'avg_ms': d['avg_ms'], 'share_of_step': d['share'],
                'algorithmic_per_launch': table[dom][1], 'event_pair_overhead_ms': round(ovh, 5),
                'profiled_batch': {'lines': batches[pb]['n'], 'width': batches[pb]['w']},
                # the row-block chain kernels launch ceil(rows / 96) workgroups of one per CU: `frac` prices them against the WHOLE chip
                'cus_occupied': cus_occupied(dom, hp, rows), 'frac_of_occupied_cus': round(d['frac'] * 256.0 / cus_occupied(dom, hp, rows), 5),
                'traffic_source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over this command, 2*FETCH+WRITE bytes per launch '
                                  '(newest profiles/*_pmc_traffic.csv; `traffic_profile.stale` compares that round\'s kernel duration with this run\'s)'}

        # the same kernel in its 48-row form (what one batch in flight runs: twice the workgroups, every weight byte streamed twice):
        # a shorter launch on more CUs at more CU-time; `value` runs the 96-row form
        if dom.startswith('chain') and not args.no_extra_legs:
            eng.set_chain_rows(48)
            torch.cuda.synchronize(dev)
            for _ in range(3):
                eng.forward(batches[pb]['x'], batches[pb]['lens'])
            eng.profile(True)
            for _ in range(args.profile_steps):
                lg, ol = eng.forward(batches[pb]['x'], batches[pb]['lens'])
                eng.ctc_greedy(lg, ol)
            p48 = eng.profile_read()
            eng.profile(False)
            eng.set_chain_rows(0)
            if dom in p48:
                ms48 = max(p48[dom][0] - p48.get('event_pair_overhead', (0.0, 0))[0], 1e-6)
                ach = table[dom][1] / (ms48 * 1e-3) / 1e12
                roof['rows48'] = {'avg_ms': round(ms48, 5), 'achieved': round(ach, 3), 'frac': round(ach / d['peak'], 5),
                                  'cus_occupied': min(256, 2 * cus_occupied(dom, hp, rows))}

    # ---- the other single-GPU workloads of BASELINE.json's `configs` (never `value`): configs[3]'s wide model on a mixed-width bucketed
    # queue and the reference's default model (default_specs.py:48-61) on the metric's batch shape -- each a bounded run of THIS script
    # in a fresh child process (its own model, its own prewarm), reduced to value / CER / the dominant kernel's roofline fraction
    others = None
    if rank == 0 and world == 1 and args.config == 'cfg2' and not args.no_extra_legs and not args.no_other_configs:
        import subprocess
        for e in engines:
            e.set_graph(False)
        torch.cuda.synchronize(dev)
        others = {}
        for cfg, extra_args in (('cfg4', ['--steps', '50', '--warmup', '5', '--lines', '320']), ('cfg1', ['--steps', '100', '--warmup', '5'])):
            cmd = [sys.executable, os.path.abspath(__file__), '--config', cfg, '--gpus', '1', '--no-cpu-baseline', '--no-extra-legs', '--no-other-configs',
                   '--profile-steps', '2', '--prewarm-ms', '200', '--dtype', args.dtype] + extra_args
            t_c = time.perf_counter()
            try:
                env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=150, env=env)
                ln = [x for x in r.stdout.splitlines() if x.startswith('{')]
                if r.returncode != 0 or not ln:
                    others[cfg] = {'error': f'rc {r.returncode}: ' + r.stderr[-300:]}
                    continue
                j = json.loads(ln[-1])
                rf = j.get('roofline') or {}
                others[cfg] = {'value': j['value'], 'unit': 'lines/s', 'steps': j['steps'], 'ms_per_step': j['ms_per_step'], 'workload': j['config']['workload'],
                               'lines_per_step': j['config']['lines_per_step_per_gpu'], 'streams_per_gpu': j['config']['streams_per_gpu'],
                               'gflop_per_line': j['config']['gflop_per_line'], 'gflop_per_line_padded': j['config']['gflop_per_line_padded'],
                               'achieved_tflops_whole_path_padded': j['achieved_tflops_whole_path_padded'],
                               'cer_vs_reference': j['cer_vs_reference'], 'lines_identical_to_reference': (j.get('cer') or {}).get('lines_identical_to_reference'),
                               'fixture_lines': (j.get('cer') or {}).get('lines'),
                               'roofline': {k: rf.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_profile', 'avg_ms', 'profiled_batch', 'cus_occupied', 'frac_of_occupied_cus')},
                               'wall_s': round(time.perf_counter() - t_c, 1)}
            except subprocess.TimeoutExpired:
                others[cfg] = {'error': 'no line within 150 s'}

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:        # rank 0 at ANY world size: the host cores' rate beside every line (the other ranks wait at the last barrier)
        cpu = cpu_baseline(hp, state, max_w if args.config != 'cfg4' else 1400, sample_lines=min(32, args.batch) if args.config != 'cfg4' else 8)

    if rank == 0:
        per_line = [flops_per_line(hp, w) for w in widths]
        padded = sum(flops_per_line(hp, b['w']) * b['n'] for b in batches) / lines_per_cycle
        gflop = float(np.mean(per_line)) / 1e9
        out = {
            'metric': METRIC, 'value': round(value, 2), 'unit': 'lines/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': workload, 'lines_per_step_per_gpu': round(lines_per_cycle / NB, 2), 'batches_in_queue': NB,
                       'gflop_per_line': round(gflop, 3), 'gflop_per_line_padded': round(padded / 1e9, 3),
                       'parallelism': f'{world} independent rank(s), weights by one RCCL broadcast', 'streams_per_gpu': S, 'steps_queued_per_stream': args.queue_depth,
                       'hipgraph_replay': bool(use_graph), 'prewarm_ms': args.prewarm_ms, 'rccl_ranks_seen': ranks_seen if use_dist else None,
                       'launched_by': 'bench.py launch_ranks' if os.environ.get('LOCAL_WORLD_SIZE') and 'TORCHELASTIC_RUN_ID' not in os.environ and use_dist
                                      else ('torch.distributed.run' if use_dist else 'in-process'),
                       'gpu_max_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'), 'gpu_max_hw_queues_preset_by_caller': _HWQ_PRESET is not None},
            'achieved_tflops_whole_path': round(value * gflop / 1e3, 2),
            'achieved_tflops_whole_path_padded': round(value * padded / 1e12, 2),
            'cer_vs_reference': cer['cer_vs_reference'] if cer else None, 'cer': cer,
            **extra,
            'latency': latency,
            'other_configs': others,
            'roofline': roof, 'cpu_baseline': cpu, 'kernels': kernels,
            'labels_emitted_last_step': int(sum(len(r) for r in recs)),
        }
        line = (json.dumps(out) + '\n').encode()
        while line:
            line = line[os.write(json_fd, line):]
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
