#!/usr/bin/env python3
"""
bench.py -- throughput of the recognition hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: 32 synthetic 96x1200 lines (BASELINE.json
configs[1], the configuration the metric is quoted on) -> conformer encoder (D=256, 12 blocks, bf16
operands) -> decoder -> CTC greedy labels, through the C ABI of libcocr_hip.so.  Inputs are resident
in HBM when the timed region starts.  With N > 1 every rank owns one GPU, receives the packed weights
by one RCCL broadcast and processes its own independent batches (no data-path collective): weak scaling.

Rank 0 prints ONE JSON line: metric/value (whole-job lines/s), the roofline of the dominant kernel
(measured live with HIP events on the forward's stream), and the CPU baseline (the oracle on the host
cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

# One hardware queue per batch in flight: the HIP runtime multiplexes its streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues,
# and two of this bench's streams sharing a queue serialise "independent" batches (measured: 4 streams on the default 4 queues --
# one is taken by the null stream -- 30.2k lines/s, on 8 queues 38.0k; 3 streams on 4 queues 36.6k).  Read at runtime start-up.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from conformer_ocr_amd import synth  # noqa: E402
from conformer_ocr_amd.engine import HipRecognizer  # noqa: E402
from conformer_ocr_amd.spec import flops_per_line, out_len  # noqa: E402

MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}     # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def kernel_table(hp, n, w, dtype):
    """Algorithmic work per launch of every kernel family for a batch of n lines of width w:
    (bound, flops or bytes per launch).  GEMMs: 2*M*N*K flop.  HBM-bound kernels: one read + one write
    of their operand."""
    es = 2 if dtype == 'bf16' else 4
    D, C, ff, h = hp.encoder_dim, hp.subsampling_conv_channels, hp.feed_forward_expansion_factor * hp.encoder_dim, hp.num_attention_heads
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
        'chain_ffn_qkv': ('mfma', 4.0 * M * D * ff + 6.0 * M * D * D),
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


def cus_occupied(family, rows):
    """CUs a launch of `family` can occupy (MI355X: 256): the 96-row chain kernels run one workgroup per CU."""
    if family.startswith('chain_') and rows >= 4800:
        return min(256, -(-rows // 96))
    return 256


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC summary (collected offline with
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` on this bench command; tools/summarize_rocprof.py), or None."""
    import csv
    import glob
    names = {'ffn_fused': 'ffn_fused_kernel', 'attention': 'relpos_attention_kernel', 'dwconv': 'dwconv_bn_silu', 'frontend_conv12': 'frontend_conv12',
             'frontend_fused': 'frontend96_kernel',
             'chain_pw2_ffn_ffn_qkv': 'chain96_kernel<6, 31, 0, 1, 1, 3>', 'chain_attn_out_glu': 'chain96_kernel<6, 0, 0, 2, -1, -1>',
             'chain_ffn_qkv': 'chain96_kernel<6, 0, 1, 3, -1, -1>', 'chain_pw2_ffn': 'chain96_kernel<6, 31, 0, 1, -1, -1>'}
    if family not in names:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.csv')))
    if not files:
        return None
    for r in csv.DictReader(open(files[-1])):
        if names[family] in r['kernel']:
            return int(r['hbm_bytes_per_launch(2*fetch+write)'])
    return None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--width', type=int, default=1200)
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--profile-steps', type=int, default=3)
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel from the host instead of replaying a captured hipGraph')
    ap.add_argument('--stagger-us', type=float, default=0.0, help='host-side offset between the first batches of a run (multi-stream only)')
    ap.add_argument('--streams', type=int, default=4, help='independent batches in flight (one packed model + HIP stream each)')
    args = ap.parse_args()
    # stdout carries exactly ONE JSON line: libraries that print banners to fd 1 (RCCL at communicator creation) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_dist = 'RANK' in os.environ          # launched by torch.distributed.run (any world size, also 1)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    hp = synth.hparams(args.config)
    S = max(1, args.streams)
    engines = [HipRecognizer(hp, dev, args.dtype) for _ in range(S)]
    eng = engines[0]
    state = None
    if rank == 0:
        state = synth.make_state_dict(hp, seed=1236, decoder_gain=8.0)
        for e in engines:
            e.load_state(state)
            e.finalize()
    else:
        for e in engines:
            e.finalize_empty()
    if use_dist:
        from conformer_ocr_amd.dist import broadcast_weights
        for e in engines:
            broadcast_weights(e, src=0)

    # per-rank independent synthetic batches, resident in HBM (float32 (N,H,W), what the reference's loader hands over)
    img, lens = synth.make_lines(args.batch, hp.height, args.width, seed=1000 + rank)
    x = torch.from_numpy(img[:, 0]).to(dev)
    lens32 = lens.astype(np.int32)
    for e in engines:
        e.reserve(args.batch, args.width)
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    T_out = eng.out_len(args.width)
    outs = [torch.empty((args.batch, T_out, hp.num_classes), dtype=torch.float32, device=dev) for _ in range(S)]
    for e in engines:
        e.set_graph(not args.no_graph)

    stagger_s = args.stagger_us * 1e-6 if S > 1 else 0.0

    def run_steps(n):
        """n steps; step i is enqueued on stream i % S (forward + greedy decode + D2H of the label records) and its
        records are collected S steps later, so S independent batches overlap on the GPU."""
        pending, recs = [], None
        for i in range(n):
            e = engines[i % S]
            with torch.cuda.stream(streams[i % S]):
                logits, out_lens = e.forward(x, lens32, out=outs[i % S])
                pending.append((e, e.ctc_greedy_async(logits, out_lens)))
            if i < S - 1 and stagger_s > 0:
                # the first S batches would start in lockstep (all frontends, then all attention kernels, ... at the same time:
                # measured 2x the steady-state time for that first group); offset them by a fraction of a step like the steady state
                t_end = time.perf_counter() + stagger_s
                while time.perf_counter() < t_end:
                    pass
            if len(pending) >= S:
                pe, h = pending.pop(0)
                recs = pe.collect(h)
        for pe, h in pending:
            recs = pe.collect(h)
        return recs

    def step():
        logits, out_lens = eng.forward(x, lens32)
        return eng.ctc_greedy(logits, out_lens)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    recs = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    lines_total = args.batch * args.steps * world
    value = lines_total / dt

    # ---- roofline of the dominant kernel: HIP events around every launch, on the forward's stream
    roof, kernels = None, {}
    if rank == 0 and args.profile_steps > 0:
        torch.cuda.synchronize(dev)
        eng.profile(True)
        for _ in range(args.profile_steps):
            step()
        prof = eng.profile_read()
        eng.profile(False)
        table = kernel_table(hp, args.batch, args.width, args.dtype)
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
        roof = {'kernel': dom, 'bound': 'hbm' if d['bound'] == 'hbm' else 'mfma', 'achieved': d['achieved'], 'peak': d['peak'],
                'unit': d['unit'], 'frac': d['frac'], 'traffic': pmc_traffic(dom), 'avg_ms': d['avg_ms'], 'share_of_step': d['share'],
                'algorithmic_per_launch': table[dom][1], 'event_pair_overhead_ms': round(ovh, 5),
                # the 96-row chain kernels launch ceil(rows / 96) workgroups of one per CU: `frac` prices them against the WHOLE chip
                'cus_occupied': cus_occupied(dom, args.batch * T_out), 'frac_of_occupied_cus': round(d['frac'] * 256.0 / cus_occupied(dom, args.batch * T_out), 5),
                'traffic_source': 'profiles/*_pmc_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
                'separate passes, 2*FETCH+WRITE bytes per launch)'}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(hp, state, args.width, sample_lines=args.batch)

    if rank == 0:
        gflop = flops_per_line(hp, args.width) / 1e9
        out = {
            'metric': 'text lines/sec (whole node), 96x1200 bf16 batch', 'value': round(value, 2), 'unit': 'lines/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{args.config}: conformer D={hp.encoder_dim} L={hp.num_encoder_layers} heads={hp.num_attention_heads} '
                                   f'sub_ch={hp.subsampling_conv_channels}, batch {args.batch} x 96x{args.width} per GPU, '
                                   f'forward + CTC greedy', 'lines_per_step_per_gpu': args.batch, 'gflop_per_line': round(gflop, 3),
                       'parallelism': f'{world} independent rank(s), weights by one RCCL broadcast', 'streams_per_gpu': S},
            'achieved_tflops_whole_path': round(value * gflop / 1e3, 2),
            'roofline': roof, 'cpu_baseline': cpu, 'kernels': kernels,
            'labels_emitted_last_step': int(sum(len(r) for r in recs)),
        }
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
