"""`python bench.py --gpus N` starts its own rank processes (bench.py launch_ranks; SURVEY.md section 8e: one process per GPU, the
reference itself selects a single device, cli/util.py:56-64).

CPU: the launcher is driven with `--dry-run` workers (gloo, world size 2): rank 0's single line is forwarded and nothing else, a
failing rank's exit code comes back and the surviving rank is ended, RANK in the environment bypasses the launcher.
GPU: `bench.py --gpus 1 --spawn` forces the child path on the one GPU of the box: a parseable line from a fresh rank process under
an RCCL group, `value` within 5 % of the in-process run of the same command."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'LOCAL_WORLD_SIZE')}
    env['MASTER_ADDR'] = '127.0.0.1'
    return env


def _run(args, timeout=240, env=None):
    return subprocess.run([sys.executable, BENCH] + args, env=env or _env(), capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(300)
def test_launcher_forwards_rank0_line_only():
    r = _run(['--gpus', '2', '--dry-run'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # rank 1's stdout text went to stderr, gloo's banner too
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['dry_run'] is True and out['value'] is None
    assert out['config']['ranks_seen'] == 2                # the all-reduce of ones over the group the launcher's environment set up
    assert 'must not reach' in r.stderr


@pytest.mark.timeout(300)
def test_launcher_propagates_a_failing_rank():
    r = _run(['--gpus', '2', '--dry-run', '--dry-run-fail-rank', '1'])
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert r.stdout.strip() == ''                          # no line for a job that did not complete on every rank
    assert 'rank 1 exited with code 3' in r.stderr
    r = _run(['--gpus', '2', '--dry-run', '--dry-run-fail-rank', '0'])
    assert r.returncode == 3 and r.stdout.strip() == ''


@pytest.mark.timeout(300)
def test_spawn_at_world_one_and_launcher_bypass_under_a_launcher():
    r = _run(['--gpus', '1', '--spawn', '--dry-run'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout)['config']['ranks_seen'] == 1
    # RANK set by an outer launcher (torch.distributed.run): this process IS a rank; a world-size mismatch is refused, not relaunched
    env = dict(_env(), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_PORT='29533')
    r = _run(['--gpus', '2', '--dry-run'], env=env)
    assert r.returncode == 2 and r.stdout.strip() == ''


@pytest.mark.timeout(300)
def test_launcher_times_out_on_a_stuck_rank():
    """A rank that never reaches the rendezvous (what a rank stuck in RCCL init looks like): after --launch-timeout the launcher names
    the ranks that are still running, ends them by PID and returns 124 -- no line, no endless wait."""
    import time
    t0 = time.time()
    r = _run(['--gpus', '2', '--dry-run', '--dry-run-hang-rank', '1', '--launch-timeout', '8'], timeout=120)
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert r.stdout.strip() == ''
    assert 'still running after 8 s' in r.stderr and '[0, 1]' in r.stderr      # rank 0 waits for rank 1 in the rendezvous: both are named
    assert time.time() - t0 < 90


@pytest.mark.timeout(300)
def test_launcher_drains_a_long_rank0_line():
    """Rank 0's line may be longer than a pipe buffer (64 KiB): the launcher reads it while the ranks run."""
    env = dict(_env(), COCR_BENCH_DRYRUN_PAD='200000')
    r = _run(['--gpus', '2', '--dry-run'], env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) > 200000
    assert json.loads(lines[0])['config']['ranks_seen'] == 2


@pytest.mark.timeout(120)
def test_launcher_refuses_more_gpus_than_visible():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(['--gpus', str(max(n, 2)), '--steps', '1', '--warmup', '0'])
    assert r.returncode == 2 and r.stdout.strip() == ''
    assert 'GPU(s) are visible' in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_forced_spawn_on_the_gpu_matches_the_in_process_run():
    args = ['--gpus', '1', '--steps', '100', '--warmup', '10', '--no-cpu-baseline', '--no-extra-legs', '--profile-steps', '0']
    a = _run(args, timeout=420)
    assert a.returncode == 0, a.stderr[-3000:]
    b = _run(args + ['--spawn'], timeout=420)
    assert b.returncode == 0, b.stderr[-3000:]
    la, lb = [ln for ln in a.stdout.splitlines() if ln.strip()], [ln for ln in b.stdout.splitlines() if ln.strip()]
    assert len(la) == 1 and len(lb) == 1
    ja, jb = json.loads(la[0]), json.loads(lb[0])
    assert jb['n_gpus'] == 1 and jb['config']['rccl_ranks_seen'] == 1 and jb['config']['launched_by'] == 'bench.py launch_ranks'
    assert ja['config']['launched_by'] == 'in-process'
    assert jb['cer_vs_reference'] == 0.0 and ja['cer_vs_reference'] == 0.0
    print('in-process', ja['value'], 'spawned', jb['value'])
    assert abs(jb['value'] - ja['value']) <= 0.05 * ja['value'], (ja['value'], jb['value'])
