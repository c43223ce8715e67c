"""GPU: the multi-GPU start-up path on the hardware that is there.  One GPU per box here, so the job is a world-size-1 "nccl" (RCCL)
process group in a FRESH child process (nothing touches the GPU before init_process_group): finalize_empty -> ONE dist.broadcast of
the packed blob -> import -> positional tables derived on the device -> forward, bit-equal to a directly finalised model.  (The N > 1
collectives are covered on the CPU with gloo, tests/test_dist_cpu.py; no scaling curve has been measured: DESIGN.md section 5.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_nccl_broadcast_start_up_in_a_fresh_process():
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dist_child.py')], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['world'] == 1 and out['shard'] == [0, 1, 2, 3, 4]
    for dtype in ('bf16', 'fp32', 'padded'):
        o = out[dtype]
        assert o['finite'] and o['same_on_all_ranks'] and o['bit_equal_to_direct_finalize'], o
    assert out['train']['loss1'] < out['train']['loss0'] and out['train']['grad_floats'] > 10000      # all-reduced flat gradient, AdamW
    # the blob carries weights only: the positional tables (9999 x D per block) are derived per device
    assert out['bf16']['blob_bytes'] < 12e6, out['bf16']
