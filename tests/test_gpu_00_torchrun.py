"""bench.py's data-parallel branch end to end, launched the way the driver launches it (python -m torch.distributed.run, one
rank per GPU), on the one GPU of the test box: --force-dp makes a world of one take the world > 1 code path (rank-sliced
draws, Engine.dp_train_step with its two overlapped bucket all-reduces on a real RCCL communicator, barrier, MAX-reduced time).
Runs first (file name) and as a fresh child process."""
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


@pytest.mark.parametrize('extra', [[], ['--dp-backend', 'torch'], ['--workload', 'config5', '--frames', '192', '--precision', 'bf16'],
                                   ['--model', 'G6', '--batch', '32', '--frames', '192', '--precision', 'bf16'],
                                   ['--global-batch', '32', '--precision', 'bf16']],
                         ids=['headline_native_rccl', 'headline_torch_distributed', 'config5_buckets_bf16', 'config4_g6_bf16', 'config3_strong_scaling_shape'])
def test_bench_dp_path_under_torchrun(extra):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dp', '--steps', '4', '--warmup', '2',
           '--no-cpu-baseline', '--no-extras'] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 1 and out['unit'] == 'utterances/s' and out['value'] > 0
    assert 'forced DP path' in out['config']['parallelism']
    assert out['roofline'] and 0 < out['roofline']['frac'] < 1
    assert out['recurrence']['launches_per_step'] in (4.0, 6.0)
    assert out['scaling'] == ('strong' if '--global-batch' in extra else 'weak')
    if '--global-batch' in extra:
        assert out['config']['global_batch'] == 32
    assert out['dp_replicas'] == {'identical': True, 'ranks': 1, 'weights_bit_sum': out['dp_replicas']['weights_bit_sum']}
    if '--dp-backend' not in extra:
        # the native RCCL path reports where each bucket's collective sat relative to the end of the backward (ss_dp_profile)
        dc = out['dp_collectives']
        assert dc and len(dc['buckets']) >= 3 and dc['buckets'][-1]['arena_offset'] == -1
        assert all(b['end_us'] >= b['start_us'] for b in dc['buckets'])
        assert sum(b['mbytes'] for b in dc['buckets']) > 10          # the whole gradient arena went through collectives (13.9 / 77.8 MB)
