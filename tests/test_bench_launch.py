"""bench.py starts its own ranks: `python bench.py --gpus N` without a launcher must form the N-rank job the
reference's tools/dist_train.sh:22-24 forms (one process per GPU), before anything touches the GPU, and fail CLEANLY when the
node has fewer GPUs.  GPU: the whole N = 2 flow on one GPU (ranks share GPU 0 over gloo: SyncBN exchanges, gradient
all-reduce and the schedule A/B child run through the real kernels)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_rank_command_and_flags():
    import bench
    cmd = bench.rank_command(['--gpus', '8', '--steps', '20', '--warmup', '5'], 8, 29511)
    assert cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']
    assert bench.strip_flag(['--a', '--dump-kernels', 'x.json', '--b'], '--dump-kernels', True) == ['--a', '--b']
    assert bench.strip_flag(['--a', '--dump-kernels=x.json', '--b'], '--dump-kernels', True) == ['--a', '--b']
    assert bench.strip_flag(['--a', '--no-graph'], '--no-graph') == ['--a']
    args = bench.parse(['--gpus', '4'])
    assert args.gpus == 4 and args.backend == 'nccl' and not args.sync_ab_child


def test_too_few_gpus_fails_cleanly():
    """No GPU here (and one on the GPU box): `--gpus 2` must say so and exit non-zero - not assert, not hang, not spawn."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('node has two GPUs')
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 3, (p.returncode, p.stderr[-500:])
    assert 'needs 2 GPUs' in p.stderr and 'Traceback' not in p.stderr and 'AssertionError' not in p.stderr
    assert p.stdout.strip() == ''


STUB_WORKER = r'''
import json, os, sys, time
rank, att, mode = int(os.environ['RANK']), os.environ['HRF_BENCH_ATTEMPT'], os.environ['STUB_MODE']
assert os.environ['HRF_BENCH_WORKER'] == '1' and os.environ['TORCHELASTIC_USE_AGENT_STORE'] == 'False'
assert os.environ['HRF_BENCH_LAUNCHED_BY_BENCH'] == '1'          # the worker leaves the schedule A/B to its supervisor
if att == 'auto':
    assert os.environ.get('HRF_SYNC_P2P') != '0' and os.environ['HRF_P2P_TIMEOUT_S'] == '60'
    if mode in ('hang', 'dead'):
        time.sleep(600)                                          # a protocol hang: only the supervisor's limit ends it
    if mode == 'fail':
        sys.exit(7)                                              # the exchange's own 60 s limit fired: every rank exits non-zero
else:
    assert att == 'rccl_packed_fallback' and os.environ['HRF_SYNC_P2P'] == '0'
    if mode == 'dead':
        sys.exit(9)
if rank == 0:
    print('some other output')
    print(json.dumps({'value': 1.0, 'n_gpus': 2, 'config': {'sync_schedule': att, 'port': os.environ['MASTER_PORT']}}))
'''


@pytest.mark.parametrize('mode', ['ok', 'hang', 'fail', 'dead'])
def test_rank_supervisor_timeout_and_fresh_child_fallback(mode, tmp_path):
    """a rank process started by an external launcher is a supervisor that stays off the GPU; the real rank is a
    child with a wall-clock limit, and a hang / failure on the default (peer-to-peer) SyncBN schedule ends in a FRESH child job
    on the RCCL packed schedule - or, when that fails too, in a non-zero exit status and no result line, within the budget.
    Two supervisors (RANK 0 / 1) run here against a stand-in worker (no GPU): the decisions are taken per rank from the own
    child's exit status."""
    import time
    stub = tmp_path / 'stub_worker.py'
    stub.write_text(STUB_WORKER)
    procs = []
    t0 = time.time()
    for rank in (0, 1):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29720',
                   HRF_BENCH_WORKER_CMD=f'{sys.executable} {stub}', STUB_MODE=mode, HRF_BENCH_ATTEMPT_TIMEOUT_S='6',
                   HRF_BENCH_BUDGET_S='90')
        for k in ('HRF_BENCH_WORKER', 'HRF_BENCH_LAUNCHED_BY_BENCH', 'HRF_SYNC_P2P', 'HRF_P2P_TIMEOUT_S'):
            env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                                       '--no-sync-ab'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=120) for p in procs]
    wall = time.time() - t0
    assert wall < 60, wall
    if mode == 'dead':
        assert all(p.returncode != 0 for p in procs)
        assert not any(ln.startswith('{') for o, _ in outs for ln in o.splitlines())          # no result line: never a made-up one
        assert 'attempt "auto" failed' in outs[0][1] and 'attempt "rccl_packed_fallback" failed' in outs[0][1]
        return
    assert [p.returncode for p in procs] == [0, 0], [o[1][-600:] for o in outs]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and outs[1][0].strip() == ''                                        # ONE line, from rank 0
    line = json.loads(lines[0])
    tried = line['config']['attempts']
    if mode == 'ok':
        assert [t['attempt'] for t in tried] == ['auto'] and line['config']['sync_schedule'] == 'auto'
        assert line['config']['port'] == '29737'                                               # a rendezvous of the child's own
    else:
        assert [t['attempt'] for t in tried] == ['auto', 'rccl_packed_fallback'] and line['config']['sync_schedule'] == 'rccl_packed_fallback'
        assert tried[0]['rc'] == (None if mode == 'hang' else 7) and tried[1]['rc'] == 0
        assert line['config']['port'] == '29738'
        assert 'attempt "auto" failed' in outs[0][1] and 'attempt "auto" failed' in outs[1][1]


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_gloo():
    """N = 2 end to end on ONE GPU: self-launch, SyncBN (packed exchanges, lock-step strands), gradient all-reduce, JSON relay,
    and the A/B child of the per-lane-communicator schedule with its gradient-equivalence check - over gloo (RCCL refuses two
    ranks on one device), eager launches."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1',
                        '--height', '64', '--width', '96', '--no-roofline', '--sync-ab-timeout', '600'],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-1000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 4 and line['finite'] is True
    # HRF_SYNC_P2P unset = auto: two ranks -> the peer-to-peer exchange after its handshake (the ranks share a GPU: IPC works)
    assert line['config']['collectives_per_step'] >= 1 and 'peer-to-peer' in line['config']['sync_schedule']
    assert line['config']['p2p_exchanges_per_step'] > 100
    ab = line['sync_ab']
    assert 'error' not in ab, ab
    assert ab['grad_rel_l2_vs_main_lane'] < 1e-6 and ab['ms_per_step'] > 0
    px = ab['p2p']                                       # third arm: the peer-to-peer exchange (IPC inboxes, no communicator)
    assert 'error' not in px, px
    assert px['grad_rel_l2_vs_main_lane'] < 1e-6 and px['p2p_exchanges_per_step'] > 100 and px['ms_per_step'] > 0
    assert px['main_lane_ms_per_step'] > 0 and px['main_lane_collectives_per_step'] > 4
    print(json.dumps({k: line[k] for k in ('value', 'ms_per_step', 'sync_ab')}))


def test_rank_supervisor_takes_its_worker_along_on_sigterm(tmp_path):
    """A supervisor ended by its launcher (SIGTERM: another rank failed) kills the worker's process group: the worker runs in a
    session of its own (so that a time-out can kill the whole group) and would otherwise keep its GPU after the job is gone."""
    import signal
    import time
    stub = tmp_path / 'stub_worker.py'
    pidfile = tmp_path / 'worker.pid'
    stub.write_text(f"import os, time\nopen(r'{pidfile}', 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29760',
               HRF_BENCH_WORKER_CMD=f'{sys.executable} {stub}', HRF_BENCH_ATTEMPT_TIMEOUT_S='120', HRF_BENCH_BUDGET_S='300')
    for k in ('HRF_BENCH_WORKER', 'HRF_BENCH_LAUNCHED_BY_BENCH'):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-sync-ab'],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    t0 = time.time()
    while not pidfile.exists() and time.time() - t0 < 60:
        time.sleep(0.2)
    assert pidfile.exists()
    time.sleep(0.3)
    wpid = int(pidfile.read_text())
    p.send_signal(signal.SIGTERM)
    p.communicate(timeout=30)
    assert p.returncode == 128 + signal.SIGTERM
    for _ in range(50):                                   # the worker is gone (reaped by init)
        try:
            os.kill(wpid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        os.kill(wpid, signal.SIGKILL)
        raise AssertionError('worker survived its supervisor')
