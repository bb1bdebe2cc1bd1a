"""bench.py's N > 1 control flow executed on CPU (two ranks over gloo, the test double of the C ABI): rank-0 build,
rendezvous from RANK / WORLD_SIZE / MASTER_*, GradSync attach, barrier + synchronize around the timed region, MAX-reduce of
the elapsed time, exactly one JSON line from rank 0 -- so that the first multi-rank execution of the script is not on the
driver's clock (the 2 / 4 / 8-GPU RCCL run itself can only happen there)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def test_bench_two_ranks_dry_run_prints_one_valid_line():
  port = _free_port()
  procs = []
  for rank in range(2):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port), OMP_NUM_THREADS='2')
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1',
                                   '--batch', '2', '--dry-run-cpu', '--comm-dtype', 'bf16'], env=env, cwd=ROOT,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
  outs = [p.communicate(timeout=900) for p in procs]
  for p, (so, se) in zip(procs, outs):
    assert p.returncode == 0, se[-2000:]
  lines0 = [l for l in outs[0][0].splitlines() if l.startswith('{')]
  lines1 = [l for l in outs[1][0].splitlines() if l.startswith('{')]
  assert len(lines0) == 1 and not lines1, 'rank 0 prints exactly one JSON line, the other ranks none'
  r = json.loads(lines0[0])
  for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config'):
    assert k in r, k
  assert r['n_gpus'] == 2 and r['steps'] == 1 and r['scaling'] == 'weak' and r['config']['global_batch'] == 4
  assert r['config']['parallelism'] == 'dp2' and r['value'] > 0
  dp = r['dp']
  assert dp['world'] == 2 and dp['comm_dtype'] == 'bf16' and dp['buckets'] >= 2 and dp['bytes_per_step'] > 80e6
  assert 'DRY RUN' in r['data']
  # the N > 1 line verifies itself: every rank reported in, with its own clock, and the exchange was measured at this N
  assert dp['world_seen_by_backend'] == 2 and [q['rank'] for q in dp['ranks']] == [0, 1]
  assert all(q['ms_per_step'] > 0 and q['host'] for q in dp['ranks'])
  assert dp['rank_ms_per_step_min'] <= dp['rank_ms_per_step_max'] and dp['ms_per_step_without_exchange'] > 0
  assert dp['exchange_ms_exposed'] is not None


def test_bench_rejects_a_rank_count_mismatch():
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run-cpu'], cwd=ROOT,
                     capture_output=True, text=True, env=dict(os.environ, WORLD_SIZE='1'), timeout=300)
  assert r.returncode != 0 and 'torch.distributed.run' in (r.stderr + r.stdout)
