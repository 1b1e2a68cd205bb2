"""bench.py emits ONE workload at every N (the N = 1 and N > 1 lines were different workloads, so the driver
could not form a scaling curve): the same metric, unit, scaling mode and keys from `--gpus 1` and from a 2-rank run - here a
small rehearsal (64 timesteps, 32^3 grid) with both ranks sharing the one GPU over the socket control plane."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(n, extra=()):
    env = dict(os.environ, VINTERP_DIST_BACKEND='socket', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(n), '--records', '64', '--c3-grid', '32',
                        '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-secondary'] + list(extra),
                       env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_one_workload_at_every_n():
    one, two = _bench(1), _bench(2)
    for line, n in ((one, 1), (two, 2)):
        assert line['metric'] == 'fit+eval timesteps/sec' and line['unit'] == 'timesteps/s'
        assert line['scaling'] == 'strong' and line['dtype'] == 'f64' and line['data'] == 'synthetic'
        assert line['n_gpus'] == n and line['steps'] == 1 and line['vs_baseline'] is None
        assert line['config']['timesteps'] == 64 and line['config']['timesteps_per_rank'] == 64 // n
        assert line['value'] > 0 and abs(line['value'] - 64 / (line['ms_per_step'] * 1e-3)) <= 1e-6 * line['value']
        # the dominant part of the step - the fit - carries the roofline; the evaluation product is the second object
        rf, re_ = line['roofline'], line['roofline_eval']
        assert rf['bound'] == 'lds' and rf['achieved'] > 0 and 0 < rf['frac'] < 1 and rf['rounds_per_step'] > 0
        assert abs(rf['frac'] - rf['achieved'] / rf['peak']) <= 1e-12
        assert re_['bound'] == 'mfma' and re_['achieved'] > 0 and 0 < re_['frac'] < 1
        assert len(line['per_rank_step_s']) == n
    assert one['config']['workload'] == two['config']['workload']
    assert set(one) - {'cpu_baseline'} == set(two) - {'cpu_baseline'}
    assert set(one['roofline']) == set(two['roofline']) and set(one['roofline_eval']) == set(two['roofline_eval'])
