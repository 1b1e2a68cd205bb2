"""world_size = 2 rehearsal of the multi-GPU plumbing on CPU (gloo): broadcast of shared parameters,
contiguous sharding of records, gather of per-rank rows - the only communication the path has."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from volumetricinterp_amd.parallel import shard_bounds

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    for T in (0, 1, 7, 8, 1000, 10000):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_bounds(T, r, world)
                assert 0 <= lo <= hi <= T
                cover += list(range(lo, hi))
            assert cover == list(range(T))


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from volumetricinterp_amd.parallel import Comm, shard_bounds
    comm = Comm(backend=os.environ.get('TEST_BACKEND', 'gloo'))
    rng = np.random.default_rng(1)
    T, P, N = 7, 11, 5
    shared = dict(lat=rng.uniform(70, 80, P), R=rng.standard_normal((N, N))) if comm.rank == 0 else {}
    shared = comm.broadcast_arrays(shared)
    value = np.random.default_rng(2).standard_normal((T, P))      # every rank can regenerate its records
    lo, hi = shard_bounds(T, comm.rank, comm.world)
    # stand-in for the per-record fit: any function of (shared, record) with no cross-record state
    local = np.stack([shared['R'] @ np.full(N, value[t] @ shared['lat']) for t in range(lo, hi)]) if hi > lo \\
        else np.zeros((0, N))
    full = comm.gather_rows(local, T)
    tmax = comm.max_over_ranks(1.0 + comm.rank)
    comm.barrier()
    if comm.rank == 0:
        np.savez(sys.argv[1], full=full, lat=shared['lat'], R=shared['R'], tmax=tmax)
    comm.close()
''')


import pytest


@pytest.mark.parametrize('backend,world', [('gloo', 2), ('socket', 2), ('socket', 3)])
def test_multi_rank_equals_single_process(tmp_path, backend, world):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % REPO)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'out.npz')
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), TEST_BACKEND=backend, VINTERP_RDV_PATH=str(tmp_path / 'rdv.sock'))
        procs.append(subprocess.Popen([sys.executable, str(script), out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()
    got = np.load(out)
    rng = np.random.default_rng(1)
    T, P, N = 7, 11, 5
    lat, R = rng.uniform(70, 80, P), rng.standard_normal((N, N))
    value = np.random.default_rng(2).standard_normal((T, P))
    want = np.stack([R @ np.full(N, value[t] @ lat) for t in range(T)])
    np.testing.assert_array_equal(got['lat'], lat)
    np.testing.assert_array_equal(got['R'], R)
    np.testing.assert_array_equal(got['full'], want)          # sharded result == single-process result, bit for bit
    assert float(got['tmax']) == float(world)
