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


CALC_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from volumetricinterp_amd.interpolate import Interpolate
    from volumetricinterp_amd.parallel import Comm

    T, P, N = 7, 13, 4

    class FakeModel(object):
        nbasis = N
        def __init__(self):
            self.calls = 0
            self.eval_reg_matricies = {'curvature': self.omega}
        def omega(self):
            self.calls += 1
            return np.arange(N * N, dtype=np.float64).reshape(N, N) + 0.5

    class Fake(Interpolate):
        """calc_coeffs with the device fit replaced by a deterministic per-record function (no GPU here)."""
        def __init__(self):
            self.model = FakeModel()
            self.model_name = 'fake'
            self.regularization_list = ['curvature']
            self.filename = 'unused'
        def read_datafile(self, filename):
            rng = np.random.default_rng(5)
            utime = np.stack([np.arange(T) * 60., np.arange(T) * 60. + 60.], axis=1)
            return (utime, rng.uniform(70, 80, P), rng.uniform(250, 270, P), rng.uniform(1e5, 5e5, P),
                    rng.standard_normal((T, P)), rng.uniform(1, 2, (T, P)))
        def compute_hull(self, lat, lon, alt):
            self.hull_vert = np.zeros((4, 3))
        def fit_records(self, lat, lon, alt, value, error, reg_matricies, calccov=True, record_slice=None):
            R = reg_matricies['curvature']
            n = value.shape[0]
            C = np.stack([R @ np.full(N, value[t] @ lat) for t in range(n)])
            return dict(Coeffs=C, Covariance=np.stack([np.outer(c, c) for c in C]), chi_sq=(value / error).sum(axis=1),
                        reg_params=[{'curvature': float(abs(value[t, 0]))} for t in range(n)])

    world = int(os.environ.get('WORLD_SIZE', '1'))
    comm = Comm(backend='socket') if world > 1 else None
    it = Fake()
    it.calc_coeffs(comm=comm)
    if comm is None or comm.rank == 0:
        np.savez(sys.argv[1], Coeffs=it.Coeffs, Covariance=it.Covariance, chi_sq=it.chi_sq, time=it.time,
                 alpha=np.array([p['curvature'] for p in it.reg_params]), omega_calls=it.model.calls)
    else:
        assert it.model.calls == 0                   # only rank 0 evaluates the regularisation matrices
        assert it.Coeffs.shape == (T, N)             # every rank ends up with the full result
    if comm is not None:
        comm.barrier()
        comm.close()
''')


@pytest.mark.parametrize('world', [2, 3, 8])
def test_sharded_calc_coeffs_equals_single_process(tmp_path, world):
    """Interpolate.calc_coeffs(comm=...) - the record loop sharded over ranks (8 ranks > 7 records: one rank has an
    empty block) - gives exactly the single-process arrays."""
    script = tmp_path / 'worker.py'
    script.write_text(CALC_WORKER % REPO)
    outs = {}
    for w in (1, world):
        out = str(tmp_path / ('out%d.npz' % w))
        procs = []
        for r in range(w):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(w), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT='0', VINTERP_RDV_PATH=str(tmp_path / ('rdv%d.sock' % w)))
            procs.append(subprocess.Popen([sys.executable, str(script), out], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT))
        for p in procs:
            o, _ = p.communicate(timeout=300)
            assert p.returncode == 0, o.decode()
        outs[w] = np.load(out)
    for k in ('Coeffs', 'Covariance', 'chi_sq', 'time', 'alpha'):
        np.testing.assert_array_equal(outs[world][k], outs[1][k])
    assert int(outs[world]['omega_calls']) == 1


def test_control_plane_framing_and_private_directory(tmp_path, monkeypatch):
    """Wire format of the socket control plane (no pickle): frames and arrays survive a round trip; the rendezvous
    directory is private to the user and a foreign or group-writable one is refused."""
    import stat
    from volumetricinterp_amd import parallel as P
    parts = [b'', b'abc', bytes(range(256)) * 3]
    assert P._unpack_parts(P._pack_parts(parts)) == parts
    a = np.arange(24, dtype=np.float64).reshape(2, 3, 4) * 0.5
    b = P._unpack_array(P._pack_array(a))
    assert b.shape == a.shape and np.array_equal(a, b)
    assert P._unpack_array(P._pack_array(np.zeros((0, 5)))).shape == (0, 5)
    monkeypatch.delenv('XDG_RUNTIME_DIR', raising=False)
    d = P._private_socket_dir()
    st = os.lstat(d)
    assert st.st_uid == os.getuid() and not (st.st_mode & 0o077) and stat.S_ISDIR(st.st_mode)
    monkeypatch.setenv('XDG_RUNTIME_DIR', str(tmp_path))
    assert P._private_socket_dir() == str(tmp_path)
