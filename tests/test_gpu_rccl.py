"""RCCL entry points.  On the one-GPU box: a 1-rank communicator exercises dlopen, id generation, init and a
broadcast through the same code the multi-GPU bench uses (the N > 1 control plane is covered on CPU).  With two or
more GPUs: a 2-rank communicator on devices 0 and 1 (test_two_rank_broadcast)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_broadcast():
    from volumetricinterp_amd import _lib
    ctx = _lib.get_context()
    buf = C.create_string_buffer(128)
    _lib.check(_lib.lib.vi_rccl_unique_id(buf), 'vi_rccl_unique_id')
    _lib.check(_lib.lib.vi_rccl_init(ctx.handle, 1, 0, buf.raw), 'vi_rccl_init')
    x = np.arange(1000, dtype=np.float64)
    d = ctx.to_device(x)
    _lib.check(_lib.lib.vi_rccl_bcast_f64(ctx.handle, d.ptr, x.size, 0), 'vi_rccl_bcast_f64')
    np.testing.assert_array_equal(d.download(), x)
    _lib.check(_lib.lib.vi_rccl_destroy(ctx.handle), 'vi_rccl_destroy')


_TWO_RANK_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ['VI_REPO'])
from volumetricinterp_amd import _lib
from volumetricinterp_amd.parallel import Comm
rank = int(os.environ['RANK'])
ctx = _lib.get_context(int(os.environ['LOCAL_RANK']))
comm = Comm(backend='rccl', ctx=ctx)
assert comm.world == 2
assert comm.rccl_ready, 'RCCL did not initialise on 2 ranks: %s' % comm.notes
src = {'a': np.arange(5000, dtype=np.float64).reshape(50, 100) * 0.5, 'b': np.array([3.25])} if rank == 0 else {}
got = comm.broadcast_arrays(src)
assert got['a'].shape == (50, 100) and np.array_equal(got['a'], np.arange(5000).reshape(50, 100) * 0.5)
assert got['b'][0] == 3.25
assert comm.max_over_ranks(float(rank + 1)) == 2.0
comm.barrier()
comm.close()
print('rank %d ok' % rank)
'''


def test_two_rank_broadcast(tmp_path):
    """RCCL with more than one rank: two fresh processes on devices 0 and 1 create a communicator from the id rank 0
    generated (shipped over the control socket) and broadcast the shared parameters with ncclBroadcast - the one collective
    of the multi-GPU path.  Needs two GPUs; skipped on the one-GPU box."""
    import os
    import socket
    import subprocess
    import sys
    from volumetricinterp_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip('needs two GPUs (device_count = %d)' % _lib.device_count())
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'worker.py'
    script.write_text(_TWO_RANK_WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', LOCAL_WORLD_SIZE='2', VI_REPO=repo,
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0',
                   VINTERP_RDV_PATH=str(tmp_path / 'rdv.sock'))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o.decode('utf-8', 'replace'))
    assert all(p.returncode == 0 for p in procs), outs
    assert 'rank 0 ok' in outs[0] and 'rank 1 ok' in outs[1], outs
