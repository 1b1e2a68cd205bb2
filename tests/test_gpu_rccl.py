"""RCCL entry points on the one-GPU box: a 1-rank communicator exercises dlopen, id generation, init and a
broadcast through the same code the multi-GPU bench uses (the N > 1 control plane is covered on CPU)."""
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
