"""The C-ABI library loads and exports every symbol include/vinterp.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(REPO, 'include', 'vinterp.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(vi_[a-z0-9_]+)\s*\(', txt)))


def test_header_declares_the_documented_surface():
    names = declared_functions()
    for must in ('vi_ctx_create', 'vi_model_create', 'vi_basis_f64', 'vi_eval_f64', 'vi_normal_eq_f64',
                 'vi_solve_trunc_f64', 'vi_chi2_f64', 'vi_cov_f64', 'vi_last_error'):
        assert must in names


def test_library_exports_every_declared_symbol():
    from volumetricinterp_amd import _lib, fitengine  # noqa: F401
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.vi_abi_version() == 1
    # every declared function has a ctypes signature registered on the Python side
    unbound = [n for n in declared_functions() if n not in _lib.EXPORTS]
    assert not unbound, unbound


def test_no_gpu_means_loud_failure_not_fallback():
    from volumetricinterp_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(_lib.VinterpError, match='no HIP device'):
        _lib.get_context()
    import io
    from volumetricinterp_amd.models.sphharmlag import Model
    m = Model(io.StringIO('[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\n'
                          'MAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'))
    import numpy as np
    with pytest.raises(_lib.VinterpError):
        m.basis(np.zeros(3), np.zeros(3), np.zeros(3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, 'volumetricinterp_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', src, flags=re.M), os.path.join(root, f)
