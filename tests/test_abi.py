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
    assert lib.vi_abi_version() == 2
    # every declared function has a ctypes signature registered on the Python side
    unbound = [n for n in declared_functions() if n not in _lib.EXPORTS]
    assert not unbound, unbound


def test_brent_kernel_range_is_a_host_side_question():
    """vi_brent_warm_supported: pure host arithmetic (LDS budget of a k_brent_warm workgroup), no GPU needed.  FitEngine asks it
    before choosing the device-side iteration, so a record size the kernel cannot hold goes to the host-driven rounds instead
    of raising."""
    from volumetricinterp_amd import _lib, fitengine  # noqa: F401
    sup = _lib.lib.vi_brent_warm_supported
    assert sup(144, 2600) == 1 and sup(144, 16385) == 1 and sup(32, 550) == 1
    assert sup(144, 30_000_000) == 0          # 117 000 partial sums: more than a CU's LDS beside the system
    assert sup(1152, 12800) == 0              # outside the in-LDS Jacobi range
    assert sup(0, 10) == 0 and sup(144, 0) == 0


def test_a_stale_library_is_refused(tmp_path):
    """_lib checks vi_abi_version() right after loading: signatures changed in round 3 without a version bump (ADVICE), and
    VINTERP_LIB makes loading another build easy.  A library that reports another version must not be bound."""
    import subprocess
    import sys
    src = tmp_path / 'stale.c'
    src.write_text('int vi_abi_version(void) { return 1; }\n')
    so = tmp_path / 'libstale.so'
    subprocess.check_call(['gcc', '-shared', '-fPIC', str(src), '-o', str(so)])
    r = subprocess.run([sys.executable, '-c', 'import volumetricinterp_amd._lib'], env=dict(os.environ, VINTERP_LIB=str(so)),
                       cwd=REPO, capture_output=True, text=True)
    assert r.returncode != 0 and 'ABI version 1' in r.stderr and 'VINTERP_LIB' in r.stderr, r.stderr


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
