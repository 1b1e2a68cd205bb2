"""CPU oracle for the fit + evaluate hot path of amisr/volumetricinterp.

TEST INFRASTRUCTURE ONLY.  Nothing under ``volumetricinterp_amd/`` may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and there only as the checker.

This is a NumPy/SciPy restatement of the reference's algorithm (same
algorithmic structure: an N-iteration special-function loop for the basis,
A^T W A rebuilt for every regularisation parameter, SVD-truncated ``lstsq``,
five scale factors x bracket walk + Brent).  Every function cites the
reference file:line it follows.  SciPy is the reference's own third-party
dependency (``scipy.special.lpmv`` etc.) and is present on the GPU box;
``pymap3d`` is absent and is restated as the WGS84 closed form.

Parity pin: the oracle is checked in ``tests/test_oracle_golden.py`` against
golden vectors produced by importing the reference itself in the build
container (``tools/gen_golden.py`` -> ``tests/golden/*.npz``), and against the
known-answer values recorded in SURVEY.md section 8c.
"""

from .geodesy import geodetic2ecef            # noqa: F401
from .sphharmlag import SphHarmLagOracle      # noqa: F401
from .radbasfun import RadBasFunOracle        # noqa: F401
from .fit import (eval_C, chi2objfunct, chi2_search, find_reg_param,   # noqa: F401
                  fit_records, compute_hull_vertices, gcvobjfunct, gcv_search)
from .evaluate import get_C, check_hull, evaluate, evaluate_gradient, evaluate_error   # noqa: F401
