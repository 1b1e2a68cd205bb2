"""Command line entry point: fit every record of the file named in a config and save the coefficients.

Mirror of the reference CLI ``volumetricinterp/run_volumetricinterp.py:14-35`` (console script
``volumetricinterp``) for the fit path; ``--validate`` (cartopy plots, validate.py) is out of scope here.

    python -m volumetricinterp_amd.run_volumetricinterp config.ini

Multi-GPU (one node): launch one process per GPU with the usual launcher,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m volumetricinterp_amd.run_volumetricinterp config.ini

Every rank fits a contiguous block of the file's records on GPU LOCAL_RANK; rank 0 broadcasts the regularisation
matrices (RCCL) and writes the coefficient file (parallel.py).
"""
import argparse
import sys


def main(argv=None):
    parser = argparse.ArgumentParser(description='Fit the 3-D analytic model to every record of an AMISR file '
                                                 'on an MI355X and save the coefficient file.')
    parser.add_argument('config_file', help='configuration file (same keys as the reference example_config.ini)')
    parser.add_argument('--validate', action='store_true',
                        help='(reference option; plotting is not part of volumetricinterp_amd)')
    args = parser.parse_args(argv)
    if args.validate:
        print('--validate draws cartopy maps in the reference (validate.py); it is not part of this package.',
              file=sys.stderr)
        return 2
    from .interpolate import Interpolate
    from .parallel import Comm, env_rank
    rank, world, local_rank = env_rank()
    if world > 1:
        import os
        from . import _lib
        ctx = _lib.get_context(local_rank)
        comm = Comm(backend=os.environ.get('VINTERP_DIST_BACKEND', 'rccl'), ctx=ctx)
        try:
            interp = Interpolate(args.config_file, ctx=ctx)
            interp.calc_coeffs(comm=comm)
            if rank == 0:
                interp.saveh5()
            comm.barrier()
        finally:
            comm.close()
        return 0
    interp = Interpolate(args.config_file)
    interp.calc_coeffs()
    interp.saveh5()
    return 0


if __name__ == '__main__':
    sys.exit(main())
