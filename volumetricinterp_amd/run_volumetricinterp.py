"""Command line entry point: fit every record of the file named in a config and save the coefficients.

Mirror of the reference CLI ``volumetricinterp/run_volumetricinterp.py:14-35`` (console script
``volumetricinterp``) for the fit path; ``--validate`` (cartopy plots, validate.py) is out of scope here.

    python -m volumetricinterp_amd.run_volumetricinterp config.ini
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
    interp = Interpolate(args.config_file)
    interp.calc_coeffs()
    interp.saveh5()
    return 0


if __name__ == '__main__':
    sys.exit(main())
