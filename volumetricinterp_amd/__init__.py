"""volumetricinterp_amd - MI355X-native fit + evaluate hot path of amisr/volumetricinterp.

Mirrors the reference package surface (volumetricinterp/__init__.py:1-5): ``Interpolate`` and
``Estimate``.  The compute path is libvinterp.so (HIP, gfx950); importing the classes without the
built library raises - there is no CPU fallback.
"""
__version__ = '0.1.0'


def __getattr__(name):
    if name == 'Estimate':
        from .estimate import Estimate
        return Estimate
    if name == 'Interpolate':
        from .interpolate import Interpolate
        return Interpolate
    raise AttributeError(name)
