"""noLZSS.core: the one implementation in `nolzss_amd.core`, bound to the compiled module
`noLZSS._noLZSS` (pybind11 -> C ABI -> HIP) instead of the ctypes mirror."""
from nolzss_amd.core import __all__, bind

from . import _noLZSS

globals().update(bind(_noLZSS))
