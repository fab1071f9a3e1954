"""noLZSS -- the reference package's import names over the MI355X-native engine.

`from noLZSS import factorize, factorize_file, count_factors` resolves exactly as with the reference
(reference: src/noLZSS/__init__.py:9-20): a compiled pybind11 module `noLZSS._noLZSS`
(nolzss_amd/csrc/pybind_shim.cpp, the binding of INTEGRATION.md section 1, built by
`make -C nolzss_amd/csrc`) with `core`, `utils`, `genomics` and `parallel` on top of it.  The compiled
module binds the factorize path; the names of the reference module it does not bind itself are taken
from the ctypes mirror `nolzss_amd._noLZSS` -- every one of them calls the same C ABI
(include/nolzss_hip.h) of libnolzss_hip.so.  There is no CPU fallback: without the compiled module or
the HIP library the import fails.
"""
import os as _os
from pathlib import Path as _Path

# The compiled module is linked against nolzss_amd/libnolzss_hip.so (rpath); NOLZSS_LIB makes the ctypes mirror load
# ANOTHER build of the library (A/B measurements, tools/ab_variant.sh).  Two builds in one process -- two arenas,
# two sets of device state -- is never what a caller wants: this package refuses it.
if _os.environ.get("NOLZSS_LIB"):
    _default = _Path(__file__).resolve().parent.parent / "nolzss_amd" / "libnolzss_hip.so"
    if _Path(_os.environ["NOLZSS_LIB"]).resolve() != _default:
        raise ImportError(f"noLZSS: NOLZSS_LIB={_os.environ['NOLZSS_LIB']} names another build than the one noLZSS._noLZSS is "
                          f"linked against ({_default}); use the nolzss_amd package for variant libraries, or unset it")

from nolzss_amd import _noLZSS as _mirror  # first: loads libnolzss_hip.so (and the HIP runtime it shares with torch)
from . import _noLZSS  # compiled (pybind11); ImportError if it has not been built

_native_set_device = _noLZSS.set_device
for _name in dir(_mirror):
    if not _name.startswith("_") and not hasattr(_noLZSS, _name):
        setattr(_noLZSS, _name, getattr(_mirror, _name))


def set_device(device: int) -> None:
    """Extension: HIP device used by this process (default NOLZSS_DEVICE / LOCAL_RANK / 0)."""
    _native_set_device(int(device))
    _mirror.set_device(int(device))


_noLZSS.set_device = set_device

from ._noLZSS import __version__  # noqa: E402
from .core import *  # noqa: E402,F401,F403
from .utils import *  # noqa: E402,F401,F403
from .core import __all__ as _core_all  # noqa: E402
from .utils import __all__ as _utils_all  # noqa: E402

__all__ = list(_core_all) + list(_utils_all) + ["__version__", "set_device"]
