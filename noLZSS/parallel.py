"""noLZSS.parallel: the thread-parallel API names (reference: src/noLZSS/parallel.py); on the GPU the
thread count is ignored and the results equal the sequential ones."""
from nolzss_amd.parallel import *  # noqa: F401,F403
from nolzss_amd import parallel as _p

__all__ = [n for n in dir(_p) if n.startswith("parallel_")] + ["Factor"]
