"""CPU oracle (TEST INFRASTRUCTURE ONLY): see oracle/ref_fft.h.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this package.  Nothing under ``fft_wgpu_amd/`` imports it.
"""
from .oracle import *  # noqa: F401,F403
