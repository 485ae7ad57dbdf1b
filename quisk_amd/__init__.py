"""quisk_amd -- MI355X-native receive-DSP engine behind Quisk's `_quisk` entry points.

The compute path is the HIP library `libqkengine.so` (C ABI in include/qk_engine.h); this package is
the thin host-side mirror used by tests, bench.py and Python callers.  There is no CPU fallback.
"""
from .engine import Engine, QkError, lib, plan_decimation, get_filter_rate, decim_rate  # noqa: F401
from .rxfilters import make_filter_coef, filter_center, rx_filter, MODES, MODE_INDEX, DEFAULT_BW  # noqa: F401
