"""Receive band-pass design, host side.

In Quisk the GUI designs the receive filter and hands the taps to the C module (quisk.py:4750-4801 designs,
quisk.py:4809-4830 picks the centre, quisk.py:4866-4868 calls `QS.set_filters(filtI, filtQ, bw, lower_edge, 0)`).
An unchanged quisk.py keeps doing that.  This module gives callers without a GUI (tests, bench.py, the facade's
own users) taps of the same design, written here as array expressions:

    prototype   : a low-pass of half-width bw/2.  Either one of the tabulated designs of the reference's
                  filters.py (data, kept in data/filter_prototypes.npz and keyed by the bandwidth scaled to a
                  24 ksps rate), or a Blackman-windowed periodic-sinc (Dirichlet kernel) of odd length.
    band-pass   : the prototype heterodyned to |centre| Hz, 2 h[i] exp(-j w (i - mid)); its real part filters the
                  I channel and its imaginary part the Q channel (cRxFilterOut, quisk.c:1154-1192).

Parity status: unpinned (the reference holds no vectors for it).  Engine and oracle are both
fed the taps this module makes, so the parity tests do not depend on it.
"""
import os

import numpy as np

MODES = ['CWL', 'CWU', 'LSB', 'USB', 'AM', 'FM', 'EXT', 'DGT-U', 'DGT-L', 'DGT-IQ', 'IMD', 'FDV-U', 'FDV-L', 'DGT-FM']
MODE_INDEX = {m: i for i, m in enumerate(MODES)}      # rx_mode_type, quisk.h:48-63
DEFAULT_BW = {0: 500, 1: 500, 2: 2800, 3: 2800, 4: 8000, 5: 16000, 7: 3200, 8: 3200, 9: 6000, 13: 16000}   # quisk.py:3579-3588

_LOWER = ('CWL', 'LSB', 'DGT-L', 'FDV-L')
_CENTRED = ('AM', 'FM', 'DGT-IQ', 'DGT-FM')
_protos = None


def prototypes():
    """{key: taps} of the tabulated low-pass designs; key = bandwidth * 24000 // rate // 2."""
    global _protos
    if _protos is None:
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "filter_prototypes.npz"))
        _protos = {int(k[2:]): np.asarray(z[k], dtype=np.float64) for k in z.files}
    return _protos


def _windowed_dirichlet(rate, bw, length=None):
    """Odd-length low-pass: Dirichlet kernel with `bw * length // rate` bins passed, under a Blackman window.
    Default length: four transition widths of a 1.5 shape factor, capped at 1000, made odd."""
    if length is None:
        transition = 0.25 * bw / rate            # (bw / 2 / rate) * (1.5 - 1)
        length = min(int(4.0 / transition), 1000) // 2 * 2 + 1
    k = np.arange((-length) // 2, length // 2 + 1, dtype=np.float64)      # floor division: an odd length gives length + 1 taps
    passed = bw * length // rate
    with np.errstate(divide="ignore", invalid="ignore"):
        kernel = np.sin(np.pi * k * passed / length) / np.sin(np.pi * k / length) / length
    kernel[k == 0] = float(passed) / length
    blackman = 0.42 + 0.5 * np.cos(2.0 * np.pi * k / length) + 0.08 * np.cos(4.0 * np.pi * k / length)
    return kernel * blackman


def make_filter_coef(rate, N, bw, center):
    """(I taps, Q taps) of a pass band `bw` Hz wide centred on |center| Hz at sample rate `rate`; N = None picks the
    length.  center == 0 returns the low-pass itself for both channels."""
    table = prototypes().get(bw * 24000 // rate // 2)
    h = table if table is not None else _windowed_dirichlet(rate, bw, N)
    if not center:
        return h.copy(), h.copy()
    mid = (len(h) - 1.0) / 2.0
    rot = np.exp(-2j * np.pi * abs(center) / rate * (np.arange(len(h)) - mid))
    z = 2.0 * rot * h
    return np.ascontiguousarray(z.real), np.ascontiguousarray(z.imag)


def filter_center(mode, bandwidth, cw_tone=600):
    """Centre of the pass band for a mode name (quisk.py:4809-4830); negative for the lower-sideband modes."""
    if mode in _CENTRED:
        return 0
    if mode in ('CWU', 'CWL'):
        c = max(cw_tone, bandwidth // 2)
    elif mode in ('DGT-U', 'DGT-L', 'FDV-U', 'FDV-L'):
        c = max(1500, bandwidth // 2)
    else:
        c = 300 + bandwidth // 2
    return -c if mode in _LOWER else c


def rx_filter(sample_rate, mode, bw=None):
    """(I taps, Q taps, bandwidth, lower edge) for receive mode `mode` (number) the way the GUI sets a receiver up
    (quisk.py:4837-4868): the filter is designed at get_filter_rate(mode, bw), the bandwidth capped at half of it."""
    from .engine import get_filter_rate
    bw = bw or DEFAULT_BW[mode]
    frate = get_filter_rate(sample_rate, mode, bw)
    bw = min(bw, frate // 2)
    center = filter_center(MODES[mode], bw)
    fi, fq = make_filter_coef(frate, None, bw, center)
    return fi, fq, bw, center - bw // 2
