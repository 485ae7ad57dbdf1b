"""Receive band-pass design, host side.

In Quisk the GUI designs the RX filter and hands the taps to the C module:
`MakeFilterCoef` (quisk.py:4750-4801), `GetFilterCenter` (quisk.py:4809-4830), then
`QS.set_filters(filtI, filtQ, bw, lower_edge, 0)` (quisk.py:4866-4868).  An unchanged quisk.py keeps doing
that; this module restates the same arithmetic for callers that have no GUI (tests, bench.py).  The
prototype low-pass taps are the data of the reference's filters.py (`Filters` dict), stored in
data/filter_prototypes.npz by tests/golden/make_golden.py.
"""
import cmath
import math
import os

import numpy as np

MODES = ['CWL', 'CWU', 'LSB', 'USB', 'AM', 'FM', 'EXT', 'DGT-U', 'DGT-L', 'DGT-IQ', 'IMD', 'FDV-U', 'FDV-L', 'DGT-FM']
MODE_INDEX = {m: i for i, m in enumerate(MODES)}      # rx_mode_type, quisk.h:48-63

_protos = None


def prototypes():
    global _protos
    if _protos is None:
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "filter_prototypes.npz"))
        _protos = {int(k[2:]): z[k] for k in z.files}
    return _protos


def make_filter_coef(rate, N, bw, center):
    """I/Q taps of a rectangular pass band `bw` Hz wide centred on |center| Hz at sample rate `rate`."""
    center = abs(center)
    lowpass = bw * 24000 // rate // 2
    protos = prototypes()
    if lowpass in protos:
        filtD = [float(v) for v in protos[lowpass]]
    else:
        if N is None:
            shape = 1.5
            trans = (bw / 2.0 / rate) * (shape - 1.0)
            N = int(4.0 / trans)
            if N > 1000:
                N = 1000
            N = (N // 2) * 2 + 1
        K = bw * N // rate
        filtD = []
        for k in range(-N // 2, N // 2 + 1):
            if k == 0:
                z = float(K) / N
            else:
                z = 1.0 / N * math.sin(math.pi * k * K / N) / math.sin(math.pi * k / N)
            w = 0.42 + 0.5 * math.cos(2. * math.pi * k / N) + 0.08 * math.cos(4. * math.pi * k / N)
            filtD.append(z * w)
    if center:
        tune = -1j * 2.0 * math.pi * center / rate
        NN = len(filtD)
        D = (NN - 1.0) / 2.0
        filtI, filtQ = [], []
        for i in range(NN):
            z = 2.0 * cmath.exp(tune * (i - D)) * filtD[i]
            filtI.append(z.real)
            filtQ.append(z.imag)
        return np.array(filtI), np.array(filtQ)
    return np.array(filtD), np.array(filtD)


def filter_center(mode, bandwidth, cw_tone=600):
    """GetFilterCenter, quisk.py:4809-4830 (mode as a name from MODES)."""
    if mode in ('CWU', 'CWL'):
        center = max(cw_tone, bandwidth // 2)
    elif mode in ('AM', 'FM', 'DGT-IQ', 'DGT-FM'):
        center = 0
    elif mode in ('DGT-U', 'DGT-L', 'FDV-U', 'FDV-L'):
        center = max(1500, bandwidth // 2)
    else:
        center = 300 + bandwidth // 2
    if mode in ('CWL', 'LSB', 'DGT-L', 'FDV-L'):
        center = -center
    return center
