"""CPU checks of the oracle's any-length DFT, sub-receiver graph and bandscope (quisk.c:4468-4600).
The reference transforms with FFTW3, which is absent here: the DFT itself is pinned to numpy's (double), the
binning and scaling arithmetic to an independent numpy restatement of the reference's loops."""
import numpy as np
import pytest

import _bind as B
import ctypes as C


@pytest.mark.parametrize("n", [6, 12, 100, 960, 1001, 4099, 9600, 4096, 3 * 5 * 7 * 11])
def test_oracle_fft_any_length(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    y = x.copy()
    B.oracle().qko_fft(y, n)
    ref = np.fft.fft(x)
    assert np.max(np.abs(y - ref)) < 1e-13 * np.max(np.abs(ref)) * max(1.0, np.log2(n))


@pytest.mark.parametrize("dw", [64, 100, 333])
def test_oracle_multirx_graph_against_numpy(dw):
    rng = np.random.default_rng(dw)
    N = 8 * dw
    x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)) * 1e6
    got = B.oracle_multirx_graph(x, dw)
    w = 0.5 + 0.5 * np.cos(2 * np.pi * (np.arange(N) - N // 2) / N)
    mag = np.abs(np.fft.fft(x * w))
    order = np.concatenate([np.arange(N // 2, N), np.arange(0, N // 2)])
    d1 = mag[order].reshape(dw, 8).sum(axis=1)
    want = np.maximum(20 * np.log10(d1) - 20 * (np.log10(N) + 31 * np.log10(2.0)), -200)
    assert np.max(np.abs(got - want)) < 1e-9


def test_oracle_bandscope_against_numpy():
    size, gw, clock = 2048, 640, 122880000
    rng = np.random.default_rng(1)
    ob = B.OracleBscope(size)
    acc = np.zeros(size // 2 + 2)
    w = 0.5 + 0.5 * np.cos(2 * np.pi * (np.arange(size) - size // 2) / size)
    mx = 0.0
    for _ in range(3):
        v = rng.uniform(-0.7, 0.7, size)
        ob.add(v)
        acc[:size // 2 + 1] += np.abs(np.fft.rfft(v * w))
        mx = max(mx, np.abs(v).max())
    n, row, lvl = ob.release(clock, 1.0, 0.0, gw)
    assert n == 3 and lvl == mx
    L = size // 2 + 1
    rate = clock / 2.0
    want = np.empty(gw)
    for i in range(gw):
        d1 = L / rate * (0.0 + i / gw * rate)
        d2 = L / rate * (0.0 + (i + 1) / gw * rate)
        j1, j2 = int(np.floor(d1)), int(np.floor(d2))
        if j1 == j2:
            s = (d2 - d1) * acc[j1]
        else:
            s = (j1 + 1 - d1) * acc[j1] + acc[j1 + 1:j2].sum() + (d2 - j2) * acc[min(j2, L)]
        s *= 1.0 / (L / gw) / 3 / size
        want[i] = -200.0 if s <= 1e-10 else 20 * np.log10(s)
    assert np.max(np.abs(row - want)) < 1e-9
    assert ob.release(clock, 1.0, 0.0, gw)[0] == 0
    ob.close()
