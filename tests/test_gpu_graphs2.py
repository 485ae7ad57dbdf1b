"""Spectra beyond the power-of-two display FFT, on the GPU against the oracle:
  * get_graph with Quisk's real sizes fft_size = data_width * fft_mult (quisk.py:3858-3890: factors 3, 5, 7 ...),
  * get_multirx_graph (quisk.c:4468-4530), 8 * multirx_data_width points,
  * get_bandscope (quisk.c:4532-4600), real ADC blocks of 512 * k samples.
The reference's transforms are FFTW3 (unpinned); the contract is the DFT in double: the oracle uses a mixed-radix
decimation in time, the engine Bluestein's identity over its power-of-two FFT -- two different algorithms.
Tolerance 1e-7 dB on every row (bins more than 200 dB down clamp to the reference's floor)."""
import numpy as np
import pytest

import _bind as B
from quisk_amd import Engine
from quisk_amd.synth import make_iq

pytestmark = pytest.mark.gpu
TOL_DB = 1e-7


@pytest.mark.parametrize("N,W", [(960, 960), (3000, 1000), (4200, 600), (9600, 1200), (1001, 143), (70, 35), (6, 6)])
def test_get_graph_mixed_radix_sizes(N, W):
    fs = 192000
    eng = Engine(fs, n_receivers=0, max_block=4 * N, fft_size=N, data_width=W)
    og = B.OracleGraph(N, W, fs)
    x = make_iq(4 * N, fs, [20000, -50000, 333], [3, 3, 3], 21)
    eng.process(x[:N + N // 3])         # frames straddle the block boundaries
    eng.process(x[N + N // 3:])
    og.add_frames(x)
    for zoom, deltaf in ((1.0, 0.0),):
        n1, r1, _ = eng.get_graph(zoom, deltaf)
        n2, r2, _ = og.release(zoom, deltaf)
        assert n1 == n2 == 4
        assert np.max(np.abs(r1 - r2)) < TOL_DB
    eng.close(); og.close()


def test_get_graph_mixed_radix_zoom_and_meter():
    fs, N, W = 96000, 2880, 960
    eng = Engine(fs, n_receivers=1, max_block=2 * N, fft_size=N, data_width=W)
    og = B.OracleGraph(N, W, fs)
    from _rxsetup import rx_filter
    fi, fq, bw, off = rx_filter(fs, 3)
    eng.set_mode(0, 3); eng.set_tune(0, 12000); eng.set_filters(0, fi, fq, bw, off)
    og.set_passband(12000, off, bw)
    x = make_iq(6 * N, fs, [12000, -30000], [3, 3], 5)
    for i in range(0, len(x), 2 * N):
        eng.process(x[i:i + 2 * N])
    og.add_frames(x)
    n1, r1, s1 = eng.get_graph(0.5, 3000.0)
    n2, r2, s2 = og.release(0.5, 3000.0)
    assert n1 == n2 == 6
    assert np.max(np.abs(r1 - r2)) < TOL_DB
    assert abs(s1 - s2) < TOL_DB
    eng.close(); og.close()


@pytest.mark.parametrize("dw", [64, 512, 100, 1200, 333, 2048])
def test_multirx_graph(dw):
    fs = 48000
    eng = Engine(fs, n_receivers=0, max_block=4096, fft_size=64, data_width=64)
    x = make_iq(8 * dw, fs, [5000, -12000], [3, 3], 7 + dw)
    got = eng.multirx_graph(x, dw)
    want = B.oracle_multirx_graph(x, dw)
    assert np.max(np.abs(got - want)) < TOL_DB
    # a second frame of another width re-plans; the first width again gives the same row
    other = eng.multirx_graph(make_iq(8 * 96, fs, [1000], [3], 1), 96)
    assert other.shape == (96,)
    assert np.array_equal(eng.multirx_graph(x, dw), got)
    eng.close()


def test_multirx_graph_silence_is_floor():
    eng = Engine(48000, n_receivers=0, max_block=4096, fft_size=64, data_width=64)
    assert np.all(eng.multirx_graph(np.zeros(8 * 200, dtype=np.complex128), 200) == -200.0)
    eng.close()


@pytest.mark.parametrize("size", [2048, 16384, 512 * 3, 512 * 12])
def test_bandscope(size):
    clock = 122880000
    eng = Engine(48000, n_receivers=0, max_block=4096, fft_size=64, data_width=64)
    ob = B.OracleBscope(size)
    eng.bandscope_init(size)
    rng = np.random.default_rng(size)
    t = np.arange(size)
    assert eng.get_bandscope(clock, 1.0, 0.0, 800)[0] == 0          # nothing yet
    for blk in range(3):
        v = 0.4 * np.sin(2 * np.pi * 0.0831 * t + blk) + 0.05 * np.sin(2 * np.pi * 0.31 * t) + 1e-3 * rng.standard_normal(size)
        eng.add_bscope_samples(v)
        ob.add(v)
    for zoom, deltaf, gw in ((1.0, 0.0, 800), ):
        n1, r1, l1 = eng.get_bandscope(clock, zoom, deltaf, gw)
        n2, r2, l2 = ob.release(clock, zoom, deltaf, gw)
        assert n1 == n2 == 3
        assert l1 == l2
        assert np.max(np.abs(r1 - r2)) < TOL_DB
    # zoomed and shifted, new average
    for blk in range(2):
        v = 0.2 * np.cos(2 * np.pi * 0.2 * t + blk) + 1e-3 * rng.standard_normal(size)
        eng.add_bscope_samples(v)
        ob.add(v)
    n1, r1, l1 = eng.get_bandscope(clock, 0.25, 5.0e6, 1111)
    n2, r2, l2 = ob.release(clock, 0.25, 5.0e6, 1111)
    assert n1 == n2 == 2 and l1 == l2
    assert np.max(np.abs(r1 - r2)) < TOL_DB
    eng.close(); ob.close()


def test_bandscope_int_samples_and_errors():
    size = 2048
    eng = Engine(48000, n_receivers=0, max_block=4096, fft_size=64, data_width=64)
    with pytest.raises(Exception):
        eng.add_bscope_samples(np.zeros(size))                       # not initialised (quisk.c:2851)
    eng.bandscope_init(size)
    rng = np.random.default_rng(3)
    ints = rng.integers(-30000, 30000, size).astype(np.int16)
    eng.add_bscope_samples_int(ints.astype("<i2").tobytes(), 2, big_endian=False)
    n1, r1, l1 = eng.get_bandscope(122880000, 1.0, 0.0, 500)
    eng.add_bscope_samples_int(ints.astype(">i2").tobytes(), 2, big_endian=True)
    n2, r2, l2 = eng.get_bandscope(122880000, 1.0, 0.0, 500)
    ob = B.OracleBscope(size)
    ob.add((ints.astype(np.int32) << 16).astype(np.float64) / 2147483647.0)       # left-justified, / CLIP32
    n3, r3, l3 = ob.release(122880000, 1.0, 0.0, 500)
    assert n1 == n2 == n3 == 1 and l1 == l2 == l3
    assert np.max(np.abs(r1 - r3)) < TOL_DB and np.array_equal(r1, r2)
    with pytest.raises(Exception):
        eng.add_bscope_samples(np.zeros(size + 1))                   # wrong block length (quisk.c:2867)
    eng.close(); ob.close()
