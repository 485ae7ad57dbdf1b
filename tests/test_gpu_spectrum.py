"""Display spectrum (get_graph, quisk.c:4602-4759) on the GPU against the oracle: window, FFT, magnitude average,
S-meter, pixel binning, dB scaling.  The reference's FFT is FFTW3 (unpinned); the contract is the DFT in double.
Tolerance: 1e-7 dB absolute on the graph row (1e-8 relative on a bin's averaged magnitude, which leaves room for
the FFT's own rounding ~1e-15 N^(1/2) of the frame peak on bins 120 dB below it) and on the S-meter."""
import numpy as np
import pytest

import _bind as B
from _rxsetup import rx_filter
from quisk_amd import Engine
from quisk_amd.synth import make_iq

pytestmark = pytest.mark.gpu
TOL_DB = 1e-7


def _pair(fs, N, W, nrx=0):
    eng = Engine(fs, n_receivers=nrx, max_block=max(N, 65536), fft_size=N, data_width=W)
    og = B.OracleGraph(N, W, fs)
    return eng, og


def _cmp(eng, og, zoom=1.0, deltaf=0.0):
    n1, row1, sm1 = eng.get_graph(zoom, deltaf)
    n2, row2, sm2 = og.release(zoom, deltaf)
    assert n1 == n2 > 0
    assert np.max(np.abs(row1 - row2)) < TOL_DB
    assert abs(sm1 - sm2) < TOL_DB
    return n1


@pytest.mark.parametrize("N", [64, 512, 1024, 2048, 4096])
def test_small_fft_sizes(N):
    fs = 192000
    eng, og = _pair(fs, N, N)
    x = make_iq(3 * N, fs, [20000, -50000], [3, 3], 11)
    eng.process(x)
    og.add_frames(x)
    assert _cmp(eng, og) == 3
    eng.close(); og.close()


def test_c1_graph_4096_counts_zoom_and_smeter():
    """Config 1's graph: fft_size = data_width = 4096 at 192 ksps, one USB receiver (its pass band feeds the S-meter);
    averaged over 1 and 7 frames, fed in 960-sample blocks; zoom 1, zoom 0.5 with an offset (the in-place aliasing case)."""
    fs, N = 192000, 4096
    eng, og = _pair(fs, N, N, nrx=1)
    fi, fq, bw, off = rx_filter(fs, 3)
    tune = 19200
    eng.set_mode(0, 3); eng.set_tune(0, tune); eng.set_filters(0, fi, fq, bw, off)
    og.set_passband(tune, off, bw)
    x = make_iq(9 * N, fs, [tune], [3], 1001)
    feed = lambda lo, hi: [eng.process(x[i:min(i + 960, hi)]) for i in range(lo, hi, 960)]
    feed(0, N); og.add_frames(x[:N])
    assert _cmp(eng, og) == 1
    feed(N, 8 * N); og.add_frames(x[N:8 * N])
    assert eng.graph_pending() == 7
    assert _cmp(eng, og, zoom=0.5, deltaf=5000.0) == 7
    feed(8 * N, 9 * N); og.add_frames(x[8 * N:9 * N])
    assert _cmp(eng, og, zoom=0.25, deltaf=-12345.0) == 1
    assert eng.get_graph()[0] == 0          # nothing pending: like get_graph returning None
    eng.close(); og.close()


@pytest.mark.parametrize("N,W", [(8192, 1024), (65536, 2048), (1 << 20, 4096)])
def test_four_step_sizes(N, W):
    """Sizes above one LDS tile (four-step FFT), spectrum-only engine; config 3 is N = 2^20, W = 4096 at 61.44 Msps."""
    fs = 61440000
    eng, og = _pair(fs, N, W)
    nfr = 2
    x = make_iq(nfr * N, fs, [5000000, -20000000, 123456], [3, 3, 3], 1003)
    # ragged feeding: frames straddle block boundaries
    pos, sizes = 0, [N // 2 + 17, N, 1000]
    while pos < len(x):
        for s in sizes:
            if pos >= len(x):
                break
            eng.process(x[pos:pos + s]); pos += s
    og.add_frames(x)
    assert _cmp(eng, og) == nfr
    x2 = make_iq(N, fs, [5000000], [3], 5)
    eng.process(x2[:N // 2]); eng.process(x2[N // 2:]); og.add_frames(x2)
    assert _cmp(eng, og, zoom=0.5, deltaf=1.0e6) == 1
    eng.close(); og.close()


@pytest.mark.parametrize("N,W", [(8192, 1024), (131072, 2048), (1 << 20, 4096)])
def test_four_step_batched_frames(N, W):
    """Blocks that hold several whole frames: the four-step passes take up to eight frames per launch, add their magnitudes
    up in registers and write the average once (a launch that starts a new average overwrites it, a later one adds); the
    S-meter sums the frames' pass-band bins frame after frame in the reference's order (one USB receiver supplies the band)."""
    fs = 1536000 if N < (1 << 20) else 61440000
    eng = Engine(fs, n_receivers=1, max_block=5 * N, fft_size=N, data_width=W)
    og = B.OracleGraph(N, W, fs)
    fi, fq, bw, off = rx_filter(fs, 3)
    tune = int(0.11 * fs)
    eng.set_mode(0, 3); eng.set_tune(0, tune); eng.set_filters(0, fi, fq, bw, off)
    og.set_passband(tune, off, bw)
    x = make_iq(13 * N, fs, [tune, -int(0.3 * fs)], [3, 3], 1003)
    eng.process(x[:5 * N])                  # five frames in one launch pair: a new average
    eng.process(x[5 * N:8 * N])             # three more: added to it
    og.add_frames(x[:8 * N])
    assert eng.graph_pending() == 8
    assert _cmp(eng, og) == 8
    eng.process(x[8 * N:8 * N + 1000])      # a ragged start, then frames from the staging buffer and straight from the block
    eng.process(x[8 * N + 1000:13 * N])
    og.add_frames(x[8 * N:13 * N])
    assert _cmp(eng, og, zoom=0.5, deltaf=fs / 16.0) == 5
    eng.close(); og.close()


@pytest.mark.parametrize("N,W,blocks", [(4096, 1024, 3), (16384, 2048, 4), (3000, 600, 5)])
def test_scan_blocks_stitching(N, W, blocks):
    """change_scan (quisk.c:3865-3872): frames numbered 0 .. scan_blocks-1 are stitched into one row (quisk.c:4682-4702) -- the
    valid centre part of each, scan_blocks bins summed into one, assigned to the block's slot; the count advances once per
    completed scan; the S-meter window follows quisk.c:4660-4663.  Single-workgroup, four-step and Bluestein sizes."""
    fs = 192000
    eng = Engine(fs, n_receivers=1, max_block=2 * N, fft_size=N, data_width=W)
    og = B.OracleGraph(N, W, fs)
    fi, fq, bw, off = rx_filter(fs, 3)
    eng.set_mode(0, 3); eng.set_tune(0, 5000); eng.set_filters(0, fi, fq, bw, off)
    og.set_passband(5000, off, bw)
    args = (blocks, fs, 0.84, 7000000, int(fs * 0.84), 7050000 - 7100000, 7100000)     # blocks, rate, valid, vfo0, deltaf, tx tune, vfo_screen
    eng.change_scan(*args); og.change_scan(*args)
    nfr = 2 * blocks + 1                                      # two whole scans and the first block of a third
    x = make_iq(nfr * N, fs, [20000, -50000, 70000], [3, 3, 4], 41)
    eng.process(x[:N]); eng.process(x[N:N + N // 2]); eng.process(x[N + N // 2:3 * N])
    for f in range(3, nfr):
        eng.process(x[f * N:(f + 1) * N])
    og.add_frames(x)
    assert eng.graph_pending() == 2
    assert _cmp(eng, og) == 2
    eng.change_scan(0); og.change_scan(0, fs, 0.84, 0, 0)     # back to plain averaging
    eng.process(x[:2 * N]); og.add_frames(x[:2 * N])
    assert _cmp(eng, og, zoom=0.5, deltaf=1000.0) == 2
    eng.close(); og.close()


@pytest.mark.parametrize("N,W", [(4096, 4096), (1 << 16, 1024)])
def test_marked_release(N, W):
    """qk_graph_mark: the row of the frames in front of the mark is released behind them and collected later, while the
    frames behind the mark go into the next average (single-workgroup and four-step sizes; zoom with the in-place case)."""
    fs = 192000
    eng, og = _pair(fs, N, W)
    x = make_iq(6 * N, fs, [20000, -50000], [3, 3], 17)
    feed = lambda lo, hi: [eng.process(x[i:i + N]) for i in range(lo, hi, N)]
    feed(0, 2 * N)
    assert eng.graph_mark(0.5, 3000.0) == 2
    assert eng.graph_pending() == 0
    with pytest.raises(Exception):
        eng.graph_mark(1.0, 0.0)                   # one row at a time
    feed(2 * N, 5 * N)                             # queued behind the mark: the next average
    assert eng.graph_pending() == 3
    og.add_frames(x[:2 * N])
    assert _cmp(eng, og, zoom=0.5, deltaf=3000.0) == 2      # (the arguments of the collecting call are not used)
    og.add_frames(x[2 * N:5 * N])
    assert _cmp(eng, og) == 3
    assert eng.graph_mark(1.0, 0.0) == 0           # nothing waiting
    feed(5 * N, 6 * N)
    og.add_frames(x[5 * N:])
    assert _cmp(eng, og) == 1
    eng.close(); og.close()
