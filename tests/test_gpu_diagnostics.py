"""The diagnostics that the receive path feeds (SURVEY.md section 8b lists them on the boundary): add_tone / AddTestTone
(quisk.c:3022-3034, 1194-1229), get_filter (4761-4848), measure_audio (3011-3020), measure_frequency (3000-3009, 4850-4920),
get_audio_graph (903-991) and the transform behind dft / idft (4922-5046), each against the oracle's restatement of the
cited lines."""
import ctypes as C

import numpy as np
import pytest

import _bind as B
from _rxsetup import TOL, configure_engine, make_oracle_rx, rx_filter
from quisk_amd import Engine
from quisk_amd.engine import TAP_PLAY
from quisk_amd.synth import make_iq

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [3, 4, 5])
def test_add_tone(mode):
    """The tone in front of the FFT staging and the mixer, plain / AM- / FM-modulated by rxMode; retuned mid-stream (the vectors
    carry on, quisk.c:1197-1198) and switched off again.  Audio in front of the AGC and the graph row against the oracle fed
    the same tone."""
    fs, blk, N, W = 192000, 9600, 4096, 512
    tune = 21000
    x = make_iq(blk * 10, fs, [tune], [mode], 17) * 0.01
    eng = Engine(fs, n_receivers=1, max_block=blk, fft_size=N, data_width=W)
    configure_engine(eng, [mode], [tune])
    ora = make_oracle_rx(fs, mode, tune)
    g = B.OracleGraph(N, W, fs)
    L = B.oracle()
    tone = (C.c_double * 4)()
    L.qko_tone_init(tone)
    peak, worst, toned = 0.0, 0.0, []
    for b in range(10):
        freq = {0: tune + 1000, 4: tune + 1700, 8: 0}.get(b)
        if freq is not None:
            eng.add_tone(freq); cur = freq
        part = np.array(x[b * blk:(b + 1) * blk], dtype=np.complex128)
        eng.process(part)
        L.qko_add_test_tone(tone, part, len(part), cur, fs, mode)
        toned.append(part.copy())
        pre, _ = ora.process_parts(part)
        got = eng.tap(TAP_PLAY, 0, planes=1)[0]
        assert len(got) == len(pre)
        peak = max(peak, float(np.max(np.abs(pre.real))))
        worst = max(worst, float(np.max(np.abs(got - pre.real))) / peak)
    assert worst < TOL, worst
    allx = np.concatenate(toned)
    g.add_frames(allx[:len(allx) // N * N])
    cnt, row, _ = eng.get_graph(1.0, 0.0)
    ocnt, orow, _ = g.release(1.0, 0.0)
    assert cnt == ocnt and cnt > 10
    assert np.max(np.abs(row - orow)) < 1e-7
    ora.close(); g.close(); eng.close()


@pytest.mark.parametrize("mode,W,N", [(3, 512, 4096), (0, 1000, 3000), (4, 256, 256)])
def test_get_filter(mode, W, N):
    fs = 48000
    eng = Engine(fs, n_receivers=1, max_block=4800, fft_size=N, data_width=W)
    fi, fq, bw, off = rx_filter(fs, mode)
    eng.set_mode(0, mode); eng.set_filters(0, fi, fq, bw, off)
    got = eng.get_filter(W)
    want = np.zeros(W)
    B.oracle().qko_get_filter(np.ascontiguousarray(fi), np.ascontiguousarray(fq), len(fi), W, N, want)
    ok = want > -139.0                                   # (the floor is a threshold on values around 1e-7: compare above it)
    assert ok.sum() > W // 20
    assert np.max(np.abs(got[ok] - want[ok])) < 1e-6
    eng.close()


@pytest.mark.parametrize("mode", [3, 1, 4, 5, 9])
def test_measure_audio(mode):
    fs, blk = 192000, 9600
    nblk = 90
    x = make_iq(blk * nblk, fs, [30000], [mode], 3)
    eng = Engine(fs, n_receivers=2, max_block=blk)
    configure_engine(eng, [mode, 3], [30000, -20000], bws=[6000 if mode == 9 else None, None])
    fi, fq, bw, _ = rx_filter(fs, mode, 6000 if mode == 9 else None)
    ora = B.OracleRx(fs, 48000, mode=mode, tune=30000, fi=fi, fq=fq, bandwidth=bw)
    L = B.oracle()
    seen = set()
    for b in range(nblk):
        if b == 45:
            assert eng.measure_audio(2) == pytest.approx(L.qko_rx_measure_audio(ora.h, 2), rel=1e-9)     # a longer average from here
        part = x[b * blk:(b + 1) * blk]
        eng.process(part); ora.process(part)
        got, want = eng.measure_audio(-1), L.qko_rx_measure_audio(ora.h, -1)
        assert got == pytest.approx(want, rel=1e-9, abs=1e-12), b
        seen.add(round(want, 6))
    assert len(seen) >= 3 and max(seen) > 0            # the value did change at the ends of several averaging stretches
    ora.close(); eng.close()


def test_measure_frequency():
    """a carrier 137.3 Hz above the receiver's tune: three half-bands, 12000-point frames at 6 ksps, the centroid of the peak"""
    fs, blk = 192000, 48000
    tune, offs = 25000, 137.3
    n = blk * 34
    t = np.arange(n) / fs
    rng = np.random.default_rng(4)
    x = np.round(2.0 ** 22 * np.exp(2j * np.pi * (tune + offs) * t) + 2.0 ** 10 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
    eng = Engine(fs, n_receivers=1, max_block=blk)
    configure_engine(eng, [3], [tune])
    L = B.oracle()
    orx = make_oracle_rx(fs, 3, tune)
    mf = L.qko_mfreq_new()
    eng.measure_frequency(2)                              # a value after every frame
    vals = []
    for b in range(34):
        part = x[b * blk:(b + 1) * blk]
        eng.process(part)
        w = np.array(part, dtype=np.complex128)
        L.qko_rx_mix(orx.h, w, len(w))
        m = L.qko_rx_decimate(orx.h, w, len(w))
        want = L.qko_measure_freq(mf, w[:m].copy(), m, 48000, 2, 0, tune)
        got = eng.measure_frequency(-1)
        assert got == pytest.approx(want, abs=1e-6), b
        vals.append(got)
    assert abs(vals[-1] - (tune + offs)) < 0.5 and len(set(vals)) >= 3
    L.qko_mfreq_delete(mf); orx.close(); eng.close()


def test_audio_graph_lsb():
    fs, blk, W = 192000, 9600, 512
    x = make_iq(blk * 24, fs, [-30000], [2], 8)
    eng = Engine(fs, n_receivers=1, max_block=blk, fft_size=4096, data_width=W)
    configure_engine(eng, [2], [-30000])
    ora = make_oracle_rx(fs, 2, -30000)
    L = B.oracle()
    ag = L.qko_audio_graph_new(W)
    L.qko_rx_set_audio_graph(ora.h, ag)
    rows = 0
    for b in range(24):
        part = x[b * blk:(b + 1) * blk]
        eng.process(part); ora.process(part)
        if b % 5 == 4:                                    # the GUI asks now and then; in between a finished graph waits
            got = eng.get_audio_graph()
            want = np.zeros(W)
            have = L.qko_audio_graph_get(ag, want)
            assert (got is not None) == bool(have), b
            if have:
                rows += 1
                assert np.max(np.abs(got - want)) < 1e-7, b
    assert rows >= 3
    L.qko_rx_set_audio_graph(ora.h, None)
    L.qko_audio_graph_delete(ag); ora.close(); eng.close()


@pytest.mark.parametrize("n", [1, 8, 31, 1000, 4096, 12000])
def test_fft_any(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    eng = Engine(48000, n_receivers=0, max_block=4096, fft_size=64, data_width=64)
    X = eng.fft_any(x)
    assert np.max(np.abs(X - np.fft.fft(x))) < 1e-11 * max(1.0, np.max(np.abs(X)))
    y = eng.fft_any(X, inverse=True) / n
    assert np.max(np.abs(y - x)) < 1e-11
    eng.close()
