"""The audio-rate hooks between the detector and the audio interpolators: the SSB squelch with its one-frame delay line
(quisk.c:1022-1116, 993-1020), the automatic notch filter (dAutoNotch, quisk.c:730-899) and the graph-based IsSquelch
(quisk.c:1231-1252), against the oracle's restatements."""
import numpy as np
import pytest

import _bind as B
from _rxsetup import configure_engine, make_oracle_rx, rx_filter, TOL
from quisk_amd import Engine
from quisk_amd.engine import TAP_PLAY
from quisk_amd.synth import make_iq

pytestmark = pytest.mark.gpu


def _noise(n, rng, amp):
    return amp * (rng.standard_normal(n) + 1j * rng.standard_normal(n))


def test_ssb_squelch_opens_on_a_signal_and_closes_on_noise():
    """USB, LSB and AM receivers: band noise keeps the squelch closed (arithmetic / geometric mean ratio 0.57 < level * 0.005),
    a tone opens it for one second; muted blocks, the squelch flag and the delayed audio follow the oracle."""
    fs, blk, nblk = 192000, 9600, 60
    modes, tunes = [3, 2, 4], [30000, -41000, 60000]
    rng = np.random.default_rng(21)
    t = np.arange(blk * nblk)
    key = np.repeat(((np.arange(nblk) // 12) % 2 == 1), blk)                  # 0.6 s of noise, 0.6 s with the signals, ...
    x = np.zeros(len(t), dtype=np.complex128)
    for m, f in zip(modes, tunes):
        side = -1000.0 if m == 2 else 1000.0
        car = np.exp(2j * np.pi * ((f + side) / fs) * t)
        if m == 4:
            car = np.exp(2j * np.pi * (f / fs) * t) * (1 + 0.5 * np.sin(2 * np.pi * 1000 / fs * t))
        x += 2.0 ** 22 * key * car
    x = np.round(x + _noise(len(t), rng, 2.0 ** 14))
    eng = Engine(fs, n_receivers=3, max_block=blk)
    configure_engine(eng, modes, tunes)
    oras = [make_oracle_rx(fs, m, tn) for m, tn in zip(modes, tunes)]
    level = 160                                                                # threshold 0.8
    eng.set_ssb_squelch(1, level)
    for o in oras:
        o.set_ssb_squelch(1, level)
    flags = [[], [], []]
    for b in range(nblk):
        part = x[b * blk:(b + 1) * blk]
        eng.process(part)
        for r, o in enumerate(oras):
            pre_o, fin_o = o.process_parts(part)
            want = o.process(part[:0])                                         # (nothing: keeps the interface exercised)
            got = eng.audio(r)
            pre_g = eng.tap(TAP_PLAY, r, planes=1)[0]
            assert len(got) == len(fin_o) == len(pre_g)
            scale = max(float(np.max(np.abs(pre_o))), 1.0)
            assert float(np.max(np.abs(pre_g - pre_o.real))) / scale < TOL, (b, r)       # the delayed audio in front of the AGC
            assert eng.squelch_active(r) == o.squelch_active(), (b, r)
            flags[r].append(o.squelch_active())
            if o.squelch_active():
                assert not np.any(got), (b, r)
    for r in range(3):
        assert 0 < sum(flags[r]) < nblk, (r, flags[r])                          # closed in some blocks, open in others
        assert flags[r][0] == 0                                                 # the first block only builds the plan
    eng.set_ssb_squelch(0, 0)
    eng.process(x[:blk])
    assert all(eng.squelch_active(r) == 0 for r in range(3))
    for o in oras:
        o.close()
    eng.close()


@pytest.mark.parametrize("mode,tune,fs", [(3, 30000, 192000), (1, -20000, 192000), (4, 50000, 192000), (5, 10000, 96000), (7, 25000, 192000)])
def test_auto_notch_against_the_oracle(mode, tune, fs):
    """A steady heterodyne beside a weaker, moving tone: the notch finds the steady line after a few 2048-sample frames and
    cuts it (and a second one); the audio in front of the AGC follows the oracle's dAutoNotch through every frame, filter
    redesign included.  CW leaves its own sidetone alone (rit_freq)."""
    blk, nblk = 48000, 14
    rng = np.random.default_rng(33)
    t = np.arange(blk * nblk)
    if mode == 4:
        x = 2.0 ** 24 * np.exp(2j * np.pi * (tune / fs) * t) * (1 + 0.4 * np.sin(2 * np.pi * 1500 / fs * t) + 0.2 * np.sin(2 * np.pi * (600 + 200 * t / len(t)) / fs * t))
    elif mode == 5:
        x = 2.0 ** 24 * np.exp(1j * (2 * np.pi * (tune / fs) * t + 1.5 * np.sin(2 * np.pi * 1300 / fs * t) + 0.5 * np.sin(2 * np.pi * 500 / fs * t)))
    else:
        sgn = -1.0 if mode in (0, 2, 8) else 1.0
        f1, f2 = (650.0, 900.0) if mode == 1 else (1500.0, 700.0)
        x = 2.0 ** 24 * np.exp(2j * np.pi * ((tune + sgn * f1) / fs) * t) + 2.0 ** 22 * np.exp(2j * np.pi * ((tune + sgn * f2) / fs) * t * (1 + 1e-7 * t))
    x = np.round(x + _noise(len(t), rng, 2.0 ** 12))
    eng = Engine(fs, n_receivers=2, max_block=blk)
    configure_engine(eng, [mode, 3], [tune, -70000 if fs > 150000 else -30000])
    o = make_oracle_rx(fs, mode, tune)
    rit = 650 if mode == 1 else 0
    eng.set_rit_freq(rit)
    eng.set_auto_notch(0, 1)
    o.set_auto_notch(1, rit)
    plain = make_oracle_rx(fs, mode, tune)                                      # the same receiver without the notch
    worst, cut = 0.0, 0.0
    for b in range(nblk):
        part = x[b * blk:(b + 1) * blk]
        eng.process(part)
        pre_o, _ = o.process_parts(part)
        pre_p, _ = plain.process_parts(part)
        pre_g = eng.tap(TAP_PLAY, 0, planes=1)[0]
        assert len(pre_g) == len(pre_o)
        scale = max(float(np.max(np.abs(pre_p))), 1.0)
        worst = max(worst, float(np.max(np.abs(pre_g - pre_o.real))) / scale)
        if b >= nblk - 3:
            cut = max(cut, float(np.sqrt(np.mean(pre_o.real ** 2)) / max(np.sqrt(np.mean(pre_p.real ** 2)), 1e-300)))
    print("auto-notch mode %d: worst pre-AGC error %.2e; notched / plain RMS at the end %.3f" % (mode, worst, cut))
    assert worst < TOL
    # (the reference's overlap-save path has a gain of 2048 / 102 = 20.08, "empirical", quisk.c:895: with the dominant line cut
    # the level still falls)
    if mode == 3:
        assert cut < 0.9 * 20.08
    o.close(); plain.close(); eng.close()


def test_is_squelch_reads_the_released_graph_row():
    fs, N, W = 192000, 4096, 1024
    eng = Engine(fs, n_receivers=1, max_block=N, fft_size=N, data_width=W)
    fi, fq, bw, off = rx_filter(fs, 3)
    eng.set_mode(0, 3); eng.set_tune(0, 20000); eng.set_filters(0, fi, fq, bw, off)
    assert eng.is_squelch(20000) == 1                                          # no row released yet: meter == 0
    x = make_iq(N, fs, [20000], [3], 3)
    eng.process(x)
    n, row, _ = eng.get_graph(1.0, 0.0)
    assert n == 1
    for level in (-150.0, -20.0):
        eng.set_squelch(level)
        for freq in (21000, -60000, 95990):
            ibw = max(5000 * W // fs, 1)
            i1 = int(freq * W / fs + W / 2.0 - ibw / 2.0 + 0.5)
            meter = float(np.sum(row[i1:i1 + ibw])) / ibw if (i1 >= 0 and i1 + ibw < W) else 0.0
            assert eng.is_squelch(freq) == int(meter == 0 or meter < level), (level, freq)
    eng.close()
