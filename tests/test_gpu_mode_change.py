"""A receiver whose mode (and with it the receive filter) changes mid-stream, against ONE oracle receiver whose mode is
switched the same way.  The reference keeps one set of filter objects per bank whatever the mode (Storage[bank],
quisk.c:1554-1574, 1730-1747; the band-pass rings quisk.c:1124-1127, 1161-1165; HalfBand7..9 / Agc1 / cFracDecim's statics in
quisk_process_samples): a mode that comes back finds its filters holding the samples of its last visit, objects shared by
two modes are handed over warm, counters stand where they stopped.  Parity from the first block behind every switch:
pre-AGC within 1e-9 of the running peak, the AGC bit-exact on identical input, equal counts."""
import numpy as np
import pytest

from _rxsetup import AgcTwin, TOL, make_oracle_rx, rx_filter
from quisk_amd import Engine
from quisk_amd.engine import TAP_PLAY
from quisk_amd.synth import make_iq

pytestmark = pytest.mark.gpu


def _run(fs, schedule, sizes, playback=48000, seed=5, others=()):
    """schedule: [(mode, n_blocks)]; others: modes of further receivers that never change (they keep their classes populated)"""
    tune = int(0.11 * fs)
    schedule = [(e + (None,))[:3] for e in schedule]
    nblk = sum(k for _, k, _ in schedule)
    x = make_iq(sum(sizes[i % len(sizes)] for i in range(nblk)), fs, [tune], [3], seed)
    nrx = 1 + len(others)
    eng = Engine(fs, n_receivers=nrx, playback_rate=playback, max_block=max(sizes))
    ora = make_oracle_rx(fs, schedule[0][0], tune, playback)
    twin = AgcTwin(playback, schedule[0][0])
    for r, m in enumerate(others, start=1):
        fi, fq, bw, off = rx_filter(fs, m)
        eng.set_mode(r, m); eng.set_tune(r, -tune); eng.set_filters(r, fi, fq, bw, off)
    eng.set_tune(0, tune)
    pos, b, peak, worst, total = 0, 0, 0.0, 0.0, 0
    for mode, k, want_bw in schedule:
        fi, fq, bw, off = rx_filter(fs, mode, want_bw)
        eng.set_mode(0, mode); eng.set_filters(0, fi, fq, bw, off)              # quisk.py OnBtnMode -> set_rx_mode, set_filters
        ora.L.qko_rx_set_mode(ora.h, mode)
        assert ora.L.qko_rx_set_filter(ora.h, np.ascontiguousarray(fi), np.ascontiguousarray(fq), len(fi), bw) == 0
        twin.fm = int(mode in (5, 13))
        for _ in range(k):
            n = sizes[b % len(sizes)]; blk = x[pos:pos + n]; pos += n; b += 1
            eng.process(blk)
            pre_o, fin_o = ora.process_parts(blk)
            pre_g = eng.tap(TAP_PLAY, 0, planes=1)[0]
            fin_g = eng.audio(0)
            assert len(pre_g) == len(pre_o) == len(fin_g), (b, mode, len(pre_g), len(pre_o))
            if not len(pre_o):
                continue
            total += len(pre_o)
            peak = max(peak, float(np.max(np.abs(pre_o.real))))
            err = float(np.max(np.abs(pre_g - pre_o.real))) / peak
            worst = max(worst, err)
            assert err < TOL, "block %d (mode %d): pre-AGC error %.3e" % (b, mode, err)
            t = twin.run(pre_g)
            assert np.array_equal(t.view(np.float64), fin_g.view(np.float64)), "block %d (mode %d): AGC not bit-exact" % (b, mode)
    ora.close(); eng.close()
    return worst, total


def test_usb_am_fm_usb_one_receiver():
    """USB -> AM -> FM -> USB -> LSB -> AM: filtDecim48to24 goes from the 24 k stream (SSB) to the 48 k stream (AM) and back, the
    dRxFilterOut ring from AM to FM, HalfBand7 from SSB to AM, HalfBand5 / the cRxFilterOut ring / filtAudio24p4 wait out the
    AM and FM stretches with their last SSB samples in them."""
    worst, total = _run(192000, [(3, 3), (4, 3), (5, 3), (3, 3), (2, 2), (4, 2)], [9600, 4801, 9599, 7, 9600])
    assert total > 10 * 2000
    print("mode changes: worst pre-AGC error %.2e over %d samples" % (worst, total))


def test_filter_length_changes_and_cw():
    """The band-pass rings (bufFilterI/Q, bufFilterC, quisk.c:1124-1127, 1161-1165) are rings of the CURRENT filter length
    with a running index: when the length changes -- every mode change does that in Quisk, and every bandwidth button --
    the cells are read in another order, the index is reset only if it lies beyond the new length (quisk.c:1139-1140,
    1176-1177), and cells beyond a shorter length come back when the filter grows again.  USB 164 taps -> CWU 194 -> USB 164
    -> USB 2 kHz (230) -> USB 3 kHz (153) -> USB 1.8 kHz (108) -> USB 230 again -> AM 115 -> AM 10 kHz (92) -> FM 16 kHz (115)
    -> CWL 200 Hz (770 taps) -> USB; blocks short enough that a ring is caught in every position."""
    sched = [(3, 2), (1, 2), (3, 2), (3, 2, 2000), (3, 2, 3000), (3, 1, 1800), (3, 2, 2000), (4, 2), (4, 2, 10000), (5, 2),
             (0, 2, 200), (3, 2)]
    worst, total = _run(192000, sched, [9600, 4801, 1203, 9599, 7, 2222], seed=8)
    assert total > 15000
    print("length changes: worst pre-AGC error %.2e over %d samples" % (worst, total))


def test_mode_change_at_a_rate_with_the_fractional_decimator():
    """250 ksps: the 24/25 resampler and cFracDecim behind the demodulator; the position and the four samples of cFracDecim
    are statics of their own (quisk.c:575-576) and go along from class to class; play rate 96 k: HalfBand7 of
    quisk_process_samples too."""
    worst, total = _run(250000, [(3, 2), (5, 2), (4, 2), (3, 2)], [12500, 6001, 12500], playback=96000, seed=6)
    assert total > 6 * 4000


def test_mode_change_beside_receivers_that_stay():
    """Receiver 0 changes between classes that other receivers populate (their audio must not move); its filters' contents
    go along, its counters are the class's: compared with the oracle only where the phases agree is not possible in
    general, so here the OTHER receivers are checked to be bit-identical to an engine in which receiver 0 never moved."""
    fs, blk = 192000, 9600
    x = make_iq(blk * 8, fs, [21120, -21120], [3, 4], 9)
    outs = []
    for moving in (False, True):
        eng = Engine(fs, n_receivers=3, max_block=blk)
        for r, m in enumerate([3, 4, 3]):
            fi, fq, bw, off = rx_filter(fs, m)
            eng.set_mode(r, m); eng.set_tune(r, 21120 if r != 1 else -21120); eng.set_filters(r, fi, fq, bw, off)
        got = []
        for b in range(8):
            if moving and b in (2, 4, 6):
                m = {2: 4, 4: 5, 6: 3}[b]
                fi, fq, bw, off = rx_filter(fs, m)
                eng.set_mode(0, m); eng.set_filters(0, fi, fq, bw, off)
            eng.process(x[b * blk:(b + 1) * blk])
            got.append([eng.audio(r) for r in (1, 2)])
        outs.append(got)
        eng.close()
    for a, b in zip(*outs):
        for u, v in zip(a, b):
            assert np.array_equal(u.view(np.float64), v.view(np.float64))
