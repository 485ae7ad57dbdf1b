"""Input conditioning on the full-rate stream -- DCremove (sound.c:165-230), spectrum inversion (quisk.c:2315-2320) and
the noise blanker (quisk.c:615-720) -- on the GPU against the oracle, seen through what follows it: the front end's output (mixer + half-bands; tolerance 1e-9
of the peak), the demodulated audio and the spectrum row.  The one-pole DC blocker is one serial chain over the whole
stream in the reference and a segmented scan here (different rounding order, alpha < 1)."""
import numpy as np
import pytest

import _bind as B
from _rxsetup import PairRunner, TOL
from quisk_amd import Engine
from quisk_amd.engine import TAP_FRONTEND
from quisk_amd.synth import make_iq

pytestmark = pytest.mark.gpu


def _run(fs, mode, tune, sizes, dc_bw, invert, seed, offset=(3.0e6 + 1.5e6j), nb=0, pulses=None):
    x = make_iq(sum(sizes), fs, [tune, -tune // 2], [mode, mode], seed) + offset
    if pulses is not None:
        x = x + pulses(len(x))
    eng = Engine(fs, n_receivers=1, max_block=max(sizes), fft_size=4096, data_width=1024)
    pr = PairRunner(eng, [mode], [tune])
    eng.set_dc_remove(dc_bw)
    eng.set_invert_spectrum(invert)
    eng.set_noise_blanker(nb)
    cond = B.OracleConditioner(fs, dc_bw, invert, nb)
    og = B.OracleGraph(4096, 1024, fs)
    from _rxsetup import rx_filter
    _, _, bw, off = rx_filter(fs, mode)
    og.set_passband(tune, off, bw)
    pos, frames = 0, []
    for s in sizes:
        blk = x[pos:pos + s]
        pos += s
        yb = cond(blk)
        frames.append(yb)
        # PairRunner.step feeds the engine `blk` and the oracle receivers the conditioned block
        eng.process(blk)
        for r, o in enumerate(pr.oras):
            pre_o, fin_o = o.process_parts(yb)
            pre_g = eng.tap(4, r, planes=1)[0]
            assert len(pre_g) == len(pre_o)
            if len(pre_o):
                pr.samples += len(pre_o)
                pr.peak[r] = max(pr.peak[r], float(np.max(np.abs(pre_o.real))))
                pr.pre_err = max(pr.pre_err, float(np.max(np.abs(pre_g - pre_o.real))) / max(pr.peak[r], 1.0))
    allx = np.concatenate(frames)
    nfr = len(allx) // 4096
    og.add_frames(allx[:nfr * 4096])
    n1, row1, sm1 = eng.get_graph(1.0, 0.0)
    n2, row2, sm2 = og.release(1.0, 0.0)
    assert n1 == n2 == nfr
    assert np.max(np.abs(row1 - row2)) < 1e-7 and abs(sm1 - sm2) < 1e-7
    assert pr.samples > 0 and pr.pre_err < TOL, pr.pre_err
    pr.close(); og.close()


@pytest.mark.parametrize("dc_bw,invert", [(100, False), (100, True), (0, True), (20, False), (1000, True)])
def test_dc_blocker_and_inversion(dc_bw, invert):
    _run(192000, 3, 21000, [9600, 4097, 12000, 999, 9600, 9600, 2048], dc_bw, invert, 17)


def test_dc_blocker_long_blocks_at_1536k():
    """Blocks longer than many scan segments, ragged lengths (a partial last segment and lane)."""
    _run(1536000, 3, 300000, [65536, 100001, 1023, 1025, 65536], 100, False, 3)


def test_dc_average_mode():
    """dc_remove_bw == 1: nothing is subtracted during the first second, then the average of each two-second stretch."""
    fs = 48000
    _run(fs, 3, 5000, [12000] * 18, 1, False, 9)


def test_dc_blocker_removes_the_offset():
    fs = 192000
    eng = Engine(fs, n_receivers=0, max_block=65536, fft_size=4096, data_width=4096)
    eng.set_dc_remove(100)
    x = make_iq(65536 * 2, fs, [30000], [3], 1) + (5.0e7 + 0j)
    eng.process(x[:65536]); eng.get_graph()
    eng.process(x[65536:])
    n, row, _ = eng.get_graph()
    dc_bin = row[2048]
    eng2 = Engine(fs, n_receivers=0, max_block=65536, fft_size=4096, data_width=4096)
    eng2.process(x[:65536]); eng2.get_graph()
    eng2.process(x[65536:])
    _, row2, _ = eng2.get_graph()
    assert row2[2048] - dc_bin > 40.0            # the carrier at DC is gone (more than 40 dB down)
    assert abs(row[2048 + 640] - row2[2048 + 640]) < 0.05      # the 30 kHz tone is untouched
    eng.close(); eng2.close()


def _pulse_train(fs, seed):
    """Impulse noise: single-sample spikes, bursts that flicker above and below the limit, pulses closer together than the
    blanker's half window (ramps that overlap), all far above the signal."""
    hw = int(fs * 500e-6 + 0.5)

    def make(n):
        rng = np.random.default_rng(seed)
        p = np.zeros(n, complex)
        pos = 5 * hw
        while pos < n - 4 * hw:
            kind = rng.integers(0, 4)
            amp = 2.0 ** rng.uniform(29, 31)
            if kind == 0:                                   # one spike
                p[pos] += amp
            elif kind == 1:                                 # a burst with gaps: several pulse starts within one window
                ln = int(rng.integers(2, hw))
                p[pos:pos + ln] += amp * (rng.random(ln) < 0.6) * np.exp(2j * np.pi * rng.random(ln))
            elif kind == 2:                                 # two spikes less than a half window apart
                p[pos] += amp
                p[pos + int(rng.integers(1, hw))] += amp * 1j
            else:                                           # a solid burst longer than the half window
                p[pos:pos + hw + int(rng.integers(1, hw))] += amp
            pos += int(rng.integers(hw // 2, 12 * hw))
        return p
    return make


@pytest.mark.parametrize("level", [1, 2, 3])
def test_noise_blanker_192k(level):
    """quisk_noise_blanker = 1, 2, 3 on impulse noise, ragged blocks (also shorter than the blanker's delay line); seen
    through the front end (1e-9 of the peak) and the spectrum row: one pulse flag decided differently would show at once."""
    fs = 192000
    _run(fs, 3, 21000, [9600, 4097, 100, 12000, 999, 37, 9600, 9600, 2048], 0, False, 23 + level, nb=level,
         pulses=_pulse_train(fs, 100 + level))


def test_noise_blanker_with_dc_blocker_and_inversion_at_1536k():
    fs = 1536000
    _run(fs, 3, 300000, [65536, 100001, 1023, 65536], 100, True, 5, nb=2, pulses=_pulse_train(fs, 7))


def test_noise_blanker_at_48k_tiny_blocks():
    """The smallest half window (24 samples at 48 ksps: the history is one segment) and blocks of a few samples."""
    fs = 48000
    _run(fs, 3, 5000, [4800, 1, 2, 63, 64, 65, 4800, 333, 4800, 4800], 0, False, 31, nb=3, pulses=_pulse_train(fs, 41))


def test_noise_blanker_on_quiet_input_only_delays():
    """No pulses: behind its start-up the blanker is a delay line of 3 x 500 us."""
    fs = 192000
    eng = Engine(fs, n_receivers=1, max_block=19200)
    from _rxsetup import configure_engine
    configure_engine(eng, [3], [21000])
    eng2 = Engine(fs, n_receivers=1, max_block=19200)
    configure_engine(eng2, [3], [21000])
    eng.set_noise_blanker(1)
    x = make_iq(19200 * 3, fs, [21000], [3], 2)
    d = 3 * 96
    xd = np.concatenate([np.zeros(d, complex), x])[:len(x)]
    for i in range(3):
        eng.process(x[i * 19200:(i + 1) * 19200])
        eng2.process(xd[i * 19200:(i + 1) * 19200])
        a, b = eng.tap(TAP_FRONTEND, 0), eng2.tap(TAP_FRONTEND, 0)
        # (the very first samples meet an empty ring, whose mean is small: they count as pulses and are blanked)
        assert a.shape == b.shape and (i == 0 or np.array_equal(a, b))
    eng.close(); eng2.close()
