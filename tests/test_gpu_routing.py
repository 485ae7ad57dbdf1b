"""Second-receiver routing (split RX/TX and the multirx play channel, quisk.c:2410-2494, with Buffer2Chan's count
equalisation, quisk.c:1452-1487), per-channel squelch muting (quisk.c:2585-2601) and the key-up envelope
(quisk.c:2602-2612), against the oracle's receivers combined by the oracle's restatement of Buffer2Chan."""
import ctypes as C

import numpy as np
import pytest

import _bind as B
from _rxsetup import configure_engine, make_oracle_rx, TOL
from quisk_amd import Engine
from quisk_amd.synth import make_iq

pytestmark = pytest.mark.gpu


def _steady(n, fs, tunes, modes, seed):
    """carriers of equal strength well above the noise: the AGCs settle alike in oracle and engine"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    x = np.zeros(n, dtype=np.complex128)
    for f, m in zip(tunes, modes):
        side = -1000.0 if m in (0, 2, 8) else 1000.0
        car = np.exp(1j * (2 * np.pi * (f + side) * t + rng.uniform(0, 6.28)))
        if m == 4:
            car = car * (1 + 0.5 * np.sin(2 * np.pi * 700.0 * t))
        if m == 5:
            car = car * np.exp(1j * 3.0 * np.sin(2 * np.pi * 700.0 * t))
        x += 2.0 ** 25 * car
    x += 2.0 ** 10 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return np.round(x.real) + 1j * np.round(x.imag)


@pytest.mark.parametrize("method,modes", [(1, [3, 3]), (2, [3, 1]), (3, [2, 4]), (1, [1, 5])])
def test_stereo_route_against_buffer2chan(method, modes):
    """Receiver 0 and its partner in different chain classes deliver different counts per block (CW / SSB / AM / FM chains
    emit on different phases): the FIFOs equalise them."""
    fs = 192000
    tunes = [30000, -45000]
    sizes = [4800, 4801, 9999, 4800, 1, 4800, 7777, 4800, 4800]
    x = _steady(sum(sizes), fs, tunes, modes, 5)
    eng = Engine(fs, n_receivers=2, max_block=max(sizes))
    configure_engine(eng, modes, tunes)
    eng.set_stereo_route(0, 1, method)
    oras = [make_oracle_rx(fs, m, t) for m, t in zip(modes, tunes)]
    b2c = B.OB2C()
    L = B.oracle()
    pos, worst, total, uneven = 0, 0.0, 0, False
    for n in sizes:
        blk = x[pos:pos + n]; pos += n
        eng.process(blk)
        a = oras[0].process(blk).real.copy(); b = oras[1].process(blk).real.copy()
        uneven = uneven or len(a) != len(b)
        cap = max(len(a), len(b)) + 12000
        aa = np.zeros(cap); bb = np.zeros(cap); aa[:len(a)] = a; bb[:len(b)] = b
        nout = L.qko_buffer2chan(C.byref(b2c), aa, len(a), bb, len(b))
        want = {1: aa[:nout] + 1j * bb[:nout], 2: bb[:nout] + 1j * aa[:nout], 3: bb[:nout] + 1j * bb[:nout]}[method]
        got = eng.audio(0)
        assert len(got) == nout
        partner = eng.audio(1)                            # the partner keeps its own (mono) row
        assert len(partner) == len(b) and (len(b) == 0 or np.max(np.abs(partner.real - b)) <= 1e-6 * max(np.max(np.abs(b)), 1))
        if nout and pos > 40000:                          # past the AGC's start-up ramp (tests/_rxsetup.py, 3.)
            worst = max(worst, float(np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1.0)))
            total += nout
    assert total > 1000 and worst < TOL
    if 1 in modes and modes[0] != modes[1]:
        assert uneven                                     # CW beside another class: the FIFOs did have work
    eng.set_stereo_route(0, 1, 0)                         # off: mono again
    eng.process(x[:4800])
    got = eng.audio(0)
    assert np.array_equal(got.real, got.imag)
    for o in oras:
        o.close()
    eng.close()


def test_squelched_partner_is_silent_in_its_channel_only_and_key_up_envelope():
    fs = 96000
    tunes, modes = [20000, -30000], [3, 5]
    n = 4800
    x = _steady(12 * n, fs, [tunes[0]], [3], 9)           # nothing on the FM partner's frequency: its squelch closes
    eng = Engine(fs, n_receivers=2, max_block=n)
    configure_engine(eng, modes, tunes)
    eng.set_squelch(-30.0)
    eng.set_stereo_route(0, 1, 1)
    orx = make_oracle_rx(fs, 3, tunes[0])
    env, L = 1.0, B.oracle()
    for b in range(12):
        blk = x[b * n:(b + 1) * n]
        if b == 8:
            eng.key_up(); env = 0.0                       # quisk.c:2240-2241
        eng.process(blk)
        want = orx.process(blk)
        want = (want.real + 0j).astype(np.complex128)     # left = receiver 0; right = the squelched partner: silence
        env = L.qko_keyup_envelope(env, want, len(want), 48000)
        got = eng.audio(0)
        if b >= 2:
            assert eng.squelch_active(1) == 1
            assert np.all(got.imag == 0.0) and np.max(np.abs(got.real)) > 0
        if b >= 6:
            assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want))
        if b == 8:
            assert abs(got[0].real) < abs(got[200].real) * 0.05        # the ramp starts near zero ...
    assert env == 1.0                                                   # ... and is over after 5 ms
    orx.close(); eng.close()
