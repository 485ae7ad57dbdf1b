"""Parity at the widths that bench.py times (BASELINE configs[1] and the per-GPU slice of configs[3]), the overlapped
four-stream pipeline that bench.py drives (back-to-back qk_process_samples_device calls with no synchronisation in
between), and the ping-pong parity across qk_reset_stats."""
import numpy as np
import pytest

from _hip import DevBuf
from _rxsetup import PairRunner, configure_engine
from quisk_amd import Engine
from quisk_amd.engine import TAP_PLAY
from quisk_amd.synth import make_iq, tune_offsets, mode_mix

pytestmark = pytest.mark.gpu


def _wide(fs, nrx, mix, sizes, seed, carriers):
    tunes = tune_offsets(nrx, fs)
    modes = mode_mix(nrx, mix)
    step = max(1, nrx // carriers)
    x = make_iq(sum(sizes), fs, tunes[::step], modes[::step], seed)
    eng = Engine(fs, n_receivers=nrx, max_block=max(sizes))
    pr = PairRunner(eng, modes, tunes)
    pos = 0
    for n in sizes:
        pr.step(x[pos:pos + n])
        pos += n
    pr.close()
    rep = pr.check()
    print("pre-AGC %.2e; end to end %r" % (pr.pre_err, rep))
    assert rep["receivers"] == nrx
    return pr


def test_config2_width_64_usb_receivers():
    """BASELINE configs[1] at its full width AND at bench.py's geometry: 64 USB receivers on one 1.536 Msps stream, blocks of
    2^20 samples (what bench.py times; the front end's span partition depends on receiver count and block length both) and
    ragged ones; every receiver against its own oracle."""
    sizes = [1 << 20, 1 << 20, 100001, 7, 1 << 20]
    pr = _wide(1536000, 64, "ssb", sizes, 1002, carriers=64)
    assert pr.samples >= 64 * (sum(sizes) // 32 - 1)      # 32 input samples per audio sample, less the stages' phases


def test_config4_slice_128_mixed_receivers_61440k():
    """The per-GPU share of BASELINE configs[3]: 128 mixed SSB/AM/FM receivers at 61.44 Msps, blocks of 2^20."""
    pr = _wide(61440000, 128, "mixed", [1 << 20, 1 << 20, 500000, 1 << 20], 1004, carriers=32)
    assert pr.samples > 128 * 2500


def _device_blocks(x, block):
    nb = len(x) // block
    dev = DevBuf(np.ascontiguousarray(x[:nb * block]))
    return dev, [dev.at(16 * block * b) for b in range(nb)]


def test_overlapped_pipeline_equals_synchronised_blocks():
    """The path bench.py times: K blocks through qk_process_samples_device with nothing waiting in between (front end of
    block b+1 beside the chain of block b beside the AGC of block b-1, twin buffers, the apply stream), against a second
    engine fed the same blocks with qk_sync after each.  Mixed SSB / AM / FM / DGT-IQ receivers at a rate whose decimated
    rate is not 48 kHz (the fractional decimator's host-side positions ride the same pipeline): the last block's audio and
    AGC input must be bit-identical."""
    fs, nrx, block, K = 1333333, 8, 140000, 6
    tunes = tune_offsets(nrx, fs)
    modes = np.array([3, 4, 5, 9, 2, 4, 5, 3])
    bws = [None, None, None, 6000, None, None, None, None]
    x = make_iq(block * K, fs, tunes, modes, 321)
    dev, ptrs = _device_blocks(x, block)
    outs = []
    for synced in (False, True):
        eng = Engine(fs, n_receivers=nrx, max_block=block)
        configure_engine(eng, modes, tunes, bws=bws)
        for b in range(K):
            eng.process_device(ptrs[b], block)
            if synced:
                eng.sync()
        eng.sync()
        audio, counts = eng.audio_all()
        taps = [eng.tap(TAP_PLAY, r, planes=2 if modes[r] == 9 else 1) for r in range(nrx)]
        outs.append((audio.copy(), counts.copy(), taps))
        eng.close()
    (a0, c0, t0), (a1, c1, t1) = outs
    assert np.array_equal(c0, c1) and int(c0.min()) > 0
    for r in range(nrx):
        assert np.array_equal(a0[r, :c0[r]].view(np.float64), a1[r, :c1[r]].view(np.float64)), r
        assert np.array_equal(t0[r], t1[r]), r
    dev.free()


def test_reset_stats_between_blocks_keeps_the_pipeline_parity():
    """qk_reset_stats after an odd number of blocks (with no synchronisation of the caller's own) must not flip the
    pipeline's ping-pong sets: FM receivers, squelch on; the squelch flag read back is the last block's, and the
    following blocks equal those of an engine that was never reset."""
    fs, nrx, block = 192000, 2, 9600
    tunes, modes = [30000, -41000], [5, 5]
    rng = np.random.default_rng(3)
    t = np.arange(block * 8)
    key = np.ones(len(t))
    key[block * 2:block * 3] = 0.0                                            # a quiet block: the squelch closes
    x = sum(2.0 ** 24 * key * np.exp(1j * (2 * np.pi * (f / fs) * t + 3.0 * np.sin(2 * np.pi * 1000 / fs * t))) for f in tunes)
    x = np.round(x + (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t))) * 2.0 ** 10)
    dev, ptrs = _device_blocks(x, block)
    res = []
    for reset in (False, True):
        eng = Engine(fs, n_receivers=nrx, max_block=block)
        configure_engine(eng, modes, tunes)
        eng.set_squelch(-60.0)
        flags = []
        for b in range(8):
            eng.process_device(ptrs[b], block)
            if b in (2, 4):                 # after an odd number of blocks (3, 5)
                if reset:
                    eng.reset_stats()
                flags.append([eng.squelch_active(r) for r in range(nrx)])
            if b == 6 and reset:
                eng.reset_stats()           # and once with nothing synchronising behind it
        eng.sync()
        audio, counts = eng.audio_all()
        res.append((flags, audio.copy(), counts.copy()))
        eng.close()
    assert res[0][0] == res[1][0]
    assert res[0][0][0] == [1, 1] and res[0][0][1] == [0, 0], res[0][0]     # closed on the quiet block, open again later
    assert np.array_equal(res[0][2], res[1][2])
    for r in range(nrx):
        m = int(res[0][2][r])
        assert m > 0 and np.array_equal(res[0][1][r, :m].view(np.float64), res[1][1][r, :m].view(np.float64)), r
    dev.free()
