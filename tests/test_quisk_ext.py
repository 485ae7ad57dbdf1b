"""The `_quisk` CPython extension (quisk_amd/ext/_quisk.so, source quisk_amd/csrc/quisk_module.c): the reference's module
interface for the receive path in C -- method table (quisk.c:5164-5266), exception, capsule `_quisk.QUISK_C_API`
(quisk.h:326-355) and the exported C symbols quisk_process_samples / cRxFilterOut (quisk.h:258, 215) -- driven the way
quisk.py, a hardware module and a C sample-source plug-in (tests/plugin/qk_test_source.c) drive it, against the oracle."""
import ctypes as C
import os
import sys
import time
import types

import numpy as np
import pytest

import _bind as B
from _rxsetup import rx_filter, make_oracle_rx, rel_err, TOL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT_DIR = os.path.join(ROOT, "quisk_amd", "ext")
sys.path.insert(0, EXT_DIR)
sys.path.insert(0, os.path.join(ROOT, "tests", "plugin"))
import _quisk as QS                     # noqa: E402  (the C extension: quisk_amd/ext/_quisk.so)
import qk_test_source as SRC            # noqa: E402  (binds through the capsule at import)
from quisk_amd.synth import make_iq     # noqa: E402

PATH_NAMES = ("record_app", "set_params", "set_sparams", "add_rx_samples", "read_sound", "start_sound", "close_sound", "set_tune",
              "set_rx_mode", "set_filters", "set_agc", "set_multirx_mode", "set_multirx_freq", "set_multirx_play_method",
              "set_multirx_play_channel", "set_split_rxtx", "get_graph", "get_multirx_graph", "add_bscope_samples", "get_bandscope",
              "get_filter_rate", "get_smeter", "get_state", "get_overrange", "invert_spectrum", "set_squelch", "set_ssb_squelch",
              "set_auto_notch", "set_kill_audio", "set_noise_blanker", "set_volume", "change_rate", "change_rates", "change_scan",
              "is_key_down", "open_sound", "open_key", "set_key_down")


def test_it_is_the_c_extension_with_the_path_names_the_exception_and_the_capsule():
    assert QS.__file__.endswith(os.path.join("ext", "_quisk.so"))
    for name in PATH_NAMES:
        assert callable(getattr(QS, name)), name
    assert issubclass(QS.error, Exception) and QS.error.__module__ == "_quisk"
    assert type(QS.QUISK_C_API).__name__ == "PyCapsule" and "_quisk.QUISK_C_API" in repr(QS.QUISK_C_API)
    L = C.CDLL(QS.__file__)             # the symbols sound.c / freedv.c link against
    for sym in ("quisk_process_samples", "cRxFilterOut", "quisk_sound_state", "QuiskGetConfigInt", "quisk_sample_source",
                "quisk_sample_source4", "QuiskTimeSec"):
        assert getattr(L, sym) is not None


def test_errors_without_an_engine_and_without_a_gpu():
    QS.close_sound()
    with pytest.raises(QS.error, match="record_app"):
        QS.set_tune(1000, 0)
    with pytest.raises(TypeError):
        QS.set_tune("a", 0)              # PyArg_ParseTuple "ii"
    assert QS.get_filter_rate(3, 2800) >= 0 and QS.is_key_down() == 0
    assert QS.set_volume(0.5) is None and QS.change_rate(96000, 1) is None       # outside the path: accepted
    try:                                 # without a GPU the module's own exception carries the engine's message: no CPU path
        QS.record_app(None, types.SimpleNamespace(), 1024, 1024, 4096, 64, 192000, 0)
    except QS.error as exc:
        assert "no HIP device" in str(exc)
    QS.close_sound()


def _conf(**kw):
    d = dict(graph_refresh=1000, playback_rate=48000, quisk_amd_receivers=3)
    d.update(kw)
    return types.SimpleNamespace(**d)


def _raw32(x):
    raw = np.empty(2 * len(x), dtype="<i4")
    raw[0::2] = x.real.astype(np.int64).astype(np.int32)
    raw[1::2] = x.imag.astype(np.int64).astype(np.int32)
    return raw.tobytes()


def _audio(rx=0):
    return np.frombuffer(QS.get_audio(rx), dtype=np.complex128)


@pytest.mark.gpu
def test_receive_path_graph_and_raw_scope_through_the_extension():
    fs, mode, tune = 192000, 3, 23000
    QS.record_app(None, _conf(), 1024, 1024, 4096, 64, fs, 0)
    QS.set_params(rx_bytes=4, rx_endian=0)
    fi, fq, bw, off = rx_filter(fs, mode)
    QS.set_rx_mode(mode); QS.set_tune(tune, 0)
    QS.set_filters(list(fi), tuple(fq), bw, off, 0)
    QS.set_agc(80.0)
    assert QS.get_filter_rate(mode, bw) == 12000
    orx = make_oracle_rx(fs, mode, tune)
    og = B.OracleGraph(4096, 1024, fs)
    og.set_passband(tune, off, bw)
    x = make_iq(16 * 4096, fs, [tune, -40000], [mode, mode], 77)
    worst = 0.0
    for i in range(0, len(x), 4800):
        xb = x[i:i + 4800]
        assert QS.add_rx_samples(_raw32(xb)) is None
        assert QS.read_sound() == len(xb)
        want = orx.process(xb)
        got = _audio(0)
        assert len(got) == len(want)
        if len(want):
            worst = max(worst, rel_err(got, want))
    assert worst < TOL
    og.add_frames(x)
    time.sleep(0.01)
    row = QS.get_graph(1, 1.0, 0.0)
    n, want_row, want_sm = og.release(1.0, 0.0)
    assert isinstance(row, tuple) and len(row) == 1024 and n == 16
    assert np.max(np.abs(np.array(row) - want_row)) < 1e-7
    assert abs(QS.get_smeter() - want_sm) < 1e-7
    assert QS.get_graph(1, 1.0, 0.0) is None
    # soft failures of add_rx_samples drop the whole buffer (quisk.c:2798-2803)
    assert QS.add_rx_samples(b"\0" * 12) is None and QS.read_sound() == 0            # odd number of bytes for 2 x 4-byte samples
    assert QS.add_rx_samples(b"\0" * (8 * 52801)) is None and QS.read_sound() == 0   # more than SAMP_BUFFER_SIZE * 8 / 10
    # raw scope: use_fft == 0 returns the first data_width samples of the oldest of at most four queued frames
    # (quisk.c:4642-4649); a fifth frame is dropped and counted (quisk.c:2342-2345)
    assert QS.get_graph(0, 1.0, 0.0) is None
    y = make_iq(5 * 4096, fs, [1000], [3], 5)
    QS.add_rx_samples(_raw32(y[:3 * 4096])); QS.read_sound()
    QS.add_rx_samples(_raw32(y[3 * 4096:])); QS.read_sound()
    for f in range(4):
        raw = QS.get_graph(0, 1.0, 0.0)
        assert len(raw) == 1024 and isinstance(raw[0], complex)
        assert np.array_equal(np.array(raw), y[f * 4096:f * 4096 + 1024])
    assert QS.get_graph(0, 1.0, 0.0) is None
    st = QS.get_state()
    assert len(st) == 17 and st[2] == fs and st[14] == 1
    assert QS.get_graph(1, 1.0, 0.0) is None               # back to spectra: the average starts again
    orx.close(); og.close()
    QS.close_sound()


@pytest.mark.gpu
def test_c_plugin_feeds_samples_through_the_capsule_and_exported_symbols():
    fs, mode, tune = 96000, 2, -7000
    QS.record_app(None, _conf(graph_refresh=25), 512, 512, 2048, 64, fs, 0)
    fi, fq, bw, off = rx_filter(fs, mode)
    QS.set_rx_mode(mode); QS.set_tune(tune, 0); QS.set_filters(list(fi), list(fq), bw, off, 0)
    x = np.round(make_iq(30000, fs, [tune], [mode], 9))
    SRC.load(x.tobytes(), 4000)
    SRC.attach()                                          # quisk_sample_source(start, stop, read), capsule entry 7
    QS.start_sound()
    started, stopped, rate_seen, refresh_seen, clock_ok = SRC.state()
    assert (started, stopped, rate_seen, refresh_seen, clock_ok) == (1, 0, fs, 25, 1)
    orx = make_oracle_rx(fs, mode, tune)
    worst, total = 0.0, 0
    while True:
        n = QS.read_sound()                               # pulls a block from the plug-in's read callback
        if n == 0:
            break
        want = orx.process(x[total:total + n])
        total += n
        got = _audio(0)
        assert len(got) == len(want)
        if len(want):
            worst = max(worst, rel_err(got, want))
    assert total == len(x) and worst < TOL
    SRC.count_read_error()                                # pt_quisk_sound_state->read_error++ from the plug-in
    assert QS.get_state()[8] == 1
    # quisk_process_samples (quisk.h:258): in place, returns the audio count
    QS.set_params(rx_bytes=4, rx_endian=0)                # Python is the sample source again
    L = C.CDLL(QS.__file__)
    L.quisk_process_samples.argtypes = [np.ctypeslib.ndpointer(dtype=np.complex128), C.c_int]
    x2 = np.round(make_iq(4800, fs, [tune], [mode], 10))
    buf = np.zeros(66000, dtype=np.complex128); buf[:4800] = x2
    m = L.quisk_process_samples(buf, 4800)
    want = orx.process(x2)
    assert m == len(want) and rel_err(buf[:m], want) < TOL
    # cRxFilterOut: the block form against the oracle's per-sample loop, then the per-sample symbol continuing the same history
    z = np.round(make_iq(700, 12000, [1500], [3], 11))
    o2 = make_oracle_rx(fs, mode, tune)
    ref = z.copy(); o2.L.qko_rx_bandpass_block(o2.h, ref, len(ref))
    got = np.frombuffer(QS.rx_filter_block(z[:690].tobytes(), 1, 0), dtype=np.complex128)
    assert rel_err(got, ref[:690]) < 1e-12

    class Cd(C.Structure):
        _fields_ = [("re", C.c_double), ("im", C.c_double)]
    L.cRxFilterOut.restype = Cd
    L.cRxFilterOut.argtypes = [Cd, C.c_int, C.c_int]
    for i in range(690, 700):
        r = L.cRxFilterOut(Cd(z[i].real, z[i].imag), 1, 0)
        assert abs(complex(r.re, r.im) - ref[i]) <= 1e-12 * np.max(np.abs(ref))
    orx.close(); o2.close()
    SRC.attach()
    QS.close_sound()
    assert SRC.state()[1] == 1                            # the plug-in's stop callback ran


@pytest.mark.gpu
def test_split_rxtx_and_multirx_play_through_the_extension():
    """set_split_rxtx / set_multirx_play_method / set_multirx_play_channel (quisk.c:4315, 4450, 4458) reach the router:
    the second ear carries the receiver tuned to the transmit offset, or the chosen sub-receiver (quisk.c:2410-2494)."""
    fs, mode, rx_t, tx_t = 96000, 3, 12000, -20000
    QS.record_app(None, _conf(quisk_amd_receivers=4), 512, 512, 2048, 64, fs, 0)
    QS.set_params(rx_bytes=4, rx_endian=0)
    fi, fq, bw, off = rx_filter(fs, mode)
    QS.set_rx_mode(mode); QS.set_filters(list(fi), list(fq), bw, off, 0); QS.set_filters(list(fi), list(fq), bw, off, 1)
    QS.set_tune(rx_t, tx_t)
    QS.set_multirx_mode(0, mode); QS.set_multirx_freq(0, 30000)
    t = np.arange(12 * 4800) / fs
    x = np.round(2.0 ** 24 * (np.exp(2j * np.pi * (rx_t + 1000) * t) + np.exp(2j * np.pi * (tx_t + 700) * t) + np.exp(2j * np.pi * (30000 + 400) * t)))
    o_rx, o_tx, o_sub = make_oracle_rx(fs, mode, rx_t), make_oracle_rx(fs, mode, tx_t), make_oracle_rx(fs, mode, 30000)
    L = o_rx.L
    pair = L.qko_pair_new()                               # Buffer2Chan's FIFOs and Agc2 (quisk.c:1452-1487, 2181)
    QS.set_split_rxtx(1)                                  # stereo, higher frequency is real: tx < rx -> left = receive side
    worst = 0.0
    for b in range(12):
        blk = x[b * 4800:(b + 1) * 4800]
        if b == 6:
            QS.set_split_rxtx(0); QS.set_multirx_play_method(2); QS.set_multirx_play_channel(0)     # left = sub-receiver 0, right = receiver 0
            L.qko_pair_reset_fifos(pair)                  # a change of play channel re-initialises Buffer2Chan (quisk.c:2230-2231)
        QS.add_rx_samples(_raw32(blk)); QS.read_sound()
        work = np.zeros(len(blk) + 30000, np.complex128); work[:len(blk)] = blk
        pre = np.zeros_like(work)
        if b < 6:                                         # the reference's order: one pair through cFracDecim / interpolators / Agc1 + Agc2
            m = L.qko_rx_pair_process(o_rx.h, o_tx.h, pair, work, len(blk), 1, pre)
            o_sub.process(blk)                            # (the sub-receiver runs beside it, as in the engine)
        else:
            m = L.qko_rx_pair_process(o_rx.h, o_sub.h, pair, work, len(blk), 2, pre)
        got = _audio(0)
        assert len(got) == m
        if m:
            worst = max(worst, rel_err(got, work[:m]))
    assert worst < 1e-6, worst                            # from block 0, through both AGCs (steady carriers: no AGC decisions at stake)
    L.qko_pair_delete(pair)
    for o in (o_rx, o_tx, o_sub):
        o.close()
    QS.close_sound()


@pytest.mark.gpu
def test_change_rates_and_change_scan_through_the_extension():
    """change_rates (quisk.c:3874-3886) moves the mixer by vfo_screen - vfo_audio (quisk.c:2354); change_scan (3865-3872) stitches
    the frames of a sweep with the transmit tune and vfo_screen in the slot arithmetic (4661): against the oracle."""
    fs, N, W, mode = 192000, 4096, 1024, 3
    QS.record_app(None, _conf(), W, W, N, 64, fs, 0)
    QS.set_params(rx_bytes=4, rx_endian=0)
    fi, fq, bw, off = rx_filter(fs, mode)
    QS.set_rx_mode(mode); QS.set_filters(list(fi), tuple(fq), bw, off, 0); QS.set_agc(80.0)
    rx_tune, tx_tune, vfo_audio, vfo_screen = 21000, 7050000 - 7100000, 7098000, 7100000
    QS.set_tune(rx_tune, tx_tune)
    QS.change_rates(fs, vfo_audio, fs, vfo_screen)
    orx = make_oracle_rx(fs, mode, rx_tune + vfo_screen - vfo_audio)
    blocks = 3
    scan = (blocks, fs, 0.84, 7000000, int(fs * 0.84))
    QS.change_scan(*scan)
    og = B.OracleGraph(N, W, fs)
    og.set_passband(rx_tune + vfo_screen - vfo_audio, off, bw)
    og.change_scan(*scan, tx_tune, vfo_screen)
    x = make_iq(2 * blocks * N, fs, [23000, -50000, 70000], [3, 3, 4], 43)
    worst = 0.0
    for i in range(0, len(x), N):
        xb = x[i:i + N]
        QS.add_rx_samples(_raw32(xb)); assert QS.read_sound() == N
        want = orx.process(xb)
        got = _audio(0)
        assert len(got) == len(want)
        if len(want):
            worst = max(worst, rel_err(got, want))
    assert worst < TOL
    og.add_frames(x)
    time.sleep(0.01)
    row = QS.get_graph(1, 1.0, 0.0)
    n, want_row, _ = og.release(1.0, 0.0)
    assert n == 2 and np.max(np.abs(np.array(row) - want_row)) < 1e-7
    QS.change_scan(0, fs, 0.84, 0, 0)
    orx.close(); og.close()
    QS.close_sound()


@pytest.mark.gpu
def test_diagnostics_through_the_extension():
    """dft / idft (Xdft, quisk.c:4922-5014: order of frequency, zero at index (size - 1) / 2, Blackman window, 1 / N on the
    inverse), get_filter, get_tx_filter, measure_audio, measure_frequency, get_audio_graph, add_tone: the entries that
    quisk.py calls while it runs (quisk.py:3204, 4178-4194, 4693, 4702, 5277, 5915), with the reference's return types."""
    fs, W = 48000, 256
    QS.record_app(None, _conf(), W, W, 1024, 64, fs, 0)
    QS.set_params(rx_bytes=4, rx_endian=0)
    fi, fq, bw, off = rx_filter(fs, 3)
    QS.set_rx_mode(3); QS.set_filters(list(fi), list(fq), bw, off, 0)
    rng = np.random.default_rng(2)
    for size in (8, 31, 1000):
        x = rng.standard_normal(size) + 1j * rng.standard_normal(size)
        j0 = (size - 1) // 2
        X = QS.dft(list(x))
        assert isinstance(X, list) and len(X) == size
        want = np.roll(np.fft.fft(np.roll(x, -j0)), j0)
        assert np.max(np.abs(np.array(X) - want)) < 1e-10
        y = QS.idft(X)
        assert np.max(np.abs(np.array(y) - x)) < 1e-12
        w = np.array([0.42 + 0.5 * np.cos(2 * np.pi * i / size) + 0.08 * np.cos(4 * np.pi * i / size) for i in range(size)])
        Xw = QS.dft(tuple(x.real), 1)                      # floats are accepted, the window applies in rotated order
        assert np.max(np.abs(np.array(Xw) - np.roll(np.fft.fft(np.roll(x.real, -j0) * w), j0))) < 1e-10
    assert QS.dft([]) == ()
    with pytest.raises(QS.error):
        QS.dft(["a"])
    resp = QS.get_filter()
    assert isinstance(resp, tuple) and len(resp) == W and max(resp) > -30.0 and min(resp) < -80.0       # (shaped by the head of the 1024-point display window: quisk.c:4816-4817)
    tx = QS.get_tx_filter()
    assert isinstance(tx, tuple) and len(tx) == W
    assert QS.measure_audio(1) == 0.0 and QS.measure_frequency(0) == 0.0 and QS.measure_frequency(-1) == 0.0
    assert QS.get_audio_graph() is None
    QS.set_tune(4000, 0)
    QS.add_tone(5000)
    blk = np.zeros(4800, complex)
    for _ in range(12):
        QS.add_rx_samples(_raw32(blk)); QS.read_sound()
    rows = QS.get_graph(1, 1.0, 0.0)
    k = int(np.argmax(rows))
    assert abs(k - (W // 2 + 5000 * W // fs)) <= 1          # the tone is where it was put, at -40 dBFS
    assert abs(max(rows) - (-40.0)) < 3.0
    assert QS.measure_audio(-1) > 0.0                        # the USB receiver 1 kHz below the tone hears it
    QS.add_tone(0)
    QS.close_sound()
