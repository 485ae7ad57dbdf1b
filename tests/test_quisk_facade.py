"""quisk_amd._quisk: the reference's `_quisk` module interface for the receive path (method table quisk.c:5164-5266).
The calls below are the ones quisk.py and the hardware modules make (record_app, set_params, add_rx_samples,
read_sound, set_tune, set_rx_mode, set_filters, set_agc, get_graph, get_filter_rate ...), checked against the oracle."""
import types

import numpy as np
import pytest

import _bind as B
from _rxsetup import rx_filter, make_oracle_rx, rel_err, TOL
from quisk_amd import _quisk as QS
from quisk_amd.synth import make_iq


def test_names_of_the_path_exist():
    for name in ("record_app", "set_params", "add_rx_samples", "read_sound", "set_tune", "set_rx_mode", "set_filters",
                 "set_agc", "set_multirx_mode", "set_multirx_freq", "get_graph", "get_multirx_graph", "add_bscope_samples",
                 "get_bandscope", "get_filter_rate", "get_smeter", "get_state", "get_overrange", "close_sound",
                 "set_volume", "set_squelch", "change_rate", "error"):
        assert hasattr(QS, name), name
    assert issubclass(QS.error, Exception)


def test_calls_before_record_app_raise_the_module_error():
    QS.close_sound()
    with pytest.raises(QS.error):
        QS.set_tune(1000, 0)


def _conf(**kw):
    d = dict(graph_refresh=1000, playback_rate=48000, quisk_amd_receivers=3)
    d.update(kw)
    return types.SimpleNamespace(**d)


@pytest.mark.gpu
def test_receive_path_through_the_module_interface():
    fs, mode, tune = 192000, 3, 23000
    QS.record_app(None, _conf(), 1024, 1024, 4096, 64, fs, 0)
    QS.set_params(rx_bytes=4, rx_endian=0)
    fi, fq, bw, off = rx_filter(fs, mode)
    QS.set_rx_mode(mode)
    QS.set_tune(tune, 0)
    QS.set_filters(list(fi), list(fq), bw, off, 0)
    QS.set_agc(80.0)
    assert QS.get_filter_rate(mode, bw) == 12000
    orx = make_oracle_rx(fs, mode, tune)
    og = B.OracleGraph(4096, 1024, fs)
    og.set_passband(tune, off, bw)
    x = make_iq(16 * 4096, fs, [tune, -40000], [mode, mode], 77)
    block = 4800
    worst = 0.0
    for i in range(0, len(x), block):
        xb = x[i:i + block]
        raw = np.empty(2 * len(xb), dtype="<i4")
        raw[0::2] = xb.real.astype(np.int64).astype(np.int32)
        raw[1::2] = xb.imag.astype(np.int64).astype(np.int32)
        assert QS.add_rx_samples(raw.tobytes()) is None
        assert QS.read_sound() == len(xb)
        want = orx.process(xb)
        got = QS.get_audio(0)
        assert len(got) == len(want)
        if len(want):
            worst = max(worst, rel_err(got, want))
    assert worst < TOL         # a steady tone: the AGC's one start-up ramp ends alike in both (tests/_rxsetup.py, 3.)
    og.add_frames(x)
    import time
    time.sleep(0.01)
    row = QS.get_graph(1, 1.0, 0.0)
    n, want_row, want_sm = og.release(1.0, 0.0)
    assert row is not None and len(row) == 1024 and n == 16
    assert np.max(np.abs(np.array(row) - want_row)) < 1e-7
    assert abs(QS.get_smeter() - want_sm) < 1e-7
    assert QS.get_graph(1, 1.0, 0.0) is None               # nothing new
    st = QS.get_state()
    assert len(st) == 17 and st[2] == fs and st[14] == 0
    orx.close(); og.close()
    QS.close_sound()


@pytest.mark.gpu
def test_set_filters_errors_and_multirx_and_bandscope():
    fs = 96000
    QS.record_app(None, _conf(), 512, 640, 1536, 100, fs, 0)       # fft_size = 3 * data_width: a mixed-radix size
    with pytest.raises(QS.error, match="Filter I is not a sequence"):
        QS.set_filters(3, [1.0], 1000, 0, 0)
    with pytest.raises(QS.error, match="must be equal"):
        QS.set_filters([1.0, 2.0], [1.0], 1000, 0, 0)
    with pytest.raises(QS.error, match="Filter size must be less than 10001"):
        QS.set_filters([0.0] * 10001, [0.0] * 10001, 1000, 0, 0)
    with pytest.raises(TypeError):
        QS.set_params(no_such_key=1)
    # sub-receiver graph: round robin over the receivers that have a full frame
    assert QS.get_multirx_graph() == ((), -1)
    f0 = make_iq(800, fs, [5000], [3], 1)
    f1 = make_iq(800, fs, [-9000], [3], 2)
    QS.add_multirx_samples(0, f0[:300]); QS.add_multirx_samples(0, f0[300:]); QS.add_multirx_samples(1, f1)
    import time
    time.sleep(0.005)
    row, idx = QS.get_multirx_graph()
    assert idx == 0 and np.max(np.abs(np.array(row) - B.oracle_multirx_graph(f0, 100))) < 1e-7
    time.sleep(0.005)
    row, idx = QS.get_multirx_graph()
    assert idx == 1 and np.max(np.abs(np.array(row) - B.oracle_multirx_graph(f1, 100))) < 1e-7
    # bandscope
    assert QS.get_bandscope(122880000, 1.0, 0.0) is None
    QS.set_params(bscope_bytes=2, bscope_endian=1, bscope_size=2048)
    ints = np.random.default_rng(5).integers(-20000, 20000, 2048).astype(">i2")
    QS.add_bscope_samples(ints.tobytes())
    time.sleep(0.005)
    row = QS.get_bandscope(122880000, 1.0, 0.0)
    ob = B.OracleBscope(2048)
    ob.add((ints.astype(np.int32) << 16).astype(np.float64) / 2147483647.0)
    n, want, _ = ob.release(122880000, 1.0, 0.0, 640)
    assert n == 1 and np.max(np.abs(np.array(row) - want)) < 1e-7
    ob.close()
    # mixed-radix display size runs
    x = make_iq(1536 * 2, fs, [1000], [3], 3)
    raw = np.empty(2 * len(x), dtype="<i4"); raw[0::2] = x.real.astype(np.int32); raw[1::2] = x.imag.astype(np.int32)
    QS.set_params(rx_bytes=4, rx_endian=0)
    QS.add_rx_samples(raw.tobytes()); QS.read_sound()
    time.sleep(0.005)
    assert len(QS.get_graph(1, 1.0, 0.0)) == 512
    assert QS.get_overrange() == 0
    QS.set_params(clip=1)
    assert QS.get_overrange() == 1 and QS.get_overrange() == 0
    # conditioning setters of the path reach the engine (quisk.c:4246, 4174; sound.c:1253)
    QS.set_noise_blanker(2); QS.invert_spectrum(1); QS.set_sparams(dc_remove_bw=100)
    QS.add_rx_samples(raw.tobytes()); assert QS.read_sound() == len(x)
    QS.set_noise_blanker(0); QS.invert_spectrum(0); QS.set_sparams(dc_remove_bw=0)
    QS.close_sound()
