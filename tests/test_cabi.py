"""The C-ABI library loads and exports every symbol include/qk_engine.h declares (no compute: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "qk_engine.h")
LIB = os.path.join(ROOT, "quisk_amd", "libqkengine.so")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(qk_[a-z0-9_]+|quisk_[A-Za-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("qk_create", "qk_destroy", "qk_set_tune", "qk_set_mode", "qk_set_filters", "qk_set_agc",
                 "qk_process_samples", "qk_process_samples_device", "qk_process_samples_int", "qk_pull_audio",
                 "qk_get_graph", "qk_get_filter_rate", "qk_plan_decimation", "quisk_cCplxSampleFilter", "quisk_ccfFilter"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build the library first (make / __graft_entry__.build())"
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_pure_planning_functions_match_the_oracle():
    """PlanDecimation / get_filter_rate / decimated rate are host-only code in the library: check them against
    the oracle's restatement over the rates the reference documents (and the odd ones)."""
    import _bind as B
    import quisk_amd as q
    L = B.oracle()
    rates = [41000, 44100, 48000, 50000, 53333, 96000, 111111, 133333, 185185, 192000, 240000, 250000, 370370, 384000,
             480000, 740740, 768000, 960000, 1333333, 1536000, 2000000, 3072000, 61440000]
    for r in rates:
        d = [ctypes.c_int() for _ in range(3)]
        best = L.qko_plan_decimation(r, *[ctypes.byref(v) for v in d])
        assert q.plan_decimation(r) == (best, d[0].value, d[1].value, d[2].value)
        assert q.decim_rate(r) == L.qko_decim_rate(r)
        for mode in range(14):
            for bw in (500, 2800, 6000):
                assert q.get_filter_rate(r, mode, bw) == L.qko_get_filter_rate(r, mode, bw)
    # values worked out by hand from quisk.c:1509-1547
    assert q.plan_decimation(192000) == (48000, 2, 0, 0)
    assert q.plan_decimation(1536000) == (48000, 5, 0, 0)
    assert q.plan_decimation(61440000) == (49152, 4, 1, 2)
    assert q.plan_decimation(250000) == (48000, 0, 0, 1)


def test_engine_fails_loudly_without_a_gpu():
    import quisk_amd as q
    try:
        eng = q.Engine(192000)
    except q.QkError as exc:
        assert "HIP" in str(exc) or "device" in str(exc)
        return
    eng.close()
    pytest.skip("a GPU is present")


def test_qk_create_judges_its_arguments_before_the_device():
    """qk_create refuses a bad configuration with QK_EINVAL and its own message whether or not a GPU is there (the argument
    checks come first): sample rate, playback rate (quisk.c:2535-2553), the fp32 selector, the short-block selector."""
    from quisk_amd.engine import lib, QkConfig
    L = lib()
    L.qk_last_error.restype = ctypes.c_char_p

    def create(**kw):
        base = dict(device=0, sample_rate=192000, playback_rate=48000, n_receivers=1, max_block=960, fft_size=0, data_width=0)
        res = kw.pop("reserved", {})
        base.update(kw)
        cfg = QkConfig(**base)
        for k, v in res.items():
            cfg.reserved[k] = v
        h = ctypes.c_void_p()
        rc = L.qk_create(ctypes.byref(h), ctypes.byref(cfg))
        msg = (L.qk_last_error() or b"").decode()
        if rc == 0:
            L.qk_destroy(h)
        return rc, msg

    QK_EINVAL = -1
    for kw, needle in ((dict(sample_rate=8000), "sample_rate"), (dict(playback_rate=44100), "playback_rate"),
                       (dict(max_block=0), "max_block"), (dict(reserved={1: 2}), "reserved[1]"), (dict(reserved={2: 3}), "reserved[2]"),
                       (dict(reserved={2: -1}), "reserved[2]")):
        rc, msg = create(**kw)
        assert rc == QK_EINVAL and needle in msg, (kw, rc, msg)
