"""The CPU restatement (oracle/) against the reference's filter.c: bit-exact.

Two pins: the committed golden vectors (made by tests/golden/make_golden.py from the compiled
reference) and, where oracle/_ref exists, the compiled reference itself on fresh random input.
"""
import ctypes as C
import numpy as np
import pytest

import _bind as B

G = np.load(B.GOLDEN + "/filter_kernels.npz")
TABS = np.load(B.GOLDEN + "/coef_tables.npz")
SPLIT = 1000


def _tab(name):
    h = np.ascontiguousarray(B.oracle_table(name))
    return h, h.ctypes.data_as(B.c_dp)


def _blocks(fn, x, mult=1):
    outs = []
    for blk in (x[:SPLIT], x[SPLIT:]):
        buf = np.zeros(len(blk) * mult + 8, dtype=x.dtype)
        buf[:len(blk)] = blk
        m = fn(buf, len(blk))
        outs.append(buf[:m].copy())
    return np.concatenate(outs)


def _same(a, b):
    assert a.shape == b.shape
    assert np.array_equal(a.view(np.float64), b.view(np.float64))


def test_coefficient_tables_match_reference():
    for name in TABS.files:
        _same(B.oracle_table(name), TABS[name])


def _cases(prefix):
    return sorted({k.rsplit(":", 1)[0] for k in G.files if k.startswith(prefix + ":")})


@pytest.mark.parametrize("key", _cases("cdecimate"))
def test_cdecimate(key):
    _, name, d = key.split(":")
    L = B.oracle()
    h, hp = _tab(name)
    f = B.OCFir()
    L.qko_cfir_init(C.byref(f), hp, len(h))
    _same(_blocks(lambda b, n: L.qko_cdecimate(b, n, C.byref(f), int(d)), G[key + ":in"]), G[key + ":out"])
    L.qko_cfir_free(C.byref(f))


@pytest.mark.parametrize("key", _cases("ccdecimate"))
def test_ccdecimate_and_tune(key):
    _, name, d, upper = key.split(":")
    L = B.oracle()
    h, hp = _tab(name)
    f = B.OCFir()
    L.qko_cfir_init(C.byref(f), hp, len(h))
    L.qko_filt_tune(C.cast(C.byref(f), C.POINTER(B.ODFir)), 1500.0 / 48000, int(upper))
    cc = np.ctypeslib.as_array(C.cast(f.hc, C.POINTER(C.c_double)), shape=(2 * len(h),)).copy()
    _same(cc[0::2] + 1j * cc[1::2], G[key + ":coefs"])
    _same(_blocks(lambda b, n: L.qko_ccdecimate(b, n, C.byref(f), int(d)), G[key + ":in"]), G[key + ":out"])


def test_ddecimate():
    key = "ddecimate:qk_h_aud24p3_100:2"
    L = B.oracle()
    h, hp = _tab("qk_h_aud24p3_100")
    f = B.ODFir()
    L.qko_dfir_init(C.byref(f), hp, len(h))
    _same(_blocks(lambda b, n: L.qko_ddecimate(b, n, C.byref(f), 2), G[key + ":in"]), G[key + ":out"])


@pytest.mark.parametrize("key", _cases("dfilter"))
def test_dfilter(key):
    L = B.oracle()
    h, hp = _tab(key.split(":")[1])
    f = B.ODFir()
    L.qko_dfir_init(C.byref(f), hp, len(h))
    _same(_blocks(lambda b, n: L.qko_dfilter(b, n, C.byref(f)), G[key + ":in"]), G[key + ":out"])


@pytest.mark.parametrize("key", _cases("dinterpolate"))
def test_dinterpolate(key):
    _, name, interp = key.split(":")
    L = B.oracle()
    h, hp = _tab(name)
    f = B.ODFir()
    L.qko_dfir_init(C.byref(f), hp, len(h))
    _same(_blocks(lambda b, n: L.qko_dinterpolate(b, n, C.byref(f), int(interp)), G[key + ":in"], int(interp)),
          G[key + ":out"])


def test_cinterpolate():
    key = "cinterpolate:qk_h_aud24p3_100:3"
    L = B.oracle()
    h, hp = _tab("qk_h_aud24p3_100")
    f = B.OCFir()
    L.qko_cfir_init(C.byref(f), hp, len(h))
    _same(_blocks(lambda b, n: L.qko_cinterpolate(b, n, C.byref(f), 3), G[key + ":in"], 3), G[key + ":out"])


@pytest.mark.parametrize("key", _cases("cinterpdecim"))
def test_cinterpdecim(key):
    _, name, interp, d = key.split(":")
    L = B.oracle()
    h, hp = _tab(name)
    f = B.OCFir()
    L.qko_cfir_init(C.byref(f), hp, len(h))
    _same(_blocks(lambda b, n: L.qko_cinterpdecim(b, n, C.byref(f), int(interp), int(d)), G[key + ":in"], 2),
          G[key + ":out"])


def test_halfbands():
    L = B.oracle()
    f = B.OCHB()
    _same(_blocks(lambda b, n: L.qko_cdecim2hb45(b, n, C.byref(f)), G["cdecim2hb45:in"]), G["cdecim2hb45:out"])
    f2 = B.OCHB()
    _same(_blocks(lambda b, n: L.qko_cinterp2hb45(b, n, C.byref(f2)), G["cinterp2hb45:in"], 2), G["cinterp2hb45:out"])
    f3 = B.ODHB()
    _same(_blocks(lambda b, n: L.qko_dinterp2hb45(b, n, C.byref(f3)), G["dinterp2hb45:in"], 2), G["dinterp2hb45:out"])


@pytest.mark.skipif(not B.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_live_reference_long_stream_ragged_blocks():
    """Fresh random stream, ragged block sizes incl. empty and 1-sample blocks, oracle == compiled reference."""
    L, R = B.oracle(), B.ref()
    rng = np.random.default_rng(7)
    x = np.round((rng.standard_normal(20000) + 1j * rng.standard_normal(20000)) * 2.0 ** 26)
    sizes = [0, 1, 2, 3, 97, 1, 4096, 0, 5, 7001, 333, 8464]
    h, hp = _tab("qk_h_dec3_147")
    fo, fr = B.OCFir(), B.RCFilter()
    L.qko_cfir_init(C.byref(fo), hp, len(h))
    R.quisk_filt_cInit(C.byref(fr), hp, len(h))
    ho, hr = B.OCHB(), B.RCHB()
    pos = 0
    for s in sizes:
        a = x[pos:pos + s].copy(); b = a.copy(); pos += s
        a = np.concatenate([a, np.zeros(4, complex)]); b = a.copy()
        na = L.qko_cdecim2hb45(a, s, C.byref(ho)); nb = R.quisk_cDecim2HB45(b, s, C.byref(hr))
        assert na == nb
        _same(a[:na], b[:nb])
        ma = L.qko_cdecimate(a, na, C.byref(fo), 3); mb = R.quisk_cDecimate(b, nb, C.byref(fr), 3)
        assert ma == mb
        _same(a[:ma], b[:mb])


def test_dd_out_per_sample_filter():
    """quisk_dD_out (filter.c:303-322) against the compiled reference's per-sample outputs."""
    L = B.oracle()
    key = "dd_out:qk_h_aud24p6_36"
    h, hp = _tab("qk_h_aud24p6_36")
    f = B.ODFir()
    L.qko_dfir_init(C.byref(f), hp, len(h))
    x = np.array(G[key + ":in"], copy=True)
    L.qko_dd_out_block(x[:SPLIT], SPLIT, C.byref(f))
    tail = np.ascontiguousarray(x[SPLIT:])
    L.qko_dd_out_block(tail, len(tail), C.byref(f))
    _same(np.concatenate([x[:SPLIT], tail]), G[key + ":out"])
    L.qko_dfir_free(C.byref(f))


@pytest.mark.parametrize("upper", [1, 0])
def test_dc_out_after_filt_tune(upper):
    """quisk_dC_out (filter.c:60-81) with the complex taps of quisk_filt_tune (filter.c:35-58)."""
    L = B.oracle()
    key = "dc_out:qk_h_aud24p3_100:%d" % upper
    h, hp = _tab("qk_h_aud24p3_100")
    f = B.ODFir()
    L.qko_dfir_init(C.byref(f), hp, len(h))
    L.qko_filt_tune(C.byref(f), 1650.0 / 48000, upper)
    x = np.ascontiguousarray(G[key + ":in"])
    out = np.zeros(len(x), dtype=np.complex128)
    L.qko_dc_out_block(x[:SPLIT], SPLIT, C.byref(f), out[:SPLIT])
    rest = np.zeros(len(x) - SPLIT, dtype=np.complex128)
    L.qko_dc_out_block(np.ascontiguousarray(x[SPLIT:]), len(x) - SPLIT, C.byref(f), rest)
    out[SPLIT:] = rest
    _same(out, G[key + ":out"])
    L.qko_dfir_free(C.byref(f))
