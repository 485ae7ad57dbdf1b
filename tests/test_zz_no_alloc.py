"""The entry point that allocates nothing (quisk_cCplxSampleFilter_noalloc, include/qk_engine.h) against the existing
host-buffer entry point quisk_cCplxSampleFilter: same kernel, so the outputs, the carried history and the phase must be
bit-identical -- on two calls in a row, the second with a smaller input (the staging array keeps the first call's
contents), and beyond the fixed staging size the call must refuse."""
import ctypes as C

import numpy as np
import pytest

from _hip import DevBuf, free_bytes, stream_create, stream_destroy, sync
from quisk_amd import lib

pytestmark = pytest.mark.gpu

MAX_SAMPLES = 1 << 19       # QK_NOALLOC_MAX_SAMPLES


def _host_call(L, x, h, decim, hist, phase):
    buf = np.array(x, dtype=np.complex128, copy=True)
    ph = C.c_int(phase)
    n = L.quisk_cCplxSampleFilter(buf.ctypes.data, len(buf), h.ctypes.data, len(h), decim, hist.ctypes.data, C.byref(ph))
    assert n >= 0, L.qk_last_error()
    return buf[:n].copy(), ph.value



@pytest.mark.parametrize("ntaps,decim", [(98, 2), (245, 5), (55, 1)])
def test_noalloc_entry_point_equals_the_allocating_one(ntaps, decim):
    L = lib()
    rng = np.random.default_rng(ntaps)
    h = rng.standard_normal(ntaps)
    sizes = [60000, 4097]                      # the second call is smaller and continues the first call's state
    x = (rng.standard_normal(sum(sizes)) + 1j * rng.standard_normal(sum(sizes))) * 2.0 ** 20
    # reference: the existing (allocating, host-buffer) entry point
    hist_h = np.zeros(ntaps - 1, dtype=np.complex128)
    phase_h, want, pos = 0, [], 0
    for n in sizes:
        y, phase_h = _host_call(L, x[pos:pos + n], h, decim, hist_h, phase_h)
        want.append(y)
        pos += n
    # the allocation-free entry point: every buffer on the device
    d_h = DevBuf(h)
    d_hist = DevBuf(nbytes=16 * (ntaps - 1))
    phase = C.c_int(0)
    pos = 0
    for k, n in enumerate(sizes):
        d_x = DevBuf(x[pos:pos + n])
        pos += n
        free_before = free_bytes()
        m = L.quisk_cCplxSampleFilter_noalloc(C.c_void_p(d_x.ptr), n, C.c_void_p(d_h.ptr), ntaps, decim,
                                              C.c_void_p(d_hist.ptr), C.byref(phase), None)
        assert m == len(want[k]), (m, L.qk_last_error())
        sync()
        if k > 0:                               # (the first call may load the code object)
            assert free_bytes() == free_before
        got = d_x.to_host(np.complex128)[:m]
        assert np.array_equal(got.view(np.float64), want[k].view(np.float64)), "call %d differs" % k
    assert phase.value == phase_h
    assert np.array_equal(d_hist.to_host(), hist_h.view(np.float64))


def test_noalloc_entry_point_refuses_oversized_blocks():
    L = lib()
    n = MAX_SAMPLES
    d_x = DevBuf(nbytes=16 * n)
    d_h = DevBuf(np.ones(8))
    d_hist = DevBuf(nbytes=16 * 7)
    phase = C.c_int(0)
    rc = L.quisk_cCplxSampleFilter_noalloc(C.c_void_p(d_x.ptr), n, C.c_void_p(d_h.ptr), 8, 1,
                                           C.c_void_p(d_hist.ptr), C.byref(phase), None)
    assert rc == -4 and b"scratch" in L.qk_last_error()          # QK_ERANGE
    assert phase.value == 0


def test_noalloc_on_a_created_stream_and_the_priming_call():
    """count = 0 is the warm-up call the header asks for (it looks the staging array up and leaves everything alone, the history
    included); then a call on a stream of the caller's own, against the allocating entry point."""
    L = lib()
    ntaps, decim, n = 147, 3, 30001
    rng = np.random.default_rng(5)
    h = rng.standard_normal(ntaps)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 2.0 ** 18
    hist0 = (rng.standard_normal(ntaps - 1) + 1j * rng.standard_normal(ntaps - 1)) * 2.0 ** 18
    hist_h = hist0.copy()
    want, phase_h = _host_call(L, x, h, decim, hist_h, 1)
    d_h, d_hist, d_x = DevBuf(h), DevBuf(hist0), DevBuf(x)
    phase = C.c_int(1)
    m0 = L.quisk_cCplxSampleFilter_noalloc(C.c_void_p(d_x.ptr), 0, C.c_void_p(d_h.ptr), ntaps, decim, C.c_void_p(d_hist.ptr), C.byref(phase), None)
    sync()
    assert m0 == 0 and phase.value == 1
    assert np.array_equal(d_hist.to_host(np.complex128), hist0)                 # nothing in, nothing moved
    st = stream_create()
    free_before = free_bytes()
    m = L.quisk_cCplxSampleFilter_noalloc(C.c_void_p(d_x.ptr), n, C.c_void_p(d_h.ptr), ntaps, decim, C.c_void_p(d_hist.ptr), C.byref(phase), st)
    sync()
    assert free_bytes() == free_before
    assert m == len(want) and phase.value == phase_h
    assert np.array_equal(d_x.to_host(np.complex128)[:m].view(np.float64), want.view(np.float64))
    assert np.array_equal(d_hist.to_host(np.complex128), hist_h)
    stream_destroy(st)
