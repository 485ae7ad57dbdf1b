"""The filter.c drop-ins of the C ABI (quisk_cCplxSampleFilter = quisk_cDecimate, quisk_ccfFilter = quisk_cCDecimate
after quisk_filt_tune) against the golden vectors produced by the compiled reference, and the integer unpack path of
add_rx_samples (quisk.c:2777-2839)."""
import ctypes as C

import numpy as np
import pytest

import _bind as B
from _rxsetup import PairRunner
from quisk_amd import Engine, lib
from quisk_amd.synth import make_iq

pytestmark = pytest.mark.gpu
G = np.load(B.GOLDEN + "/filter_kernels.npz")
SPLIT = 1000


def _run(fn, x, coefs, ntaps, decim):
    hist = np.zeros(2 * (ntaps - 1))
    phase = C.c_int(0)
    outs = []
    for blk in (x[:SPLIT], x[SPLIT:], x[:0]):
        buf = np.array(blk, dtype=np.complex128, copy=True)
        if len(buf) == 0:
            buf = np.zeros(1, np.complex128)
        m = fn(buf.ctypes.data, len(blk), coefs.ctypes.data, ntaps, decim, hist.ctypes.data, C.byref(phase))
        assert m >= 0, lib().qk_last_error()
        outs.append(buf[:m].copy())
    return np.concatenate(outs)


@pytest.mark.parametrize("key", sorted({k.rsplit(":", 1)[0] for k in G.files if k.startswith("cdecimate:")}))
def test_real_coefficient_dropin(key):
    _, name, d = key.split(":")
    h = np.ascontiguousarray(B.oracle_table(name))
    got = _run(lib().quisk_cCplxSampleFilter, G[key + ":in"], h, len(h), int(d))
    want = G[key + ":out"]
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 1e-13


@pytest.mark.parametrize("key", sorted({k.rsplit(":", 1)[0] for k in G.files if k.startswith("ccdecimate:")}))
def test_complex_coefficient_dropin(key):
    cc = np.ascontiguousarray(G[key + ":coefs"])
    d = int(key.split(":")[2])
    got = _run(lib().quisk_ccfFilter, G[key + ":in"], cc.view(np.float64), len(cc), d)
    want = G[key + ":out"]
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 1e-13


@pytest.mark.parametrize("nbytes,big", [(2, False), (3, False), (3, True), (4, False), (4, True), (1, False)])
def test_integer_sample_input(nbytes, big):
    """add_rx_samples formats: 1..4-byte signed I/Q, either endianness, left-justified into int32."""
    fs, block = 192000, 4800
    x = make_iq(block * 4, fs, [30000], [3], 3)
    ints = np.stack([x.real, x.imag], axis=1).astype(np.int64)
    ints = (ints >> (8 * (4 - nbytes))).astype(np.int64)                 # keep the top nbytes
    raw = bytearray()
    for v in ints.reshape(-1):
        raw += int(v).to_bytes(nbytes, "big" if big else "little", signed=True)
    xq = ((ints << (8 * (4 - nbytes))).astype(np.float64)).view(np.complex128).reshape(-1)
    eng = Engine(fs, n_receivers=1, max_block=block)
    ref = Engine(fs, n_receivers=1, max_block=block)
    from _rxsetup import configure_engine
    configure_engine(eng, [3], [30000]); configure_engine(ref, [3], [30000])
    step = block * 2 * nbytes
    for i in range(4):
        eng.process_int(bytes(raw[i * step:(i + 1) * step]), nbytes, big)
        ref.process(xq[i * block:(i + 1) * block])
        assert np.array_equal(eng.audio(0), ref.audio(0))
    eng.close(); ref.close()
