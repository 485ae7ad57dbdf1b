#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Runs in the build container only.

  filter_kernels.npz   inputs and outputs of every filter.c kernel on the receive path, produced by
                       the reference's own filter.c (oracle/_ref/libquisk_filter_ref.so, compiled
                       unmodified by oracle/Makefile).  Each case is run as two unequal blocks so the
                       carried state (ring position, decimation phase, half-band toggle) is pinned.
  coef_tables.npz      the coefficient tables read from the same library (numbers only).
  filter_prototypes.npz  the RX band-pass prototype taps: the `Filters` dict of the reference's
                       filters.py (a pure data module, imported from /root/reference here).

The reference cannot travel, so only these data files are committed, with this script.
"""
import ctypes as C
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _bind as B  # noqa: E402

TABLES = {  # reference symbol -> (taps, project name)
    "quiskFilt48dec24Coefs": (98, "qk_h_dec2_98"), "quiskFilt144D3Coefs": (147, "qk_h_dec3_147"),
    "quiskFilt240D5CoefsSharp": (245, "qk_h_dec5_245"), "quiskFilt300D5Coefs": (125, "qk_h_rs65_125"),
    "quiskFilt53D1Coefs": (55, "qk_h_sdriq53_55"), "quiskFilt111D2Coefs": (114, "qk_h_sdriq111_114"),
    "quiskFilt133D2Coefs": (136, "qk_h_sdriq133_136"), "quiskFilt167D3Coefs": (174, "qk_h_sdriq167_174"),
    "quiskFilt185D3Coefs": (189, "qk_h_sdriq185_189"), "quiskAudio24p3Coefs": (100, "qk_h_aud24p3_100"),
    "quiskAudio24p4Coefs": (50, "qk_h_aud24p4_50"), "quiskAudio24p6Coefs": (36, "qk_h_aud24p6_36"),
    "quiskAudioFmHpCoefs": (309, "qk_h_fmhp_309"), "quiskFilt16dec8Coefs": (62, "qk_h_dec16to8_62"),
}
SPLIT = 1000
N = 2047


def cnoise(rng, n):
    return np.round((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 2.0 ** 24).astype(np.complex128)


def run_blocks(fn, x, cap_mult=1):
    """call fn(buf, count) -> nout on two unequal blocks; buf is in/out"""
    outs = []
    for blk in (x[:SPLIT], x[SPLIT:]):
        buf = np.zeros(len(blk) * cap_mult + 8, dtype=x.dtype)
        buf[:len(blk)] = blk
        m = fn(buf, len(blk))
        outs.append(buf[:m].copy())
    return np.concatenate(outs)


def main():
    R = B.ref()
    rng = np.random.default_rng(20241)
    out = {}
    tabs = {}
    keep = []
    for sym, (n, name) in TABLES.items():
        tabs[name] = B.ref_table(sym, n)

    def ctab(sym):
        n = TABLES[sym][0]
        arr = (C.c_double * n).in_dll(R, sym)
        keep.append(arr)
        return C.cast(arr, B.c_dp), n

    def cfilter(sym):
        f = B.RCFilter()
        p, n = ctab(sym)
        R.quisk_filt_cInit(C.byref(f), p, n)
        return f

    def dfilter(sym):
        f = B.RDFilter()
        p, n = ctab(sym)
        R.quisk_filt_dInit(C.byref(f), p, n)
        return f

    # quisk_cDecimate
    for sym, d in (("quiskFilt48dec24Coefs", 2), ("quiskFilt144D3Coefs", 3), ("quiskFilt240D5CoefsSharp", 5),
                   ("quiskFilt53D1Coefs", 1), ("quiskFilt185D3Coefs", 3)):
        x = cnoise(rng, N)
        f = cfilter(sym)
        key = "cdecimate:%s:%d" % (TABLES[sym][1], d)
        out[key + ":in"] = x
        out[key + ":out"] = run_blocks(lambda b, n: R.quisk_cDecimate(b, n, C.byref(f), d), x)
    # quisk_cCDecimate after quisk_filt_tune, both sidebands
    for upper in (1, 0):
        x = cnoise(rng, N)
        f = cfilter("quiskAudio24p3Coefs")
        R.quisk_filt_tune(C.cast(C.byref(f), C.POINTER(B.RDFilter)), 1500.0 / 48000, upper)
        key = "ccdecimate:qk_h_aud24p3_100:3:%d" % upper
        out[key + ":in"] = x
        out[key + ":out"] = run_blocks(lambda b, n: R.quisk_cCDecimate(b, n, C.byref(f), 3), x)
        nt = f.nTaps
        cc = np.ctypeslib.as_array(C.cast(f.cpxCoefs, C.POINTER(C.c_double)), shape=(2 * nt,)).copy()
        out[key + ":coefs"] = cc[0::2] + 1j * cc[1::2]
    # quisk_dDecimate, quisk_dFilter
    x = np.round(rng.standard_normal(N) * 2.0 ** 24)
    f = dfilter("quiskAudio24p3Coefs")
    out["ddecimate:qk_h_aud24p3_100:2:in"] = x
    out["ddecimate:qk_h_aud24p3_100:2:out"] = run_blocks(lambda b, n: R.quisk_dDecimate(b, n, C.byref(f), 2), x)
    for sym in ("quiskAudio24p6Coefs", "quiskAudioFmHpCoefs"):
        x = np.round(rng.standard_normal(N) * 2.0 ** 24)
        f = dfilter(sym)
        key = "dfilter:%s" % TABLES[sym][1]
        out[key + ":in"] = x
        out[key + ":out"] = run_blocks(lambda b, n: R.quisk_dFilter(b, n, C.byref(f)), x)
    # interpolators
    for sym, L in (("quiskAudio24p4Coefs", 2), ("quiskAudio24p3Coefs", 3)):
        x = np.round(rng.standard_normal(N) * 2.0 ** 24)
        f = dfilter(sym)
        key = "dinterpolate:%s:%d" % (TABLES[sym][1], L)
        out[key + ":in"] = x
        out[key + ":out"] = run_blocks(lambda b, n: R.quisk_dInterpolate(b, n, C.byref(f), L), x, cap_mult=L)
    x = cnoise(rng, N)
    f = cfilter("quiskAudio24p3Coefs")
    out["cinterpolate:qk_h_aud24p3_100:3:in"] = x
    out["cinterpolate:qk_h_aud24p3_100:3:out"] = run_blocks(lambda b, n: R.quisk_cInterpolate(b, n, C.byref(f), 3), x, cap_mult=3)
    # the 6/5 then 4/5 resampler pair
    for sym, L, D in (("quiskFilt300D5Coefs", 6, 5), ("quiskFilt240D5CoefsSharp", 4, 5)):
        x = cnoise(rng, N)
        f = cfilter(sym)
        key = "cinterpdecim:%s:%d:%d" % (TABLES[sym][1], L, D)
        out[key + ":in"] = x
        out[key + ":out"] = run_blocks(lambda b, n: R.quisk_cInterpDecim(b, n, C.byref(f), L, D), x, cap_mult=2)
    # half-bands
    x = cnoise(rng, N)
    hb = B.RCHB()
    out["cdecim2hb45:in"] = x
    out["cdecim2hb45:out"] = run_blocks(lambda b, n: R.quisk_cDecim2HB45(b, n, C.byref(hb)), x)
    x = cnoise(rng, N)
    hb2 = B.RCHB()
    out["cinterp2hb45:in"] = x
    out["cinterp2hb45:out"] = run_blocks(lambda b, n: R.quisk_cInterp2HB45(b, n, C.byref(hb2)), x, cap_mult=2)
    x = np.round(rng.standard_normal(N) * 2.0 ** 24)
    hb3 = B.RDHB()
    out["dinterp2hb45:in"] = x
    out["dinterp2hb45:out"] = run_blocks(lambda b, n: R.quisk_dInterp2HB45(b, n, C.byref(hb3)), x, cap_mult=2)

    # the per-sample filters quisk_dD_out and quisk_dC_out (the latter after quisk_filt_tune)
    x = np.round(rng.standard_normal(N) * 2.0 ** 24)
    f = dfilter("quiskAudio24p6Coefs")
    out["dd_out:qk_h_aud24p6_36:in"] = x
    out["dd_out:qk_h_aud24p6_36:out"] = np.array([R.quisk_dD_out(float(v), C.byref(f)) for v in x])
    for upper in (1, 0):
        x = np.round(rng.standard_normal(N) * 2.0 ** 24)
        f = dfilter("quiskAudio24p3Coefs")
        R.quisk_filt_tune(C.byref(f), 1650.0 / 48000, upper)
        key = "dc_out:qk_h_aud24p3_100:%d" % upper
        vals = [R.quisk_dC_out(float(v), C.byref(f)) for v in x]
        out[key + ":in"] = x
        out[key + ":out"] = np.array([complex(v.re, v.im) for v in vals])
        nt = f.nTaps
        cc = np.ctypeslib.as_array(C.cast(f.cpxCoefs, C.POINTER(C.c_double)), shape=(2 * nt,)).copy()
        out[key + ":coefs"] = cc[0::2] + 1j * cc[1::2]

    np.savez_compressed(os.path.join(HERE, "filter_kernels.npz"), **out)
    np.savez_compressed(os.path.join(HERE, "coef_tables.npz"), **tabs)

    # RX band-pass prototypes: data of the reference's filters.py
    sys.path.insert(0, "/root/reference")
    import filters  # noqa: E402  (pure data module)
    protos = {"bw%d" % k: np.array(v, dtype=np.float64) for k, v in filters.Filters.items()}
    np.savez_compressed(os.path.join(HERE, "filter_prototypes.npz"), **protos)
    print("wrote %d kernel arrays, %d tables, %d prototypes" % (len(out), len(tabs), len(protos)))


if __name__ == "__main__":
    main()
