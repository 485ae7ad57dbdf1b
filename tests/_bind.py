"""ctypes bindings for the checker libraries (test infrastructure).

  ORACLE : oracle/libqk_oracle.so              this project's CPU restatement
  REF    : oracle/_ref/libquisk_filter_ref.so  the reference's filter.c compiled unmodified
           (exists only where oracle/Makefile could see /root/reference, or was prebuilt)
"""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "libqk_oracle.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libquisk_filter_ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

c_dp = C.POINTER(C.c_double)
cdp = np.ctypeslib.ndpointer(dtype=np.complex128, flags="C_CONTIGUOUS")
ddp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
idp = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


# ----------------------------------------------------------------------------- oracle structs
class OCFir(C.Structure):
    _fields_ = [("h", c_dp), ("hc", C.c_void_p), ("ntaps", C.c_int), ("decim_index", C.c_int),
                ("ring", C.c_void_p), ("pos", C.c_int), ("scratch", C.c_void_p), ("nscratch", C.c_int)]


class ODFir(C.Structure):
    _fields_ = OCFir._fields_


class OCHB(C.Structure):
    _fields_ = [("toggle", C.c_int), ("samples", C.c_double * 44), ("center", C.c_double * 22),
                ("scratch", C.c_void_p), ("nscratch", C.c_int)]


class ODHB(C.Structure):
    _fields_ = [("toggle", C.c_int), ("samples", C.c_double * 22), ("center", C.c_double * 11),
                ("scratch", C.c_void_p), ("nscratch", C.c_int)]


class OAgc(C.Structure):
    _fields_ = [("max_out", C.c_double), ("sample_rate", C.c_int), ("buf_size", C.c_int),
                ("index_read", C.c_int), ("index_start", C.c_int), ("is_clipping", C.c_int),
                ("themax", C.c_double), ("gain", C.c_double), ("delta", C.c_double),
                ("target_gain", C.c_double), ("time_release", C.c_double), ("ring", C.c_void_p)]


class OFrac(C.Structure):
    _fields_ = [("dindex", C.c_double), ("c", C.c_double * 8)]


class OGraph(C.Structure):
    _fields_ = [("fft_size", C.c_int), ("data_width", C.c_int), ("sample_rate", C.c_int),
                ("window", c_dp), ("avg", c_dp), ("count", C.c_int), ("meter", C.c_double),
                ("rx_tune", C.c_int), ("filter_start_offset", C.c_int), ("filter_bandwidth", C.c_int),
                ("scan_blocks", C.c_int), ("scan_sample_rate", C.c_int), ("scan_vfo0", C.c_int), ("scan_deltaf", C.c_int),
                ("tx_tune", C.c_int), ("vfo_screen", C.c_int), ("next_block", C.c_int), ("scan_valid", C.c_double), ("tmp", c_dp)]


class ODcRemove(C.Structure):
    _fields_ = [("old_sample_rate", C.c_int), ("old_bandwidth", C.c_int), ("alpha", C.c_double),
                ("dc_remove", C.c_double * 2), ("dc_average", C.c_double * 2), ("dc_sum", C.c_double * 2),
                ("dc_count", C.c_int), ("dc_key_delay", C.c_int)]


class ONoiseBlanker(C.Structure):
    _fields_ = [("ring", C.c_void_p), ("mags", C.c_void_p), ("sum", C.c_double), ("size", C.c_int), ("hwindow", C.c_int),
                ("state", C.c_int), ("index", C.c_int), ("win_index", C.c_int), ("sample_rate", C.c_int)]


class OB2C(C.Structure):
    _fields_ = [("nbuf1", C.c_int), ("nbuf2", C.c_int), ("buf1", C.c_double * 12000), ("buf2", C.c_double * 12000)]


class OBscope(C.Structure):
    _fields_ = [("size", C.c_int), ("count", C.c_int), ("window", c_dp), ("avg", c_dp), ("the_max", C.c_double)]


_oracle = None


def oracle():
    global _oracle
    if _oracle is not None:
        return _oracle
    if not os.path.exists(ORACLE_PATH):
        raise RuntimeError("oracle library missing: run `make -C oracle` (or __graft_entry__.build())")
    L = C.CDLL(ORACLE_PATH)
    P = C.POINTER
    L.qko_table.restype = c_dp
    L.qko_table.argtypes = [C.c_char_p, P(C.c_int)]
    L.qko_cfir_init.argtypes = [P(OCFir), c_dp, C.c_int]
    L.qko_dfir_init.argtypes = [P(ODFir), c_dp, C.c_int]
    L.qko_cfir_free.argtypes = [P(OCFir)]
    L.qko_dfir_free.argtypes = [P(ODFir)]
    L.qko_filt_tune.argtypes = [P(ODFir), C.c_double, C.c_int]
    for nm in ("qko_cdecimate", "qko_ccdecimate", "qko_cinterpolate"):
        getattr(L, nm).argtypes = [cdp, C.c_int, P(OCFir), C.c_int]
    L.qko_ddecimate.argtypes = [ddp, C.c_int, P(ODFir), C.c_int]
    L.qko_dinterpolate.argtypes = [ddp, C.c_int, P(ODFir), C.c_int]
    L.qko_dfilter.argtypes = [ddp, C.c_int, P(ODFir)]
    L.qko_dc_out_block.argtypes = [ddp, C.c_int, P(ODFir), cdp]
    L.qko_dd_out_block.argtypes = [ddp, C.c_int, P(ODFir)]
    L.qko_hermes_unpack.argtypes = [C.c_void_p, C.c_int, cdp, C.c_void_p, C.c_long, P(C.c_int)]
    L.qko_cinterpdecim.argtypes = [cdp, C.c_int, P(OCFir), C.c_int, C.c_int]
    L.qko_cdecim2hb45.argtypes = [cdp, C.c_int, P(OCHB)]
    L.qko_cinterp2hb45.argtypes = [cdp, C.c_int, P(OCHB)]
    L.qko_dinterp2hb45.argtypes = [ddp, C.c_int, P(ODHB)]
    L.qko_plan_decimation.argtypes = [C.c_int, P(C.c_int), P(C.c_int), P(C.c_int)]
    L.qko_decim_rate.argtypes = [C.c_int]
    L.qko_get_filter_rate.argtypes = [C.c_int, C.c_int, C.c_int]
    L.qko_rx_new.restype = C.c_void_p
    L.qko_rx_new.argtypes = [C.c_int, C.c_int]
    L.qko_rx_free.argtypes = [C.c_void_p]
    L.qko_rx_set_tune.argtypes = [C.c_void_p, C.c_int]
    L.qko_rx_set_mode.argtypes = [C.c_void_p, C.c_int]
    L.qko_rx_set_filter.argtypes = [C.c_void_p, ddp, ddp, C.c_int, C.c_int]
    L.qko_rx_set_agc.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.qko_rx_set_squelch.argtypes = [C.c_void_p, C.c_double]
    L.qko_rx_set_kill_audio.argtypes = [C.c_void_p, C.c_int]
    L.qko_rx_squelch_active.argtypes = [C.c_void_p]
    L.qko_rx_set_ssb_squelch.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.qko_rx_set_own_resampler.argtypes = [C.c_void_p, C.c_int]
    L.qko_rx_set_auto_notch.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.qko_rx_process.argtypes = [C.c_void_p, cdp, C.c_int]
    L.qko_rx_pre_agc.argtypes = [C.c_void_p, cdp, C.c_int]
    L.qko_rx_agc.argtypes = [C.c_void_p, cdp, C.c_int]
    L.qko_rx_mix.argtypes = [C.c_void_p, cdp, C.c_int]
    L.qko_rx_decimate.argtypes = [C.c_void_p, cdp, C.c_int]
    L.qko_rx_demodulate.argtypes = [C.c_void_p, cdp, ddp, C.c_int]
    L.qko_rx_bandpass_block.argtypes = [C.c_void_p, cdp, C.c_int]
    L.qko_rx_decim_srate.argtypes = [C.c_void_p]
    L.qko_rx_filter_srate.argtypes = [C.c_void_p]
    L.qko_agc_init.argtypes = [P(OAgc), C.c_double, C.c_int]
    L.qko_agc_free.argtypes = [P(OAgc)]
    L.qko_agc_process.argtypes = [P(OAgc), cdp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
    L.qko_agc_trace.argtypes = [P(OAgc), cdp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, ddp, idp, idp]
    L.qko_fracdecim_init.argtypes = [P(OFrac)]
    L.qko_fracdecim.argtypes = [P(OFrac), cdp, C.c_int, C.c_double]
    L.qko_fft.argtypes = [cdp, C.c_int]
    L.qko_multirx_graph.argtypes = [cdp, C.c_int, ddp]
    L.qko_dcremove_init.argtypes = [P(ODcRemove)]
    L.qko_dcremove.argtypes = [P(ODcRemove), cdp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.qko_invert_spectrum.argtypes = [cdp, C.c_int]
    L.qko_nb_init.argtypes = [P(ONoiseBlanker)]
    L.qko_nb_free.argtypes = [P(ONoiseBlanker)]
    L.qko_noise_blanker.argtypes = [P(ONoiseBlanker), cdp, C.c_int, C.c_int, C.c_int]
    L.qko_bscope_init.argtypes = [P(OBscope), C.c_int]
    L.qko_bscope_free.argtypes = [P(OBscope)]
    L.qko_bscope_add.argtypes = [P(OBscope), ddp]
    L.qko_bscope_release.argtypes = [P(OBscope), C.c_int, C.c_double, C.c_double, ddp, C.c_int, P(C.c_double)]
    L.qko_hann_window.argtypes = [ddp, C.c_int]
    L.qko_graph_init.argtypes = [P(OGraph), C.c_int, C.c_int, C.c_int]
    L.qko_graph_free.argtypes = [P(OGraph)]
    L.qko_graph_set_passband.argtypes = [P(OGraph), C.c_int, C.c_int, C.c_int]
    L.qko_graph_add_frame.argtypes = [P(OGraph), cdp]
    L.qko_graph_change_scan.argtypes = [P(OGraph), C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
    L.qko_graph_release.argtypes = [P(OGraph), C.c_double, C.c_double, ddp, P(C.c_double)]
    L.qko_buffer2chan.argtypes = [P(OB2C), ddp, C.c_int, ddp, C.c_int]
    L.qko_add_test_tone.argtypes = [C.c_void_p, cdp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.qko_tone_init.argtypes = [C.c_void_p]
    L.qko_get_filter.argtypes = [ddp, ddp, C.c_int, C.c_int, C.c_int, ddp]
    L.qko_mfreq_new.restype = C.c_void_p
    L.qko_mfreq_delete.argtypes = [C.c_void_p]
    L.qko_measure_freq.argtypes = [C.c_void_p, cdp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.qko_measure_freq.restype = C.c_double
    L.qko_audio_graph_new.restype = C.c_void_p
    L.qko_audio_graph_new.argtypes = [C.c_int]
    L.qko_audio_graph_delete.argtypes = [C.c_void_p]
    L.qko_audio_graph_get.argtypes = [C.c_void_p, ddp]
    L.qko_rx_set_audio_graph.argtypes = [C.c_void_p, C.c_void_p]
    L.qko_rx_measure_audio.argtypes = [C.c_void_p, C.c_int]
    L.qko_rx_measure_audio.restype = C.c_double
    L.qko_pair_new.restype = C.c_void_p
    L.qko_pair_delete.argtypes = [C.c_void_p]
    L.qko_pair_reset_fifos.argtypes = [C.c_void_p]
    L.qko_rx_pair_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, cdp, C.c_int, C.c_int, cdp]
    L.qko_keyup_envelope.argtypes = [C.c_double, cdp, C.c_int, C.c_int]
    L.qko_keyup_envelope.restype = C.c_double
    L.qko_rx_process_many.argtypes = [P(C.c_void_p), C.c_int, cdp, C.c_int, cdp, C.c_long, idp]
    _oracle = L
    return L


def oracle_table(name):
    n = C.c_int(0)
    p = oracle().qko_table(name.encode(), C.byref(n))
    if not p:
        raise KeyError(name)
    return np.ctypeslib.as_array(p, shape=(n.value,)).copy()


class OracleRx:
    """One reference-semantics receiver (quisk_process_samples for bank 0)."""

    def __init__(self, sample_rate, playback_rate=48000, mode=3, tune=0, fi=None, fq=None, bandwidth=0,
                 agc_gain=80.0, agc_release_time=1.0):
        self.L = oracle()
        self.h = self.L.qko_rx_new(sample_rate, playback_rate)
        self.ratio = playback_rate / sample_rate
        self.L.qko_rx_set_mode(self.h, mode)
        self.L.qko_rx_set_tune(self.h, tune)
        self.L.qko_rx_set_agc(self.h, agc_gain, agc_release_time)
        if fi is not None:
            fi = np.ascontiguousarray(fi, dtype=np.float64)
            fq = np.ascontiguousarray(fq, dtype=np.float64)
            assert self.L.qko_rx_set_filter(self.h, fi, fq, len(fi), bandwidth) == 0

    def set_tune(self, hz):
        self.L.qko_rx_set_tune(self.h, hz)

    def set_squelch(self, level_db):
        self.L.qko_rx_set_squelch(self.h, level_db)

    def set_kill_audio(self, on):
        self.L.qko_rx_set_kill_audio(self.h, int(on))

    def set_ssb_squelch(self, enabled, level):
        self.L.qko_rx_set_ssb_squelch(self.h, int(enabled), int(level))

    def set_auto_notch(self, on, rit_freq=0):
        self.L.qko_rx_set_auto_notch(self.h, int(on), int(rit_freq))

    def squelch_active(self):
        return self.L.qko_rx_squelch_active(self.h)

    def _cap(self, n):
        """work-buffer size: the chain runs in place; only a play rate above the input rate makes it grow"""
        return max(n, int(n * self.ratio * 1.1)) + 4096

    def process(self, x):
        """x: complex128 block (<= 66000); returns the stereo audio block."""
        buf = np.array(x, dtype=np.complex128, copy=True)
        cap = max(len(buf), 16)
        work = np.zeros(self._cap(cap), dtype=np.complex128)
        work[:len(buf)] = buf
        m = self.L.qko_rx_process(self.h, work, len(buf))
        return work[:m].copy()

    def process_parts(self, x):
        """(pre-AGC stereo block at the play rate, final block)"""
        n = len(x)
        work = np.zeros(self._cap(n), dtype=np.complex128)
        work[:n] = x
        m = self.L.qko_rx_pre_agc(self.h, work, n)
        pre = work[:m].copy()
        self.L.qko_rx_agc(self.h, work, m)
        return pre, work[:m].copy()

    def process_stream(self, x, block):
        outs = [self.process(x[i:i + block]) for i in range(0, len(x), block)]
        return np.concatenate(outs) if outs else np.zeros(0, np.complex128)

    def close(self):
        if self.h:
            self.L.qko_rx_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------- reference structs
class RCFilter(C.Structure):      # struct quisk_cFilter, filter.h:1-10
    _fields_ = [("dCoefs", c_dp), ("cpxCoefs", C.c_void_p), ("nBuf", C.c_int), ("nTaps", C.c_int),
                ("decim_index", C.c_int), ("cSamples", C.c_void_p), ("ptcSamp", C.c_void_p), ("cBuf", C.c_void_p)]


class RDFilter(C.Structure):      # struct quisk_dFilter, filter.h:12-21
    _fields_ = [("dCoefs", c_dp), ("cpxCoefs", C.c_void_p), ("nBuf", C.c_int), ("nTaps", C.c_int),
                ("decim_index", C.c_int), ("dSamples", C.c_void_p), ("ptdSamp", C.c_void_p), ("dBuf", C.c_void_p)]


class RCd(C.Structure):           # complex double returned by value
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class RCHB(C.Structure):          # struct quisk_cHB45Filter, filter.h:23-29
    _fields_ = [("cBuf", C.c_void_p), ("nBuf", C.c_int), ("toggle", C.c_int),
                ("samples", C.c_double * 44), ("center", C.c_double * 22)]


class RDHB(C.Structure):          # struct quisk_dHB45Filter, filter.h:31-37
    _fields_ = [("dBuf", C.c_void_p), ("nBuf", C.c_int), ("toggle", C.c_int),
                ("samples", C.c_double * 22), ("center", C.c_double * 11)]


_ref = None


def have_ref():
    return os.path.exists(REF_PATH)


def ref():
    global _ref
    if _ref is not None:
        return _ref
    L = C.CDLL(REF_PATH)
    P = C.POINTER
    L.quisk_filt_cInit.argtypes = [P(RCFilter), c_dp, C.c_int]
    L.quisk_filt_dInit.argtypes = [P(RDFilter), c_dp, C.c_int]
    L.quisk_filt_tune.argtypes = [P(RDFilter), C.c_double, C.c_int]
    for nm in ("quisk_cDecimate", "quisk_cCDecimate", "quisk_cInterpolate"):
        getattr(L, nm).argtypes = [cdp, C.c_int, P(RCFilter), C.c_int]
    L.quisk_dDecimate.argtypes = [ddp, C.c_int, P(RDFilter), C.c_int]
    L.quisk_dInterpolate.argtypes = [ddp, C.c_int, P(RDFilter), C.c_int]
    L.quisk_dFilter.argtypes = [ddp, C.c_int, P(RDFilter)]
    L.quisk_dD_out.argtypes = [C.c_double, P(RDFilter)]
    L.quisk_dD_out.restype = C.c_double
    L.quisk_dC_out.argtypes = [C.c_double, P(RDFilter)]
    L.quisk_dC_out.restype = RCd
    L.quisk_cInterpDecim.argtypes = [cdp, C.c_int, P(RCFilter), C.c_int, C.c_int]
    L.quisk_cDecim2HB45.argtypes = [cdp, C.c_int, P(RCHB)]
    L.quisk_cInterp2HB45.argtypes = [cdp, C.c_int, P(RCHB)]
    L.quisk_dInterp2HB45.argtypes = [ddp, C.c_int, P(RDHB)]
    _ref = L
    return L


def ref_table(symbol, n):
    arr = (C.c_double * n).in_dll(ref(), symbol)
    return np.array(arr[:], dtype=np.float64)


class OracleGraph:
    """get_graph restated (quisk.c:4602-4759): feed whole frames, then release."""

    def __init__(self, fft_size, data_width, sample_rate):
        self.L = oracle()
        self.g = OGraph()
        self.N, self.W = fft_size, data_width
        self.L.qko_graph_init(C.byref(self.g), fft_size, data_width, sample_rate)

    def set_passband(self, rx_tune, start_offset, bandwidth):
        self.L.qko_graph_set_passband(C.byref(self.g), int(rx_tune), int(start_offset), int(bandwidth))

    def change_scan(self, blocks, rate, valid, vfo0, deltaf, tx_tune=0, vfo_screen=0):
        self.L.qko_graph_change_scan(C.byref(self.g), int(blocks), int(rate), float(valid), int(vfo0), int(deltaf), int(tx_tune), int(vfo_screen))

    def add_frames(self, x):
        assert len(x) % self.N == 0
        for i in range(0, len(x), self.N):
            fr = np.array(x[i:i + self.N], dtype=np.complex128, copy=True)
            self.L.qko_graph_add_frame(C.byref(self.g), fr)

    def release(self, zoom=1.0, deltaf=0.0):
        out = np.zeros(self.W)
        sm = C.c_double(0)
        n = self.L.qko_graph_release(C.byref(self.g), zoom, deltaf, out, C.byref(sm))
        return n, out, sm.value

    def close(self):
        self.L.qko_graph_free(C.byref(self.g))


def oracle_multirx_graph(iq, data_width):
    """get_multirx_graph restated (quisk.c:4468-4530)."""
    L = oracle()
    x = np.array(iq, dtype=np.complex128, copy=True)
    assert len(x) == 8 * data_width
    out = np.zeros(data_width)
    L.qko_multirx_graph(x, int(data_width), out)
    return out


class OracleBscope:
    """init_bandscope / get_bandscope restated (quisk.c:2758-2775, 4532-4600)."""

    def __init__(self, size):
        self.L = oracle()
        self.b = OBscope()
        self.L.qko_bscope_init(C.byref(self.b), int(size))

    def add(self, samples):
        v = np.ascontiguousarray(samples, dtype=np.float64)
        assert len(v) == self.b.size
        self.L.qko_bscope_add(C.byref(self.b), v)

    def release(self, clock, zoom, deltaf, graph_width):
        out = np.zeros(graph_width)
        lvl = C.c_double(0)
        n = self.L.qko_bscope_release(C.byref(self.b), int(clock), float(zoom), float(deltaf), out, int(graph_width), C.byref(lvl))
        return n, out, lvl.value

    def close(self):
        self.L.qko_bscope_free(C.byref(self.b))


class OracleConditioner:
    """DCremove (sound.c:165-230), spectrum inversion (quisk.c:2315-2320), NoiseBlanker (quisk.c:615-720), as the reference
    applies them in front of the FFT staging and the mixer."""

    def __init__(self, sample_rate, dc_remove_bw=0, invert=False, noise_blanker=0):
        self.L = oracle()
        self.fs, self.bw, self.invert, self.nb_level = sample_rate, dc_remove_bw, invert, noise_blanker
        self.d = ODcRemove()
        self.L.qko_dcremove_init(C.byref(self.d))
        self.nb = ONoiseBlanker()
        self.L.qko_nb_init(C.byref(self.nb))

    def __call__(self, x):
        y = np.array(x, dtype=np.complex128, copy=True)
        self.L.qko_dcremove(C.byref(self.d), y, len(y), self.fs, self.bw, 0)
        if self.invert:
            self.L.qko_invert_spectrum(y, len(y))
        self.L.qko_noise_blanker(C.byref(self.nb), y, len(y), self.fs, self.nb_level)      # quisk.c:2322
        return y
