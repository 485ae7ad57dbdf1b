"""ctypes mirror of include/qk_engine.h.  Loads quisk_amd/libqkengine.so and fails loudly if it is missing:
there is no CPU path in this package."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QK_ENGINE_LIB") or os.path.join(_HERE, "libqkengine.so")    # (QK_ENGINE_LIB: an instrumented build of the same library)


class QkError(RuntimeError):
    pass


class QkConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("sample_rate", C.c_int), ("playback_rate", C.c_int),
                ("n_receivers", C.c_int), ("max_block", C.c_int), ("fft_size", C.c_int),
                ("data_width", C.c_int), ("reserved", C.c_int * 8)]


class QkStats(C.Structure):
    _fields_ = [("samples_in", C.c_int64), ("blocks", C.c_int64), ("fft_frames", C.c_int64),
                ("fft_error", C.c_int64), ("last_block_ms", C.c_double), ("block_ms_sum", C.c_double),
                ("blocks_timed", C.c_int64), ("kernel_ms_sum", C.c_double * 8), ("kernel_launches", C.c_int64 * 8)]


KT_FRONTEND, KT_DECIM, KT_DEMOD, KT_AUDIO, KT_AGC, KT_FFT = range(6)
TAP_FRONTEND, TAP_DECIM, TAP_DEMOD, TAP_AUDIO48, TAP_PLAY, TAP_ROUTE = range(6)

_lib = None


def lib():
    """The loaded C-ABI library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QkError("%s is missing: build it with `make` or __graft_entry__.build()" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.qk_last_error.restype = C.c_char_p
    L.qk_version.restype = C.c_char_p
    L.qk_create.argtypes = [C.POINTER(vp), C.POINTER(QkConfig)]
    L.qk_destroy.argtypes = [vp]
    L.qk_destroy.restype = None
    L.qk_set_tune.argtypes = [vp, C.c_int, C.c_int]
    L.qk_set_mode.argtypes = [vp, C.c_int, C.c_int]
    L.qk_set_filters.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int]
    L.qk_set_agc.argtypes = [vp, C.c_int, C.c_double]
    L.qk_set_agc_release_time.argtypes = [vp, C.c_double]
    L.qk_process_samples.argtypes = [vp, vp, C.c_int]
    L.qk_process_samples_device.argtypes = [vp, vp, C.c_int]
    L.qk_process_samples_int.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int]
    L.qk_sync.argtypes = [vp]
    L.qk_audio_count.argtypes = [vp, C.c_int]
    L.qk_pull_audio.argtypes = [vp, C.c_int, vp, C.c_int]
    L.qk_pull_audio_all.argtypes = [vp, vp, C.c_size_t, ip]
    L.qk_audio_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.qk_get_graph.argtypes = [vp, C.c_double, C.c_double, vp, C.c_int, dp]
    L.qk_graph_pending.argtypes = [vp]
    L.qk_graph_mark.argtypes = [vp, C.c_double, C.c_double]
    L.qk_set_ssb_squelch.argtypes = [vp, C.c_int, C.c_int]
    L.qk_set_auto_notch.argtypes = [vp, C.c_int, C.c_int]
    L.qk_set_rit_freq.argtypes = [vp, C.c_int]
    L.qk_is_squelch.argtypes = [vp, C.c_int]
    L.qk_set_stereo_route.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.qk_key_up.argtypes = [vp]
    L.qk_change_scan.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
    L.qk_set_graph_raw.argtypes = [vp, C.c_int]
    L.qk_get_graph_raw.argtypes = [vp, vp, C.c_int]
    L.qk_rx_filter_block.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int]
    L.qk_set_dc_remove.argtypes = [vp, C.c_int]
    L.qk_set_invert_spectrum.argtypes = [vp, C.c_int]
    L.qk_set_noise_blanker.argtypes = [vp, C.c_int]
    L.qk_set_squelch.argtypes = [vp, C.c_double]
    L.qk_set_kill_audio.argtypes = [vp, C.c_int]
    L.qk_squelch_active.argtypes = [vp, C.c_int]
    L.qk_multirx_graph.argtypes = [vp, vp, C.c_int, vp]
    L.qk_bandscope_init.argtypes = [vp, C.c_int]
    L.qk_add_bscope_samples.argtypes = [vp, vp, C.c_int]
    L.qk_add_bscope_samples_int.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int]
    L.qk_get_bandscope.argtypes = [vp, C.c_int, C.c_double, C.c_double, vp, C.c_int, dp]
    L.qk_get_smeter.argtypes = [vp]
    L.qk_get_smeter.restype = C.c_double
    L.qk_graph_device.argtypes = [vp, C.POINTER(vp)]
    L.qk_get_filter_rate.argtypes = [C.c_int, C.c_int, C.c_int]
    L.qk_plan_decimation.argtypes = [C.c_int, ip, ip, ip]
    L.qk_decim_rate.argtypes = [C.c_int]
    L.qk_get_stats.argtypes = [vp, C.POINTER(QkStats)]
    L.qk_set_timing.argtypes = [vp, C.c_int]
    L.qk_reset_stats.argtypes = [vp]
    L.qk_export_audio_device.argtypes = [vp, vp, C.c_size_t]
    L.qk_debug_tap.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    L.qk_debug_agc_passes.argtypes = [vp, C.c_int]
    L.qk_debug_agc_passes.restype = C.c_longlong
    L.quisk_cCplxSampleFilter.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, ip]
    L.quisk_ccfFilter.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, ip]
    L.qk_cInterpolate.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp]
    L.qk_dC_out_block.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
    L.qk_dD_out_block.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.qk_process_hermes.argtypes = [vp, vp, C.c_int, C.c_int, vp, ip, ip]
    L.quisk_cCplxSampleFilter_noalloc.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, ip, vp]
    L.qk_comm_unique_id.argtypes = [vp]
    L.qk_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.qk_comm_destroy.argtypes = [vp]
    L.qk_gather_audio.argtypes = [vp, C.c_int, vp, vp]
    L.qk_gather_graph.argtypes = [vp, C.c_int, vp]
    L.qk_add_tone.argtypes = [vp, C.c_int]
    L.qk_get_filter.argtypes = [vp, vp, C.c_int]
    L.qk_measure_audio.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.qk_measure_frequency.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.qk_get_audio_graph.argtypes = [vp, vp, C.c_int]
    L.qk_fft_any.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise QkError("qk error %d: %s" % (rc, lib().qk_last_error().decode()))
    return rc


def plan_decimation(rate):
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    best = lib().qk_plan_decimation(rate, C.byref(a), C.byref(b), C.byref(c))
    return best, a.value, b.value, c.value


def get_filter_rate(rate, mode, bw):
    return lib().qk_get_filter_rate(rate, mode, bw)


def decim_rate(rate):
    return lib().qk_decim_rate(rate)


class Engine:
    """N receivers on one wideband I/Q stream (the reference's quisk_process_samples, batched)."""

    def __init__(self, sample_rate, n_receivers=1, playback_rate=48000, max_block=66000, fft_size=0,
                 data_width=0, device=0, max_filter_taps=0, fp32=False, short_blocks=None):
        self.L = lib()
        cfg = QkConfig(device=device, sample_rate=sample_rate, playback_rate=playback_rate,
                       n_receivers=n_receivers, max_block=max_block, fft_size=fft_size, data_width=data_width)
        cfg.reserved[0] = max_filter_taps
        cfg.reserved[1] = 1 if fp32 else 0          # optional fp32 mode (BASELINE configs[4]); fp64 is the reference's arithmetic
        cfg.reserved[2] = 0 if short_blocks is None else (1 if short_blocks else 2)   # short-block mode: None = when max_block <= 4096
        self.fp32 = bool(fp32)
        self.h = C.c_void_p()
        _check(self.L.qk_create(C.byref(self.h), C.byref(cfg)))
        self.n_receivers = n_receivers
        self.sample_rate = sample_rate
        self.playback_rate = playback_rate
        self.max_block = max_block
        self.data_width = data_width
        self.out_cap = int(max_block * playback_rate // sample_rate + 256)

    def close(self):
        if getattr(self, "h", None):
            self.L.qk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setters (names of the reference's Python entry points) ----
    def set_tune(self, rx, hz):
        _check(self.L.qk_set_tune(self.h, rx, int(hz)))

    def set_mode(self, rx, mode):
        _check(self.L.qk_set_mode(self.h, rx, int(mode)))

    def set_filters(self, rx, filt_i, filt_q, bandwidth, start_offset):
        fi = np.ascontiguousarray(filt_i, dtype=np.float64)
        fq = np.ascontiguousarray(filt_q, dtype=np.float64)
        if len(fi) != len(fq):
            raise QkError("The size of filters I and Q must be equal")      # quisk.c:4212-4215
        _check(self.L.qk_set_filters(self.h, rx, fi.ctypes.data, fq.ctypes.data, len(fi), int(bandwidth), int(start_offset)))

    def set_agc(self, rx, gain):
        _check(self.L.qk_set_agc(self.h, rx, float(gain)))

    def set_dc_remove(self, bandwidth):
        """dc_remove_bw of DCremove (sound.c:165-230): 0 off, 1 running average, > 1 one-pole blocker of that bandwidth."""
        _check(self.L.qk_set_dc_remove(self.h, int(bandwidth)))

    def set_invert_spectrum(self, on):
        _check(self.L.qk_set_invert_spectrum(self.h, int(bool(on))))

    def set_squelch(self, level_db):
        """set_squelch (quisk.c:4348-4354): FM squelch level in dB re full scale."""
        _check(self.L.qk_set_squelch(self.h, float(level_db)))

    def squelch_active(self, rx=0):
        return _check(self.L.qk_squelch_active(self.h, rx))

    def set_kill_audio(self, on):
        """set_kill_audio (quisk.c:4364-4370)."""
        _check(self.L.qk_set_kill_audio(self.h, int(bool(on))))

    def set_ssb_squelch(self, enabled, level):
        """set_ssb_squelch (quisk.c:4356-4362)."""
        _check(self.L.qk_set_ssb_squelch(self.h, int(enabled), int(level)))

    def set_auto_notch(self, rx, on):
        """set_auto_notch (quisk.c:4237-4244), per receiver."""
        _check(self.L.qk_set_auto_notch(self.h, int(rx), int(on)))

    def set_rit_freq(self, hz):
        _check(self.L.qk_set_rit_freq(self.h, int(hz)))

    def is_squelch(self, freq):
        """IsSquelch (quisk.c:1231-1252)."""
        return _check(self.L.qk_is_squelch(self.h, int(freq)))

    def set_stereo_route(self, rx, partner_rx, method):
        """Second-receiver routing (quisk.c:2410-2494): 1 = (rx, partner), 2 = (partner, rx), 3 = partner on both, 4 = rx on both through the stereo
        path, 0 = off."""
        _check(self.L.qk_set_stereo_route(self.h, int(rx), int(partner_rx), int(method)))

    def route_tap(self, rx):
        """(debug) the pair of receiver rx's live stereo route in front of its two AGCs: [2][count], left and right"""
        return self.tap(TAP_ROUTE, rx, planes=2, cap=self.out_cap + 32768)

    # ---- multi-GPU: the output gather over RCCL, inside the library (include/qk_engine.h) ----
    @staticmethod
    def comm_unique_id():
        """128 bytes made by rank 0 (ncclGetUniqueId), to be handed to every rank"""
        buf = C.create_string_buffer(128)
        _check(lib().qk_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid, rank, world):
        _check(self.L.qk_comm_init(self.h, C.c_char_p(bytes(uid)), int(rank), int(world)))

    def comm_destroy(self):
        _check(self.L.qk_comm_destroy(self.h))

    def gather_audio(self, dst_rank, dst_dev_ptr, counts_dev_ptr):
        """every rank's last audio block to dst_rank: [world][n_receivers][stride] complex + [world][n_receivers] counts (device)"""
        _check(self.L.qk_gather_audio(self.h, int(dst_rank), C.c_void_p(dst_dev_ptr), C.c_void_p(counts_dev_ptr)))

    def gather_graph(self, dst_rank, dst_dev_ptr):
        _check(self.L.qk_gather_graph(self.h, int(dst_rank), C.c_void_p(dst_dev_ptr)))

    def add_tone(self, freq_hz):
        """add_tone (quisk.c:3022-3034): the test tone of AddTestTone at freq_hz from the centre; 0 = off."""
        _check(self.L.qk_add_tone(self.h, int(freq_hz)))

    def get_filter(self, data_width):
        """get_filter (quisk.c:4761-4848): frequency response of receiver 0's band-pass, dB, negative frequencies first."""
        out = np.zeros(int(data_width), dtype=np.float64)
        _check(self.L.qk_get_filter(self.h, out.ctypes.data, int(data_width)))
        return out

    def measure_audio(self, time):
        v = C.c_double(0.0)
        _check(self.L.qk_measure_audio(self.h, int(time), C.byref(v)))
        return v.value

    def measure_frequency(self, mode):
        v = C.c_double(0.0)
        _check(self.L.qk_measure_frequency(self.h, int(mode), C.byref(v)))
        return v.value

    def get_audio_graph(self):
        """get_audio_graph (quisk.c:966-991): data_width dB values, or None when no graph is ready."""
        out = np.zeros(self.data_width, dtype=np.float64)
        got = _check(self.L.qk_get_audio_graph(self.h, out.ctypes.data, self.data_width))
        return out if got else None

    def fft_any(self, x, inverse=False):
        """the transform behind dft / idft (quisk.c:4922-5014): forward, or FFTW's unnormalised backward"""
        x = np.ascontiguousarray(x, dtype=np.complex128)
        out = np.empty_like(x)
        _check(self.L.qk_fft_any(self.h, x.ctypes.data, len(x), int(bool(inverse)), out.ctypes.data))
        return out

    def key_up(self):
        """The key went up: the 5 ms envelope of quisk.c:2602-2612 starts with the next block."""
        _check(self.L.qk_key_up(self.h))

    def set_noise_blanker(self, level):
        """set_noise_blanker (quisk.c:4246-4252): 0 off, 1 / 2 / 3 = pulse limit 6 / 4 / 2.5."""
        _check(self.L.qk_set_noise_blanker(self.h, int(level)))

    def set_agc_release_time(self, seconds):
        _check(self.L.qk_set_agc_release_time(self.h, float(seconds)))

    def set_timing(self, on, groups=None):
        """on: time every kernel group; groups: iterable of KT_* to time only those (cheaper)."""
        v = int(bool(on))
        if on and groups is not None:
            v = 0
            for g in groups:
                v |= 2 << int(g)
        _check(self.L.qk_set_timing(self.h, v))

    # ---- data path ----
    def process(self, iq):
        """One block of complex128 samples in host memory."""
        x = np.ascontiguousarray(iq, dtype=np.complex128)
        _check(self.L.qk_process_samples(self.h, x.ctypes.data, len(x)))

    def process_device(self, dev_ptr, n):
        """One block already resident in device memory (dev_ptr: integer address of n complex doubles)."""
        _check(self.L.qk_process_samples_device(self.h, C.c_void_p(int(dev_ptr)), int(n)))

    def process_int(self, raw, nbytes_per_component, big_endian=False):
        buf = np.frombuffer(raw, dtype=np.uint8)
        _check(self.L.qk_process_samples_int(self.h, buf.ctypes.data, buf.size, nbytes_per_component, int(big_endian)))

    def process_hermes(self, packets, multirx_count=0, want_sub=False):
        """read_rx_udp10's wire format (quisk.c:3341-3550): `packets` = whole 1032-byte Hermes receive packets.
        Returns (samples processed, read errors, overrange count, sub-receiver samples or None)."""
        buf = np.frombuffer(packets, dtype=np.uint8)
        npk = buf.size // 1032
        nrec = 504 // ((multirx_count + 1) * 6 + 2)
        sub = np.zeros((multirx_count, npk * 2 * nrec), dtype=np.complex128) if (want_sub and multirx_count) else None
        err, over = C.c_int(0), C.c_int(0)
        n = _check(self.L.qk_process_hermes(self.h, buf.ctypes.data, npk, int(multirx_count), sub.ctypes.data if sub is not None else None,
                                            C.byref(err), C.byref(over)))
        return n, err.value, over.value, sub

    def sync(self):
        _check(self.L.qk_sync(self.h))

    def audio(self, rx):
        cap = max(self.out_cap, _check(self.L.qk_audio_count(self.h, rx)))     # (a stereo route's block can hold what its FIFOs kept back)
        out = np.empty(cap, dtype=np.complex128)
        m = _check(self.L.qk_pull_audio(self.h, rx, out.ctypes.data, cap))
        return out[:m].copy()

    def audio_all(self):
        out = np.empty((self.n_receivers, self.out_cap), dtype=np.complex128)
        counts = (C.c_int * self.n_receivers)()
        _check(self.L.qk_pull_audio_all(self.h, out.ctypes.data, self.out_cap, counts))
        return out, np.array(counts[:], dtype=np.int64)

    def audio_device(self):
        p, s = C.c_void_p(), C.c_size_t()
        _check(self.L.qk_audio_device(self.h, C.byref(p), C.byref(s)))
        return p.value, s.value

    def process_stream(self, iq, block, rx=None):
        """Feed `iq` in blocks; returns the concatenated audio of receiver rx (or a list for all)."""
        outs = [[] for _ in range(self.n_receivers)]
        which = range(self.n_receivers) if rx is None else [rx]
        for i in range(0, len(iq), block):
            self.process(iq[i:i + block])
            for r in which:
                outs[r].append(self.audio(r))
        cat = [np.concatenate(o) if o else np.zeros(0, np.complex128) for o in outs]
        return cat if rx is None else cat[rx]

    def get_graph(self, zoom=1.0, deltaf=0.0):
        """(count, dB row, smeter); count == 0 means no data, like get_graph returning None."""
        out = np.empty(max(self.data_width, 1), dtype=np.float64)
        sm = C.c_double(0.0)
        n = _check(self.L.qk_get_graph(self.h, float(zoom), float(deltaf), out.ctypes.data, self.data_width, C.byref(sm)))
        return n, out, sm.value

    def graph_mark(self, zoom=1.0, deltaf=0.0):
        """Queue the release of the frames averaged so far and start a new average; the next get_graph returns that row
        and waits for nothing queued after the mark.  Returns the number of frames marked."""
        return _check(self.L.qk_graph_mark(self.h, float(zoom), float(deltaf)))

    def change_scan(self, scan_blocks, scan_sample_rate=0, scan_valid=0.84, scan_vfo0=0, scan_deltaf=0, tx_tune=0, vfo_screen=0):
        """change_scan (quisk.c:3865-3872): stitch the frames of a VFO sweep; scan_blocks = 0 returns to plain averaging."""
        _check(self.L.qk_change_scan(self.h, int(scan_blocks), int(scan_sample_rate), float(scan_valid), int(scan_vfo0), int(scan_deltaf),
                                     int(tx_tune), int(vfo_screen)))

    def set_graph_raw(self, on):
        """get_graph's use_fft == 0 mode (quisk.c:4642-4649)."""
        _check(self.L.qk_set_graph_raw(self.h, int(bool(on))))

    def get_graph_raw(self):
        """The first data_width samples of the oldest queued frame, or None."""
        out = np.empty(self.data_width, dtype=np.complex128)
        n = _check(self.L.qk_get_graph_raw(self.h, out.ctypes.data, self.data_width))
        return out if n else None

    def rx_filter_block(self, rx, bank, samples):
        """cRxFilterOut (quisk.c:1154-1192) over a block, with the history of `bank`."""
        x = np.ascontiguousarray(samples, dtype=np.complex128)
        out = np.empty_like(x)
        _check(self.L.qk_rx_filter_block(self.h, int(rx), int(bank), x.ctypes.data, out.ctypes.data, len(x)))
        return out

    def smeter(self):
        """get_smeter (quisk.c:2737-2742): the value of the last graph release."""
        return float(self.L.qk_get_smeter(self.h))

    def multirx_graph(self, iq, data_width):
        """get_multirx_graph's transform of one sub-receiver frame: 8 * data_width complex samples -> dB row."""
        iq = np.ascontiguousarray(iq, dtype=np.complex128)
        if len(iq) != 8 * data_width:
            raise ValueError("a multirx frame is 8 * data_width samples")
        out = np.empty(data_width, dtype=np.float64)
        _check(self.L.qk_multirx_graph(self.h, iq.ctypes.data, int(data_width), out.ctypes.data))
        return out

    def bandscope_init(self, size):
        _check(self.L.qk_bandscope_init(self.h, int(size)))

    def add_bscope_samples(self, samples):
        """One bandscope block: real samples scaled to +-1 (float64), or raw bytes via add_bscope_samples_int."""
        v = np.ascontiguousarray(samples, dtype=np.float64)
        _check(self.L.qk_add_bscope_samples(self.h, v.ctypes.data, len(v)))

    def add_bscope_samples_int(self, raw, nbytes, big_endian=False):
        buf = np.frombuffer(raw, dtype=np.uint8)
        _check(self.L.qk_add_bscope_samples_int(self.h, buf.ctypes.data, len(buf), int(nbytes), int(bool(big_endian))))

    def get_bandscope(self, clock, zoom, deltaf, graph_width):
        """(count, dB row, adc_level); count == 0 means no data (get_bandscope returning None)."""
        out = np.empty(graph_width, dtype=np.float64)
        lvl = C.c_double(0.0)
        n = _check(self.L.qk_get_bandscope(self.h, int(clock), float(zoom), float(deltaf), out.ctypes.data, int(graph_width), C.byref(lvl)))
        return n, out, lvl.value

    def reset_stats(self):
        _check(self.L.qk_reset_stats(self.h))

    def export_audio_device(self, dst_dev_ptr, dst_stride):
        _check(self.L.qk_export_audio_device(self.h, C.c_void_p(int(dst_dev_ptr)), int(dst_stride)))

    def graph_pending(self):
        return _check(self.L.qk_graph_pending(self.h))

    def stats(self):
        st = QkStats()
        _check(self.L.qk_get_stats(self.h, C.byref(st)))
        return st

    def tap(self, which, rx, planes=2, cap=None):
        cap = cap or (self.max_block + 64)
        out = np.zeros((planes, cap), dtype=np.float64)
        m = _check(self.L.qk_debug_tap(self.h, which, rx, out.ctypes.data, cap))
        return out[:, :m].copy()
