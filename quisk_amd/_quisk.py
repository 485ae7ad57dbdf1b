"""_quisk -- the reference's extension-module interface for the receive path, over libqkengine.so.

quisk.py and the quisk_hardware_*.py modules reach the receive DSP only through `import _quisk as QS`
(quisk.py:33, quisk_hardware_model.py:15).  This module has the same function names, argument order and error
behaviour for the hot path (method table QuiskMethods, quisk.c:5164-5266), so a maintainer can put it on the path
as `_quisk` (e.g. `sys.modules['_quisk'] = quisk_amd._quisk`) and leave the Python layer unchanged.  It is host
logic only: every sample goes through the C ABI of include/qk_engine.h into the HIP kernels; there is no CPU
fallback (record_app raises `error` when no GPU is present).

What maps to what
  record_app(app, conf, data_width, graph_width, fft_size, multirx_data_width, rate, handle)   quisk.c:5048-5113
        -> qk_create with 1 + QUISK_MAX_RECEIVERS receivers (bank 0 + the sub-receivers, quisk.h:21)
  set_params(rx_bytes=, rx_endian=, bscope_bytes=, bscope_endian=, bscope_size=, ...)           quisk.c:2920-2958
  add_rx_samples(buffer) / read_sound()                                                         quisk.c:2777-2839, 3902-3912
        -> bytes are queued (at most 52800 samples pending, quisk.c:2810) and read_sound() pushes them through
           qk_process_samples_int: the int -> complex double unpack happens on the GPU
  set_tune, set_rx_mode, set_filters, set_agc                                                   quisk.c:4323, 4262, 4192, 4184
  set_multirx_freq / set_multirx_mode                                                           quisk.c:4438, 4426
        -> receivers 1..9.  In the reference each sub-receiver has its own Hermes DDC stream; here they are further
           receivers on the shared wideband stream, tuned by offset like receiver 0.  Filter slot 1 (the slot of
           the sub-receiver being played, quisk.c:2472) goes to all of them.
  get_graph(use_fft, zoom, deltaf)                                                              quisk.c:4602-4759
        -> None until 1/graph_refresh seconds have passed and at least one frame is averaged (quisk.c:4712)
  get_multirx_graph(), add_bscope_samples(buffer), get_bandscope(clock, zoom, deltaf)           quisk.c:4468-4600, 2841-2900
  get_filter_rate(mode, bw), get_smeter(), get_state(), get_overrange()                         quisk.c:2661, 2737, 2618
  set_sparams(dc_remove_bw=), invert_spectrum(i)                                                sound.c:1253, quisk.c:4174
Audio goes to the sound card in the reference (quisk_play_*); here the last block of every receiver is kept and
`get_audio(rx)` returns it (an addition: there is no such call in the reference).
Everything else in QuiskMethods that is outside the path is accepted and ignored (INERT below).
"""
import time

import numpy as np

from . import engine as _eng


class error(Exception):
    """_quisk.error (quisk.c:5311-5317)."""


QUISK_MAX_RECEIVERS = 9          # quisk.h:21
MAX_FILTER_SIZE = 10001          # quisk.c:34
SAMP_BUFFER_SIZE = 66000         # quisk.h:15
MAX_PENDING = 52800              # add_rx_samples keeps SAMP_BUFFER_SIZE * 8 / 10 samples at most, quisk.c:2810

_S = None                        # the one engine of the process, like the file-scope state of quisk.c


class _State:
    def __init__(self):
        self.eng = None
        self.conf = None
        self.app = None
        self.data_width = self.graph_width = self.fft_size = self.multirx_data_width = self.rate = 0
        self.rx_bytes, self.rx_endian = 3, 1          # py_sample_rx_bytes / endian defaults, quisk.c:203-204
        self.bscope_bytes, self.bscope_endian, self.bscope_size = 2, 1, 0
        self.pending = bytearray()
        self.graph_time0 = 0.0
        self.use_fft = 1
        self.multirx_frames = [bytearray() for _ in range(QUISK_MAX_RECEIVERS)]
        self.multirx_count = 0
        self.multirx_next = 0
        self.multirx_time0 = 0.0
        self.bscope_time0 = 0.0
        self.rx_mode = 3
        self.rx_tune = self.tx_tune = 0
        self.vfo_screen = self.vfo_audio = 0
        self.scan = None               # change_scan's arguments while a scan is on
        self.split_rxtx = 0
        self.rit_freq = 0
        self.nrx = 0
        self.play_method = 0
        self.play_channel = -1
        self.f0 = None                 # filter slot 0 as last set: (fi, fq, bw, start_offset)
        self.overrange = 0
        self.read_error = 0
        self.last_counts = None


def _st():
    if _S is None or _S.eng is None:
        raise error("record_app has not been called")
    return _S


def _refresh(conf):
    return float(getattr(conf, "graph_refresh", 7) or 7)


def record_app(app, conf, data_width, graph_width, fft_size, multirx_data_width, rate, handle=0):
    """quisk.c:5048-5113 ("OOiiiiik").  conf.playback_rate (default 48000) and conf.agc_release_time are read like
    the reference reads them (quisk.c:3833); conf.quisk_amd_receivers overrides the receiver count."""
    global _S
    close_sound()
    s = _State()
    s.app, s.conf = app, conf
    s.data_width, s.graph_width, s.fft_size = int(data_width), int(graph_width), int(fft_size)
    s.multirx_data_width, s.rate = int(multirx_data_width), int(rate)
    nrx = int(getattr(conf, "quisk_amd_receivers", 1 + QUISK_MAX_RECEIVERS))
    try:
        s.nrx = nrx                                     # the receivers the caller sees; one more is kept for the split RX/TX partner
        s.eng = _eng.Engine(s.rate, n_receivers=nrx + 1, playback_rate=int(getattr(conf, "playback_rate", 48000)),
                            max_block=SAMP_BUFFER_SIZE, fft_size=s.fft_size, data_width=s.data_width,
                            device=int(getattr(conf, "quisk_amd_device", 0)),
                            max_filter_taps=int(getattr(conf, "quisk_amd_max_filter_taps", 0)),
                            short_blocks=bool(int(getattr(conf, "quisk_amd_short_blocks", 1 if s.rate <= 384000 else 0))))
    except Exception as exc:                          # no GPU, bad sizes ...: the module's own exception type
        raise error(str(exc))
    rel = getattr(conf, "agc_release_time", None)
    if rel is not None:
        s.eng.set_agc_release_time(float(rel))
    s.graph_time0 = time.time()
    _S = s


def close_sound():
    """quisk.c:3914-3930: stop and free; here it releases the engine."""
    global _S
    if _S is not None and _S.eng is not None:
        _S.eng.close()
    _S = None


def set_params(**kw):
    """quisk.c:2920-2958: all keywords optional."""
    s = _st()
    known = ("quisk_is_vna", "rx_bytes", "rx_endian", "read_error", "clip", "bscope_bytes", "bscope_endian", "bscope_size",
             "bandscopeScale", "hermes_pause")
    for k in kw:
        if k not in known:
            raise TypeError("'%s' is an invalid keyword argument for this function" % k)
    if "rx_bytes" in kw:
        s.rx_bytes = int(kw["rx_bytes"])
    if "rx_endian" in kw:
        s.rx_endian = int(kw["rx_endian"])
    if "bscope_bytes" in kw:
        s.bscope_bytes = int(kw["bscope_bytes"])
    if "bscope_endian" in kw:
        s.bscope_endian = int(kw["bscope_endian"])
    if kw.get("read_error", 0):
        s.read_error += 1                             # quisk.c:2938-2939
    if kw.get("clip", 0):
        s.overrange += 1                              # quisk.c:2940-2941
    size = int(kw.get("bscope_size", 0) or 0)
    if size > 0:
        if s.bscope_size == 0:                        # quisk.c:2944-2947
            s.bscope_size = size
            s.eng.bandscope_init(size)
        elif size != s.bscope_size:
            print("Illegal attempt to change bscope_size")   # quisk.c:2948-2950


def add_rx_samples(samples):
    """quisk.c:2777-2839: interleaved I, Q signed integers of rx_bytes bytes.  Soft failures print and return None."""
    s = _st()
    try:
        view = memoryview(samples).cast("B")
    except TypeError:
        print("add_rx_samples: Invalid object sent as samples")
        return None
    unit = 2 * s.rx_bytes
    if len(view) % unit:                               # quisk.c:2798-2800: the whole buffer is dropped
        print("add_rx_samples: Odd number of bytes in sample buffer")
    elif len(s.pending) // unit + len(view) // unit > MAX_PENDING:      # quisk.c:2801-2803
        print("add_rx_samples: buffer is too full")
    else:
        s.pending += view
    return None


def read_sound():
    """quisk.c:3902-3912 -> quisk_read_sound -> quisk_process_samples: returns the number of samples read."""
    s = _st()
    unit = 2 * s.rx_bytes
    n = len(s.pending) // unit
    if n == 0:
        return 0
    raw = bytes(s.pending[:n * unit])
    del s.pending[:n * unit]
    s.eng.process_int(raw, s.rx_bytes, big_endian=bool(s.rx_endian))
    return n


def get_audio(rx=0):
    """Not in the reference (its audio goes to the sound card): the last block's stereo audio of one receiver."""
    return _st().eng.audio(rx)


def set_tune(rx_hz, tx_hz=0):
    """quisk.c:4323-4329 ("ii"): receive tuning offset from the centre of the stream; the transmit side is off the path."""
    s = _st()
    s.rx_tune, s.tx_tune = int(rx_hz), int(tx_hz)
    _apply_tune_and_scan(s)
    if s.split_rxtx:
        _update_route(s)


def set_rx_mode(mode):
    """quisk.c:4262-4269."""
    s = _st()
    s.rx_mode = int(mode)
    try:
        s.eng.set_mode(0, int(mode))
        if s.split_rxtx:
            _update_route(s)
    except Exception as exc:
        raise error(str(exc))


def set_filters(filterI, filterQ, bw, start_offset, nFilter):
    """quisk.c:4192-4235 ("OOiii"), with its messages."""
    s = _st()
    if not hasattr(filterI, "__len__") or not hasattr(filterI, "__getitem__"):
        raise error("Filter I is not a sequence")
    if not hasattr(filterQ, "__len__") or not hasattr(filterQ, "__getitem__"):
        raise error("Filter Q is not a sequence")
    size = len(filterI)
    if size != len(filterQ):
        raise error("The size of filters I and Q must be equal")
    if size >= MAX_FILTER_SIZE:
        raise error("Filter size must be less than %d" % MAX_FILTER_SIZE)
    fi = np.array([float(v) for v in filterI], dtype=np.float64)
    fq = np.array([float(v) for v in filterQ], dtype=np.float64)
    nFilter = int(nFilter)
    if nFilter == 0:
        targets = [0]
        s.f0 = (fi, fq, int(bw), int(start_offset))
    elif nFilter == 1:
        targets = list(range(1, s.nrx))
    else:
        targets = []                                   # slot 2 feeds the external digital-mode path (quisk.c:2515): off the path
    try:
        for r in targets:
            s.eng.set_filters(r, fi, fq, int(bw), int(start_offset))
        if nFilter == 0 and s.split_rxtx:
            _update_route(s)
    except Exception as exc:
        raise error(str(exc))


def set_agc(gain):
    """quisk.c:4184-4190: agcReleaseGain, one value for every receiver as in the reference."""
    s = _st()
    for r in range(s.eng.n_receivers):
        s.eng.set_agc(r, float(gain))


def set_multirx_mode(index, mode):
    """quisk.c:4426-4436: silently ignored beyond QUISK_MAX_RECEIVERS."""
    s = _st()
    index = int(index)
    if index < QUISK_MAX_RECEIVERS and 1 + index < s.nrx:
        try:
            s.eng.set_mode(1 + index, int(mode))
        except Exception as exc:
            raise error(str(exc))


def set_multirx_freq(index, freq):
    """quisk.c:4438-4448."""
    s = _st()
    index = int(index)
    if index < QUISK_MAX_RECEIVERS and 1 + index < s.nrx:
        s.eng.set_tune(1 + index, int(freq))


def get_graph(use_fft, zoom, deltaf):
    """quisk.c:4602-4759 ("idd"): tuple of data_width dB values, or None when nothing is due."""
    s = _st()
    if int(use_fft) != s.use_fft:                      # change in data return type; re-initialise (quisk.c:4618-4621)
        s.use_fft = int(use_fft)
    if not s.use_fft:
        raise error("raw (use_fft == 0) graph data are not kept on the GPU path")
    if s.eng.graph_pending() <= 0:
        return None
    if time.time() - s.graph_time0 < 1.0 / _refresh(s.conf):       # quisk.c:4712
        return None
    n, row, _ = s.eng.get_graph(float(zoom), float(deltaf))
    if n <= 0:
        return None
    s.graph_time0 = time.time()
    return tuple(float(v) for v in row)


def get_smeter():
    """quisk.c:2737-2742."""
    return _st().eng.smeter()


def add_multirx_samples(index, samples):
    """The sub-receiver frames that read_rx_udp10 collects (quisk.c:3520-3527) arrive here as complex128 arrays:
    frame `index` fills up to 8 * multirx_data_width samples and further samples are dropped until it is used."""
    s = _st()
    index = int(index)
    if not 0 <= index < QUISK_MAX_RECEIVERS:
        return None
    need = 8 * s.multirx_data_width * 16
    buf = s.multirx_frames[index]
    if len(buf) < need:
        v = np.ascontiguousarray(samples, dtype=np.complex128).tobytes()
        buf += v[:need - len(buf)]
    s.multirx_count = max(s.multirx_count, index + 1)
    return None


def get_multirx_graph():
    """quisk.c:4468-4530: (tuple, index) when a frame of the receiver whose turn it is is complete and
    1 / graph_refresh / count seconds have passed (quisk.c:3543-3547, 4485); else ((), -1)."""
    s = _st()
    if s.multirx_count <= 0 or s.multirx_data_width <= 0:
        return (), -1
    idx = s.multirx_next
    need = 8 * s.multirx_data_width * 16
    if len(s.multirx_frames[idx]) < need:
        return (), -1
    if time.time() - s.multirx_time0 < 1.0 / _refresh(s.conf) / s.multirx_count:
        return (), -1
    s.multirx_time0 = time.time()
    frame = np.frombuffer(bytes(s.multirx_frames[idx]), dtype=np.complex128)
    s.multirx_frames[idx] = bytearray()
    row = s.eng.multirx_graph(frame, s.multirx_data_width)
    s.multirx_next = (idx + 1) % s.multirx_count       # quisk.c:3538-3541
    return tuple(float(v) for v in row), idx


def add_bscope_samples(samples):
    """quisk.c:2841-2900."""
    s = _st()
    if s.bscope_size <= 0:
        print("add_bscope_samples: The bandscope was not initialized with InitBscope()")
        return None
    try:
        view = memoryview(samples).cast("B")
    except TypeError:
        print("add_bscope_samples: Invalid object sent as samples")
        return None
    if len(view) != s.bscope_size * s.bscope_bytes:
        print("add_bscope_samples: Wrong number of bytes in sample buffer")
        return None
    s.eng.add_bscope_samples_int(bytes(view), s.bscope_bytes, big_endian=bool(s.bscope_endian))
    return None


def get_bandscope(clock, zoom, deltaf):
    """quisk.c:4532-4600 ("idd"): tuple of graph_width dB values or None."""
    s = _st()
    if s.bscope_size <= 0:
        return None
    if time.time() - s.bscope_time0 < 1.0 / _refresh(s.conf):      # quisk.c:4559
        return None
    n, row, _ = s.eng.get_bandscope(int(clock), float(zoom), float(deltaf), s.graph_width)
    if n <= 0:
        return None
    s.bscope_time0 = time.time()
    return tuple(float(v) for v in row)


def get_filter_rate(mode, bandwidth):
    """quisk.c:2661-2735 ("ii")."""
    s = _st()
    return _eng.get_filter_rate(s.rate, int(mode), int(bandwidth))


def get_state():
    """quisk.c:2618-2643: the 17-tuple; counters the engine does not have are zero."""
    s = _st()
    st = s.eng.stats()
    return (s.rate, s.rate, s.rate, 2, 2, "", 0, "", s.read_error, 0, 0, 0, 0, int(st.blocks), int(st.fft_error), 0, 0)


def get_overrange():
    """quisk.c:2744-2752: returns the count and clears it."""
    s = _st()
    n, s.overrange = s.overrange, 0
    return n


def set_sparams(**kw):
    """sound.c:1253-1262: dc_remove_bw (and digital_output_level, which is off the path)."""
    if "dc_remove_bw" in kw:
        _st().eng.set_dc_remove(int(kw["dc_remove_bw"]))


def invert_spectrum(on):
    """quisk.c:4174-4182."""
    _st().eng.set_invert_spectrum(int(on))


def set_squelch(level):
    """quisk.c:4348-4354: the FM squelch level."""
    _st().eng.set_squelch(float(level))


def set_kill_audio(on):
    """quisk.c:4364-4370."""
    _st().eng.set_kill_audio(int(on))


def set_noise_blanker(level):
    """quisk.c:4246-4252."""
    _st().eng.set_noise_blanker(int(level))


def _update_route(s):
    """Which receiver is heard beside receiver 0, and in which ear (quisk.c:2410-2494), as csrc/quisk_module.c does it: split
    RX/TX demodulates the stream a second time at the transmit offset + RIT with receiver 0's mode and filter (quisk.c:2411,
    2417-2418) on a receiver of the engine kept for that alone (index nrx); the multirx play channel is sub-receiver 1 + index."""
    method, partner = 0, 0
    if 1 <= s.split_rxtx <= 4:
        partner = s.nrx
        rx_left = (s.tx_tune < s.rx_tune) if s.split_rxtx == 1 else (s.tx_tune >= s.rx_tune)       # quisk.c:2424, 2438
        method = 4 if s.split_rxtx == 3 else 3 if s.split_rxtx == 4 else (1 if rx_left else 2)
        s.eng.set_mode(partner, s.rx_mode)
        s.eng.set_tune(partner, s.tx_tune + s.rit_freq)
        if s.f0 is not None:
            s.eng.set_filters(partner, *s.f0)
    elif s.play_channel >= 0 and 1 + s.play_channel < s.nrx:
        partner = 1 + s.play_channel
        method = 1 if s.play_method == 1 else 2 if s.play_method == 2 else 3                        # quisk.c:2474-2493
    s.eng.set_stereo_route(0, partner, method)


def set_split_rxtx(split):
    """quisk.c:4372-4379."""
    s = _st()
    s.split_rxtx = int(split)
    try:
        _update_route(s)
    except Exception as exc:
        raise error(str(exc))


def set_multirx_play_method(method):
    """quisk.c:4450-4456."""
    s = _st()
    s.play_method = int(method)
    try:
        _update_route(s)
    except Exception as exc:
        raise error(str(exc))


def set_multirx_play_channel(channel):
    """quisk.c:4458-4466: -1 = none."""
    s = _st()
    s.play_channel = int(channel)
    if s.play_channel >= QUISK_MAX_RECEIVERS:
        s.play_channel = -1
    try:
        _update_route(s)
    except Exception as exc:
        raise error(str(exc))


def set_ssb_squelch(enabled, level):
    """quisk.c:4356-4362."""
    _st().eng.set_ssb_squelch(int(enabled), int(level))


def set_auto_notch(on):
    """quisk.c:4237-4244: the reference's notch works on bank 0."""
    _st().eng.set_auto_notch(0, int(on))


def set_sidetone(ctrl, volume, rit, keyup):
    """quisk.c:4331-4346 ("idii"): of the four values the receive path uses rit_freq."""
    if _S is not None and _S.eng is not None:
        _S.eng.set_rit_freq(int(rit))
        _S.rit_freq = int(rit)
        if _S.split_rxtx:
            _update_route(_S)                           # the second channel is tuned to tx + rit (quisk.c:2411)


def _apply_tune_and_scan(s):
    """The mixer's offset is rx_tune_freq + vfo_screen - vfo_audio (quisk.c:2354); the scan's slots depend on the transmit
    tune and vfo_screen (quisk.c:4661)."""
    s.eng.set_tune(0, s.rx_tune + s.vfo_screen - s.vfo_audio)
    if s.scan is not None and s.scan[0] > 0 and s.fft_size > 0:
        s.eng.change_scan(*s.scan, s.tx_tune, s.vfo_screen)


def change_rates(rate, vfo_audio, fft_rate, vfo_screen):
    """quisk.c:3874-3886 ("iiii").  The engine's rate is fixed by record_app: another one is reported and ignored."""
    s = _st()
    if int(rate) != s.rate or int(fft_rate) != int(rate):
        print("change_rates: the sample rate is fixed by record_app (%d); %d / %d ignored" % (s.rate, int(rate), int(fft_rate)))
    s.vfo_audio, s.vfo_screen = int(vfo_audio), int(vfo_screen)
    _apply_tune_and_scan(s)


def change_scan(scan_blocks, scan_sample_rate, scan_valid, scan_vfo0, scan_deltaf):
    """quisk.c:3865-3872 ("iidii")."""
    s = _st()
    s.scan = (int(scan_blocks), int(scan_sample_rate), float(scan_valid), int(scan_vfo0), int(scan_deltaf))
    if s.fft_size > 0:
        s.eng.change_scan(*s.scan, s.tx_tune, s.vfo_screen)


def add_tone(freq):
    """quisk.c:3022-3034."""
    _st().eng.add_tone(int(freq))


def get_filter():
    """quisk.c:4761-4848: a tuple of data_width dB values."""
    s = _st()
    return tuple(float(v) for v in s.eng.get_filter(s.data_width))


def get_tx_filter():
    """microphone.c:719-807: the transmit chain is off the path; the reference's shape at its floor."""
    return tuple(-140.0 for _ in range(_st().data_width))


def get_audio_graph():
    """quisk.c:966-991: None until a graph is ready."""
    g = _st().eng.get_audio_graph()
    return None if g is None else tuple(float(v) for v in g)


def measure_frequency(mode):
    """quisk.c:3000-3009."""
    return _st().eng.measure_frequency(int(mode))


def measure_audio(time):
    """quisk.c:3011-3020."""
    return _st().eng.measure_audio(int(time))


def _xdft(seq, inverse, window):
    """Xdft, quisk.c:4922-5014: order of frequency with zero at index (size - 1) / 2, Blackman window, 1 / N on the inverse."""
    try:
        size = len(seq)
    except TypeError:
        raise error("DFT input data is not a sequence")
    if size <= 0:
        return ()
    j0 = (size - 1) // 2
    try:
        x = np.array([complex(seq[(j0 + i) % size]) for i in range(size)], dtype=np.complex128)
    except (TypeError, ValueError):
        raise error("DFT input data is not a complex/float/int number")
    w = None
    if window:
        w = np.zeros(size + 1)
        for i in range(size // 2 + 1):
            w[i] = w[size - i] = 0.42 + 0.50 * np.cos(2. * np.pi * i / size) + 0.08 * np.cos(4. * np.pi * i / size)
        w = w[:size]
    if not inverse and w is not None:
        x = x * w
    y = _st().eng.fft_any(x, inverse)
    if inverse:
        y = y * (w / size) if w is not None else y / size
    out = [0j] * size
    for i in range(size):
        out[(j0 + i) % size] = complex(y[i])
    return out


def dft(seq, window=0):
    return _xdft(seq, False, window)


def idft(seq, window=0):
    return _xdft(seq, True, window)


def sound_devices():
    return ([], [])


pa_sound_devices = sound_devices


def sound_errors():
    return []


def ip_interfaces():
    return []


def get_hermes_adc():
    return 0.0


def get_hermes_TFRC():
    return (0.0, 0.0, 0.0, 0.0)


def get_hermeslite_writepointer():
    return 0


def get_hermeslite_response():
    return bytearray(5)


def hermes_to_pc():
    return bytearray(20)


def freedv_get_snr():
    return 0.0


def freedv_get_version():
    return -1


def freedv_get_rx_char():
    return ""


def _inert(*args, **kw):
    return None


def _zero(*args, **kw):
    return 0


is_key_down = get_hardware_ptt = is_vox = _zero       # receive only: the key is up


def get_squelch(freq):
    """get_squelch (quisk.c:2645-2652) = IsSquelch on the last released graph row."""
    return _st().eng.is_squelch(int(freq))


# Entries of QuiskMethods outside the receive path: accepted and ignored.
INERT = ("set_volume", "set_enable_bandscope",
         "test_1", "test_2", "test_3", "set_spot_level", "set_imd_level", "set_key_down", "set_ctcss", "set_tx_audio", "set_record_state",
         "set_file_name", "set_transmit_mode", "set_fdx", "set_sample_bytes",
         "change_rate", "start_sound", "open_sound", "open_key", "close_key",
         "open_rx_udp", "close_rx_udp", "set_hermes_id", "set_hermes_filters", "set_alex_hpf", "set_alex_lpf",
         "set_hermeslite_writepointer", "clear_hermeslite_response", "set_hermeslite_writequeue", "pc_to_hermes",
         "pc_to_hermeslite_writequeue", "freedv_open", "freedv_close", "freedv_set_options", "set_freedv_mode",
         "record_graph", "ImmediateChange", "set_ampl_phase", "set_udp_tx_correct", "set_mic_out_volume", "tx_hold_state",
         "open_wav_file_play", "capt_channels", "play_channels", "micplay_channels", "mixer_set", "set_PTT")
for _name in INERT:
    globals()[_name] = _inert
del _name
