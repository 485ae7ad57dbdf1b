/* quisk_module.c -- the `_quisk` CPython extension for the receive path, over libqkengine.so.
 *
 * quisk.py and every quisk_hardware_*.py reach the receive DSP through `import _quisk as QS` (quisk.py:33,
 * quisk_hardware_model.py:15); C sample-source plug-ins (sdriqpkg/sdriq.c:826, soapypkg/soapy.c:199, perseuspkg/perseus.c:286,
 * afedrinet/afedrinet_io.c:343) bind through the capsule `_quisk.QUISK_C_API` with import_quisk_api.c.  This module
 * carries that interface: the method names and PyArg formats of QuiskMethods (quisk.c:5164-5266) for the path, the
 * exception `_quisk.error` (quisk.c:5311-5317), the eleven-entry capsule (quisk.h:326-355), and the two C symbols other
 * objects link against, quisk_process_samples (quisk.h:258) and cRxFilterOut (quisk.h:215).  Host logic only: every
 * sample goes through the C ABI of include/qk_engine.h into the HIP kernels; there is no CPU DSP in here.
 *
 * Entries of QuiskMethods outside the receive path (sound cards, transmit, Hermes control, FreeDV ...) are accepted
 * and ignored, so that an unchanged quisk.py can call them.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "../../include/qk_engine.h"

#define QK_MAX_SUBRX 9                 /* QUISK_MAX_RECEIVERS, quisk.h:21 */
#define QK_SAMP_BUFFER 66000           /* SAMP_BUFFER_SIZE, quisk.h:15 */
#define QK_PENDING_MAX (QK_SAMP_BUFFER * 8 / 10)    /* add_rx_samples keeps at most this many samples, quisk.c:2801 */
#define QK_MAX_FILTER 10001            /* MAX_FILTER_SIZE, quisk.h:10 */

/* The sample-source interface of C plug-ins (quisk.h:250-253). */
typedef void (*ty_sample_start)(void);
typedef void (*ty_sample_stop)(void);
typedef int (*ty_sample_read)(double _Complex *);
typedef int (*ty_sample_write)(double _Complex *, int);
typedef int (*ty_dvoice_codec_rx)(double _Complex *, double *, int, int);
typedef int (*ty_dvoice_codec_tx)(double _Complex *, double *, int);

/* Capsule entry 0 is a pointer to this record; plug-ins read and count into it through pt_quisk_sound_state
 * (import_quisk_api.c:37), so the member order and types are those of struct sound_conf, quisk.h:124-157. */
struct sound_conf {
    char dev_capt_name[128], dev_play_name[128];
    int sample_rate, playback_rate, data_poll_usec, latency_millisecs;
    unsigned int rate_min, rate_max, chan_min, chan_max;
    int read_error, write_error, underrun_error, overrange, latencyCapt, latencyPlay, interupts;
    char msg1[128], err_msg[128];
    char mic_dev_name[128], name_of_mic_play[128], mic_ip[32];
    int mic_sample_rate, mic_playback_rate, tx_audio_port, mic_read_error, mic_channel_I, mic_channel_Q;
    double mic_out_volume;
    char IQ_server[32];
    int verbose_pulse;
};
struct sound_conf quisk_sound_state;

static PyObject *QuiskError;

static struct {
    qk_engine *qk;
    PyObject *app, *conf;
    int data_width, graph_width, fft_size, multirx_data_width, nrx;
    int rx_bytes, rx_endian, bs_bytes, bs_endian, bs_size, is_vna;
    double bandscope_scale;
    unsigned char *pend; size_t pend_bytes;         /* integer samples from add_rx_samples, waiting for read_sound */
    double _Complex *cbuf;                           /* one block for a C sample source */
    double *row;                                     /* graph row / raw row scratch */
    int use_fft;
    double graph_t0, bscope_t0, mrx_t0, graph_refresh;
    ty_sample_start src_start; ty_sample_stop src_stop; ty_sample_read src_read; ty_sample_write src_write;
    ty_dvoice_codec_rx dv_rx; ty_dvoice_codec_tx dv_tx;
    double *mrx_frame[QK_MAX_SUBRX]; int mrx_fill[QK_MAX_SUBRX]; int mrx_count, mrx_next;
    int rx_mode;
    /* second-receiver routing (quisk.c:2410-2494): split_rxtx 0..4, multirx_play_method 0..2, multirx_play_channel or -1 */
    int split_rxtx, play_method, play_channel, rx_tune, tx_tune, rit_freq, partner_dirty, multirx_state;
    double adc_level;
    int vfo_screen, vfo_audio;                      /* change_rates, quisk.c:3874-3886 */
    int scan_blocks, scan_rate, scan_vfo0, scan_deltaf; double scan_valid;      /* change_scan, quisk.c:3865-3872 */
    double *f0; int f0_n, f0_bw, f0_off;            /* the taps of filter slot 0, for the split partner */
} G = { .rx_bytes = 3, .rx_endian = 1, .bs_bytes = 2, .bs_endian = 1, .use_fft = 1, .graph_refresh = 7, .rx_mode = 3, .play_channel = -1 };

/* ---------------------------------------------------------------- capsule functions (quisk.h:340-349) */
double QuiskTimeSec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return (double)ts.tv_sec + ts.tv_nsec * 1e-9;
}
void QuiskSleepMicrosec(int usec) { if (usec > 0) usleep((useconds_t)usec); }
void QuiskPrintTime(const char *str, int index)
{   /* elapsed time since the last call with the same index and since the first call; a NULL string restarts the clocks */
    static double first, last[10];
    const double now = QuiskTimeSec();
    if (index < 0 || index > 9) index = 0;
    if (!str) { first = now; for (int i = 0; i < 10; i++) last[i] = now; return; }
    if (first == 0) { first = now; for (int i = 0; i < 10; i++) last[i] = now; }
    if (str[0]) printf("%12.6lf  %9.3lf  %9.3lf  %s\n", now - first, (now - first) * 1e3, (now - last[index]) * 1e3, str);
    last[index] = now;
}
/* configuration values come from the `conf` object given to record_app; called with the GIL held (GUI thread only,
 * as the reference's, utility.c:11-13) */
int QuiskGetConfigInt(const char *name, int deflt)
{
    if (!G.conf) return deflt;
    PyObject *v = PyObject_GetAttrString(G.conf, name);
    if (!v) { PyErr_Clear(); return deflt; }
    long r = PyLong_Check(v) ? PyLong_AsLong(v) : (PyFloat_Check(v) ? (long)PyFloat_AsDouble(v) : deflt);
    if (PyErr_Occurred()) { PyErr_Clear(); r = deflt; }
    Py_DECREF(v);
    return (int)r;
}
double QuiskGetConfigDouble(const char *name, double deflt)
{
    if (!G.conf) return deflt;
    PyObject *v = PyObject_GetAttrString(G.conf, name);
    if (!v) { PyErr_Clear(); return deflt; }
    double r = PyFloat_Check(v) ? PyFloat_AsDouble(v) : (PyLong_Check(v) ? (double)PyLong_AsLong(v) : deflt);
    if (PyErr_Occurred()) { PyErr_Clear(); r = deflt; }
    Py_DECREF(v);
    return r;
}
char *QuiskGetConfigString(const char *name, char *deflt)
{   /* the returned text lives until the next call, like the reference's static buffer */
    static char text[256];
    if (!G.conf) return deflt;
    PyObject *v = PyObject_GetAttrString(G.conf, name);
    if (!v) { PyErr_Clear(); return deflt; }
    char *r = deflt;
    if (PyUnicode_Check(v)) {
        const char *u = PyUnicode_AsUTF8(v);
        if (u) { strncpy(text, u, sizeof text - 1); text[sizeof text - 1] = 0; r = text; } else PyErr_Clear();
    }
    Py_DECREF(v);
    return r;
}
void quisk_sample_source4(ty_sample_start start, ty_sample_stop stop, ty_sample_read rd, ty_sample_write wr)
{   /* sound.c:402-415: the plug-in's callbacks replace the sound card as the source of I/Q samples */
    G.src_start = start; G.src_stop = stop; G.src_read = rd; G.src_write = wr;
}
void quisk_sample_source(ty_sample_start start, ty_sample_stop stop, ty_sample_read rd) { quisk_sample_source4(start, stop, rd, NULL); }
void quisk_dvoice_freedv(ty_dvoice_codec_rx rx, ty_dvoice_codec_tx tx) { G.dv_rx = rx; G.dv_tx = tx; }   /* codec is off the path */
int quisk_is_key_down(void) { return 0; }        /* receive only */

/* ---------------------------------------------------------------- exported C entry points of the path */
/* quisk_process_samples (quisk.h:258; quisk.c:2149): n input samples at the stream rate are replaced, in place, by the
 * stereo audio of receiver 0 at the play rate; returns the audio count.  The buffer holds SAMP_BUFFER_SIZE samples. */
int quisk_process_samples(double _Complex *cSamples, int nSamples)
{
    if (!G.qk || !cSamples || nSamples <= 0) return 0;
    if (qk_process_samples(G.qk, (const double *)cSamples, nSamples) != QK_OK) return 0;   /* no error returns at this level in the reference */
    const int m = qk_pull_audio(G.qk, 0, (double *)cSamples, QK_SAMP_BUFFER);
    return m > 0 ? m : 0;
}
/* cRxFilterOut (quisk.h:215; quisk.c:1154): one sample through the receive band-pass of filter slot nFilter with the
 * history of `bank`.  One sample is one small GPU launch: kept for link compatibility (freedv.c:233); code that has a
 * block of samples calls qk_rx_filter_block. */
double _Complex cRxFilterOut(double _Complex sample, int bank, int nFilter)
{
    double in[2] = { creal(sample), cimag(sample) }, out[2] = { 0, 0 };
    const int rx = (nFilter == 1 && G.nrx > 1) ? 1 : 0;
    if (!G.qk || qk_rx_filter_block(G.qk, rx, bank, in, out, 1) != 1) return sample;
    return out[0] + out[1] * I;
}

/* ---------------------------------------------------------------- helpers */
static int need_engine(void)
{
    if (G.qk) return 1;
    PyErr_SetString(QuiskError, "record_app has not been called");
    return 0;
}
static PyObject *engine_error(void) { PyErr_SetString(QuiskError, qk_last_error()); return NULL; }
static void release_engine(void)
{
    if (G.src_stop && G.qk) G.src_stop();
    if (G.qk) { qk_destroy(G.qk); G.qk = NULL; }
    for (int i = 0; i < QK_MAX_SUBRX; i++) { free(G.mrx_frame[i]); G.mrx_frame[i] = NULL; G.mrx_fill[i] = 0; }
    free(G.pend); G.pend = NULL; G.pend_bytes = 0;
    free(G.cbuf); G.cbuf = NULL;
    free(G.row); G.row = NULL;
    G.mrx_count = G.mrx_next = 0; G.bs_size = 0;
    free(G.f0); G.f0 = NULL; G.f0_n = 0;
    G.split_rxtx = 0; G.play_method = 0; G.play_channel = -1;
    Py_CLEAR(G.app); Py_CLEAR(G.conf);
}

/* Which receiver is heard beside receiver 0, and in which ear (quisk.c:2410-2494).  Split RX/TX demodulates the same stream a
 * second time at the transmit offset + RIT with receiver 0's mode and filter slot 0 (quisk.c:2411, 2417-2418): the engine is
 * made with one receiver more than the sub-receivers need (index G.nrx), kept for that alone.  The multirx play channel is
 * sub-receiver 1 + index.  The partner's mode and filter are sent again only when they changed (a filter upload drains the
 * engine's pipeline); its tune follows every set_tune. */
static int update_route(void)
{
    if (!G.qk) return QK_OK;
    int method = 0, partner = 0;
    if (G.split_rxtx >= 1 && G.split_rxtx <= 4) {
        partner = G.nrx;
        const int rx_left = G.split_rxtx == 1 ? G.tx_tune < G.rx_tune : G.tx_tune >= G.rx_tune;     /* quisk.c:2424, 2438 */
        method = G.split_rxtx == 3 ? 4 : G.split_rxtx == 4 ? 3 : (rx_left ? 1 : 2);
        if (G.partner_dirty) {
            if (qk_set_mode(G.qk, partner, G.rx_mode) != QK_OK) return QK_EINVAL;
            if (G.f0 && qk_set_filters(G.qk, partner, G.f0, G.f0 + G.f0_n, G.f0_n, G.f0_bw, G.f0_off) != QK_OK) return QK_EINVAL;
            G.partner_dirty = 0;
        }
        if (qk_set_tune(G.qk, partner, G.tx_tune + G.rit_freq) != QK_OK) return QK_EINVAL;
    } else if (G.play_channel >= 0 && 1 + G.play_channel < G.nrx) {
        partner = 1 + G.play_channel;
        method = G.play_method == 1 ? 1 : G.play_method == 2 ? 2 : 3;                                  /* quisk.c:2474-2493 */
    }
    return qk_set_stereo_route(G.qk, 0, partner, method);
}

/* ---------------------------------------------------------------- methods of the path */
static PyObject *record_app(PyObject *self, PyObject *args)
{   /* quisk.c:5048-5113 */
    PyObject *app, *conf;
    int dw, gw, fft, mdw, rate;
    unsigned long handle;
    if (!PyArg_ParseTuple(args, "OOiiiiik", &app, &conf, &dw, &gw, &fft, &mdw, &rate, &handle)) return NULL;
    release_engine();
    Py_INCREF(app); Py_INCREF(conf);
    G.app = app; G.conf = conf;
    G.data_width = dw; G.graph_width = gw; G.fft_size = fft; G.multirx_data_width = mdw;
    G.graph_refresh = QuiskGetConfigDouble("graph_refresh", 7.0);
    if (!(G.graph_refresh > 0)) G.graph_refresh = 7.0;
    qk_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = QuiskGetConfigInt("quisk_amd_device", 0);
    cfg.sample_rate = rate;
    cfg.playback_rate = QuiskGetConfigInt("playback_rate", 48000);
    G.nrx = QuiskGetConfigInt("quisk_amd_receivers", 1 + QK_MAX_SUBRX);
    cfg.n_receivers = G.nrx + 1;                    /* + the split RX/TX partner (update_route) */
    G.partner_dirty = 1;
    cfg.max_block = QK_SAMP_BUFFER;
    cfg.fft_size = fft; cfg.data_width = dw;
    cfg.reserved[0] = QuiskGetConfigInt("quisk_amd_max_filter_taps", 0);
    cfg.reserved[1] = QuiskGetConfigInt("quisk_amd_fp32", 0);
    /* polls of 5 ms (data_poll_usec) are a few thousand samples up to a few hundred ksps: the engine's short-block mode */
    cfg.reserved[2] = QuiskGetConfigInt("quisk_amd_short_blocks", rate <= 384000 ? 1 : 2);
    if (qk_create(&G.qk, &cfg) != QK_OK) { G.qk = NULL; Py_CLEAR(G.app); Py_CLEAR(G.conf); return engine_error(); }
    const double rel = QuiskGetConfigDouble("agc_release_time", 1.0);          /* quisk.c:3833 */
    if (rel > 0) qk_set_agc_release_time(G.qk, rel);
    memset(&quisk_sound_state, 0, sizeof quisk_sound_state);
    quisk_sound_state.sample_rate = rate;
    quisk_sound_state.playback_rate = cfg.playback_rate;
    quisk_sound_state.data_poll_usec = QuiskGetConfigInt("data_poll_usec", 5000);
    quisk_sound_state.rate_min = quisk_sound_state.rate_max = (unsigned)rate;
    quisk_sound_state.chan_min = quisk_sound_state.chan_max = 2;
    G.pend = malloc((size_t)QK_PENDING_MAX * 8);
    G.cbuf = malloc(sizeof(double _Complex) * QK_SAMP_BUFFER);
    G.row = malloc(sizeof(double) * 2 * (size_t)(dw > gw ? dw : gw) + 64);
    if (!G.pend || !G.cbuf || !G.row) { release_engine(); return PyErr_NoMemory(); }
    G.pend_bytes = 0; G.use_fft = 1;
    G.graph_t0 = G.bscope_t0 = G.mrx_t0 = QuiskTimeSec();
    Py_RETURN_NONE;
}

static PyObject *close_sound(PyObject *self, PyObject *args)
{   /* quisk.c:3914-3930 */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    release_engine();
    Py_RETURN_NONE;
}

static PyObject *start_sound(PyObject *self, PyObject *args)
{
    if (!PyArg_ParseTuple(args, "")) return NULL;
    if (G.src_start) G.src_start();          /* quisk_start_sound, sound.c:1195 on */
    Py_RETURN_NONE;
}

static PyObject *set_params(PyObject *self, PyObject *args, PyObject *kw)
{   /* quisk.c:2924-2966: keywords only, all optional */
    static char *names[] = { "quisk_is_vna", "rx_bytes", "rx_endian", "read_error", "clip", "bscope_bytes", "bscope_endian",
                             "bscope_size", "bandscopeScale", "hermes_pause", NULL };
    int nbytes = -1, read_error = -1, clip = -1, bscope_size = -1, hermes_pause = -1;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "|iiiiiiiidi", names, &G.is_vna, &nbytes, &G.rx_endian, &read_error, &clip,
                                     &G.bs_bytes, &G.bs_endian, &bscope_size, &G.bandscope_scale, &hermes_pause))
        return NULL;
    if (nbytes != -1) {
        G.rx_bytes = nbytes;
        quisk_sample_source4(NULL, NULL, NULL, NULL);      /* Python is the sample source: add_rx_samples feeds read_sound */
    }
    if (read_error != -1) quisk_sound_state.read_error++;
    if (clip != -1) quisk_sound_state.overrange++;
    if (bscope_size > 0) {
        if (G.bs_size == 0) {
            if (!need_engine()) return NULL;
            if (qk_bandscope_init(G.qk, bscope_size) != QK_OK) return engine_error();
            G.bs_size = bscope_size;
        } else if (bscope_size != G.bs_size) {
            printf("Illegal attempt to change bscope_size\n");
        }
    }
    if (hermes_pause != -1) {       /* quisk.c:2952-2962: returns the previous quisk_multirx_state; 20 and above = paused */
        const int prev = G.multirx_state;
        if (hermes_pause) { if (G.multirx_state < 20) G.multirx_state = 20; }
        else if (G.multirx_state >= 20) G.multirx_state = 0;
        return PyLong_FromLong(prev);
    }
    Py_RETURN_NONE;
}

static PyObject *add_rx_samples(PyObject *self, PyObject *args)
{   /* quisk.c:2777-2839: soft failures print and return None; a refused buffer is dropped whole */
    PyObject *samples;
    Py_buffer view;
    if (!PyArg_ParseTuple(args, "O", &samples)) return NULL;
    if (!need_engine()) return NULL;
    if (!PyObject_CheckBuffer(samples)) { printf("add_rx_samples: Invalid object sent as samples\n"); Py_RETURN_NONE; }
    if (PyObject_GetBuffer(samples, &view, PyBUF_SIMPLE) != 0) { PyErr_Clear(); printf("add_rx_samples: Can not view sample buffer\n"); Py_RETURN_NONE; }
    const Py_ssize_t unit = 2 * (Py_ssize_t)G.rx_bytes;
    if (G.rx_bytes < 1 || G.rx_bytes > 4 || view.len % unit != 0)
        printf("add_rx_samples: Odd number of bytes in sample buffer\n");
    else if ((Py_ssize_t)(G.pend_bytes / unit) + view.len / unit > QK_PENDING_MAX)
        printf("add_rx_samples: buffer is too full\n");
    else {
        memcpy(G.pend + G.pend_bytes, view.buf, (size_t)view.len);      /* the int -> complex double unpack happens on the GPU */
        G.pend_bytes += (size_t)view.len;
    }
    PyBuffer_Release(&view);
    Py_RETURN_NONE;
}

static PyObject *read_sound(PyObject *self, PyObject *args)
{   /* quisk.c:3902-3912 -> quisk_read_sound (sound.c:640-700) -> quisk_process_samples; the GIL is released around it */
    int n = 0, rc = QK_OK;
    if (!PyArg_ParseTuple(args, "")) return NULL;
    if (!need_engine()) return NULL;
    Py_BEGIN_ALLOW_THREADS
    if (G.src_read) {                                       /* a C plug-in delivers complex doubles (quisk.h:252) */
        n = G.src_read(G.cbuf);
        if (n > QK_SAMP_BUFFER) n = QK_SAMP_BUFFER;
        if (n > 0) rc = qk_process_samples(G.qk, (const double *)G.cbuf, n);
    } else if (G.pend_bytes > 0) {
        n = (int)(G.pend_bytes / (2 * (size_t)G.rx_bytes));
        rc = qk_process_samples_int(G.qk, G.pend, G.pend_bytes, G.rx_bytes, G.rx_endian != 0);
        G.pend_bytes = 0;
    }
    Py_END_ALLOW_THREADS
    if (rc != QK_OK) return engine_error();
    return PyLong_FromLong(n);
}

static PyObject *get_audio(PyObject *self, PyObject *args)
{   /* not in the reference (its audio goes to the sound card): the last block's stereo audio of one receiver, as the bytes
     * of interleaved (left, right) doubles = numpy complex128 */
    int rx = 0;
    if (!PyArg_ParseTuple(args, "|i", &rx)) return NULL;
    if (!need_engine()) return NULL;
    const int n = qk_audio_count(G.qk, rx);
    if (n < 0) return engine_error();
    PyObject *b = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)n * 16);
    if (!b) return NULL;
    if (n > 0 && qk_pull_audio(G.qk, rx, (double *)PyBytes_AS_STRING(b), n) != n) { Py_DECREF(b); return engine_error(); }
    return b;
}

static int apply_tune_and_scan(void);
static PyObject *set_tune(PyObject *self, PyObject *args)
{   /* quisk.c:4323-4329: receive and transmit tuning offsets; the transmit side is off the path */
    int rx_hz, tx_hz;
    if (!PyArg_ParseTuple(args, "ii", &rx_hz, &tx_hz)) return NULL;
    if (!need_engine()) return NULL;
    G.rx_tune = rx_hz; G.tx_tune = tx_hz;
    if (apply_tune_and_scan() != QK_OK) return engine_error();
    if (G.split_rxtx && update_route() != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *set_rx_mode(PyObject *self, PyObject *args)
{   /* quisk.c:4262-4269 */
    int mode;
    if (!PyArg_ParseTuple(args, "i", &mode)) return NULL;
    if (!need_engine()) return NULL;
    if (qk_set_mode(G.qk, 0, mode) != QK_OK) return engine_error();
    G.rx_mode = mode;
    G.partner_dirty = 1;
    if (G.split_rxtx && update_route() != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
/* The mixer's offset is rx_tune_freq + vfo_screen - vfo_audio (quisk.c:2354); the scan's slots depend on the transmit tune
 * and vfo_screen (quisk.c:4661): both are applied again whenever one of their inputs changes. */
static int apply_tune_and_scan(void)
{
    if (!G.qk) return QK_OK;
    if (qk_set_tune(G.qk, 0, G.rx_tune + G.vfo_screen - G.vfo_audio) != QK_OK) return QK_EINVAL;
    if (G.scan_blocks > 0 && G.fft_size > 0 &&
        qk_change_scan(G.qk, G.scan_blocks, G.scan_rate, G.scan_valid, G.scan_vfo0, G.scan_deltaf, G.tx_tune, G.vfo_screen) != QK_OK)
        return QK_EINVAL;
    return QK_OK;
}

static PyObject *change_rates(PyObject *self, PyObject *args)
{   /* quisk.c:3874-3886 ("iiii"): sample_rate, vfo_audio, fft_sample_rate, vfo_screen.  The engine's rate is fixed by
     * record_app: another one is reported and ignored; the two VFOs move the mixer and the scan. */
    int rate, vfo_audio, fft_rate, vfo_screen;
    if (!PyArg_ParseTuple(args, "iiii", &rate, &vfo_audio, &fft_rate, &vfo_screen)) return NULL;
    if (!need_engine()) return NULL;
    if (rate != (int)quisk_sound_state.sample_rate || fft_rate != rate)
        printf("change_rates: the sample rate is fixed by record_app (%d); %d / %d ignored\n", (int)quisk_sound_state.sample_rate, rate, fft_rate);
    G.vfo_audio = vfo_audio; G.vfo_screen = vfo_screen;
    if (apply_tune_and_scan() != QK_OK) return engine_error();
    Py_RETURN_NONE;
}

static PyObject *change_scan(PyObject *self, PyObject *args)
{   /* quisk.c:3865-3872 ("iidii") */
    if (!PyArg_ParseTuple(args, "iidii", &G.scan_blocks, &G.scan_rate, &G.scan_valid, &G.scan_vfo0, &G.scan_deltaf)) return NULL;
    if (!need_engine()) return NULL;
    if (G.fft_size <= 0) Py_RETURN_NONE;
    if (qk_change_scan(G.qk, G.scan_blocks, G.scan_rate, G.scan_valid, G.scan_vfo0, G.scan_deltaf, G.tx_tune, G.vfo_screen) != QK_OK)
        return engine_error();
    Py_RETURN_NONE;
}

static PyObject *set_split_rxtx(PyObject *self, PyObject *args)
{   /* quisk.c:4372-4379 */
    if (!PyArg_ParseTuple(args, "i", &G.split_rxtx)) return NULL;
    G.partner_dirty = 1;
    if (!need_engine()) return NULL;
    if (update_route() != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *set_multirx_play_method(PyObject *self, PyObject *args)
{   /* quisk.c:4450-4456 */
    if (!PyArg_ParseTuple(args, "i", &G.play_method)) return NULL;
    if (!need_engine()) return NULL;
    if (update_route() != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *set_multirx_play_channel(PyObject *self, PyObject *args)
{   /* quisk.c:4458-4466: -1 = none */
    if (!PyArg_ParseTuple(args, "i", &G.play_channel)) return NULL;
    if (G.play_channel >= QK_MAX_SUBRX) G.play_channel = -1;
    if (!need_engine()) return NULL;
    if (update_route() != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *key_up(PyObject *self, PyObject *args)
{   /* an addition: the transmit side tells the receive path that the key went up (quisk.c:2240-2241) */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    if (!need_engine()) return NULL;
    if (qk_key_up(G.qk) != QK_OK) return engine_error();
    Py_RETURN_NONE;
}

static PyObject *set_filters(PyObject *self, PyObject *args)
{   /* quisk.c:4192-4235, with its messages */
    PyObject *fI, *fQ;
    int bw, start_offset, nFilter;
    if (!PyArg_ParseTuple(args, "OOiii", &fI, &fQ, &bw, &start_offset, &nFilter)) return NULL;
    if (!need_engine()) return NULL;
    if (PySequence_Check(fI) != 1) { PyErr_SetString(QuiskError, "Filter I is not a sequence"); return NULL; }
    if (PySequence_Check(fQ) != 1) { PyErr_SetString(QuiskError, "Filter Q is not a sequence"); return NULL; }
    const Py_ssize_t size = PySequence_Size(fI);
    if (size != PySequence_Size(fQ)) { PyErr_SetString(QuiskError, "The size of filters I and Q must be equal"); return NULL; }
    if (size >= QK_MAX_FILTER) {
        char msg[64];
        snprintf(msg, sizeof msg, "Filter size must be less than %d", QK_MAX_FILTER);
        PyErr_SetString(QuiskError, msg);
        return NULL;
    }
    double *ti = malloc(sizeof(double) * 2 * (size_t)(size + 1)), *tq = ti + size + 1;
    if (!ti) return PyErr_NoMemory();
    for (Py_ssize_t i = 0; i < size; i++) {
        PyObject *a = PySequence_GetItem(fI, i), *b = PySequence_GetItem(fQ, i);
        ti[i] = a ? PyFloat_AsDouble(a) : 0; tq[i] = b ? PyFloat_AsDouble(b) : 0;
        Py_XDECREF(a); Py_XDECREF(b);
        if (PyErr_Occurred()) { free(ti); return NULL; }
    }
    /* slot 0 = receiver 0, slot 1 = the sub-receivers (quisk.c:2472); slot 2 feeds the external digital path (off the path) */
    int rc = QK_OK;
    if (nFilter == 0) {
        rc = qk_set_filters(G.qk, 0, ti, tq, (int)size, bw, start_offset);
        free(G.f0);
        G.f0 = malloc(sizeof(double) * 2 * (size_t)(size + 1));
        if (G.f0) { memcpy(G.f0, ti, sizeof(double) * (size_t)size); memcpy(G.f0 + size, tq, sizeof(double) * (size_t)size); G.f0_n = (int)size; G.f0_bw = bw; G.f0_off = start_offset; }
        G.partner_dirty = 1;
        if (rc == QK_OK && G.split_rxtx) rc = update_route();
    }
    else if (nFilter == 1)
        for (int r = 1; r < G.nrx && rc == QK_OK; r++) rc = qk_set_filters(G.qk, r, ti, tq, (int)size, bw, start_offset);
    free(ti);
    if (rc != QK_OK) return engine_error();
    Py_RETURN_NONE;
}

static PyObject *set_agc(PyObject *self, PyObject *args)
{   /* quisk.c:4184-4190: one agcReleaseGain for every bank */
    double gain;
    if (!PyArg_ParseTuple(args, "d", &gain)) return NULL;
    if (!need_engine()) return NULL;
    for (int r = 0; r <= G.nrx; r++) if (qk_set_agc(G.qk, r, gain) != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *set_multirx_mode(PyObject *self, PyObject *args)
{   /* quisk.c:4426-4436: silently ignored beyond the receivers that exist */
    int index, mode;
    if (!PyArg_ParseTuple(args, "ii", &index, &mode)) return NULL;
    if (!need_engine()) return NULL;
    if (index >= 0 && index < QK_MAX_SUBRX && 1 + index < G.nrx && qk_set_mode(G.qk, 1 + index, mode) != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *set_multirx_freq(PyObject *self, PyObject *args)
{   /* quisk.c:4438-4448 */
    int index, freq;
    if (!PyArg_ParseTuple(args, "ii", &index, &freq)) return NULL;
    if (!need_engine()) return NULL;
    if (index >= 0 && index < QK_MAX_SUBRX && 1 + index < G.nrx && qk_set_tune(G.qk, 1 + index, freq) != QK_OK) return engine_error();
    Py_RETURN_NONE;
}

static PyObject *get_graph(PyObject *self, PyObject *args)
{   /* quisk.c:4602-4759: a tuple of data_width dB values, or of data_width complex samples when use_fft == 0, or None */
    int use_fft;
    double zoom, deltaf;
    if (!PyArg_ParseTuple(args, "idd", &use_fft, &zoom, &deltaf)) return NULL;
    if (!need_engine()) return NULL;
    use_fft = use_fft != 0;
    if (use_fft != G.use_fft) {                             /* change in data return type; re-initialise (quisk.c:4618-4621) */
        if (qk_set_graph_raw(G.qk, !use_fft) != QK_OK) return engine_error();
        G.use_fft = use_fft;
    }
    const int W = G.data_width;
    if (!use_fft) {
        const int got = qk_get_graph_raw(G.qk, G.row, W);
        if (got < 0) return engine_error();
        if (got == 0) Py_RETURN_NONE;
        PyObject *t = PyTuple_New(W);
        for (int i = 0; t && i < W; i++) PyTuple_SET_ITEM(t, i, PyComplex_FromDoubles(G.row[2 * i], G.row[2 * i + 1]));
        return t;
    }
    if (qk_graph_pending(G.qk) <= 0 || QuiskTimeSec() - G.graph_t0 < 1.0 / G.graph_refresh) Py_RETURN_NONE;   /* quisk.c:4712 */
    double sm;
    const int count = qk_get_graph(G.qk, zoom, deltaf, G.row, W, &sm);
    if (count < 0) return engine_error();
    if (count == 0) Py_RETURN_NONE;
    G.graph_t0 = QuiskTimeSec();
    PyObject *t = PyTuple_New(W);
    for (int i = 0; t && i < W; i++) PyTuple_SET_ITEM(t, i, PyFloat_FromDouble(G.row[i]));
    return t;
}
static PyObject *get_smeter(PyObject *self, PyObject *args)
{   /* quisk.c:2737-2742 */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    if (!need_engine()) return NULL;
    return PyFloat_FromDouble(qk_get_smeter(G.qk));
}

static PyObject *add_multirx_samples(PyObject *self, PyObject *args)
{   /* The sub-receiver frames that read_rx_udp10 collects (quisk.c:3520-3527) arrive here as complex128 buffers: frame
     * `index` fills up to 8 * multirx_data_width samples; further samples are dropped until the frame is used. */
    int index;
    Py_buffer view;
    if (!PyArg_ParseTuple(args, "iy*", &index, &view)) return NULL;
    if (!need_engine()) { PyBuffer_Release(&view); return NULL; }
    if (index >= 0 && index < QK_MAX_SUBRX && G.multirx_data_width > 0) {
        const int need = 8 * G.multirx_data_width;
        if (!G.mrx_frame[index]) { G.mrx_frame[index] = malloc(sizeof(double) * 2 * (size_t)need); G.mrx_fill[index] = 0; }
        if (G.mrx_frame[index]) {
            int take = (int)(view.len / 16);
            if (take > need - G.mrx_fill[index]) take = need - G.mrx_fill[index];
            if (take > 0) memcpy(G.mrx_frame[index] + 2 * (size_t)G.mrx_fill[index], view.buf, (size_t)take * 16);
            G.mrx_fill[index] += take > 0 ? take : 0;
            if (index + 1 > G.mrx_count) G.mrx_count = index + 1;
        }
    }
    PyBuffer_Release(&view);
    Py_RETURN_NONE;
}
static PyObject *get_multirx_graph(PyObject *self, PyObject *args)
{   /* quisk.c:4468-4530: (tuple, index) when the frame of the receiver whose turn it is is complete and
     * 1 / graph_refresh / count seconds have passed (quisk.c:3543-3547, 4485); else ((), -1) */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    if (!need_engine()) return NULL;
    const int idx = G.mrx_next, need = 8 * G.multirx_data_width, W = G.multirx_data_width;
    if (G.mrx_count <= 0 || W <= 0 || G.mrx_fill[idx] < need || QuiskTimeSec() - G.mrx_t0 < 1.0 / G.graph_refresh / G.mrx_count)
        return Py_BuildValue("(()i)", -1);
    double *row = malloc(sizeof(double) * (size_t)W);
    if (!row) return PyErr_NoMemory();
    if (qk_multirx_graph(G.qk, G.mrx_frame[idx], W, row) != QK_OK) { free(row); return engine_error(); }
    G.mrx_t0 = QuiskTimeSec();
    G.mrx_fill[idx] = 0;
    G.mrx_next = (idx + 1) % G.mrx_count;                   /* quisk.c:3538-3541 */
    PyObject *t = PyTuple_New(W);
    for (int i = 0; t && i < W; i++) PyTuple_SET_ITEM(t, i, PyFloat_FromDouble(row[i]));
    free(row);
    if (!t) return NULL;
    return Py_BuildValue("(Ni)", t, idx);
}

static PyObject *add_bscope_samples(PyObject *self, PyObject *args)
{   /* quisk.c:2841-2900 */
    PyObject *samples;
    Py_buffer view;
    if (!PyArg_ParseTuple(args, "O", &samples)) return NULL;
    if (G.bs_size <= 0) { printf("add_bscope_samples: The bandscope was not initialized with InitBscope()\n"); Py_RETURN_NONE; }
    if (!PyObject_CheckBuffer(samples)) { printf("add_bscope_samples: Invalid object sent as samples\n"); Py_RETURN_NONE; }
    if (PyObject_GetBuffer(samples, &view, PyBUF_SIMPLE) != 0) { PyErr_Clear(); printf("add_bscope_samples: Can not view sample buffer\n"); Py_RETURN_NONE; }
    int rc = QK_OK;
    if (view.len != (Py_ssize_t)G.bs_size * G.bs_bytes) printf("add_bscope_samples: Wrong number of bytes in sample buffer\n");
    else rc = qk_add_bscope_samples_int(G.qk, view.buf, (size_t)view.len, G.bs_bytes, G.bs_endian != 0);
    PyBuffer_Release(&view);
    if (rc != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *get_bandscope(PyObject *self, PyObject *args)
{   /* quisk.c:4532-4600: tuple of graph_width dB values or None */
    int clock;
    double zoom, deltaf, adc = 0;
    if (!PyArg_ParseTuple(args, "idd", &clock, &zoom, &deltaf)) return NULL;
    if (!need_engine()) return NULL;
    if (G.bs_size <= 0 || QuiskTimeSec() - G.bscope_t0 < 1.0 / G.graph_refresh) Py_RETURN_NONE;      /* quisk.c:4559 */
    const int W = G.graph_width;
    const int count = qk_get_bandscope(G.qk, clock, zoom, deltaf, G.row, W, &adc);
    if (count < 0) return engine_error();
    if (count == 0) Py_RETURN_NONE;
    G.adc_level = adc;                                      /* hermes_adc_level, quisk.c:4591 */
    G.bscope_t0 = QuiskTimeSec();
    PyObject *t = PyTuple_New(W);
    for (int i = 0; t && i < W; i++) PyTuple_SET_ITEM(t, i, PyFloat_FromDouble(G.row[i]));
    return t;
}

static PyObject *get_filter_rate(PyObject *self, PyObject *args)
{   /* quisk.c:2661-2735 */
    int mode, bw;
    if (!PyArg_ParseTuple(args, "ii", &mode, &bw)) return NULL;
    return PyLong_FromLong(qk_get_filter_rate(quisk_sound_state.sample_rate, mode, bw));
}
static PyObject *get_state(PyObject *self, PyObject *args)
{   /* quisk.c:2618-2643: the 17-tuple */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    qk_stats st;
    memset(&st, 0, sizeof st);
    if (G.qk) qk_get_stats(G.qk, &st);
    const struct sound_conf *s = &quisk_sound_state;
    return Py_BuildValue("iiiiisisiiiiiiiii", (int)s->rate_min, (int)s->rate_max, s->sample_rate, (int)s->chan_min, (int)s->chan_max,
                         s->msg1, 0, s->err_msg, s->read_error, s->write_error, s->underrun_error, s->latencyCapt, s->latencyPlay,
                         s->interupts, (int)st.fft_error, 0, s->data_poll_usec);
}
static PyObject *get_overrange(PyObject *self, PyObject *args)
{   /* quisk.c:2654-2659 -> quisk_get_overrange (sound.c): the count since the last call */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    const int n = quisk_sound_state.overrange;
    quisk_sound_state.overrange = 0;
    return PyLong_FromLong(n);
}
static PyObject *set_sparams(PyObject *self, PyObject *args, PyObject *kw)
{   /* sound.c:1253-1262: dc_remove_bw reaches the path; digital_output_level does not */
    static char *names[] = { "dc_remove_bw", "digital_output_level", NULL };
    int bw = -1; double level = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "|id", names, &bw, &level)) return NULL;
    if (bw >= 0) { if (!need_engine()) return NULL; if (qk_set_dc_remove(G.qk, bw) != QK_OK) return engine_error(); }
    Py_RETURN_NONE;
}
#define INT_SETTER(pyname, call, cite)                                           \
    static PyObject *pyname(PyObject *self, PyObject *args)                      \
    {   /* cite */                                                               \
        int v;                                                                   \
        if (!PyArg_ParseTuple(args, "i", &v)) return NULL;                       \
        if (!need_engine()) return NULL;                                         \
        if (call(G.qk, v) != QK_OK) return engine_error();                       \
        Py_RETURN_NONE;                                                          \
    }
INT_SETTER(invert_spectrum, qk_set_invert_spectrum, quisk.c:4174-4182)
INT_SETTER(set_kill_audio, qk_set_kill_audio, quisk.c:4364-4370)
INT_SETTER(set_noise_blanker, qk_set_noise_blanker, quisk.c:4246-4252)
static PyObject *set_squelch(PyObject *self, PyObject *args)
{   /* quisk.c:4348-4354 */
    double level;
    if (!PyArg_ParseTuple(args, "d", &level)) return NULL;
    if (!need_engine()) return NULL;
    if (qk_set_squelch(G.qk, level) != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *set_ssb_squelch(PyObject *self, PyObject *args)
{   /* quisk.c:4356-4362 */
    int enabled, level;
    if (!PyArg_ParseTuple(args, "ii", &enabled, &level)) return NULL;
    if (!need_engine()) return NULL;
    if (qk_set_ssb_squelch(G.qk, enabled, level) != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *set_auto_notch(PyObject *self, PyObject *args)
{   /* quisk.c:4237-4244: the reference's notch works on bank 0 */
    int on;
    if (!PyArg_ParseTuple(args, "i", &on)) return NULL;
    if (!need_engine()) return NULL;
    if (qk_set_auto_notch(G.qk, 0, on) != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *set_sidetone(PyObject *self, PyObject *args)
{   /* quisk.c:4331-4346 ("idii"): of the four values the receive path uses rit_freq (the CW notch leaves it alone) */
    int ctrl, rit, keyup;
    double volume;
    if (!PyArg_ParseTuple(args, "idii", &ctrl, &volume, &rit, &keyup)) return NULL;
    if (G.qk && qk_set_rit_freq(G.qk, rit) != QK_OK) return engine_error();
    G.rit_freq = rit;
    if (G.qk && G.split_rxtx && update_route() != QK_OK) return engine_error();      /* the second channel is tuned to tx + rit (quisk.c:2411) */
    Py_RETURN_NONE;
}
static PyObject *get_squelch(PyObject *self, PyObject *args)
{   /* quisk.c:2645-2652 -> IsSquelch */
    int freq;
    if (!PyArg_ParseTuple(args, "i", &freq)) return NULL;
    if (!need_engine()) return NULL;
    const int v = qk_is_squelch(G.qk, freq);
    if (v < 0) return engine_error();
    return PyLong_FromLong(v);
}
static PyObject *rx_filter_block(PyObject *self, PyObject *args)
{   /* cRxFilterOut over a block (an addition, like get_audio): (complex128 buffer, bank, nFilter) -> bytes */
    Py_buffer view;
    int bank, nFilter;
    if (!PyArg_ParseTuple(args, "y*ii", &view, &bank, &nFilter)) return NULL;
    if (!need_engine()) { PyBuffer_Release(&view); return NULL; }
    const int n = (int)(view.len / 16);
    PyObject *b = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)n * 16);
    const int rx = (nFilter == 1 && G.nrx > 1) ? 1 : 0;
    const int rc = b ? qk_rx_filter_block(G.qk, rx, bank, view.buf, (double *)PyBytes_AS_STRING(b), n) : 0;
    PyBuffer_Release(&view);
    if (!b) return NULL;
    if (rc != n) { Py_DECREF(b); return engine_error(); }
    return b;
}

/* Entries outside the receive path: accepted and ignored. */
/* ---------------------------------------------------------------- diagnostics fed by the path */
static PyObject *add_tone(PyObject *self, PyObject *args)
{   /* quisk.c:3022-3034 */
    int freq;
    if (!PyArg_ParseTuple(args, "i", &freq)) return NULL;
    if (!need_engine()) return NULL;
    if (qk_add_tone(G.qk, freq) != QK_OK) return engine_error();
    Py_RETURN_NONE;
}
static PyObject *float_tuple(const double *v, int n)
{
    PyObject *t = PyTuple_New(n);
    if (!t) return NULL;
    for (int i = 0; i < n; i++) PyTuple_SET_ITEM(t, i, PyFloat_FromDouble(v[i]));
    return t;
}
static PyObject *get_filter(PyObject *self, PyObject *args)
{   /* quisk.c:4761-4848: a tuple of data_width dB values */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    if (!need_engine()) return NULL;
    double *v = malloc(sizeof(double) * (size_t)G.data_width);
    if (!v) return PyErr_NoMemory();
    if (qk_get_filter(G.qk, v, G.data_width) != QK_OK) { free(v); return engine_error(); }
    PyObject *t = float_tuple(v, G.data_width);
    free(v);
    return t;
}
static PyObject *get_tx_filter(PyObject *self, PyObject *args)
{   /* microphone.c:719-807: the transmit chain is off the path; the reference's shape (data_width dB values) at its floor */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    PyObject *t = PyTuple_New(G.data_width > 0 ? G.data_width : 0);
    if (!t) return NULL;
    for (int i = 0; i < G.data_width; i++) PyTuple_SET_ITEM(t, i, PyFloat_FromDouble(-140.0));
    return t;
}
static PyObject *get_audio_graph(PyObject *self, PyObject *args)
{   /* quisk.c:966-991: None until a graph is ready */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    if (!need_engine()) return NULL;
    double *v = malloc(sizeof(double) * (size_t)G.data_width);
    if (!v) return PyErr_NoMemory();
    const int got = qk_get_audio_graph(G.qk, v, G.data_width);
    if (got < 0) { free(v); return engine_error(); }
    PyObject *t = got ? float_tuple(v, G.data_width) : (Py_INCREF(Py_None), Py_None);
    free(v);
    return t;
}
static PyObject *measure_frequency(PyObject *self, PyObject *args)
{   /* quisk.c:3000-3009 */
    int mode;
    double v = 0;
    if (!PyArg_ParseTuple(args, "i", &mode)) return NULL;
    if (!need_engine()) return NULL;
    if (qk_measure_frequency(G.qk, mode, &v) != QK_OK) return engine_error();
    return PyFloat_FromDouble(v);
}
static PyObject *measure_audio(PyObject *self, PyObject *args)
{   /* quisk.c:3011-3020 */
    int time;
    double v = 0;
    if (!PyArg_ParseTuple(args, "i", &time)) return NULL;
    if (!need_engine()) return NULL;
    if (qk_measure_audio(G.qk, time, &v) != QK_OK) return engine_error();
    return PyFloat_FromDouble(v);
}
static PyObject *xdft(PyObject *seq, int inverse, int window)
{   /* Xdft, quisk.c:4922-5014: the sequence is in order of frequency with zero at index (size - 1) / 2; Blackman window on the
     * frequency side of an inverse / the time side of a forward transform; the inverse is normalised by 1 / N; returns a list */
    if (PySequence_Check(seq) != 1) { PyErr_SetString(QuiskError, "DFT input data is not a sequence"); return NULL; }
    const Py_ssize_t size = PySequence_Size(seq);
    if (size <= 0) return PyTuple_New(0);
    if (!need_engine()) return NULL;
    double *x = malloc(sizeof(double) * 4 * (size_t)size), *y = x ? x + 2 * size : NULL;
    if (!x) return PyErr_NoMemory();
    Py_ssize_t j = (size - 1) / 2;
    for (Py_ssize_t i = 0; i < size; i++) {
        PyObject *o = PySequence_GetItem(seq, j);
        if (o && PyComplex_Check(o)) { Py_complex c = PyComplex_AsCComplex(o); x[2 * i] = c.real; x[2 * i + 1] = c.imag; }
        else if (o && PyFloat_Check(o)) { x[2 * i] = PyFloat_AsDouble(o); x[2 * i + 1] = 0; }
        else if (o && PyLong_Check(o)) { x[2 * i] = (double)PyLong_AsLong(o); x[2 * i + 1] = 0; }
        else { Py_XDECREF(o); free(x); PyErr_SetString(QuiskError, "DFT input data is not a complex/float/int number"); return NULL; }
        Py_DECREF(o);
        if (++j >= size) j = 0;
    }
    double *w = NULL;
    if (window) {
        w = malloc(sizeof(double) * (size_t)(size + 1));
        if (!w) { free(x); return PyErr_NoMemory(); }
        for (Py_ssize_t i = 0; i <= size / 2; i++)
            w[i] = w[size - i] = 0.42 + 0.50 * cos(2. * M_PI * i / size) + 0.08 * cos(4. * M_PI * i / size);
    }
    if (!inverse && w) for (Py_ssize_t i = 0; i < size; i++) { x[2 * i] *= w[i]; x[2 * i + 1] *= w[i]; }
    if (qk_fft_any(G.qk, x, (int)size, inverse, y) != QK_OK) { free(x); free(w); return engine_error(); }
    if (inverse) for (Py_ssize_t i = 0; i < size; i++) {
        if (w) { const double f = w[i] / size; y[2 * i] *= f; y[2 * i + 1] *= f; }
        else { y[2 * i] /= size; y[2 * i + 1] /= size; }
    }
    PyObject *out = PyList_New(size);
    j = (size - 1) / 2;
    for (Py_ssize_t i = 0; out && i < size; i++) {
        PyList_SET_ITEM(out, j, PyComplex_FromDoubles(y[2 * i], y[2 * i + 1]));
        if (++j >= size) j = 0;
    }
    free(x); free(w);
    return out;
}
static PyObject *dft(PyObject *self, PyObject *args)
{
    PyObject *seq; int window = 0;
    if (!PyArg_ParseTuple(args, "O|i", &seq, &window)) return NULL;
    return xdft(seq, 0, window);
}
static PyObject *idft(PyObject *self, PyObject *args)
{
    PyObject *seq; int window = 0;
    if (!PyArg_ParseTuple(args, "O|i", &seq, &window)) return NULL;
    return xdft(seq, 1, window);
}
/* Entries of the reference's table that are off the receive path (sound cards, Hermes control, FreeDV): callable with the
 * reference's arity, returning the reference's shapes with nothing in them */
static PyObject *no_args_none(PyObject *self, PyObject *args) { if (!PyArg_ParseTuple(args, "")) return NULL; Py_RETURN_NONE; }
static PyObject *sound_devices(PyObject *self, PyObject *args)
{   /* sound_alsa.c:496: (capture names, playback names) */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    return Py_BuildValue("([][])");
}
static PyObject *sound_errors(PyObject *self, PyObject *args) { if (!PyArg_ParseTuple(args, "")) return NULL; return PyList_New(0); }   /* sound.c:1284 */
static PyObject *ip_interfaces(PyObject *self, PyObject *args) { if (!PyArg_ParseTuple(args, "")) return NULL; return PyList_New(0); }  /* quisk.c:4062 */
static PyObject *get_hermes_adc(PyObject *self, PyObject *args)
{   /* quisk.c:2751-2756: hermes_adc_level, the largest bandscope sample of the last release */
    if (!PyArg_ParseTuple(args, "")) return NULL;
    return PyFloat_FromDouble(G.adc_level);
}
static PyObject *get_hermes_TFRC(PyObject *self, PyObject *args) { if (!PyArg_ParseTuple(args, "")) return NULL; return Py_BuildValue("dddd", 0.0, 0.0, 0.0, 0.0); }  /* quisk.c:2968 */
static PyObject *get_hermeslite_writepointer(PyObject *self, PyObject *args) { if (!PyArg_ParseTuple(args, "")) return NULL; return Py_BuildValue("I", 0u); }        /* quisk.c:3988 */
static PyObject *get_hermeslite_response(PyObject *self, PyObject *args) { static const char z[5] = {0}; if (!PyArg_ParseTuple(args, "")) return NULL; return PyByteArray_FromStringAndSize(z, 5); }   /* quisk.c:3995 */
static PyObject *hermes_to_pc(PyObject *self, PyObject *args) { static const char z[20] = {0}; if (!PyArg_ParseTuple(args, "")) return NULL; return PyByteArray_FromStringAndSize(z, 20); }            /* quisk.c:4011 */
static PyObject *freedv_get_snr(PyObject *self, PyObject *args) { if (!PyArg_ParseTuple(args, "")) return NULL; return PyFloat_FromDouble(0.0); }       /* freedv.c:532 */
static PyObject *freedv_get_version(PyObject *self, PyObject *args) { if (!PyArg_ParseTuple(args, "")) return NULL; return PyLong_FromLong(-1); }      /* freedv.c:62, 543: no codec library */
static PyObject *freedv_get_rx_char(PyObject *self, PyObject *args) { if (!PyArg_ParseTuple(args, "")) return NULL; return PyUnicode_FromString(""); } /* freedv.c:379 */

static PyObject *inert(PyObject *self, PyObject *args, PyObject *kw) { Py_RETURN_NONE; }
static PyObject *inert_zero(PyObject *self, PyObject *args, PyObject *kw) { return PyLong_FromLong(0); }
#define INERT(name) { name, (PyCFunction)inert, METH_VARARGS | METH_KEYWORDS, "Outside the receive path: accepted and ignored." }

static PyMethodDef QuiskMethods[] = {
    { "record_app", record_app, METH_VARARGS, "Save the App instance and create the engine." },
    { "close_sound", close_sound, METH_VARARGS, "Stop and release the engine." },
    { "start_sound", start_sound, METH_VARARGS, "Start the sample source." },
    { "set_params", (PyCFunction)set_params, METH_VARARGS | METH_KEYWORDS, "Set miscellaneous parameters." },
    { "set_sparams", (PyCFunction)set_sparams, METH_VARARGS | METH_KEYWORDS, "Set miscellaneous sound parameters." },
    { "add_rx_samples", add_rx_samples, METH_VARARGS, "Record the Rx samples received by Python code." },
    { "read_sound", read_sound, METH_VARARGS, "Read and process the waiting samples." },
    { "get_audio", get_audio, METH_VARARGS, "The last block's stereo audio of a receiver (complex128 bytes)." },
    { "set_tune", set_tune, METH_VARARGS, "Set the tuning frequency." },
    { "set_rx_mode", set_rx_mode, METH_VARARGS, "Set the receive mode: CWL, USB, AM, etc." },
    { "set_filters", set_filters, METH_VARARGS, "Set the receive audio I and Q channel filters." },
    { "set_agc", set_agc, METH_VARARGS, "Set the AGC parameters." },
    { "set_multirx_mode", set_multirx_mode, METH_VARARGS, "Select demodulation mode for sub-receivers." },
    { "set_multirx_freq", set_multirx_freq, METH_VARARGS, "Select the frequency of sub-receivers." },
    { "get_graph", get_graph, METH_VARARGS, "Return a tuple of graph data." },
    { "get_smeter", get_smeter, METH_VARARGS, "Return the S meter reading." },
    { "add_multirx_samples", add_multirx_samples, METH_VARARGS, "Record samples of a sub-receiver's graph frame." },
    { "get_multirx_graph", get_multirx_graph, METH_VARARGS, "Return a tuple of sub-receiver graph data." },
    { "add_bscope_samples", add_bscope_samples, METH_VARARGS, "Record the bandscope samples received by Python code." },
    { "get_bandscope", get_bandscope, METH_VARARGS, "Return a tuple of bandscope data." },
    { "get_filter_rate", get_filter_rate, METH_VARARGS, "Return the sample rate used for the filters." },
    { "get_state", get_state, METH_VARARGS, "Return a count of read and write errors." },
    { "get_overrange", get_overrange, METH_VARARGS, "Return the count of overrange (clip) for the ADC." },
    { "invert_spectrum", invert_spectrum, METH_VARARGS, "Invert the input RF spectrum" },
    { "set_kill_audio", set_kill_audio, METH_VARARGS, "Replace radio sound with silence." },
    { "set_noise_blanker", set_noise_blanker, METH_VARARGS, "Set the noise blanker level." },
    { "set_squelch", set_squelch, METH_VARARGS, "Set the FM squelch parameter." },
    { "set_ssb_squelch", set_ssb_squelch, METH_VARARGS, "Set the SSB squelch parameters." },
    { "set_auto_notch", set_auto_notch, METH_VARARGS, "Set the auto notch on or off." },
    { "set_sidetone", set_sidetone, METH_VARARGS, "Set the sidetone volume and frequency." },
    { "get_squelch", get_squelch, METH_VARARGS, "Get the FM squelch state, 0 or 1." },
    { "set_split_rxtx", set_split_rxtx, METH_VARARGS, "Set split for rx/tx." },
    { "set_multirx_play_method", set_multirx_play_method, METH_VARARGS, "Select how to play audio from sub-receivers." },
    { "set_multirx_play_channel", set_multirx_play_channel, METH_VARARGS, "Select which sub-receiver to play audio." },
    { "change_rates", change_rates, METH_VARARGS, "The VFOs of the audio and the screen (the sample rate is fixed by record_app)." },
    { "change_scan", change_scan, METH_VARARGS, "Stitch the frames of a VFO sweep into one graph row." },
    { "key_up", key_up, METH_VARARGS, "The key went up: start the 5 ms envelope." },
    { "rx_filter_block", rx_filter_block, METH_VARARGS, "cRxFilterOut over a block of complex128 samples." },
    { "is_key_down", (PyCFunction)inert_zero, METH_VARARGS | METH_KEYWORDS, "Receive only: the key is up." },
    { "get_hardware_ptt", (PyCFunction)inert_zero, METH_VARARGS | METH_KEYWORDS, "Receive only." },
    { "is_vox", (PyCFunction)inert_zero, METH_VARARGS | METH_KEYWORDS, "Receive only." },
    { "add_tone", add_tone, METH_VARARGS, "Add a test tone to the data." },
    { "get_filter", get_filter, METH_VARARGS, "Return the frequency response of the receive filter." },
    { "get_tx_filter", get_tx_filter, METH_VARARGS, "Return the frequency response of the transmit filter." },
    { "get_audio_graph", get_audio_graph, METH_VARARGS, "Return a tuple of the audio graph data." },
    { "measure_frequency", measure_frequency, METH_VARARGS, "Set the method, return the measured frequency." },
    { "measure_audio", measure_audio, METH_VARARGS, "Set the method, return the measured audio voltage." },
    { "dft", dft, METH_VARARGS, "Calculate the discrete Fourier transform." },
    { "idft", idft, METH_VARARGS, "Calculate the inverse discrete Fourier transform." },
    { "sound_devices", sound_devices, METH_VARARGS, "Return a list of available sound device names." },
    { "pa_sound_devices", sound_devices, METH_VARARGS, "Return a list of available PulseAudio device names." },
    { "sound_errors", sound_errors, METH_VARARGS, "Return a list of text strings with sound devices and error counts" },
    { "ip_interfaces", ip_interfaces, METH_VARARGS, "Return a list of interface data." },
    { "get_hermes_adc", get_hermes_adc, METH_VARARGS, "Return the ADC peak level." },
    { "get_hermes_TFRC", get_hermes_TFRC, METH_VARARGS, "Return the temperature, forward and reverse power and PA current." },
    { "get_hermeslite_writepointer", get_hermeslite_writepointer, METH_VARARGS, "Return the Hermes-Lite write pointer." },
    { "get_hermeslite_response", get_hermeslite_response, METH_VARARGS, "Return the Hermes-Lite response array." },
    { "hermes_to_pc", hermes_to_pc, METH_VARARGS, "Get the Hermes response C0 to C4 for C0 == 0, 8, 16, 24." },
    { "test_1", no_args_none, METH_VARARGS, "Test 1 function." },
    { "test_2", no_args_none, METH_VARARGS, "Test 2 function." },
    { "test_3", no_args_none, METH_VARARGS, "Test 3 function." },
    { "freedv_get_snr", freedv_get_snr, METH_VARARGS, "Return the signal to noise ratio in dB." },
    { "freedv_get_version", freedv_get_version, METH_VARARGS, "Return the codec2 API version." },
    { "freedv_get_rx_char", freedv_get_rx_char, METH_VARARGS, "Get text characters received from freedv." },
    INERT("set_volume"), INERT("set_enable_bandscope"), INERT("set_spot_level"), INERT("set_imd_level"),
    INERT("set_key_down"), INERT("set_PTT"), INERT("set_ctcss"), INERT("set_tx_audio"), INERT("set_record_state"), INERT("set_file_name"),
    INERT("set_transmit_mode"), INERT("set_fdx"), INERT("set_sample_bytes"), INERT("change_rate"),
    INERT("open_sound"), INERT("open_key"), INERT("open_rx_udp"), INERT("close_rx_udp"),
    INERT("set_hermes_id"), INERT("set_hermes_filters"), INERT("set_alex_hpf"), INERT("set_alex_lpf"), INERT("set_hermeslite_writepointer"),
    INERT("clear_hermeslite_response"), INERT("pc_to_hermes"), INERT("pc_to_hermeslite_writequeue"), INERT("freedv_open"),
    INERT("freedv_close"), INERT("freedv_set_options"), INERT("record_graph"), INERT("ImmediateChange"), INERT("set_ampl_phase"),
    INERT("set_udp_tx_correct"), INERT("set_mic_out_volume"), INERT("tx_hold_state"), INERT("mixer_set"), INERT("capt_channels"),
    INERT("play_channels"), INERT("micplay_channels"), INERT("open_wav_file_play"),
    { NULL, NULL, 0, NULL }
};

static struct PyModuleDef quisk_module = { PyModuleDef_HEAD_INIT, "_quisk", NULL, -1, QuiskMethods };

PyMODINIT_FUNC PyInit__quisk(void)
{   /* quisk.c:5302-5324 */
    static void *api[11];
    PyObject *m = PyModule_Create(&quisk_module);
    if (!m) return NULL;
    QuiskError = PyErr_NewException("_quisk.error", NULL, NULL);
    if (!QuiskError) { Py_DECREF(m); return NULL; }
    Py_INCREF(QuiskError);
    PyModule_AddObject(m, "error", QuiskError);
    /* the order of QUISK_API_INIT, quisk.h:351-355 */
    api[0] = &quisk_sound_state; api[1] = (void *)&QuiskGetConfigInt; api[2] = (void *)&QuiskGetConfigDouble;
    api[3] = (void *)&QuiskGetConfigString; api[4] = (void *)&QuiskTimeSec; api[5] = (void *)&QuiskSleepMicrosec;
    api[6] = (void *)&QuiskPrintTime; api[7] = (void *)&quisk_sample_source; api[8] = (void *)&quisk_dvoice_freedv;
    api[9] = (void *)&quisk_is_key_down; api[10] = (void *)&quisk_sample_source4;
    PyObject *cap = PyCapsule_New(api, "_quisk.QUISK_C_API", NULL);
    if (cap) PyModule_AddObject(m, "QUISK_C_API", cap);
    return m;
}
