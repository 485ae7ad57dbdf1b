/* qk_oracle.h -- CPU restatement of Quisk's receive-DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under quisk_amd/ may include, link or load
 * this; it is the checker the HIP path is compared with (tests/, smoke(), and the
 * cpu_baseline leg of bench.py).
 *
 * Pinning status:
 *   - filter kernels (qko_cdecimate ... qko_cinterp2hb45, qko_filt_tune): PINNED,
 *     bit-exact against the reference's own filter.c compiled unmodified into
 *     oracle/_ref/libquisk_filter_ref.so (tests/test_oracle_vs_ref.py) and against
 *     tests/golden/filter_kernels.npz generated from it.
 *   - everything restated from quisk.c (mixer, PlanDecimation, cRxFilterOut,
 *     dRxFilterOut, demodulators, cFracDecim, process_agc, graph post-processing):
 *     PARITY UNPINNED.  quisk.c needs <fftw3.h>, which this image lacks, so it is
 *     unbuildable here, and the reference ships no tests or golden vectors.  These
 *     functions follow the cited lines of quisk.c by reading.
 *   - FFT: the reference calls FFTW3 (not vendored, version unpinned).  qko_fft is
 *     the mathematical DFT in double, checked against numpy.fft in tests.
 */
#ifndef QK_ORACLE_H
#define QK_ORACLE_H
#include <complex.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef double complex qko_cd;

/* rx_mode_type, quisk.h:48-63 */
enum { QKO_CWL = 0, QKO_CWU = 1, QKO_LSB = 2, QKO_USB = 3, QKO_AM = 4, QKO_FM = 5, QKO_EXT = 6,
       QKO_DGT_U = 7, QKO_DGT_L = 8, QKO_DGT_IQ = 9, QKO_IMD = 10, QKO_FDV_U = 11, QKO_FDV_L = 12,
       QKO_DGT_FM = 13 };

#define QKO_CLIP32 2147483647.0          /* quisk.h:13 */
#define QKO_OUT_CAP (66000 * 8 / 10)     /* SAMP_BUFFER_SIZE*8/10, filter.c:135 */
#define QKO_MAX_FILTER 10001             /* quisk.h:10 */

/* ---- filter.c kernels ------------------------------------------------------------ */
typedef struct {            /* struct quisk_cFilter, filter.h:1-10 */
    const double *h; qko_cd *hc; int ntaps; int decim_index; qko_cd *ring; int pos;
    qko_cd *scratch; int nscratch;
} qko_cfir;
typedef struct {            /* struct quisk_dFilter, filter.h:12-21 */
    const double *h; qko_cd *hc; int ntaps; int decim_index; double *ring; int pos;
    double *scratch; int nscratch;
} qko_dfir;
typedef struct { int toggle; qko_cd samples[22]; qko_cd center[11]; qko_cd *scratch; int nscratch; } qko_chb45; /* filter.h:23-29 */
void qko_dc_out_block(const double *x, int n, qko_dfir *f, qko_cd *out);      /* quisk_dC_out, filter.c:60-81 */
void qko_dd_out_block(double *x, int n, qko_dfir *f);                         /* quisk_dD_out, filter.c:303-322 */
/* read_rx_udp10's record loop for one packet (quisk.c:3513-3531) */
void qko_hermes_unpack(const unsigned char *buf, int multirx_count, qko_cd *samp, qko_cd *sub, long sub_stride, int *nsamples);
typedef struct { int toggle; double samples[22]; double center[11]; double *scratch; int nscratch; } qko_dhb45; /* filter.h:31-37 */

void qko_cfir_init(qko_cfir *f, const double *h, int ntaps);           /* filter.c:9-20 */
void qko_dfir_init(qko_dfir *f, const double *h, int ntaps);           /* filter.c:22-33 */
void qko_cfir_free(qko_cfir *f);
void qko_dfir_free(qko_dfir *f);
void qko_filt_tune(qko_dfir *f, double freq, int ssb_upper);           /* filter.c:35-58 */
int qko_cdecimate(qko_cd *x, int n, qko_cfir *f, int decim);           /* filter.c:180-206 */
int qko_ccdecimate(qko_cd *x, int n, qko_cfir *f, int decim);          /* filter.c:208-234 */
int qko_ddecimate(double *x, int n, qko_dfir *f, int decim);           /* filter.c:236-262 */
int qko_dfilter(double *x, int n, qko_dfir *f);                        /* filter.c:324-347 */
int qko_cinterpolate(qko_cd *x, int n, qko_cfir *f, int interp);       /* filter.c:108-142 */
int qko_dinterpolate(double *x, int n, qko_dfir *f, int interp);       /* filter.c:144-178 */
int qko_cinterpdecim(qko_cd *x, int n, qko_cfir *f, int interp, int decim); /* filter.c:264-301 */
int qko_cdecim2hb45(qko_cd *x, int n, qko_chb45 *f);                   /* filter.c:354-394 */
int qko_dinterp2hb45(double *x, int n, qko_dhb45 *f);                  /* filter.c:397-430 */
int qko_cinterp2hb45(qko_cd *x, int n, qko_chb45 *f);                  /* filter.c:432-465 */

/* ---- quisk.c: planning ----------------------------------------------------------- */
int qko_plan_decimation(int sample_rate, int *d2, int *d3, int *d5);   /* quisk.c:1509-1547 */
int qko_decim_rate(int sample_rate);      /* quisk_decim_srate after quisk_process_decimate, quisk.c:1607-1720 */
int qko_get_filter_rate(int sample_rate, int mode, int bandwidth);     /* quisk.c:2661-2735 */

/* ---- quisk.c: one receiver ("bank") ---------------------------------------------- */
typedef struct {            /* struct AgcState, quisk.c:51-64 */
    double max_out; int sample_rate; int buf_size; int index_read, index_start, is_clipping;
    double themax, gain, delta, target_gain, time_release; qko_cd *ring;
} qko_agc;

typedef struct qko_rx qko_rx;

/* A receiver owns every piece of state quisk_process_samples keeps for bank 0 (function statics
 * at quisk.c:2161-2181, Storage[] at 1554-1574 and 1730-1747, cRxFilterOut/dRxFilterOut rings). */
qko_rx *qko_rx_new(int sample_rate, int playback_rate);
void qko_rx_free(qko_rx *rx);
void qko_rx_set_squelch(qko_rx *rx, double level_db);                  /* set_squelch, quisk.c:4348 (FM squelch level, dB) */
void qko_rx_set_kill_audio(qko_rx *rx, int on);                        /* set_kill_audio, quisk.c:4364 */
int  qko_rx_squelch_active(const qko_rx *rx);
void qko_rx_set_tune(qko_rx *rx, int tune_hz);                         /* set_tune, quisk.c:4323 */
void qko_rx_set_mode(qko_rx *rx, int mode);                            /* set_rx_mode, quisk.c:4262 */
int  qko_rx_set_filter(qko_rx *rx, const double *fi, const double *fq, int n, int bandwidth); /* set_filters, quisk.c:4192 */
void qko_rx_set_agc(qko_rx *rx, double release_gain, double release_time); /* set_agc 4184; agc_release_time 146 */
/* quisk_process_samples (quisk.c:2149-2616), key up, default options: in place, returns the number
 * of stereo output samples written to the head of x.  n <= 66000 like the reference's buffer. */
int  qko_rx_process(qko_rx *rx, qko_cd *x, int n);
/* the same in two steps: everything before the AGC (returns the count), then the AGC in place */
int  qko_rx_pre_agc(qko_rx *rx, qko_cd *x, int n);
void qko_rx_agc(qko_rx *rx, qko_cd *x, int n);
/* stages, exposed for per-stage parity tests */
void qko_rx_mix(qko_rx *rx, qko_cd *x, int n);                         /* quisk.c:2350-2361 */
int  qko_rx_decimate(qko_rx *rx, qko_cd *x, int n);                    /* quisk.c:1549-1722 */
int  qko_rx_demodulate(qko_rx *rx, qko_cd *x, double *audio, int n);   /* quisk.c:1724-2020 */
void qko_rx_set_own_resampler(qko_rx *rx, int on);                   /* see qk_oracle.c: quisk.c:1713 shares filtDecim5S */
void qko_rx_set_ssb_squelch(qko_rx *rx, int enabled, int level);     /* set_ssb_squelch, quisk.c:4356-4362 */
void qko_rx_set_auto_notch(qko_rx *rx, int on, int rit_freq);        /* set_auto_notch, quisk.c:4237-4244 (bank 0 only there) */
int  qko_rx_decim_srate(const qko_rx *rx);
int  qko_rx_filter_srate(const qko_rx *rx);

void qko_agc_init(qko_agc *a, double max_out, int sample_rate);
void qko_agc_free(qko_agc *a);
/* process_agc, quisk.c:2022-2147.  playback_rate is used when a->sample_rate is 0;
 * fm_mode is (rxMode == FM || rxMode == DGT_FM), quisk.c:2088. */
void qko_agc_process(qko_agc *a, qko_cd *x, int n, int is_cpx, int playback_rate,
                     double release_gain, double release_time, int fm_mode);
/* test aid: the same, one sample at a time, with the state after every sample */
void qko_agc_trace(qko_agc *a, qko_cd *x, int n, int is_cpx, int playback_rate, double release_gain, double release_time,
                   int fm_mode, double *gain_out, int *clip_out, int *istart_out);

typedef struct { double dindex; qko_cd c0, c1, c2, c3; } qko_fracstate; /* statics at quisk.c:575-576 */
void qko_fracdecim_init(qko_fracstate *s);
int  qko_fracdecim(qko_fracstate *s, qko_cd *x, int n, double fdecim);  /* quisk.c:570-613 */

/* ---- quisk.c: display spectrum --------------------------------------------------- */
void qko_fft(qko_cd *x, int n);               /* forward complex DFT in place, any n (radix 2 for powers of two) */
void qko_hann_window(double *w, int n);       /* quisk.c:5091-5096 */
typedef struct {
    int fft_size, data_width, sample_rate; double *window; double *avg; int count; double meter;
    int rx_tune, filter_start_offset, filter_bandwidth;
    /* change_scan (quisk.c:3865-3872): frames carry block numbers 0 .. scan_blocks-1 and are stitched (quisk.c:4682-4702) */
    int scan_blocks, scan_sample_rate, scan_vfo0, scan_deltaf, tx_tune, vfo_screen, next_block;
    double scan_valid; double *tmp;
} qko_graph;
void qko_graph_init(qko_graph *g, int fft_size, int data_width, int sample_rate);
void qko_graph_free(qko_graph *g);
void qko_graph_set_passband(qko_graph *g, int rx_tune, int filter_start_offset, int filter_bandwidth);
/* change_scan "iidii" plus the two globals its S-meter formula reads (quisk_tx_tune_freq, vfo_screen; quisk.c:4661); the next
 * frame is block 0 (the hardware's zero marker, quisk.c:3672-3679), later ones count up cyclically (quisk.c:3680-3689) */
void qko_graph_change_scan(qko_graph *g, int scan_blocks, int scan_sample_rate, double scan_valid, int scan_vfo0, int scan_deltaf,
                           int tx_tune, int vfo_screen);
/* one filled FFT buffer: window, FFT, S-meter sum, magnitude average (quisk.c:4651-4710).  frame is overwritten. */
void qko_graph_add_frame(qko_graph *g, qko_cd *frame);
/* release (quisk.c:4713-4753): pixel binning, dB scaling; returns the averaged-frame count, resets the average.
 * out has data_width doubles; *smeter receives the S-meter value (quisk.c:4728-4737). */
int qko_graph_release(qko_graph *g, double zoom, double deltaf, double *out, double *smeter);

/* add_tone + AddTestTone (quisk.c:3022-3034, 1194-1229): testtoneVector and audioVector as a struct */
typedef struct { qko_cd tone, audio; } qko_tone;
void qko_tone_init(qko_tone *t);
void qko_add_test_tone(qko_tone *t, qko_cd *x, int n, int freq, int sample_rate, int mode);
/* get_filter (quisk.c:4761-4848): data_width dB values, negative frequencies first */
void qko_get_filter(const double *fi, const double *fq, int nfilt, int data_width, int fft_size, double *out);
/* measure_freq (quisk.c:4850-4920), its statics as a struct; x: the decimated samples of a block (untouched) */
typedef struct { int index, count, nbuf; double value; qko_cd *samples, *buf; double *window, *average; qko_chb45 hb[3]; } qko_mfreq;
void qko_mfreq_init(qko_mfreq *m);
void qko_mfreq_free(qko_mfreq *m);
double qko_measure_freq(qko_mfreq *m, const qko_cd *x, int n, int srate, int mode, int rit_freq, int rx_tune);
/* quisk_calc_audio_graph / get_audio_graph (quisk.c:903-991), their statics as a struct */
typedef struct { int data_width, planned, index, count, count_limit, ready; double *window, *average; qko_cd *fft; } qko_agraph;
qko_agraph *qko_audio_graph_new(int data_width);
void qko_audio_graph_delete(qko_agraph *g);
void qko_audio_graph_add(qko_agraph *g, const double *d, int n);
int qko_audio_graph_get(qko_agraph *g, double *out);
void qko_rx_set_audio_graph(qko_rx *rx, qko_agraph *g);       /* receiver rx feeds g from its LSB branch (quisk.c:1835) */
double qko_rx_measure_audio(qko_rx *rx, int time);           /* measure_audio, quisk.c:3011-3020 */
qko_mfreq *qko_mfreq_new(void);
void qko_mfreq_delete(qko_mfreq *m);

/* Buffer2Chan (quisk.c:1452-1487), its statics as a struct (zero-initialised); and the key-up envelope (quisk.c:2602-2612) */
#define QKO_B2C_SIZE 12000
typedef struct { int nbuf1, nbuf2; double buf1[QKO_B2C_SIZE], buf2[QKO_B2C_SIZE]; } qko_b2c;
int qko_buffer2chan(qko_b2c *b, double *samp1, int count1, double *samp2, int count2);
double qko_keyup_envelope(double env, qko_cd *x, int n, int playback_rate);
/* quisk_process_samples with a second channel (split RX/TX, multirx play; quisk.c:2410-2494, 2528-2601): see qk_oracle.c */
typedef struct { qko_b2c b2c; qko_agc agc2; qko_cd *xb, *ch; double *ab; int cap; } qko_pair;
void qko_pair_init(qko_pair *p);
void qko_pair_free(qko_pair *p);
qko_pair *qko_pair_new(void);
void qko_pair_delete(qko_pair *p);
void qko_pair_reset_fifos(qko_pair *p);
int qko_rx_pair_process(qko_rx *a, qko_rx *b, qko_pair *p, qko_cd *x, int n, int method, qko_cd *pre);

/* get_multirx_graph (quisk.c:4468-4530): 8 * data_width samples (overwritten) -> data_width dB values */
void qko_multirx_graph(qko_cd *samples, int data_width, double *out);
/* bandscope: init_bandscope (quisk.c:2758-2775), get_bandscope (quisk.c:4532-4600) split into its two halves */
typedef struct { int size, count; double *window, *avg; double the_max; } qko_bscope;
void qko_bscope_init(qko_bscope *b, int size);
void qko_bscope_free(qko_bscope *b);
void qko_bscope_add(qko_bscope *b, const double *samples);
int  qko_bscope_release(qko_bscope *b, int clock, double zoom, double deltaf, double *out, int graph_width, double *adc_level);

/* input conditioning in front of quisk_process_samples: DCremove (sound.c:165-230, statics as a struct) and the spectrum
 * inversion of quisk.c:2315-2320 */
typedef struct {
    int old_sample_rate, old_bandwidth; double alpha; qko_cd dc_remove, dc_average, dc_sum; int dc_count, dc_key_delay;
} qko_dcstate;
void qko_dcremove_init(qko_dcstate *d);
void qko_dcremove(qko_dcstate *d, qko_cd *x, int n, int sample_rate, int dc_remove_bw, int key_state);
void qko_invert_spectrum(qko_cd *x, int n);
/* NoiseBlanker (quisk.c:615-720), its function statics as a struct: a delay line of 1.5 ms whose samples are ramped down
 * in front of a pulse (magnitude above `limit` times the running average), zeroed during it and ramped up behind it.
 * level 1 / 2 / 3 = limit 6 / 4 / 2.5; level <= 0 leaves the samples alone (and the delay line untouched). */
typedef struct {
    qko_cd *ring; double *mags; double sum; int size, hwindow, state, index, win_index, sample_rate;
} qko_nbstate;
void qko_nb_init(qko_nbstate *b);
void qko_nb_free(qko_nbstate *b);
void qko_noise_blanker(qko_nbstate *b, qko_cd *x, int n, int sample_rate, int level);

/* ---- batch driver used as the timed CPU baseline --------------------------------- */
/* run nrx receivers (already configured) over the same input block; audio of receiver r is written to
 * out + r*out_stride; counts[r] receives the sample count.  Receivers are processed one after the other. */
void qko_rx_process_many(qko_rx **rx, int nrx, const qko_cd *x, int n, qko_cd *out, long out_stride, int *counts);

const double *qko_table(const char *name, int *ntaps);   /* coefficient tables by project name */

#ifdef __cplusplus
}
#endif
#endif
