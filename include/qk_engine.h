/* qk_engine.h -- C ABI of the MI355X receive-DSP engine (libqkengine.so).
 *
 * This is the drop-in boundary for Quisk's receive hot path: the entry points are what the
 * reference's `_quisk` extension (quisk.c) binds for that path, generalised from the reference's
 * three demodulator "banks" (MAX_RX_CHANNELS, quisk.c:33) to N independent receivers that share one
 * wideband I/Q stream.  Plain C types only.  Each declaration cites the reference interface it
 * replaces (file:line under the reference tree).  INTEGRATION.md shows the binding a maintainer of
 * the reference would add.
 *
 * Threading: like the reference (two threads, no locks), setters may be called from a thread other
 * than the one that calls qk_process_*; the engine takes a mutex around every entry point, so
 * setters take effect at the next block boundary.
 *
 * All functions return 0 (QK_OK) or a negative error code unless stated otherwise;
 * qk_last_error() returns a message for the calling thread's last failure.
 */
#ifndef QK_ENGINE_H
#define QK_ENGINE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qk_engine qk_engine;

enum { QK_OK = 0, QK_EINVAL = -1, QK_ENOMEM = -2, QK_EHIP = -3, QK_ERANGE = -4, QK_EUNSUPPORTED = -5 };

/* receive modes: rx_mode_type, quisk.h:48-63 (same numeric values, set_rx_mode passes them through) */
enum { QK_CWL = 0, QK_CWU = 1, QK_LSB = 2, QK_USB = 3, QK_AM = 4, QK_FM = 5, QK_EXT = 6, QK_DGT_U = 7,
       QK_DGT_L = 8, QK_DGT_IQ = 9, QK_IMD = 10, QK_FDV_U = 11, QK_FDV_L = 12, QK_DGT_FM = 13 };

typedef struct qk_config {
    int device;            /* HIP device ordinal */
    int sample_rate;       /* input I/Q rate, quisk_sound_state.sample_rate (record_app `rate`, quisk.c:5070) */
    int playback_rate;     /* audio rate: 48000 x {1,2,4,8} (quisk.c:2535-2553) */
    int n_receivers;       /* number of independent receivers (reference: 3 banks); 0 = spectrum only */
    int max_block;         /* largest n accepted by qk_process_* (reference: SAMP_BUFFER_SIZE 66000, quisk.h:15) */
    int fft_size;          /* display FFT size (any, e.g. data_width * fft_mult, quisk.py:3858-3890; powers of two run the
                              direct transform) or 0 for no spectrum (record_app, quisk.c:5054) */
    int data_width;        /* graph points returned by qk_get_graph (record_app, quisk.c:5054) */
    int reserved[8];       /* [0] = largest receive-filter length accepted by qk_set_filters (0 = 1024);
                              [1] = 1: fp32 mode (BASELINE configs[4]; the reference computes in double);
                              [2] = short-block mode (one stream, fused chain launch, host-resident audio rows: for polls of a
                                    few thousand samples, the reference's own operating point): 0 = when max_block <= 4096,
                                    1 = on, 2 = off; rest zero */
} qk_config;

/* record_app (quisk.c:5048-5113): allocates the FFT buffers/window and initialises every filter state
 * (quisk_process_decimate(NULL), quisk_process_demodulate(NULL), quisk.c:5106-5107). */
int  qk_create(qk_engine **out, const qk_config *cfg);
void qk_destroy(qk_engine *e);
const char *qk_last_error(void);
const char *qk_version(void);

/* set_tune (quisk.c:4323-4329) for receiver 0 / set_multirx_freq (quisk.c:4438-4448) for the others:
 * tuning offset in integer Hz from the centre of the I/Q stream.  The running phasor is not reset
 * (rxTuneVector is a function static, quisk.c:2168). */
int qk_set_tune(qk_engine *e, int rx, int tune_hz);
/* set_rx_mode (quisk.c:4262-4269) / set_multirx_mode (quisk.c:4426-4436).  FDV_* and EXT need an external
 * codec / plug-in and return QK_EUNSUPPORTED. */
int qk_set_mode(qk_engine *e, int rx, int mode);
/* set_filters (quisk.c:4192-4235): n I taps and n Q taps of the receive band-pass, bandwidth and lower
 * edge offset in Hz.  n >= 10001 (MAX_FILTER_SIZE) is an error, as in the reference.  n == 0 = no filter. */
int qk_set_filters(qk_engine *e, int rx, const double *filt_i, const double *filt_q, int n,
                   int bandwidth, int start_offset);
/* set_agc (quisk.c:4184-4190): agcReleaseGain. */
int qk_set_agc(qk_engine *e, int rx, double release_gain);
/* agc_release_time (quisk.c:146; from the config file, quisk.c:3833).  Must be set before the first block. */
int qk_set_agc_release_time(qk_engine *e, double seconds);

/* Input conditioning of the full-rate stream, applied in front of the spectrum and the mixer, in the reference's
 * order.  qk_set_dc_remove: dc_remove_bw of DCremove (sound.c:165-230; set_sparams, sound.c:1253-1262): 0 = off (the
 * engine's default; Quisk's configuration default is 100), 1 = subtract the average of two-second stretches, > 1 = one-pole
 * DC blocker of that bandwidth in Hz.  qk_set_invert_spectrum: invert_spectrum (quisk.c:4174-4182, applied at
 * quisk.c:2315-2320): conjugate every sample. */
int qk_set_dc_remove(qk_engine *e, int bandwidth);
int qk_set_invert_spectrum(qk_engine *e, int on);
/* set_squelch (quisk.c:4348-4354): squelch_level, the FM squelch threshold in dB re full scale (-999 = never closed); an FM or
 * DGT-FM receiver is muted, behind its AGC (quisk.c:2590-2593), while the mean magnitude of its filtered samples, taken over
 * stretches of at least 2400 samples closed at block ends (quisk.c:1892-1893, 1911-1920), is below it.  qk_squelch_active:
 * MeasureSquelch[bank].squelch_active of the last block (1 = muted, 0 = open, < 0 error).  set_kill_audio (quisk.c:4364-4370):
 * replace every receiver's audio with silence (quisk.c:2585-2589).  The SSB squelch and the graph-based IsSquelch are not built. */
int qk_set_squelch(qk_engine *e, double level_db);
int qk_squelch_active(qk_engine *e, int rx);
int qk_set_kill_audio(qk_engine *e, int on);
/* set_ssb_squelch (quisk.c:4356-4362): ssb_squelch (quisk.c:1022-1116) on the detected audio of LSB / USB / AM receivers --
 * frames of 512 samples, Hann window, transform; the arithmetic over the geometric mean of the power in 300 Hz .. 300 Hz +
 * min(bandwidth, 3000) opens the squelch for one second when its logarithm exceeds level * 0.005; the audio is delayed by
 * one frame (d_delay, quisk.c:993-1020) and muted behind the AGC while the squelch is closed.  A receiver's first block
 * after creation is not analysed (the reference's first call only builds its plan).  qk_squelch_active reports it.
 * set_auto_notch (quisk.c:4237-4244; dAutoNotch, quisk.c:730-899) per receiver (the reference runs it on bank 0 only): the
 * audio passes an overlap-save filter of 2048-point frames that notches out the one or two strongest steady lines; on = 0 / 1
 * re-initialises its state.  qk_set_rit_freq: the CW sidetone offset that a CW receiver's notch leaves alone (set_sidetone's
 * rit_freq, quisk.c:4335-4343).  qk_is_squelch = IsSquelch (quisk.c:1231-1252; get_squelch, quisk.c:2645-2652): 1 when the mean
 * of the last released graph row over 5 kHz around freq is below the squelch level or off screen. */
int qk_set_ssb_squelch(qk_engine *e, int enabled, int level);
int qk_set_auto_notch(qk_engine *e, int rx, int on);
int qk_set_rit_freq(qk_engine *e, int hz);
int qk_is_squelch(qk_engine *e, int freq);
/* Second-receiver routing (quisk.c:2410-2494): the output of receiver rx becomes stereo, made of its own detected audio and that
 * of partner_rx, in the reference's order: the two streams' counts are equalised by Buffer2Chan's FIFOs at the decimated rate
 * (quisk.c:1452-1487, 2419, 2473), the pair goes through cFracDecim and the play-rate interpolators (quisk.c:2528-2553), then
 * Agc1 -- the receiver's own AGC, which carries on from mono -- runs on the left channel and Agc2 on the right
 * (quisk.c:2563-2572).
 *   method 1  left = rx, right = partner   (split_rxtx 1 / 2 with the receive side left; multirx_play_method 1)
 *   method 2  left = partner, right = rx   (split_rxtx 1 / 2 with the receive side right; multirx_play_method 2)
 *   method 3  both channels = partner      (split_rxtx 4; multirx_play_method 0)
 *   method 4  both channels = rx, through the stereo path and its two AGCs (split_rxtx 3)
 *   method 0  off: mono on both channels
 * Split RX/TX: the partner is a receiver on the same stream tuned to the transmit offset + RIT with receiver rx's mode and
 * filter (quisk.c:2410-2418); multirx play: the sub-receiver of set_multirx_play_channel.  A squelched or killed source is
 * silent in its own channel only (quisk.c:2585-2601).  The partner keeps its own (mono) output row.  While rx is in DGT-IQ,
 * which is stereo by itself, the route rests (quisk.c:2407-2409).  At most 4 routes per engine. */
int qk_set_stereo_route(qk_engine *e, int rx, int partner_rx, int method);
/* The key went up (quisk.c:2240-2241): the next 5 ms of every receiver's output are raised along the reference's ramp
 * (keyupEnvelope, quisk.c:2602-2612). */
int qk_key_up(qk_engine *e);
/* set_noise_blanker (quisk.c:4246-4252): quisk_noise_blanker, 0 = off, 1 / 2 / 3 = pulse limit 6 / 4 / 2.5 times the running
 * mean magnitude; NoiseBlanker (quisk.c:615-720) then runs behind the inversion (quisk.c:2322) and delays the stream by
 * 1.5 ms.  Switching it on starts from an empty delay line. */
int qk_set_noise_blanker(qk_engine *e, int level);

/* quisk_process_samples (quisk.h:258, quisk.c:2149-2616), for all receivers at once.
 * iq: n interleaved (re, im) doubles = C99 `complex double`, in HOST memory (copied in) ... */
int qk_process_samples(qk_engine *e, const double *iq_host, int n);
/* ... or already resident in DEVICE memory (no copy; the pointer must stay valid until qk_sync). */
int qk_process_samples_device(qk_engine *e, const double *iq_dev, int n);
/* add_rx_samples (quisk.c:2777-2839): n_bytes of interleaved I,Q signed integers of `bytes` (1..4)
 * per component, little (endian=0) or big endian, left-justified into int32 by the unpack kernel. */
int qk_process_samples_int(qk_engine *e, const void *raw_host, size_t n_bytes, int bytes, int big_endian);
int qk_sync(qk_engine *e);

/* Number of stereo output samples receiver rx produced in the last block (the return value of
 * quisk_process_samples), or a negative error. */
int qk_audio_count(qk_engine *e, int rx);
/* Copy the last block's audio of receiver rx (interleaved re=left, im=right doubles) to host memory;
 * returns the count copied.  cap is in complex samples. */
int qk_pull_audio(qk_engine *e, int rx, double *dst_host, int cap);
/* All receivers at once: dst is [n_receivers][stride] complex samples; counts[n_receivers]. */
int qk_pull_audio_all(qk_engine *e, double *dst_host, size_t stride, int *counts);
/* Device pointer and stride (in complex samples) of the audio output array, for zero-copy consumers
 * and for the RCCL gather.  With max_block <= 4096 (the short-block mode: one stream, one fused chain launch, see DESIGN.md
 * section 4) the rows live in pinned host memory that the device addresses through the same pointer. */
int qk_audio_device(qk_engine *e, const double **dev_ptr, size_t *stride);
/* Device-to-device copy of the last block's audio into a caller-owned device buffer
 * [n_receivers][dst_stride] complex samples (e.g. the send buffer of an RCCL gather). */
int qk_export_audio_device(qk_engine *e, double *dst_dev, size_t dst_stride);

/* get_graph (quisk.c:4602-4759) with the wall-clock gate (quisk.c:4712) left to the caller: returns the
 * number of FFT frames averaged since the last release (0 = no data, `out` untouched), writes
 * data_width dB values and the S-meter value (quisk.c:4728-4737), and clears the average. */
int qk_get_graph(qk_engine *e, double zoom, double deltaf, double *out, int data_width, double *smeter);
/* get_smeter (quisk.c:2737-2742): the value computed by the last qk_get_graph release. */
double qk_get_smeter(qk_engine *e);
/* First half of qk_get_graph, for a caller that keeps the GPU fed: queues the release of the frames averaged so far behind
 * them (binning, dB scaling, the S-meter sum; quisk.c:4713-4753), starts a new average and returns at once with the number of
 * frames marked (0: none waiting).  The next qk_get_graph call returns that row (its zoom / deltaf arguments are then not
 * used) and waits for nothing queued after the mark.  One row can be marked at a time (QK_EINVAL otherwise).  The reference
 * has the same split between its sound thread, which fills fft_data[] (quisk.c:2324-2348), and the GUI thread, which takes
 * what is ready (quisk.c:4651-4710). */
int qk_graph_mark(qk_engine *e, double zoom, double deltaf);

/* Frames waiting in the average (count_fft, quisk.c:4613) without releasing them. */
int qk_graph_pending(qk_engine *e);
/* Device pointer to the last released graph row (data_width doubles), for the RCCL gather. */
int qk_graph_device(qk_engine *e, const double **dev_ptr);

/* change_scan (quisk.c:3865-3872, "iidii") for hardware that sweeps its VFO: frames carry block numbers 0 .. scan_blocks-1 (the
 * next frame after this call is block 0, then cyclically, quisk.c:3672-3689) and get_graph stitches them (quisk.c:4682-4702):
 * of each frame the centre scan_valid part is kept, scan_blocks neighbouring bins summed into one, and ASSIGNED to the block's
 * slot of the average; the frame count advances once per completed scan.  The S-meter window follows quisk.c:4660-4663, which
 * reads the transmit tuning offset and vfo_screen (set_tune's second argument; change_rates): passed here.  scan_blocks = 0
 * returns to plain averaging. */
int qk_change_scan(qk_engine *e, int scan_blocks, int scan_sample_rate, double scan_valid, int scan_vfo0, int scan_deltaf,
                   int tx_tune_hz, int vfo_screen);

/* get_graph with use_fft == 0 (quisk.c:4618-4621, 4642-4649): the raw scope.  qk_set_graph_raw(1) stops the transforms and
 * keeps, of every completed frame, the first data_width samples in a queue of four (FFT_ARRAY_SIZE, quisk.c:66); a frame
 * that finds the queue full is dropped and counted in qk_stats.fft_error (quisk.c:2342-2345).  qk_get_graph_raw pops the
 * oldest: writes data_width interleaved (re, im) pairs and returns 1, or returns 0 when none is waiting.  Switching modes
 * clears the average (count_fft = 0). */
int qk_set_graph_raw(qk_engine *e, int on);
int qk_get_graph_raw(qk_engine *e, double *out, int data_width);

/* get_multirx_graph (quisk.c:4468-4530), the transform of one sub-receiver's frame: iq_host holds
 * 8 * data_width complex samples (multirx_fft_width, quisk.c:5098); periodic Hann window (4481-4482), forward FFT,
 * 8 -> 1 magnitude binning in order of frequency and dB scaling (4491-4519) into out[data_width].  Which
 * sub-receiver's turn it is and the refresh timer (quisk.c:3538-3548, 4485) stay with the caller. */
int qk_multirx_graph(qk_engine *e, const double *iq_host, int data_width, double *out);
/* Bandscope (Hermes ADC samples).  qk_bandscope_init = init_bandscope (quisk.c:2758-2775); size = 512 * blocks.
 * qk_add_bscope_samples: one block of `size` real samples already scaled to +-1 (read_rx_udp10, quisk.c:3411-3428);
 * _int: add_bscope_samples (quisk.c:2841-2900), signed integers of `bytes` bytes, scaled by 1 / CLIP32.  Each block
 * is windowed, transformed and its magnitudes added to the average at once (quisk.c:4545-4558).
 * qk_get_bandscope = the release half of get_bandscope (quisk.c:4559-4595) without its wall-clock gate: returns the
 * number of blocks averaged (0 = none, out untouched), writes graph_width dB values and the largest |sample| seen
 * (hermes_adc_level), and clears the average. */
int qk_bandscope_init(qk_engine *e, int size);
int qk_add_bscope_samples(qk_engine *e, const double *samples_host, int n);
int qk_add_bscope_samples_int(qk_engine *e, const void *raw_host, size_t n_bytes, int bytes, int big_endian);
int qk_get_bandscope(qk_engine *e, int clock, double zoom, double deltaf, double *out, int graph_width, double *adc_level);

/* add_tone (quisk.c:3022-3034) + AddTestTone (quisk.c:1194-1229): a -40 dBFS complex test tone at freq_hz from the centre of the
 * stream (AM- or FM-modulated with 1 kHz while receiver 0 is in AM / FM / DGT-FM) added to the full-rate stream in front of the
 * inversion, the noise blanker and the FFT staging; 0 = off.  The tone's vectors carry on from one setting to the next. */
int qk_add_tone(qk_engine *e, int freq_hz);
/* get_filter (quisk.c:4761-4848): the frequency response of receiver 0's receive band-pass (filter slot 0), data_width dB values,
 * negative frequencies first.  Needs an engine made with fft_size >= data_width (the response is shaped by the head of the
 * display window, as in the reference). */
int qk_get_filter(qk_engine *e, double *out, int data_width);
/* measure_audio (quisk.c:3011-3020; the sums at quisk.c:1787-1790, 1873-1876, 1901-1904, 1980-1985, closed at 2015-2018): time > 0
 * sets the averaging time in seconds (measure_audio_time, default 1); *value = measured_audio, the RMS of receiver 0's detected
 * audio over that time in millionths of full scale. */
int qk_measure_audio(qk_engine *e, int time, double *value);
/* measure_frequency (quisk.c:3000-3009; measure_freq, quisk.c:4850-4920, called from the path at quisk.c:2400-2401): mode >= 0 sets
 * measure_freq_mode (0 = off; otherwise receiver 0's decimated stream goes through three half-bands into 12000-point frames, and
 * every mode / 2 frames the strongest line within 500 Hz of the RIT offset is located by its three-bin centroid), mode < 0 only
 * reads; *value = measured_frequency in Hz from the centre of the stream. */
int qk_measure_frequency(qk_engine *e, int mode, double *value);
/* get_audio_graph (quisk.c:966-991; quisk_calc_audio_graph, quisk.c:903-964, fed from the LSB branch at quisk.c:1835): returns 1 and
 * writes data_width dB values (re full scale, negative frequencies first) when a graph of receiver 0's detected LSB audio is
 * ready, else 0.  The reference feeds it from every bank in LSB; here receiver 0 alone. */
int qk_get_audio_graph(qk_engine *e, double *out, int data_width);
/* The transform behind the module's dft / idft (Xdft, quisk.c:4922-5014): n complex points (interleaved, host memory) through the
 * engine's FFT, forward or FFTW's unnormalised backward transform; any n up to 2^22.  Rotation, window and 1 / N stay with the
 * caller. */
int qk_fft_any(qk_engine *e, const double *in_host, int n, int inverse, double *out_host);

/* get_filter_rate (quisk.c:2661-2735) and PlanDecimation (quisk.c:1509-1547): pure functions. */
int qk_get_filter_rate(int sample_rate, int mode, int bandwidth);
int qk_plan_decimation(int sample_rate, int *decim2, int *decim3, int *decim5);
/* quisk_decim_srate when quisk_process_decimate returns (quisk.c:1607-1720). */
int qk_decim_rate(int sample_rate);

/* get_state (quisk.c:2618-2643) keeps error counters; the engine's equivalents: */
typedef struct qk_stats {
    int64_t samples_in;        /* total I/Q samples consumed */
    int64_t blocks;            /* calls to qk_process_* */
    int64_t fft_frames;        /* display FFT frames transformed */
    int64_t fft_error;         /* frames dropped (fft_error, quisk.c:38); always 0: the GPU path never drops */
    /* HIP-event timing on the engine's stream (only while qk_set_timing is on): */
    double last_block_ms;      /* device time of the last timed block */
    double block_ms_sum;       /* sum over timed blocks */
    int64_t blocks_timed;
    double kernel_ms_sum[8];   /* per kernel group QK_KT_*: sum of launch durations */
    int64_t kernel_launches[8];
} qk_stats;
enum { QK_KT_FRONTEND = 0, QK_KT_DECIM = 1, QK_KT_DEMOD = 2, QK_KT_AUDIO = 3, QK_KT_AGC = 4, QK_KT_FFT = 5 };
int qk_get_stats(qk_engine *e, qk_stats *out);
/* Per-launch HIP-event timing of the kernel groups (off by default).  Events are recorded on the engine's
 * own streams and read back at qk_get_stats, so the timed region is not synchronised.  on: 0 = off, 1 = every
 * group and the whole block, otherwise a mask with bit (g + 1) set for each group QK_KT_g to time (bit 9 = whole
 * block): every timed launch costs two event records, which a benchmark may not want on every kernel. */
int qk_set_timing(qk_engine *e, int on);
int qk_reset_stats(qk_engine *e);

/* Stage taps for parity tests (debug): copy the planar output of an internal stream to the host.
 * which: QK_TAP_*; out: [planes][count] doubles; returns count per plane. */
enum { QK_TAP_FRONTEND = 0, QK_TAP_DECIM = 1, QK_TAP_DEMOD = 2, QK_TAP_AUDIO48 = 3, QK_TAP_PLAY = 4, QK_TAP_ROUTE = 5 };
int qk_debug_tap(qk_engine *e, int which, int rx, double *out_host, int cap);
/* Statistics (debug): speculation passes the AGC kernel has run for receiver rx since creation. */
long long qk_debug_agc_passes(qk_engine *e, int rx);

/* cRxFilterOut (quisk.c:1154-1192; exported, quisk.h:215; its caller outside quisk.c is freedv.c:233) in block form: the
 * receive band-pass of receiver rx (the taps of qk_set_filters: I taps on the real parts, Q taps on the imaginary parts,
 * the reference's tap order) over n complex samples in host memory, with a history of its own per bank (0..2), separate
 * from the receive chain's.  A receiver without a filter passes the samples.  Returns n. */
int qk_rx_filter_block(qk_engine *e, int rx, int bank, const double *in_host, double *out_host, int n);

/* filter.c drop-ins (filter.h:39-54) for callers that hold their samples in host memory.  The two names
 * BASELINE.json's north_star uses for them come from later Quisk releases and do not exist in this
 * snapshot ("two facts"); they are provided as thin aliases:
 *   quisk_cCplxSampleFilter : real-coefficient FIR over complex samples, decimate by `decim`
 *                             (quisk_cDecimate, filter.c:180-206; decim 1 = quisk_cFilter, filter.c:349)
 *   quisk_ccfFilter         : complex-coefficient FIR over complex samples, decimate by `decim`
 *                             (quisk_cCDecimate, filter.c:208-234; coefficients as quisk_filt_tune makes them)
 * State is explicit, like the reference's struct quisk_cFilter (filter.h:1-10): `hist` holds the last
 * ntaps-1 input samples (interleaved), `*phase` the decimation counter.  Samples are processed in place;
 * the return value is the output count. */
int quisk_cCplxSampleFilter(double *samples, int count, const double *coefs, int ntaps, int decim,
                            double *hist, int *phase);
int quisk_ccfFilter(double *samples, int count, const double *cpx_coefs, int ntaps, int decim,
                    double *hist, int *phase);

/* The remaining filter.c kernels as host-buffer drop-ins with explicit state (`hist` = the last ntaps - 1 input samples):
 *   qk_cInterpolate   quisk_cInterpolate (filter.c:108-142): complex samples, real taps, x interp in place (the buffer holds
 *                     count * interp complex samples); returns the output count, which like the reference's stops at
 *                     SAMP_BUFFER_SIZE * 8 / 10 = 52800 (filter.c:135)
 *   qk_dC_out_block   quisk_dC_out (filter.c:60-81) over a block: real samples through complex taps (as quisk_filt_tune makes
 *                     them, filter.c:35-58), one complex output per input sample
 *   qk_dD_out_block   quisk_dD_out (filter.c:303-322) over a block: real samples, real taps, in place
 * The reference calls the last two once per sample (microphone.c:373-717); a block is the sensible unit for a GPU. */
int qk_cInterpolate(double *samples, int count, const double *coefs, int ntaps, int interp, double *hist);
int qk_dC_out_block(const double *samples, int count, const double *cpx_coefs, int ntaps, double *hist, double *out_complex);
int qk_dD_out_block(double *samples, int count, const double *coefs, int ntaps, double *hist);

/* read_rx_udp10 (quisk.c:3341-3550), the Hermes / Metis wire format as the engine's input: n_packets receive packets of 1032
 * bytes (EF FE 01 06, sequence number, two 512-byte frames of sync 7F 7F 7F, C0..C4 and records of (1 + multirx_count) x
 * 24-bit Q, I samples + 2 microphone bytes).  The packets go to the GPU as they are; a kernel de-interleaves the first
 * receiver's samples (left-justified into int32, quisk.c:3516-3518) into the block that qk_process_samples_device then takes.
 * sub_out_host (optional): [multirx_count][samples] complex doubles of the sub-receivers' streams, for their graph frames
 * (quisk.c:3520-3527).  *read_errors counts bad sync bytes, *overrange the ADC overrange bits of C1 (quisk.c:3455-3460,
 * 3496-3499); sockets, sequence numbers and the transmit side stay with the caller.  Returns the number of samples processed. */
int qk_process_hermes(qk_engine *e, const void *packets, int n_packets, int multirx_count, double *sub_out_host, int *read_errors, int *overrange);

/* Multi-GPU (no counterpart in the reference, which is one process on one CPU; BASELINE.json north_star):
 * one process and one engine per GPU, the receivers sharded over the ranks, every rank fed the same I/Q stream; the only
 * exchange is the gather of the outputs to one rank, over RCCL (xGMI) from inside this library: ncclSend / ncclRecv in one
 * group, on the engine's own stream, straight out of the audio rows.  librccl is loaded on first use.
 *   qk_comm_unique_id  rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means
 *   qk_comm_init       every rank: ncclCommInitRank on the engine's device
 *   qk_gather_audio    the last block's audio of every rank's receivers to dst_rank: dst_dev (device memory on dst_rank,
 *                      ignored elsewhere) receives [world][n_receivers][stride] complex samples, stride as qk_audio_device
 *                      reports it, counts_dev [world][n_receivers] the sample counts.  Every rank has the same n_receivers.
 *   qk_gather_graph    the last released graph row (data_width doubles) of every rank to dst_rank: [world][data_width] */
#define QK_COMM_ID_BYTES 128
int qk_comm_unique_id(void *id_out);
int qk_comm_init(qk_engine *e, const void *id, int rank, int world);
int qk_comm_destroy(qk_engine *e);
int qk_gather_audio(qk_engine *e, int dst_rank, double *dst_dev, int *counts_dev);
int qk_gather_graph(qk_engine *e, int dst_rank, double *dst_dev);

/* No counterpart in the reference.  For callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  quisk_cCplxSampleFilter (above) with every buffer already in DEVICE memory:
 * samples (count complex, filtered in place), coefs (ntaps doubles), hist (ntaps - 1 complex, updated); *phase is a
 * host word.  The staging that an in-place FIR needs lives in a fixed array inside the library, QK_NOALLOC_MAX_SAMPLES
 * complex samples (8 MiB on every GPU the library runs on, held until the process ends): a call whose
 * (ntaps - 1 + count) + outputs exceeds it returns QK_ERANGE.  Calls no hipMalloc / hipFree or any allocating routine;
 * the three kernels are queued on `hip_stream` (a hipStream_t; NULL = the default stream) and nothing waits for them.
 * Every call on one GPU shares that GPU's staging array: calls that overlap in time on one GPU race; calls on one
 * stream do not.  The array is the one of the stream's device, which must be the current device (QK_EINVAL otherwise).  The
 * FIRST call on a device looks the array's address up (hipGetSymbolAddress loads the code object and materialises the
 * array): make one warm-up call per device -- count = 0 will do -- outside any graph capture, with that device current.
 * Returns the output count, like the reference's functions. */
#define QK_NOALLOC_MAX_SAMPLES (1 << 19)
int quisk_cCplxSampleFilter_noalloc(double *samples_dev, int count, const double *coefs_dev, int ntaps, int decim,
                                    double *hist_dev, int *phase, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
