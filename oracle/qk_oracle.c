/* qk_oracle.c -- CPU restatement of Quisk's receive-DSP hot path.  TEST INFRASTRUCTURE ONLY.
 * See qk_oracle.h for the pinning status of each group of functions.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: the reference is built by distutils with
 * plain x86-64 flags, so every product and sum is rounded separately).
 *
 * The arithmetic (operand order, summation order, where a value is rounded) follows the cited
 * reference lines exactly, because the filter kernels are checked bit for bit against the compiled
 * reference.  The code organisation (index-based rings, explicit re/im parts, one receiver object
 * instead of function statics) is this project's own.
 */
#include "qk_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define QK_COEF_TABLE(name, n) static const double name[n]
#include "../quisk_amd/csrc/qk_coefs.inc"

/* The 12 half-band constants: filter.c:359-362 (45-tap half band, taps 0 and 44 are zero). */
static const double HB45[12] = {
    0.000018566625444266, -0.000118469698701817, 0.000457318798253456, -0.001347840471412094,
    0.003321838571445455, -0.007198422696929033, 0.014211106939802483, -0.026424776824073383,
    0.048414810444971007, -0.096214669073304823, 0.314881034738348550, 0.500000000000000000 };

const double *qko_table(const char *name, int *ntaps)
{
#define T(nm) if (!strcmp(name, #nm)) { *ntaps = (int)(sizeof(nm) / sizeof(double)); return nm; }
    T(qk_h_dec2_98) T(qk_h_dec3_147) T(qk_h_dec5_245) T(qk_h_rs65_125) T(qk_h_sdriq53_55)
    T(qk_h_sdriq111_114) T(qk_h_sdriq133_136) T(qk_h_sdriq167_174) T(qk_h_sdriq185_189)
    T(qk_h_aud24p3_100) T(qk_h_aud24p4_50) T(qk_h_aud24p6_36) T(qk_h_fmhp_309) T(qk_h_dec16to8_62)
#undef T
    if (!strcmp(name, "hb45")) { *ntaps = 12; return HB45; }
    *ntaps = 0;
    return NULL;
}

/* ================================ filter.c kernels ================================== */

void qko_cfir_init(qko_cfir *f, const double *h, int ntaps)
{
    memset(f, 0, sizeof *f);
    f->h = h; f->ntaps = ntaps;
    f->ring = (qko_cd *)calloc((size_t)ntaps, sizeof(qko_cd));
}
void qko_dfir_init(qko_dfir *f, const double *h, int ntaps)
{
    memset(f, 0, sizeof *f);
    f->h = h; f->ntaps = ntaps;
    f->ring = (double *)calloc((size_t)ntaps, sizeof(double));
}
void qko_cfir_free(qko_cfir *f) { free(f->ring); free(f->hc); free(f->scratch); memset(f, 0, sizeof *f); }
void qko_dfir_free(qko_dfir *f) { free(f->ring); free(f->hc); free(f->scratch); memset(f, 0, sizeof *f); }

void qko_filt_tune(qko_dfir *f, double freq, int ssb_upper)
{   /* filter.c:35-58: c[i] = cexp(i 2 pi freq (i - D)) h[i]; lower sideband swaps re and im */
    int i;
    qko_cd coef, tune;
    double D;
    if (!f->hc) f->hc = (qko_cd *)malloc((size_t)f->ntaps * sizeof(qko_cd));
    tune = I * 2.0 * M_PI * freq;
    D = (f->ntaps - 1.0) / 2.0;
    for (i = 0; i < f->ntaps; i++) {
        coef = cexp(tune * (i - D)) * f->h[i];
        f->hc[i] = ssb_upper ? coef : cimag(coef) + I * creal(coef);
    }
}

/* y = sum_k h[k] ring[pos - k], accumulated from k = 0 upward (newest sample first) */
static inline qko_cd cdot_real(const qko_cfir *f, const double *h, int first, int step, int terms)
{
    double ar = 0, ai = 0;
    int j = f->pos, k;
    const double *ph = h + first;
    for (k = 0; k < terms; k++, ph += step) {
        ar += creal(f->ring[j]) * *ph;
        ai += cimag(f->ring[j]) * *ph;
        if (--j < 0) j = f->ntaps - 1;
    }
    return ar + I * ai;
}
static inline double ddot_real(const qko_dfir *f, int first, int step, int terms)
{
    double a = 0;
    int j = f->pos, k;
    const double *ph = f->h + first;
    for (k = 0; k < terms; k++, ph += step) {
        a += f->ring[j] * *ph;
        if (--j < 0) j = f->ntaps - 1;
    }
    return a;
}
#define RING_ADVANCE(f) do { if (++(f)->pos >= (f)->ntaps) (f)->pos = 0; } while (0)

void qko_dc_out_block(const double *x, int n, qko_dfir *f, qko_cd *out)
{   /* quisk_dC_out (filter.c:60-81), sample after sample: real samples through the complex taps of qko_filt_tune */
    int i, k, j;
    for (i = 0; i < n; i++) {
        qko_cd acc = 0;
        f->ring[f->pos] = x[i];
        j = f->pos;
        for (k = 0; k < f->ntaps; k++) {
            acc += f->ring[j] * f->hc[k];
            if (--j < 0) j = f->ntaps - 1;
        }
        RING_ADVANCE(f);
        out[i] = acc;
    }
}
void qko_dd_out_block(double *x, int n, qko_dfir *f)
{   /* quisk_dD_out (filter.c:303-322), sample after sample */
    int i;
    for (i = 0; i < n; i++) {
        f->ring[f->pos] = x[i];
        x[i] = ddot_real(f, 0, 1, f->ntaps);
        RING_ADVANCE(f);
    }
}
void qko_hermes_unpack(const unsigned char *buf, int multirx_count, qko_cd *samp, qko_cd *sub, long sub_stride, int *nsamples)
{   /* the record loop of read_rx_udp10 for one 1032-byte packet (quisk.c:3453, 3513-3531); samples are appended at *nsamples */
    int start, i, j, index, xr, xi, n = *nsamples;
    const int num_records = 504 / ((multirx_count + 1) * 6 + 2);
    for (start = 11; start < 1000; start += 512) {
        index = start + 5;
        for (i = 0; i < num_records; i++) {
            xi = (int)((unsigned)buf[index] << 24 | (unsigned)buf[index + 1] << 16 | (unsigned)buf[index + 2] << 8);
            xr = (int)((unsigned)buf[index + 3] << 24 | (unsigned)buf[index + 4] << 16 | (unsigned)buf[index + 5] << 8);
            samp[n] = xr + xi * I;
            index += 6;
            for (j = 0; j < multirx_count; j++) {
                xi = (int)((unsigned)buf[index] << 24 | (unsigned)buf[index + 1] << 16 | (unsigned)buf[index + 2] << 8);
                xr = (int)((unsigned)buf[index + 3] << 24 | (unsigned)buf[index + 4] << 16 | (unsigned)buf[index + 5] << 8);
                if (sub) sub[(long)j * sub_stride + n] = xr + xi * I;
                index += 6;
            }
            n++;
            index += 2;
        }
    }
    *nsamples = n;
}

int qko_cdecimate(qko_cd *x, int n, qko_cfir *f, int decim)
{
    int i, nout = 0;
    for (i = 0; i < n; i++) {
        f->ring[f->pos] = x[i];
        if (++f->decim_index >= decim) {
            f->decim_index = 0;
            x[nout++] = cdot_real(f, f->h, 0, 1, f->ntaps);
        }
        RING_ADVANCE(f);
    }
    return nout;
}

int qko_ccdecimate(qko_cd *x, int n, qko_cfir *f, int decim)
{   /* complex coefficients from qko_filt_tune (cast the filter to qko_dfir for tuning, as the reference does) */
    int i, k, j, nout = 0;
    for (i = 0; i < n; i++) {
        f->ring[f->pos] = x[i];
        if (++f->decim_index >= decim) {
            qko_cd acc = 0;
            f->decim_index = 0;
            j = f->pos;
            for (k = 0; k < f->ntaps; k++) {
                acc += f->ring[j] * f->hc[k];
                if (--j < 0) j = f->ntaps - 1;
            }
            x[nout++] = acc;
        }
        RING_ADVANCE(f);
    }
    return nout;
}

int qko_ddecimate(double *x, int n, qko_dfir *f, int decim)
{
    int i, nout = 0;
    for (i = 0; i < n; i++) {
        f->ring[f->pos] = x[i];
        if (++f->decim_index >= decim) {
            f->decim_index = 0;
            x[nout++] = ddot_real(f, 0, 1, f->ntaps);
        }
        RING_ADVANCE(f);
    }
    return nout;
}

int qko_dfilter(double *x, int n, qko_dfir *f)
{
    int i;
    for (i = 0; i < n; i++) {
        f->ring[f->pos] = x[i];
        x[i] = ddot_real(f, 0, 1, f->ntaps);
        RING_ADVANCE(f);
    }
    return n;
}

static qko_cd *c_scratch(qko_cd **buf, int *cap, const qko_cd *x, int n)
{   /* the interpolators read their input from a private copy because the output overwrites x */
    if (n > *cap) { free(*buf); *cap = n * 2; *buf = (qko_cd *)malloc((size_t)*cap * sizeof(qko_cd)); }
    memcpy(*buf, x, (size_t)n * sizeof(qko_cd));
    return *buf;
}
static double *d_scratch(double **buf, int *cap, const double *x, int n)
{
    if (n > *cap) { free(*buf); *cap = n * 2; *buf = (double *)malloc((size_t)*cap * sizeof(double)); }
    memcpy(*buf, x, (size_t)n * sizeof(double));
    return *buf;
}

int qko_cinterpolate(qko_cd *x, int n, qko_cfir *f, int interp)
{
    int i, j, nout = 0;
    const qko_cd *in = c_scratch(&f->scratch, &f->nscratch, x, n);
    for (i = 0; i < n; i++) {
        f->ring[f->pos] = in[i];
        for (j = 0; j < interp; j++) {
            qko_cd y = cdot_real(f, f->h, j, interp, f->ntaps / interp);
            if (nout < QKO_OUT_CAP) x[nout++] = y * interp;
        }
        RING_ADVANCE(f);
    }
    return nout;
}

int qko_dinterpolate(double *x, int n, qko_dfir *f, int interp)
{
    int i, j, nout = 0;
    const double *in = d_scratch(&f->scratch, &f->nscratch, x, n);
    for (i = 0; i < n; i++) {
        f->ring[f->pos] = in[i];
        for (j = 0; j < interp; j++) {
            double y = ddot_real(f, j, interp, f->ntaps / interp);
            if (nout < QKO_OUT_CAP) x[nout++] = y * interp;
        }
        RING_ADVANCE(f);
    }
    return nout;
}

int qko_cinterpdecim(qko_cd *x, int n, qko_cfir *f, int interp, int decim)
{
    int i, nout = 0;
    const qko_cd *in = c_scratch(&f->scratch, &f->nscratch, x, n);
    for (i = 0; i < n; i++) {
        f->ring[f->pos] = in[i];
        while (f->decim_index < interp) {
            qko_cd y = cdot_real(f, f->h, f->decim_index, interp, f->ntaps / interp);
            if (nout < QKO_OUT_CAP) x[nout++] = y * interp;
            f->decim_index += decim;
        }
        RING_ADVANCE(f);
        f->decim_index -= interp;
    }
    return nout;
}

int qko_cdecim2hb45(qko_cd *x, int n, qko_chb45 *f)
{   /* even-numbered inputs feed the centre-tap delay line, odd-numbered inputs the symmetric part */
    int i, k, nout = 0;
    for (i = 0; i < n; i++) {
        if (!f->toggle) {
            f->toggle = 1;
            memmove(f->center + 1, f->center, 10 * sizeof(qko_cd));
            f->center[0] = x[i];
        } else {
            qko_cd acc;
            f->toggle = 0;
            memmove(f->samples + 1, f->samples, 21 * sizeof(qko_cd));
            f->samples[0] = x[i];
            acc = (f->samples[0] + f->samples[21]) * HB45[0];
            for (k = 1; k < 11; k++)
                acc = acc + (f->samples[k] + f->samples[21 - k]) * HB45[k];
            x[nout++] = acc + f->center[10] * HB45[11];
        }
    }
    return nout;
}

int qko_dinterp2hb45(double *x, int n, qko_dhb45 *f)
{
    int i, k, nout = 0;
    const double *in = d_scratch(&f->scratch, &f->nscratch, x, n);
    for (i = 0; i < n; i++) {
        double acc = 0;
        memmove(f->samples + 1, f->samples, 21 * sizeof(double));
        f->samples[0] = in[i];
        if (nout > QKO_OUT_CAP) continue;       /* sic: '>' (filter.c:421) */
        x[nout++] = f->samples[11] * HB45[11] * 2;
        for (k = 0; k < 11; k++)
            acc += (f->samples[k] + f->samples[21 - k]) * HB45[k];
        x[nout++] = acc * 2;
    }
    return nout;
}

int qko_cinterp2hb45(qko_cd *x, int n, qko_chb45 *f)
{
    int i, k, nout = 0;
    const qko_cd *in = c_scratch(&f->scratch, &f->nscratch, x, n);
    for (i = 0; i < n; i++) {
        qko_cd acc = 0;
        memmove(f->samples + 1, f->samples, 21 * sizeof(qko_cd));
        f->samples[0] = in[i];
        if (nout > QKO_OUT_CAP) continue;
        x[nout++] = f->samples[11] * HB45[11] * 2;
        for (k = 0; k < 11; k++)
            acc += (f->samples[k] + f->samples[21 - k]) * HB45[k];
        x[nout++] = acc * 2;
    }
    return nout;
}

/* ================================ quisk.c: planning ================================= */

int qko_plan_decimation(int sample_rate, int *d2, int *d3, int *d5)
{   /* quisk.c:1509-1547: exhaustive search with truncating integer division */
    int best = sample_rate, b2 = 0, b3 = 0, b5 = 0, i2, i3, i5, i, r;
    for (i2 = 0; i2 <= 6; i2++)
        for (i3 = 0; i3 <= 3; i3++)
            for (i5 = 0; i5 <= 3; i5++) {
                r = sample_rate;
                for (i = 0; i < i2; i++) r /= 2;
                for (i = 0; i < i3; i++) r /= 3;
                for (i = 0; i < i5; i++) r /= 5;
                if (r >= 48000 && r < best) { b2 = i2; b3 = i3; b5 = i5; best = r; }
            }
    if (best >= 50000) best = best * 24 / 25;
    if (d2) { *d2 = b2; *d3 = b3; *d5 = b5; }
    return best;
}

static int rate_case(int sample_rate) { return (sample_rate + 100) / 1000; }

int qko_decim_rate(int sample_rate)
{   /* the value quisk_decim_srate holds when quisk_process_decimate returns (quisk.c:1607-1720) */
    int d2, d3, d5, r, i;
    switch (rate_case(sample_rate)) {
    case 41: return 48000;
    case 53: return sample_rate;
    case 111: case 133: return sample_rate / 2;
    case 185: return sample_rate / 3;
    case 370: return sample_rate / 6;
    case 740: return sample_rate / 12;
    case 1333: return sample_rate / 24;
    }
    qko_plan_decimation(sample_rate, &d2, &d3, &d5);
    r = sample_rate;
    for (i = 0; i < d2; i++) r /= 2;     /* (d2-1) half-bands plus the final /2 */
    for (i = 0; i < d3; i++) r /= 3;
    for (i = 0; i < d5; i++) r /= 5;
    if (r >= 50000) r = r * 24 / 25;
    return r;
}

int qko_get_filter_rate(int sample_rate, int mode, int bandwidth)
{   /* quisk.c:2661-2735, including the missing break after case 41 (falls into case 53) */
    int decim;
    switch (rate_case(sample_rate)) {
    case 41:
    case 53: decim = sample_rate; break;
    case 111: case 133: decim = sample_rate / 2; break;
    case 185: decim = sample_rate / 3; break;
    case 370: decim = sample_rate / 6; break;
    case 740: decim = sample_rate / 12; break;
    case 1333: decim = sample_rate / 24; break;
    default: decim = qko_plan_decimation(sample_rate, NULL, NULL, NULL); break;
    }
    switch (mode) {
    case QKO_CWL: case QKO_CWU: return decim / 8;
    case QKO_AM: return decim / 2;
    case QKO_FM: case QKO_DGT_FM: case QKO_DGT_IQ: return decim;
    case QKO_DGT_U: case QKO_DGT_L: return bandwidth < 3000 ? decim / 8 : decim;
    case QKO_FDV_U: case QKO_FDV_L: return 8000;
    default: return decim / 4;
    }
}

/* ================================ quisk.c: AGC, fractional decimator ================ */

void qko_agc_init(qko_agc *a, double max_out, int sample_rate)
{
    memset(a, 0, sizeof *a);
    a->max_out = max_out; a->sample_rate = sample_rate;
}
void qko_agc_free(qko_agc *a) { free(a->ring); a->ring = NULL; }

void qko_agc_process(qko_agc *a, qko_cd *x, int n, int is_cpx, int playback_rate,
                     double release_gain, double release_time, int fm_mode)
{
    int i;
    const double ceiling = a->max_out * QKO_CLIP32;
    if (!a->buf_size) {     /* first call: set up and return the block untouched (quisk.c:2033-2050) */
        if (a->sample_rate == 0) a->sample_rate = playback_rate;
        a->buf_size = a->sample_rate * 15 / 1000;
        a->index_read = a->index_start = a->is_clipping = 0;
        a->themax = 1.0; a->gain = 100; a->delta = 0; a->target_gain = 100;
        a->time_release = 1.0 - exp(-1.0 / a->sample_rate / release_time);
        a->ring = (qko_cd *)calloc((size_t)a->buf_size, sizeof(qko_cd));
        return;
    }
    for (i = 0; i < n; i++) {
        qko_cd in = x[i];
        double out_m, m;
        x[i] = a->ring[a->index_read] * a->gain;
        out_m = is_cpx ? cabs(x[i]) : fabs(creal(x[i]));
        if (out_m > QKO_CLIP32) x[i] /= out_m;             /* sic: scales to unit magnitude (quisk.c:2065) */
        a->ring[a->index_read] = in;
        m = is_cpx ? cabs(in) : fabs(creal(in));
        if (!a->is_clipping) {
            if (m * a->gain > ceiling) {                    /* start a linear ramp down */
                a->target_gain = ceiling / m;
                a->delta = (a->gain - a->target_gain) / a->buf_size;
                a->is_clipping = 1;
                a->themax = m;
                a->gain -= a->delta;
            } else if (a->index_read == a->index_start) {   /* one buffer length measured: new release target */
                double clip_gain = ceiling / a->themax;
                if (fm_mode) a->target_gain = clip_gain;
                else if (release_gain > clip_gain) a->target_gain = clip_gain;
                else a->target_gain = release_gain;
                a->themax = m;
                a->gain = a->gain * (1.0 - a->time_release) + a->target_gain * a->time_release;
            } else {
                if (a->themax < m) a->themax = m;
                a->gain = a->gain * (1.0 - a->time_release) + a->target_gain * a->time_release;
            }
        } else {
            if (m > a->themax) {
                double d;
                a->themax = m;
                a->target_gain = ceiling / m;
                d = (a->gain - a->target_gain) / a->buf_size;
                if (d > a->delta) a->delta = d;
            }
            a->gain -= a->delta;
            if (a->gain <= a->target_gain) {
                a->is_clipping = 0;
                a->gain = a->target_gain;
                a->themax = m;
                a->index_start = a->index_read;
            }
        }
        if (++a->index_read >= a->buf_size) a->index_read = 0;
    }
}

/* Test aid (no counterpart in the reference): process_agc sample by sample, recording after each sample the gain in
 * force for the NEXT sample and the discrete state (is_clipping, index_start), so that a test can say where two runs of
 * the state machine part company.  The first call of a fresh state only initialises, as qko_agc_process does. */
void qko_agc_trace(qko_agc *a, qko_cd *x, int n, int is_cpx, int playback_rate, double release_gain, double release_time,
                   int fm_mode, double *gain_out, int *clip_out, int *istart_out)
{
    int i;
    if (!a->buf_size) {
        qko_agc_process(a, x, n, is_cpx, playback_rate, release_gain, release_time, fm_mode);
        for (i = 0; i < n; i++) { gain_out[i] = a->gain; clip_out[i] = 0; istart_out[i] = 0; }
        return;
    }
    for (i = 0; i < n; i++) {
        qko_agc_process(a, x + i, 1, is_cpx, playback_rate, release_gain, release_time, fm_mode);
        gain_out[i] = a->gain; clip_out[i] = a->is_clipping; istart_out[i] = a->index_start;
    }
}

void qko_fracdecim_init(qko_fracstate *s) { memset(s, 0, sizeof *s); s->dindex = 1; }

int qko_fracdecim(qko_fracstate *s, qko_cd *x, int n, double fdecim)
{   /* quisk.c:570-613: 4-point Lagrange interpolation at fractional position dindex */
    int i, nout = 0;
    for (i = 0; i < n; i++) {
        s->c3 = x[i];
        if (s->dindex < 2) {
            double xm0 = s->dindex - 0, xm1 = s->dindex - 1, xm2 = s->dindex - 2, xm3 = s->dindex - 3;
            x[nout++] = (xm1 * xm2 * xm3 * s->c0 / -6.0 + xm0 * xm2 * xm3 * s->c1 / 2.0 +
                         xm0 * xm1 * xm3 * s->c2 / -2.0 + xm0 * xm1 * xm2 * s->c3 / 6.0);
            s->dindex += fdecim - 1;
        } else {
            s->dindex -= 1;
        }
        s->c0 = s->c1; s->c1 = s->c2; s->c2 = s->c3;
    }
    return nout;
}

/* ================================ quisk.c: one receiver ============================= */

struct qko_rx {
    int sample_rate, playback_rate, mode, tune;
    double release_gain, release_time;
    /* FM squelch (MeasureSquelch, quisk.c:203-214; squelch_level 147) and kill_audio (quisk.c:103) */
    double squelch_level, sq_rf_sum, sq_value; int sq_rf_count, sq_active, kill_audio;
    qko_cd tune_vec;                                     /* rxTuneVector, quisk.c:2168 */
    /* quisk_process_decimate Storage, quisk.c:1554-1574 */
    int planned_rate, d2, d3, d5;
    qko_chb45 hb[5];
    qko_cfir sdriq111, sdriq53, sdriq133, sdriq167, sdriq185, dec3[3], dec5, dec5b, dec5s, dec2last, rs65;
    qko_cfir rs45; int own_rs45;                         /* a resampler filter of its own instead of the shared filtDecim5S (test aid, see qko_rx_set_own_resampler) */
    int decim_srate, filter_srate;
    /* quisk_process_demodulate Storage, quisk.c:1730-1747 */
    qko_cd fm_1; double dc_remove, fm_a0, fm_a1, fm_b1, fm_x1, fm_y1;
    qko_chb45 dem_hb4, dem_hb5; qko_dhb45 dem_hb6, dem_hb7;
    qko_dfir aud24p3, aud24p4, aud12p2, aud24p6, fmhp;
    qko_cfir dem_dec2;
    /* receive band-pass: cFilterI/Q (quisk.c:94-96) and the two Storage rings (1124-1127, 1161-1165) */
    double *fi, *fq; int nfilt, bandwidth;
    int idx_c; double *ring_i, *ring_q;
    int idx_d; qko_cd *ring_c;
    qko_fracstate frac;
    /* SSB squelch (quisk.c:1022-1116; ssb_squelch_enabled / _level 148-149) with its delay line d_delay (993-1020) */
    int ssb_sq_enabled, ssb_sq_level, ssb_plan, ssb_index, ssb_open; double ssb_in[512], ssb_delay[512]; int ssb_didx, ssb_dalloc;
    /* auto-notch (dAutoNotch, quisk.c:730-899): its function statics */
    int notch_on, rit_freq; struct qko_notch *notch;
    qko_chb45 play_hb[3];                                /* HalfBand7..9, quisk.c:2178-2180 */
    qko_agc agc;                                         /* Agc1, quisk.c:2181 */
    double *audio; int naudio;
    qko_agraph *agraph;                                  /* quisk_calc_audio_graph's statics, when a test wants the audio graph */
    double ma_sum, measured_audio; long ma_count; int ma_time;   /* measure_audio (quisk.c:196-201, 1787-1790, 2015-2018) */
};

#define NOTCH_N 2048                 /* NOTCH_DATA_SIZE and the sizes derived from it, quisk.c:723-729 */
#define NOTCH_DESIGN (NOTCH_N / 4)
#define NOTCH_TAPS (NOTCH_DESIGN - 1)
#define NOTCH_FBINS (NOTCH_TAPS / 2 + 1)
#define NOTCH_START (NOTCH_TAPS - 1)
#define NOTCH_OUT (NOTCH_N - NOTCH_START)
#define NOTCH_BINS (NOTCH_N / 2 + 1)
struct qko_notch {
    int old1, count1, old2, count2, index, sig;
    double data_in[NOTCH_N], data_out[NOTCH_N], window[NOTCH_N], average[NOTCH_BINS], fltr_out[NOTCH_DESIGN];
    qko_cd fltr_fft[NOTCH_BINS];
};
static void notch_reset(struct qko_notch *s)
{   /* dAutoNotch(NULL, ...), quisk.c:761-770; the window is made with the plans (757-759) */
    int i;
    memset(s, 0, sizeof *s);
    s->index = NOTCH_START; s->sig = -1; s->count1 = s->count2 = -4;
    for (i = 0; i < NOTCH_TAPS; i++) s->window[i] = 0.50 - 0.50 * cos(2. * M_PI * i / (NOTCH_TAPS));
}
/* FFTW's r2c / c2r transforms through the complex DFT: r2c keeps bins 0 .. n/2; c2r (unnormalised) takes bins 0 .. n/2 of a
 * Hermitian spectrum and ignores the imaginary parts of bins 0 and n/2 */
static void r2c(const double *in, qko_cd *out, int n, qko_cd *work)
{
    int i;
    for (i = 0; i < n; i++) work[i] = in[i];
    qko_fft(work, n);
    for (i = 0; i <= n / 2; i++) out[i] = work[i];
}
static void c2r(const qko_cd *in, double *out, int n, qko_cd *work)
{
    int i;
    work[0] = creal(in[0]); work[n / 2] = creal(in[n / 2]);
    for (i = 1; i < n / 2; i++) { work[i] = conj(in[i]); work[n - i] = in[i]; }     /* inverse = conj(FFT(conj X)); the result is real */
    qko_fft(work, n);
    for (i = 0; i < n; i++) out[i] = creal(work[i]);
}
static void auto_notch(struct qko_notch *s, double *dsamples, int nSamples, int sidetone, int rate)
{   /* dAutoNotch, quisk.c:775-897 */
    int i, j, k, i1, i2, inp, signal, delta_sig, delta_i1, half_width;
    double d, d1, d2;
    static qko_cd work[NOTCH_N], spec[NOTCH_BINS], fdesign[NOTCH_BINS];
    static double fltr_in[NOTCH_N];
    for (inp = 0; inp < nSamples; inp++) {
        s->data_in[s->index] = dsamples[inp];
        dsamples[inp] = s->data_out[s->index];
        if (++s->index >= NOTCH_N) {
            s->index = NOTCH_START;
            r2c(s->data_in, spec, NOTCH_N, work);
            delta_sig = (300 * 2 * NOTCH_BINS + rate / 2) / rate;
            delta_i1 = (400 * 2 * NOTCH_BINS + rate / 2) / rate;
            signal = sidetone != 0 ? (abs(sidetone) * 2 * NOTCH_BINS + rate / 2) / rate : -999;
            d1 = 0; i1 = 0;
            for (i = 0; i < NOTCH_BINS; i++) {
                d = cabs(spec[i]);
                s->average[i] = 0.5 * s->average[i] + 0.5 * d;
                if (abs(i - signal) > delta_sig && s->average[i] > d1) { d1 = s->average[i]; i1 = i; }
            }
            if (abs(i1 - s->old1) < 3) s->count1++; else s->count1--;
            if (s->count1 > 4) s->count1 = 4; else if (s->count1 < -1) s->count1 = -1;
            if (s->count1 < 0) s->old1 = i1;
            d2 = 0; i2 = 0;
            for (i = 0; i < NOTCH_BINS; i++)
                if (abs(i - signal) > delta_sig && abs(i - i1) > delta_i1 && s->average[i] > d2) { d2 = s->average[i]; i2 = i; }
            if (abs(i2 - s->old2) < 3) s->count2++; else s->count2--;
            if (s->count2 > 4) s->count2 = 4; else if (s->count2 < -2) s->count2 = -2;
            if (s->count2 < 0) s->old2 = i2;
            if (s->count1 > 0 && s->count2 > 0) k = i1 + 10000 * i2;
            else if (s->count1 > 0) k = i1;
            else k = 0;
            if (s->sig != k) {                           /* make the filter: quisk.c:844-886 */
                s->sig = k;
                half_width = (100 * 2 * NOTCH_FBINS + rate / 2) / rate;
                if (half_width < 3) half_width = 3;
                /* the design array IS fltr_fft (quisk.c:849-850): bins 0 .. 255 are set; bin 256, the Nyquist bin of the
                   512-point inverse transform, keeps what the last 2048-point filter spectrum left there (0 at first) */
                for (i = 0; i < NOTCH_BINS; i++) fdesign[i] = s->fltr_fft[i];
                for (i = 0; i < NOTCH_FBINS; i++) fdesign[i] = 1.0;
                k = (i1 + 2) / 4;
                if (s->count1 > 0)
                    for (i = -half_width; i <= half_width; i++) { j = k + i; if (j >= 0 && j < NOTCH_FBINS) fdesign[j] = 0.0; }
                k = (i2 + 2) / 4;
                if (s->count1 > 0 && s->count2 > 0)
                    for (i = -half_width; i <= half_width; i++) { j = k + i; if (j >= 0 && j < NOTCH_FBINS) fdesign[j] = 0.0; }
                c2r(fdesign, s->fltr_out, NOTCH_DESIGN, work);         /* bins 0 .. NOTCH_DESIGN / 2 = NOTCH_FBINS - 1 + ... */
                memmove(s->fltr_out + NOTCH_DESIGN / 2 - 1, s->fltr_out, sizeof(double) * (NOTCH_TAPS / 2 - 1));
                for (i = NOTCH_DESIGN / 2 - 2, j = NOTCH_DESIGN / 2; i >= 0; i--, j++) s->fltr_out[i] = s->fltr_out[j];
                for (i = 0; i < NOTCH_TAPS; i++) fltr_in[i] = s->fltr_out[i] * s->window[i] / NOTCH_DESIGN;
                for (i = NOTCH_TAPS; i < NOTCH_N; i++) fltr_in[i] = 0.0;
                r2c(fltr_in, s->fltr_fft, NOTCH_N, work);
            }
            for (i = 0; i < NOTCH_BINS; i++) spec[i] *= s->fltr_fft[i];
            c2r(spec, s->data_out, NOTCH_N, work);
            memmove(s->data_in, s->data_in + NOTCH_OUT, NOTCH_START * sizeof(double));
            for (i = NOTCH_START; i < NOTCH_N; i++) s->data_out[i] /= NOTCH_N / 20;
        }
    }
}

static void ssb_squelch(qko_rx *rx, double *dsamples, int nSamples, int samp_rate)
{   /* quisk.c:1022-1116; the caller's first call only builds the plan */
    int i, bw, bw1, bw2, inp;
    double d, arith_avg, geom_avg, ratio;
    static qko_cd work[512], out[257];
    double win[512];
    if (!rx->ssb_plan) { rx->ssb_plan = 1; rx->ssb_index = 0; rx->ssb_open = 0; return; }
    for (inp = 0; inp < nSamples; inp++) {
        rx->ssb_in[rx->ssb_index++] = dsamples[inp];
        if (rx->ssb_index >= 512) {
            rx->ssb_index = 0;
            for (i = 0; i < 512; i++) win[i] = rx->ssb_in[i] * (0.50 - 0.50 * cos(2. * M_PI * i / 512));
            r2c(win, out, 512, work);
            bw = rx->bandwidth; if (bw > 3000) bw = 3000;
            bw1 = 300 * 512 / samp_rate;
            bw2 = (bw + 300) * 512 / samp_rate;
            arith_avg = 0.0; geom_avg = 0.0;
            for (i = bw1; i < bw2; i++) {
                qko_cd c = out[i] / 32767;
                d = creal(c) * creal(c) + cimag(c) * cimag(c);
                if (d > 1E-4) { arith_avg += d; geom_avg += log(d); }
            }
            if (arith_avg > 1E-4) { bw = bw2 - bw1; arith_avg = log(arith_avg / bw); geom_avg /= bw; ratio = arith_avg - geom_avg; }
            else ratio = 1.0;
            if (ratio > rx->ssb_sq_level * 0.005) rx->ssb_open = samp_rate;
        }
    }
    rx->ssb_open -= nSamples;
    if (rx->ssb_open < 0) rx->ssb_open = 0;
    rx->sq_active = rx->ssb_open == 0;
}
static void ssb_delay(qko_rx *rx, double *dsamples, int nSamples)
{   /* d_delay by SQUELCH_FFT_SIZE samples, quisk.c:993-1020 */
    int i;
    for (i = 0; i < nSamples; i++) {
        double v = rx->ssb_delay[rx->ssb_didx];
        rx->ssb_delay[rx->ssb_didx] = dsamples[i];
        dsamples[i] = v;
        if (++rx->ssb_didx >= 512) rx->ssb_didx = 0;
    }
}
static void audio_hooks(qko_rx *rx, double *a, int n, int sidetone, int squelch)
{   /* the calls between the detector and the audio interpolators, quisk.c:1792-1793, 1829-1834, 1879-1884 */
    if (rx->notch_on && rx->notch) auto_notch(rx->notch, a, n, sidetone, rx->filter_srate);
    if (squelch && rx->ssb_sq_enabled) { ssb_squelch(rx, a, n, rx->filter_srate); ssb_delay(rx, a, n); }
}
void qko_rx_set_ssb_squelch(qko_rx *rx, int enabled, int level) { rx->ssb_sq_enabled = enabled; rx->ssb_sq_level = level; }   /* quisk.c:4356-4362 */
void qko_rx_set_auto_notch(qko_rx *rx, int on, int rit_freq)
{   /* set_auto_notch, quisk.c:4237-4244: also re-initialises */
    if (!rx->notch) rx->notch = (struct qko_notch *)malloc(sizeof(struct qko_notch));
    notch_reset(rx->notch);
    rx->notch_on = on; rx->rit_freq = rit_freq;
}

qko_rx *qko_rx_new(int sample_rate, int playback_rate)
{
    int n, i;
    const double *h;
    qko_rx *rx = (qko_rx *)calloc(1, sizeof *rx);
    rx->sample_rate = sample_rate; rx->playback_rate = playback_rate;
    rx->mode = QKO_USB; rx->tune_vec = 1;
    rx->release_gain = 80; rx->release_time = 1.0;       /* quisk.c:145-146 */
    rx->squelch_level = -999.0;                          /* quisk.c:147 */
    rx->ma_time = 1;                                     /* quisk.c:201 */
#define CI(f, nm) do { h = qko_table(#nm, &n); qko_cfir_init(&rx->f, h, n); } while (0)
#define DI(f, nm) do { h = qko_table(#nm, &n); qko_dfir_init(&rx->f, h, n); } while (0)
    CI(sdriq111, qk_h_sdriq111_114); CI(sdriq53, qk_h_sdriq53_55); CI(sdriq133, qk_h_sdriq133_136);
    CI(sdriq167, qk_h_sdriq167_174); CI(sdriq185, qk_h_sdriq185_189);
    for (i = 0; i < 3; i++) CI(dec3[i], qk_h_dec3_147);
    CI(dec5, qk_h_dec5_245); CI(dec5b, qk_h_dec5_245); CI(dec5s, qk_h_dec5_245); CI(rs45, qk_h_dec5_245);
    CI(dec2last, qk_h_dec2_98); CI(rs65, qk_h_rs65_125);
    DI(aud24p3, qk_h_aud24p3_100); DI(aud24p4, qk_h_aud24p4_50); DI(aud12p2, qk_h_aud24p4_50);
    DI(aud24p6, qk_h_aud24p6_36); DI(fmhp, qk_h_fmhp_309);
    CI(dem_dec2, qk_h_dec2_98);
#undef CI
#undef DI
    rx->fm_1 = 10;                                       /* quisk.c:1763 */
    {   /* de-emphasis coefficients, quisk.c:1764-1768 */
        double www = tan(M_PI * 300.0 / 48000), nnn = 1.0 / (1.0 + www);
        rx->fm_a0 = www * nnn; rx->fm_a1 = rx->fm_a0; rx->fm_b1 = nnn * (www - 1.0);
    }
    rx->fi = (double *)calloc(QKO_MAX_FILTER, sizeof(double));
    rx->fq = (double *)calloc(QKO_MAX_FILTER, sizeof(double));
    rx->ring_i = (double *)calloc(QKO_MAX_FILTER, sizeof(double));
    rx->ring_q = (double *)calloc(QKO_MAX_FILTER, sizeof(double));
    rx->ring_c = (qko_cd *)calloc(QKO_MAX_FILTER, sizeof(qko_cd));
    qko_fracdecim_init(&rx->frac);
    qko_agc_init(&rx->agc, 0.7, 0);
    rx->decim_srate = qko_decim_rate(sample_rate);
    rx->filter_srate = 48000;                            /* quisk.c:101 */
    return rx;
}

void qko_rx_free(qko_rx *rx)
{
    int i;
    if (!rx) return;
    qko_cfir_free(&rx->sdriq111); qko_cfir_free(&rx->sdriq53); qko_cfir_free(&rx->sdriq133);
    qko_cfir_free(&rx->sdriq167); qko_cfir_free(&rx->sdriq185);
    for (i = 0; i < 3; i++) qko_cfir_free(&rx->dec3[i]);
    qko_cfir_free(&rx->dec5); qko_cfir_free(&rx->dec5b); qko_cfir_free(&rx->dec5s); qko_cfir_free(&rx->rs45);
    qko_cfir_free(&rx->dec2last); qko_cfir_free(&rx->rs65); qko_cfir_free(&rx->dem_dec2);
    qko_dfir_free(&rx->aud24p3); qko_dfir_free(&rx->aud24p4); qko_dfir_free(&rx->aud12p2);
    qko_dfir_free(&rx->aud24p6); qko_dfir_free(&rx->fmhp);
    for (i = 0; i < 5; i++) free(rx->hb[i].scratch);
    free(rx->dem_hb4.scratch); free(rx->dem_hb5.scratch); free(rx->dem_hb6.scratch); free(rx->dem_hb7.scratch);
    for (i = 0; i < 3; i++) free(rx->play_hb[i].scratch);
    free(rx->fi); free(rx->fq); free(rx->ring_i); free(rx->ring_q); free(rx->ring_c);
    qko_agc_free(&rx->agc);
    free(rx->audio); free(rx->notch);
    free(rx);
}

/* The reference runs the resampler's second stage on filtDecim5S, the state object of the third /5 stage (quisk.c:1713): at
 * rates whose plan has three /5 stages the two filters then overwrite each other's history and counters.  on = 1 gives the
 * resampler a state of its own (what the plan means; the GPU engine's stages never share state). */
void qko_rx_set_own_resampler(qko_rx *rx, int on) { rx->own_rs45 = on; }
void qko_rx_set_squelch(qko_rx *rx, double level_db) { rx->squelch_level = level_db; }       /* set_squelch, quisk.c:4348 */
void qko_rx_set_kill_audio(qko_rx *rx, int on) { rx->kill_audio = on; }                        /* set_kill_audio, quisk.c:4364 */
int  qko_rx_squelch_active(const qko_rx *rx) { return rx->sq_active; }
void qko_rx_set_tune(qko_rx *rx, int tune_hz) { rx->tune = tune_hz; }
void qko_rx_set_mode(qko_rx *rx, int mode) { rx->mode = mode; }
void qko_rx_set_agc(qko_rx *rx, double release_gain, double release_time)
{ rx->release_gain = release_gain; rx->release_time = release_time; }
int qko_rx_decim_srate(const qko_rx *rx) { return rx->decim_srate; }
int qko_rx_filter_srate(const qko_rx *rx) { return rx->filter_srate; }

int qko_rx_set_filter(qko_rx *rx, const double *fi, const double *fq, int n, int bandwidth)
{
    if (n >= QKO_MAX_FILTER || n < 0) return -1;         /* quisk.c:4216 */
    memcpy(rx->fi, fi, (size_t)n * sizeof(double));
    memcpy(rx->fq, fq, (size_t)n * sizeof(double));
    rx->nfilt = n; rx->bandwidth = bandwidth;
    return 0;
}

void qko_rx_mix(qko_rx *rx, qko_cd *x, int n)
{   /* quisk.c:2350-2361; the vector is neither renormalised nor reset on a frequency change */
    int i;
    double tune = rx->tune;
    qko_cd phase;
    if (tune == 0) return;
    phase = cexp((I * -2.0 * M_PI * tune) / rx->sample_rate);
    for (i = 0; i < n; i++) {
        x[i] *= rx->tune_vec;
        rx->tune_vec *= phase;
    }
}

int qko_rx_decimate(qko_rx *rx, qko_cd *x, int n)
{   /* quisk.c:1600-1721 */
    int i2, i3, i5, rate = rx->sample_rate;
    if (rx->planned_rate != rate) {
        rx->planned_rate = rate;
        qko_plan_decimation(rate, &rx->d2, &rx->d3, &rx->d5);
    }
    switch (rate_case(rate)) {
    case 41: rx->decim_srate = 48000; return n;
    case 53: rx->decim_srate = rate; return qko_cdecimate(x, n, &rx->sdriq53, 1);
    case 111: rx->decim_srate = rate / 2; return qko_cdecimate(x, n, &rx->sdriq111, 2);
    case 133: rx->decim_srate = rate / 2; return qko_cdecimate(x, n, &rx->sdriq133, 2);
    case 185: rx->decim_srate = rate / 3; return qko_cdecimate(x, n, &rx->sdriq185, 3);
    case 370:
        rx->decim_srate = rate / 6;
        n = qko_cdecim2hb45(x, n, &rx->hb[1]);
        return qko_cdecimate(x, n, &rx->sdriq185, 3);
    case 740:
        rx->decim_srate = rate / 12;
        n = qko_cdecim2hb45(x, n, &rx->hb[1]);
        n = qko_cdecim2hb45(x, n, &rx->hb[2]);
        return qko_cdecimate(x, n, &rx->sdriq185, 3);
    case 1333:
        rx->decim_srate = rate / 24;
        n = qko_cdecim2hb45(x, n, &rx->hb[0]);
        n = qko_cdecim2hb45(x, n, &rx->hb[1]);
        n = qko_cdecim2hb45(x, n, &rx->hb[2]);
        return qko_cdecimate(x, n, &rx->sdriq167, 3);
    }
    rx->decim_srate = rate;
    for (i2 = 0; i2 < rx->d2 - 1 && i2 < 5; i2++) {      /* all but the last /2 are half-bands */
        n = qko_cdecim2hb45(x, n, &rx->hb[i2]);
        rx->decim_srate /= 2;
    }
    for (i3 = 0; i3 < rx->d3; i3++) {
        n = qko_cdecimate(x, n, &rx->dec3[i3], 3);
        rx->decim_srate /= 3;
    }
    for (i5 = 0; i5 < rx->d5; i5++) {
        n = qko_cdecimate(x, n, i5 == 0 ? &rx->dec5 : i5 == 1 ? &rx->dec5b : &rx->dec5s, 5);
        rx->decim_srate /= 5;
    }
    if (rx->d2 > 0) {
        n = qko_cdecimate(x, n, &rx->dec2last, 2);
        rx->decim_srate /= 2;
    }
    if (rx->decim_srate >= 50000) {                      /* 24/25 resampler; the second stage shares dec5s (quisk.c:1713) */
        rx->decim_srate = rx->decim_srate * 24 / 25;
        n = qko_cinterpdecim(x, n, &rx->rs65, 6, 5);
        n = qko_cinterpdecim(x, n, rx->own_rs45 ? &rx->rs45 : &rx->dec5s, 4, 5);
    }
    return n;
}

static qko_cd rx_bandpass_iq(qko_rx *rx, qko_cd s)
{   /* cRxFilterOut, quisk.c:1154-1192: I taps on the real part, Q taps on the imaginary part;
     * the newest sample meets tap 0 and the index then runs forward through the ring */
    double ai = 0, aq = 0;
    int j, k, N = rx->nfilt;
    if (!N) return s;
    if (rx->idx_c >= N) rx->idx_c = 0;
    rx->ring_i[rx->idx_c] = creal(s);
    rx->ring_q[rx->idx_c] = cimag(s);
    j = rx->idx_c;
    for (k = 0; k < N; k++) {
        ai += rx->ring_i[j] * rx->fi[k];
        aq += rx->ring_q[j] * rx->fq[k];
        if (++j >= N) j = 0;
    }
    rx->idx_c++;
    return ai + I * aq;
}

static qko_cd rx_bandpass_real(qko_rx *rx, qko_cd s)
{   /* dRxFilterOut, quisk.c:1118-1152: I taps on the complex sample */
    double ar = 0, ai = 0;
    int j, k, N = rx->nfilt;
    if (!N) return s;
    if (rx->idx_d >= N) rx->idx_d = 0;
    rx->ring_c[rx->idx_d] = s;
    j = rx->idx_d;
    for (k = 0; k < N; k++) {
        ar += creal(rx->ring_c[j]) * rx->fi[k];
        ai += cimag(rx->ring_c[j]) * rx->fi[k];
        if (++j >= N) j = 0;
    }
    rx->idx_d++;
    return ar + I * ai;
}

void qko_rx_bandpass_block(qko_rx *rx, qko_cd *x, int n)
{   /* cRxFilterOut (quisk.c:1154-1192) sample after sample, as freedv.c:233 calls it */
    int i;
    for (i = 0; i < n; i++) x[i] = rx_bandpass_iq(rx, x[i]);
}

int qko_rx_demodulate(qko_rx *rx, qko_cd *x, double *a, int n)
{   /* quisk.c:1778-2014 for the modes that need no codec: CW, SSB, AM, FM, DGT-U/L/IQ/FM.
     * Off-by-default side branches (auto-notch, SSB squelch, audio measurement) are left out. */
    int i, narrow;
    qko_cd c, p;
    double m, d, di;
    rx->sq_active = 0;                                   /* quisk.c:1777 */
    switch (rx->mode) {
    case QKO_CWL: case QKO_CWU:
        rx->filter_srate = rx->decim_srate / 8;
        n = qko_cdecim2hb45(x, n, &rx->dem_hb5);
        n = qko_cdecim2hb45(x, n, &rx->dem_hb4);
        n = qko_cdecimate(x, n, &rx->dem_dec2, 2);
        for (i = 0; i < n; i++) {
            c = rx_bandpass_iq(rx, x[i]);
            a[i] = rx->mode == QKO_CWL ? creal(c) + cimag(c) : creal(c) - cimag(c);
            rx->ma_sum += a[i] * a[i]; rx->ma_count += 1;
        }
        audio_hooks(rx, a, n, rx->rit_freq, 0);
        n = qko_dinterpolate(a, n, &rx->aud12p2, 2);
        n = qko_dinterp2hb45(a, n, &rx->dem_hb6);
        n = qko_dinterp2hb45(a, n, &rx->dem_hb7);
        break;
    case QKO_AM:
        rx->filter_srate = rx->decim_srate / 2;
        n = qko_cdecimate(x, n, &rx->dem_dec2, 2);
        for (i = 0; i < n; i++) {
            c = rx_bandpass_real(rx, x[i]);
            m = cabs(c);
            d = m + rx->dc_remove * 0.99;                /* DC removal, quisk.c:1869-1871 */
            a[i] = d - rx->dc_remove;
            rx->dc_remove = d;
            rx->ma_sum += a[i] * a[i]; rx->ma_count += 1;
        }
        n = qko_dfilter(a, n, &rx->aud24p6);
        audio_hooks(rx, a, n, 0, 1);
        n = qko_dinterp2hb45(a, n, &rx->dem_hb7);
        break;
    case QKO_FM: case QKO_DGT_FM:
        rx->filter_srate = rx->decim_srate;
        for (i = 0; i < n; i++) {
            c = rx_bandpass_real(rx, x[i]);
            rx->sq_rf_sum += cabs(c);                    /* quisk.c:1892-1893 */
            rx->sq_rf_count += 1;
            p = c * conj(rx->fm_1);
            rx->fm_1 = c;
            di = rx->filter_srate * carg(p);
            a[i] = rx->fm_y1 = di * rx->fm_a0 + rx->fm_x1 * rx->fm_a1 - rx->fm_y1 * rx->fm_b1;
            rx->fm_x1 = di;
            rx->ma_sum += a[i] * a[i]; rx->ma_count += 1;
        }
        n = qko_ddecimate(a, n, &rx->aud24p3, 2);
        n = qko_dfilter(a, n, &rx->fmhp);
        n = qko_dinterp2hb45(a, n, &rx->dem_hb6);
        audio_hooks(rx, a, n, 0, 0);                     /* quisk.c:1909-1910 */
        if (rx->sq_rf_count >= 2400) {                   /* quisk.c:1911-1918: mean magnitude in dB re full scale */
            rx->sq_value = rx->sq_rf_sum / rx->sq_rf_count / QKO_CLIP32;
            rx->sq_value = rx->sq_value > 1.E-10 ? 20 * log10(rx->sq_value) : -200.0;
            rx->sq_rf_sum = 0; rx->sq_rf_count = 0;
        }
        rx->sq_active = rx->sq_value < rx->squelch_level;                /* quisk.c:1920 */
        break;
    case QKO_DGT_U: case QKO_DGT_L:
        narrow = rx->bandwidth < 3000;                   /* DGT_NARROW_FREQ, quisk.c:35 */
        if (narrow) {
            rx->filter_srate = rx->decim_srate / 8;
            n = qko_cdecim2hb45(x, n, &rx->dem_hb5);
            n = qko_cdecim2hb45(x, n, &rx->dem_hb4);
            n = qko_cdecimate(x, n, &rx->dem_dec2, 2);
        } else {
            rx->filter_srate = rx->decim_srate;
        }
        for (i = 0; i < n; i++) {
            c = rx_bandpass_iq(rx, x[i]);
            a[i] = rx->mode == QKO_DGT_L ? creal(c) + cimag(c) : creal(c) - cimag(c);
            rx->ma_sum += a[i] * a[i]; rx->ma_count += 1;
        }
        audio_hooks(rx, a, n, 0, 0);                     /* quisk.c:1940-1941, 1966-1967 */
        if (narrow) {
            n = qko_dinterpolate(a, n, &rx->aud12p2, 2);
            n = qko_dinterp2hb45(a, n, &rx->dem_hb6);
            n = qko_dinterp2hb45(a, n, &rx->dem_hb7);
        }
        break;
    case QKO_DGT_IQ:
        rx->filter_srate = rx->decim_srate;
        if (rx->bandwidth < 19000)
            for (i = 0; i < n; i++) x[i] = rx_bandpass_real(rx, x[i]);
        for (i = 0; i < n; i++) { rx->ma_sum += creal(x[i] * conj(x[i])); rx->ma_count += 1; }     /* quisk.c:1980-1985 */
        break;
    default:        /* LSB, USB and every mode the reference's switch sends to its default label */
        rx->filter_srate = rx->decim_srate / 4;
        n = qko_cdecim2hb45(x, n, &rx->dem_hb5);
        n = qko_cdecimate(x, n, &rx->dem_dec2, 2);
        for (i = 0; i < n; i++) {
            c = rx_bandpass_iq(rx, x[i]);
            a[i] = rx->mode == QKO_LSB ? creal(c) + cimag(c) : creal(c) - cimag(c);
            rx->ma_sum += a[i] * a[i]; rx->ma_count += 1;
        }
        audio_hooks(rx, a, n, 0, 1);
        if (rx->mode == QKO_LSB && rx->agraph) qko_audio_graph_add(rx->agraph, a, n);      /* quisk.c:1835 */
        n = qko_dinterpolate(a, n, &rx->aud24p4, 2);
        n = qko_dinterp2hb45(a, n, &rx->dem_hb7);
        break;
    }
    if (rx->ma_count >= (long)rx->filter_srate * rx->ma_time) {      /* quisk.c:2015-2018 */
        rx->measured_audio = sqrt(rx->ma_sum / rx->ma_count) / QKO_CLIP32 * 1e6;
        rx->ma_sum = 0; rx->ma_count = 0;
    }
    return n;
}
double qko_rx_measure_audio(qko_rx *rx, int time) { if (time > 0) rx->ma_time = time; return rx->measured_audio; }   /* quisk.c:3011-3020 */
void qko_rx_set_audio_graph(qko_rx *rx, qko_agraph *g) { rx->agraph = g; }

int qko_rx_pre_agc(qko_rx *rx, qko_cd *x, int n)
{   /* quisk_process_samples up to the AGC (quisk.c:2350-2553): stereo samples at the play rate */
    int i;
    if (n <= 0) return n;
    if (n * 2 > rx->naudio) {                            /* quisk.c:2198-2212 */
        free(rx->audio);
        rx->naudio = n * 2;
        rx->audio = (double *)malloc((size_t)rx->naudio * sizeof(double));
    }
    qko_rx_mix(rx, x, n);
    n = qko_rx_decimate(rx, x, n);
    n = qko_rx_demodulate(rx, x, rx->audio, n);
    if (rx->mode != QKO_DGT_IQ)                          /* mono on both channels, quisk.c:2495-2501 */
        for (i = 0; i < n; i++) x[i] = rx->audio[i] + I * rx->audio[i];
    if (rx->decim_srate != 48000) {                      /* quisk.c:2528-2532 */
        n = qko_fracdecim(&rx->frac, x, n, rx->decim_srate / 48000.0);
        rx->decim_srate = 48000;
    }
    switch (rx->playback_rate / 48000) {                 /* quisk.c:2535-2553 */
    case 8: n = qko_cinterp2hb45(x, n, &rx->play_hb[0]); n = qko_cinterp2hb45(x, n, &rx->play_hb[1]);
            n = qko_cinterp2hb45(x, n, &rx->play_hb[2]); break;
    case 4: n = qko_cinterp2hb45(x, n, &rx->play_hb[0]); n = qko_cinterp2hb45(x, n, &rx->play_hb[1]); break;
    case 2: n = qko_cinterp2hb45(x, n, &rx->play_hb[0]); break;
    default: break;
    }
    return n;
}

void qko_rx_agc(qko_rx *rx, qko_cd *x, int n)
{   /* quisk.c:2556-2575 */
    int i;
    qko_agc_process(&rx->agc, x, n, rx->mode == QKO_DGT_IQ, rx->playback_rate, rx->release_gain,
                    rx->release_time, rx->mode == QKO_FM || rx->mode == QKO_DGT_FM);
    if (rx->kill_audio || rx->sq_active)                 /* quisk.c:2585-2593: silence behind the AGC */
        for (i = 0; i < n; i++) x[i] = 0;
}

int qko_rx_process(qko_rx *rx, qko_cd *x, int n)
{
    if (n <= 0) return n;
    n = qko_rx_pre_agc(rx, x, n);
    qko_rx_agc(rx, x, n);
    return n;
}

void qko_rx_process_many(qko_rx **rx, int nrx, const qko_cd *x, int n, qko_cd *out, long out_stride, int *counts)
{
    int r;
    qko_cd *work = (qko_cd *)malloc((size_t)(n > 0 ? n : 1) * sizeof(qko_cd));
    for (r = 0; r < nrx; r++) {
        int m;
        memcpy(work, x, (size_t)n * sizeof(qko_cd));
        m = qko_rx_process(rx[r], work, n);
        memcpy(out + r * out_stride, work, (size_t)(m > 0 ? m : 0) * sizeof(qko_cd));
        counts[r] = m;
    }
    free(work);
}

/* ================================ quisk.c: display spectrum ========================= */

/* any length: recursive decimation in time over the prime factors of n (direct p-point sums at each level),
 * twiddles from one table of cexp values.  Stands where the reference calls FFTW3 on sizes such as
 * data_width * fft_mult (quisk.py:3858-3890) and 8 * multirx_data_width (quisk.c:5098). */
static void fft_any_rec(const qko_cd *in, qko_cd *out, int n, int stride, const qko_cd *w, int N)
{
    int p, m, r, k, q;
    qko_cd t[64];
    if (n == 1) { out[0] = in[0]; return; }
    for (p = 2; p * p <= n && n % p; p++) ;
    if (n % p) p = n;
    m = n / p;
    if (p > 64) {       /* a large prime factor: plain DFT of this level */
        qko_cd *tmp = (qko_cd *)malloc((size_t)n * sizeof(qko_cd));
        for (k = 0; k < n; k++) {
            qko_cd acc = 0;
            for (r = 0; r < n; r++) acc += in[(size_t)r * stride] * w[(size_t)(((long long)r * k) % n) * (N / n)];
            tmp[k] = acc;
        }
        memcpy(out, tmp, (size_t)n * sizeof(qko_cd));
        free(tmp);
        return;
    }
    for (r = 0; r < p; r++) fft_any_rec(in + (size_t)r * stride, out + (size_t)r * m, m, stride * p, w, N);
    for (k = 0; k < m; k++) {
        for (r = 0; r < p; r++) t[r] = out[(size_t)r * m + k] * w[(size_t)(((long long)r * k) % n) * (N / n)];
        for (q = 0; q < p; q++) {
            qko_cd acc = t[0];
            for (r = 1; r < p; r++) acc += t[r] * w[(size_t)((((long long)r * q) % p) * m) * (N / n)];
            out[(size_t)k + (size_t)m * q] = acc;
        }
    }
}
static void fft_any(qko_cd *x, int n)
{
    int i;
    qko_cd *w = (qko_cd *)malloc((size_t)n * sizeof(qko_cd)), *in = (qko_cd *)malloc((size_t)n * sizeof(qko_cd));
    for (i = 0; i < n; i++) w[i] = cexp(-2.0 * M_PI * I * (double)i / (double)n);
    memcpy(in, x, (size_t)n * sizeof(qko_cd));
    fft_any_rec(in, x, n, 1, w, n);
    free(w); free(in);
}

void qko_fft(qko_cd *x, int n)
{   /* plain radix-2 decimation in time; twiddles from cexp per index, so the error stays ~1e-16 log2 n */
    int i, j, len, k, bits = 0;
    qko_cd *w;
    if (n < 1) return;
    if (n & (n - 1)) { fft_any(x, n); return; }
    while ((1 << bits) < n) bits++;
    for (i = 0; i < n; i++) {
        unsigned r = 0, v = (unsigned)i;
        for (k = 0; k < bits; k++) { r = (r << 1) | (v & 1); v >>= 1; }
        if ((int)r > i) { qko_cd t = x[i]; x[i] = x[r]; x[r] = t; }
    }
    w = (qko_cd *)malloc((size_t)(n / 2 > 0 ? n / 2 : 1) * sizeof(qko_cd));
    for (i = 0; i < n / 2; i++) w[i] = cexp(-2.0 * M_PI * I * (double)i / (double)n);
    for (len = 2; len <= n; len <<= 1) {
        int half = len / 2, stride = n / len;
        for (i = 0; i < n; i += len)
            for (j = 0; j < half; j++) {
                qko_cd u = x[i + j], t;
                qko_cd b = x[i + j + half], tw = w[j * stride];
                t = (creal(b) * creal(tw) - cimag(b) * cimag(tw)) + I * (creal(b) * cimag(tw) + cimag(b) * creal(tw));
                x[i + j] = u + t;
                x[i + j + half] = u - t;
            }
    }
    free(w);
}

void qko_hann_window(double *w, int n)
{
    int i, j;
    for (i = 0, j = -n / 2; i < n; i++, j++) w[i] = 0.5 + 0.5 * cos(2. * M_PI * j / n);
}

void qko_graph_init(qko_graph *g, int fft_size, int data_width, int sample_rate)
{
    memset(g, 0, sizeof *g);
    g->fft_size = fft_size; g->data_width = data_width; g->sample_rate = sample_rate;
    g->window = (double *)malloc((size_t)fft_size * sizeof(double));
    g->avg = (double *)calloc((size_t)fft_size, sizeof(double));
    qko_hann_window(g->window, fft_size);
}
void qko_graph_free(qko_graph *g) { free(g->window); free(g->avg); free(g->tmp); memset(g, 0, sizeof *g); }
void qko_graph_change_scan(qko_graph *g, int scan_blocks, int scan_sample_rate, double scan_valid, int scan_vfo0, int scan_deltaf,
                           int tx_tune, int vfo_screen)
{
    g->scan_blocks = scan_blocks; g->scan_sample_rate = scan_sample_rate; g->scan_valid = scan_valid;
    g->scan_vfo0 = scan_vfo0; g->scan_deltaf = scan_deltaf; g->tx_tune = tx_tune; g->vfo_screen = vfo_screen;
    g->next_block = 0;
    if (!g->tmp) g->tmp = (double *)malloc((size_t)g->fft_size * sizeof(double));
}
void qko_graph_set_passband(qko_graph *g, int rx_tune, int filter_start_offset, int filter_bandwidth)
{ g->rx_tune = rx_tune; g->filter_start_offset = filter_start_offset; g->filter_bandwidth = filter_bandwidth; }

void qko_graph_add_frame(qko_graph *g, qko_cd *s)
{
    int i, j, k, n, N = g->fft_size;
    double d2;
    qko_cd c;
    for (i = 0; i < N; i++) s[i] *= g->window[i];
    qko_fft(s, N);
    /* S-meter: power over the receive pass band, quisk.c:4659-4680 */
    int block = 0;
    if (g->scan_blocks) { block = g->next_block; g->next_block = block + 1 < g->scan_blocks ? block + 1 : 0; }
    d2 = (double)g->filter_bandwidth * N / g->sample_rate;
    if (g->scan_blocks) {                                /* quisk.c:4660-4663 */
        double d1 = ((double)g->tx_tune + g->vfo_screen - g->scan_vfo0 - g->scan_deltaf * block) * N / g->scan_sample_rate;
        i = (int)(d1 - d2 / 2 + 0.5);
    } else
        i = (int)((double)(g->rx_tune + g->filter_start_offset) * N / g->sample_rate + 0.5);
    n = (int)(floor(d2) + 0.01);
    if (i > -N / 2 && i + n + 1 < N / 2) {
        for (j = 0; j < n; i++, j++) {
            c = i < 0 ? s[N + i] : s[i];
            g->meter = g->meter + (creal(c) * creal(c) + cimag(c) * cimag(c));
        }
        c = i < 0 ? s[N + i] : s[i];
        g->meter = g->meter + (creal(c) * creal(c) + cimag(c) * cimag(c)) * (d2 - n);
    }
    if (g->scan_blocks) {                                /* quisk.c:4682-4702: the block's valid centre part, scan_blocks bins to one */
        int m, m0, deltam, ii, nn;
        if (block == g->scan_blocks - 1) g->count++;
        k = 0;
        for (i = N / 2; i < N; i++) g->tmp[k++] = cabs(s[i]);
        for (i = 0; i < N / 2; i++) g->tmp[k++] = cabs(s[i]);
        m0 = (int)(N * ((1.0 - g->scan_valid) / 2.0));
        deltam = (int)(N * g->scan_valid / g->scan_blocks);
        m = m0 + block * deltam;
        ii = (int)(N * ((1.0 - g->scan_valid) / 2.0));
        for (j = 0; j < deltam; j++) {
            d2 = 0;
            for (nn = 0; nn < g->scan_blocks; nn++) d2 += g->tmp[ii++];
            g->avg[m++] = d2;
        }
        return;
    }
    /* magnitude average, negative frequencies first, quisk.c:4703-4710 */
    g->count++;
    k = 0;
    for (i = N / 2; i < N; i++) g->avg[k++] += cabs(s[i]);
    for (i = 0; i < N / 2; i++) g->avg[k++] += cabs(s[i]);
}

int qko_graph_release(qko_graph *g, double zoom, double deltaf, double *out, double *smeter)
{
    int i, j, k, n, N = g->fft_size, W = g->data_width, count = g->count;
    double d2, scale, sm;
    if (count <= 0) return 0;
    n = (int)(zoom * (double)N / W + 0.5);
    if (n < 1) n = 1;
    for (i = 0; i < W; i++) {                            /* in place, like the reference (can alias for zoom < 1) */
        k = (int)(N * (deltaf / g->sample_rate + zoom * ((double)i / W - 0.5) + 0.5) + 0.1);
        d2 = 0.0;
        for (j = 0; j < n; j++, k++)
            if (k >= 0 && k < N) d2 += g->avg[k];
        g->avg[i] = d2;
    }
    scale = 1.0 / 2147483647.0 / N;
    sm = g->meter * scale * scale / count;
    g->meter = 0;
    sm = sm > 0 ? 10.0 * log10(sm) : -160.0;
    sm += 4.25969;
    if (smeter) *smeter = sm;
    scale = log10(count) + log10(N) + 31.0 * log10(2.0);
    scale *= 20.0;
    for (i = 0; i < W; i++) {
        d2 = 20.0 * log10(g->avg[i]) - scale;
        if (d2 < -200) d2 = -200;
        out[i] = d2;
    }
    for (i = 0; i < N; i++) g->avg[i] = 0;
    g->count = 0;
    return count;
}

/* ---- test tone, filter response, frequency measurement, audio graph ---------------------- */
void qko_tone_init(qko_tone *t) { t->tone = 21474836.47; t->audio = 1.0; }       /* quisk.c:1197-1198 */
void qko_add_test_tone(qko_tone *t, qko_cd *x, int n, int freq, int sample_rate, int mode)
{   /* add_tone (quisk.c:3022-3034) + AddTestTone (quisk.c:1194-1229) */
    int i;
    qko_cd phase, aphase;
    if (!freq) return;
    phase = cexp((I * 2.0 * M_PI * freq) / sample_rate);
    aphase = cexp(I * 2.0 * M_PI * 1000 / sample_rate);
    switch (mode) {
    default:
        for (i = 0; i < n; i++) { x[i] += t->tone; t->tone *= phase; }
        break;
    case QKO_AM:
        for (i = 0; i < n; i++) { x[i] += t->tone * (1.0 + creal(t->audio)); t->tone *= phase; t->audio *= aphase; }
        break;
    case QKO_FM: case QKO_DGT_FM:
        for (i = 0; i < n; i++) { x[i] += t->tone * cexp(I * creal(t->audio)); t->tone *= phase; t->audio *= aphase; }
        break;
    }
}

void qko_get_filter(const double *fi, const double *fq, int nfilt, int data_width, int fft_size, double *out)
{   /* get_filter, quisk.c:4761-4848 */
    int i, j, k, n, freq, time, W = data_width, T = data_width + nfilt;
    double *average = (double *)malloc(sizeof(double) * T), *bufI = (double *)calloc(nfilt + 1, sizeof(double)),
           *bufQ = (double *)calloc(nfilt + 1, sizeof(double)), *window = (double *)malloc(sizeof(double) * fft_size);
    qko_cd *samples = (qko_cd *)calloc(W, sizeof(qko_cd));
    double phase, delta, d2, accI, accQ, scale;
    qko_hann_window(window, fft_size);
    for (i = 0; i < T; i++) average[i] = 0.5;
    for (freq = 1; freq < W / 2.0 - 10.0; freq++) {
        delta = 2 * M_PI / W * freq;
        phase = 0;
        for (time = 0; time < T; time++) {
            average[time] += cos(phase);
            phase += delta;
            if (phase > 2 * M_PI) phase -= 2 * M_PI;
        }
    }
    n = 0;
    for (time = 0; time < T; time++) {
        d2 = average[time];
        accI = accQ = 0;
        if (nfilt > 0) {
            bufI[n] = d2; bufQ[n] = d2;
            j = n;
            for (k = 0; k < nfilt; k++) {
                accI += bufI[j] * fi[k];
                accQ += bufQ[j] * fq[k];
                if (++j >= nfilt) j = 0;
            }
            if (++n >= nfilt) n = 0;
        }
        if (time >= nfilt) samples[time - nfilt] = accI + I * accQ;
    }
    for (i = 0; i < W; i++) samples[i] *= window[i];
    qko_fft(samples, W);
    scale = 1. / W;
    for (k = 0; k < W; k++) {
        d2 = cabs(samples[k]) * scale;
        average[k] = d2 <= 1e-7 ? -7 : log10(d2);
    }
    i = 0;
    for (k = W / 2; k < W; k++, i++) out[i] = 20.0 * average[k];
    for (k = 0; k < W / 2; k++, i++) out[i] = 20.0 * average[k];
    free(average); free(bufI); free(bufQ); free(window); free(samples);
}

#define MF_N 12000
void qko_mfreq_init(qko_mfreq *m)
{   /* measure_freq(NULL, ...), quisk.c:4866-4876 */
    int i;
    memset(m, 0, sizeof *m);
    m->samples = (qko_cd *)malloc(sizeof(qko_cd) * MF_N);
    m->window = (double *)malloc(sizeof(double) * MF_N);
    m->average = (double *)calloc(MF_N, sizeof(double));
    for (i = 0; i < MF_N; i++) m->window[i] = 0.50 - 0.50 * cos(2. * M_PI * i / (MF_N - 1));
}
void qko_mfreq_free(qko_mfreq *m) { free(m->samples); free(m->window); free(m->average); free(m->hb[0].scratch); free(m->hb[1].scratch); free(m->hb[2].scratch); free(m->buf); memset(m, 0, sizeof *m); }
double qko_measure_freq(qko_mfreq *m, const qko_cd *x, int n, int srate, int mode, int rit_freq, int rx_tune)
{   /* measure_freq, quisk.c:4877-4919; returns measured_frequency */
    int i, k, center, ipeak;
    double dmax, c3, freq;
    if (n > m->nbuf) { free(m->buf); m->nbuf = n * 2; m->buf = (qko_cd *)malloc(sizeof(qko_cd) * m->nbuf); }
    memcpy(m->buf, x, (size_t)n * sizeof(qko_cd));
    n = qko_cdecim2hb45(m->buf, n, &m->hb[0]);
    n = qko_cdecim2hb45(m->buf, n, &m->hb[1]);
    n = qko_cdecim2hb45(m->buf, n, &m->hb[2]);
    srate /= 8;
    for (i = 0; i < n && m->index < MF_N; i++, m->index++) m->samples[m->index] = m->buf[i];
    if (m->index < MF_N) return m->value;
    for (i = 0; i < MF_N; i++) m->samples[i] *= m->window[i];
    qko_fft(m->samples, MF_N);
    m->index = 0;
    m->count++;
    k = 0;
    for (i = MF_N / 2; i < MF_N; i++) m->average[k++] += cabs(m->samples[i]);
    for (i = 0; i < MF_N / 2; i++) m->average[k++] += cabs(m->samples[i]);
    if (m->count < mode / 2) return m->value;
    m->count = 0;
    dmax = 1.e-20; ipeak = 0;
    center = MF_N / 2 - rit_freq * MF_N / srate;
    k = 500; k = k * MF_N / srate;
    for (i = center - k; i <= center + k; i++)
        if (m->average[i] > dmax) { dmax = m->average[i]; ipeak = i; }
    c3 = 1.36 * (m->average[ipeak + 1] - m->average[ipeak - 1]) / (m->average[ipeak - 1] + m->average[ipeak] + m->average[ipeak + 1]);
    freq = srate * (2 * (ipeak + c3) - MF_N) / 2 / MF_N;
    freq += rx_tune;
    m->value = freq;
    memset(m->average, 0, sizeof(double) * MF_N);
    return m->value;
}

qko_mfreq *qko_mfreq_new(void) { qko_mfreq *m = (qko_mfreq *)malloc(sizeof *m); qko_mfreq_init(m); return m; }
void qko_mfreq_delete(qko_mfreq *m) { if (m) { qko_mfreq_free(m); free(m); } }
qko_agraph *qko_audio_graph_new(int data_width) { qko_agraph *g = (qko_agraph *)calloc(1, sizeof *g); g->data_width = data_width; return g; }
void qko_audio_graph_delete(qko_agraph *g) { if (g) { free(g->window); free(g->average); free(g->fft); free(g); } }
void qko_audio_graph_add(qko_agraph *g, const double *d, int n)
{   /* quisk_calc_audio_graph(pow(2, 31) - 1, NULL, dsamples, nSamples, 1), quisk.c:903-964 */
    int i, k, inp, N = g->data_width;
    double scale = pow(2, 31) - 1;
    if (!g->planned) {
        g->planned = 1;
        g->count_limit = 8000 / N / 5;
        if (g->count_limit <= 0) g->count_limit = 1;
        g->window = (double *)malloc(N * sizeof(double));
        g->average = (double *)calloc(N, sizeof(double));
        g->fft = (qko_cd *)malloc(N * sizeof(qko_cd));
        for (i = 0; i < N; i++) g->window[i] = 0.50 - 0.50 * cos(2. * M_PI * i / N);
        return;
    }
    if (g->ready) return;
    scale *= N / 2.0;
    scale *= g->count_limit;
    scale *= 0.5;
    for (inp = 0; inp < n; inp++) {
        g->fft[g->index] = d[inp] / scale;
        if (++g->index >= N) {
            g->index = 0;
            for (i = 0; i < N; i++) g->fft[i] *= g->window[i];
            qko_fft(g->fft, N);
            g->count++;
            k = 0;
            for (i = N / 2; i < N; i++) g->average[k++] += cabs(g->fft[i]);
            for (i = 0; i < N / 2; i++) g->average[k++] += cabs(g->fft[i]);
            if (g->count >= g->count_limit) { g->ready = 1; g->count = 0; }
        }
    }
}
int qko_audio_graph_get(qko_agraph *g, double *out)
{   /* get_audio_graph, quisk.c:966-991 */
    int i;
    if (!g->ready) return 0;
    for (i = 0; i < g->data_width; i++) {
        double d2 = g->average[i];
        if (d2 < 1E-10) d2 = 1E-10;
        out[i] = 20.0 * log10(d2);
        g->average[i] = 0;
    }
    g->ready = 0;
    return 1;
}

/* ---- second-receiver routing ----------------------------------------------------------- */
int qko_buffer2chan(qko_b2c *b, double *samp1, int count1, double *samp2, int count2)
{   /* Buffer2Chan, quisk.c:1452-1487: the minimum of the two counts comes out, the rest waits */
    int nout;
    if (b->nbuf1 == 0 && b->nbuf2 == 0 && count1 == count2) return count1;
    if (count1 + b->nbuf1 >= QKO_B2C_SIZE || count2 + b->nbuf2 >= QKO_B2C_SIZE) b->nbuf1 = b->nbuf2 = 0;
    memcpy(b->buf1 + b->nbuf1, samp1, (size_t)count1 * sizeof(double)); b->nbuf1 += count1;
    memcpy(b->buf2 + b->nbuf2, samp2, (size_t)count2 * sizeof(double)); b->nbuf2 += count2;
    nout = b->nbuf1 <= b->nbuf2 ? b->nbuf1 : b->nbuf2;
    memcpy(samp1, b->buf1, (size_t)nout * sizeof(double));
    b->nbuf1 -= nout; memmove(b->buf1, b->buf1 + nout, (size_t)b->nbuf1 * sizeof(double));
    memcpy(samp2, b->buf2, (size_t)nout * sizeof(double));
    b->nbuf2 -= nout; memmove(b->buf2, b->buf2 + nout, (size_t)b->nbuf2 * sizeof(double));
    return nout;
}
/* quisk_process_samples with a second channel (quisk.c:2403-2601): receiver `a` is bank 0, receiver `b` the second channel
 * (split RX/TX: the same mode and filter, tuned to the transmit frequency, quisk.c:2410-2418; or the sub-receiver of
 * multirx_play_channel, quisk.c:2463-2472).  Both are mixed, decimated and demodulated on the same input; Buffer2Chan
 * equalises the counts at the decimated rate; the pair goes through cFracDecim and the play-rate interpolators -- the
 * statics of quisk_process_samples, i.e. a's -- and Agc1 (a's) runs on the left channel, Agc2 (the pair's) on the right,
 * each as a mono process_agc (quisk.c:2563-2572); then the squelch per channel (quisk.c:2585-2601).
 * method 1: left = a, right = b; 2: left = b, right = a; 3: both = b; 4: both = a.  x: n input samples, overwritten by the
 * stereo block; pre (optional, same capacity): the pair in front of the AGCs.  Returns the count. */
void qko_pair_init(qko_pair *p) { memset(p, 0, sizeof *p); qko_agc_init(&p->agc2, 0.7, 0); }
void qko_pair_free(qko_pair *p) { qko_agc_free(&p->agc2); free(p->xb); free(p->ab); free(p->ch); p->xb = NULL; p->ab = NULL; p->ch = NULL; }
qko_pair *qko_pair_new(void) { qko_pair *p = (qko_pair *)malloc(sizeof *p); qko_pair_init(p); return p; }
void qko_pair_delete(qko_pair *p) { if (p) { qko_pair_free(p); free(p); } }
void qko_pair_reset_fifos(qko_pair *p) { p->b2c.nbuf1 = p->b2c.nbuf2 = 0; }      /* Buffer2Chan(NULL, 0, NULL, 0), quisk.c:2227-2231 */
int qko_rx_pair_process(qko_rx *a, qko_rx *b, qko_pair *p, qko_cd *x, int n, int method, qko_cd *pre)
{
    int i, na, nb, m, sq_l, sq_r, fm;
    if (n <= 0) return n;
    if (n * 2 > p->cap) {
        free(p->xb); free(p->ab); free(p->ch);
        p->cap = n * 2 + 2 * QKO_B2C_SIZE;
        p->xb = (qko_cd *)malloc((size_t)p->cap * sizeof(qko_cd));
        p->ab = (double *)malloc((size_t)p->cap * sizeof(double));
        p->ch = (qko_cd *)malloc((size_t)p->cap * sizeof(qko_cd));
    }
    if (p->cap > a->naudio) {
        free(a->audio);
        a->naudio = p->cap;
        a->audio = (double *)malloc((size_t)a->naudio * sizeof(double));
    }
    memcpy(p->xb, x, (size_t)n * sizeof(qko_cd));                 /* orig_cSamples, quisk.c:2226 */
    qko_rx_mix(a, x, n);
    na = qko_rx_decimate(a, x, n);
    na = qko_rx_demodulate(a, x, a->audio, na);
    qko_rx_mix(b, p->xb, n);                                      /* quisk.c:2411-2418 */
    nb = qko_rx_decimate(b, p->xb, n);
    nb = qko_rx_demodulate(b, p->xb, p->ab, nb);
    m = qko_buffer2chan(&p->b2c, a->audio, na, p->ab, nb);        /* quisk.c:2419 */
    for (i = 0; i < m; i++) {
        const double l = (method == 1 || method == 4) ? a->audio[i] : p->ab[i];
        const double r = (method == 2 || method == 4) ? a->audio[i] : p->ab[i];
        x[i] = l + I * r;
    }
    sq_l = (method == 1 || method == 4) ? a->sq_active : b->sq_active;
    sq_r = (method == 2 || method == 4) ? a->sq_active : b->sq_active;
    if (a->decim_srate != 48000) {                                /* quisk.c:2528-2532 */
        m = qko_fracdecim(&a->frac, x, m, a->decim_srate / 48000.0);
        a->decim_srate = 48000;
    }
    switch (a->playback_rate / 48000) {                           /* quisk.c:2535-2553 */
    case 8: m = qko_cinterp2hb45(x, m, &a->play_hb[0]); m = qko_cinterp2hb45(x, m, &a->play_hb[1]);
            m = qko_cinterp2hb45(x, m, &a->play_hb[2]); break;
    case 4: m = qko_cinterp2hb45(x, m, &a->play_hb[0]); m = qko_cinterp2hb45(x, m, &a->play_hb[1]); break;
    case 2: m = qko_cinterp2hb45(x, m, &a->play_hb[0]); break;
    default: break;
    }
    if (pre) memcpy(pre, x, (size_t)m * sizeof(qko_cd));
    fm = a->mode == QKO_FM || a->mode == QKO_DGT_FM;               /* rxMode, quisk.c:2088 */
    for (i = 0; i < m; i++) { p->ch[i] = cimag(x[i]); x[i] = creal(x[i]); }        /* quisk.c:2564-2567 */
    qko_agc_process(&a->agc, x, m, 0, a->playback_rate, a->release_gain, a->release_time, fm);
    qko_agc_process(&p->agc2, p->ch, m, 0, a->playback_rate, a->release_gain, a->release_time, fm);
    for (i = 0; i < m; i++) x[i] = creal(x[i]) + I * creal(p->ch[i]);
    if (a->kill_audio || (sq_l && sq_r)) { for (i = 0; i < m; i++) x[i] = 0; }     /* quisk.c:2585-2601 */
    else if (sq_r) { for (i = 0; i < m; i++) x[i] = creal(x[i]); }
    else if (sq_l) { for (i = 0; i < m; i++) x[i] = I * cimag(x[i]); }
    return m;
}
double qko_keyup_envelope(double env, qko_cd *x, int n, int playback_rate)
{   /* quisk.c:2602-2612; returns the new keyupEnvelope */
    int i;
    double di;
    if (env < 1.0) {
        di = 1. / (playback_rate * 5e-3);
        for (i = 0; i < n; i++) {
            env += di;
            if (env > 1.0) { env = 1.0; break; }
            x[i] *= env;
        }
    }
    return env;
}

/* ---- sub-receiver graph and bandscope ------------------------------------------------ */
void qko_multirx_graph(qko_cd *samples, int data_width, double *out)
{   /* get_multirx_graph, quisk.c:4478-4519; samples (8 * data_width of them) are overwritten */
    const int mult = 8, N = data_width * mult;
    int i, j, k;
    double d1, d2, scale;
    for (i = 0, j = -N / 2; i < N; i++, j++) samples[i] *= 0.5 + 0.5 * cos(2. * M_PI * j / N);
    qko_fft(samples, N);
    scale = log10(N) + 31.0 * log10(2.0);
    scale *= 20.0;
    j = mult; k = 0; d1 = 0;
    for (i = N / 2; i < N; i++) {
        d1 += cabs(samples[i]);
        if (--j == 0) { d2 = 20.0 * log10(d1) - scale; if (d2 < -200) d2 = -200; out[k++] = d2; d1 = 0; j = mult; }
    }
    for (i = 0; i < N / 2; i++) {
        d1 += cabs(samples[i]);
        if (--j == 0) { d2 = 20.0 * log10(d1) - scale; if (d2 < -200) d2 = -200; out[k++] = d2; d1 = 0; j = mult; }
    }
}

void qko_bscope_init(qko_bscope *b, int size)
{   /* init_bandscope, quisk.c:2758-2775 */
    int i, j;
    memset(b, 0, sizeof *b);
    b->size = size;
    b->window = (double *)malloc((size_t)size * sizeof(double));
    b->avg = (double *)calloc((size_t)size / 2 + 2, sizeof(double));
    for (i = 0, j = -size / 2; i < size; i++, j++) b->window[i] = 0.5 + 0.5 * cos(2. * M_PI * j / size);
}
void qko_bscope_free(qko_bscope *b) { free(b->window); free(b->avg); memset(b, 0, sizeof *b); }
void qko_bscope_add(qko_bscope *b, const double *samples)
{   /* the processing half of get_bandscope, quisk.c:4545-4558 (the r2c transform as a complex one) */
    int i, L = b->size / 2 + 1;
    qko_cd *x = (qko_cd *)malloc((size_t)b->size * sizeof(qko_cd));
    for (i = 0; i < b->size; i++) {
        double d1 = fabs(samples[i]);
        if (d1 > b->the_max) b->the_max = d1;
        x[i] = samples[i] * b->window[i];
    }
    qko_fft(x, b->size);
    for (i = 0; i < L; i++) b->avg[i] += cabs(x[i]);
    b->count++;
    free(x);
}
int qko_bscope_release(qko_bscope *b, int clock, double zoom, double deltaf, double *out, int graph_width, double *adc_level)
{   /* quisk.c:4559-4595 */
    int i, j, j1, j2, L = b->size / 2 + 1, count = b->count;
    double frac, scale, rate, f1, d1, d2, sample;
    if (count <= 0) return 0;
    b->avg[L] = 0.0;
    frac = (double)L / graph_width;
    scale = 1.0 / frac / count / b->size;
    rate = clock / 2.0;
    for (i = 0; i < graph_width; i++) {
        f1 = deltaf + rate / 2.0 * (1.0 - zoom);
        d1 = L / rate * (f1 + (double)i / graph_width * zoom * rate);
        d2 = L / rate * (f1 + (double)(i + 1) / graph_width * zoom * rate);
        j1 = (int)floor(d1);
        j2 = (int)floor(d2);
#define BS_AT(j) (((j) >= 0 && (j) <= L) ? b->avg[j] : 0.0)     /* the reference indexes without a check */
        if (j1 == j2) sample = (d2 - d1) * BS_AT(j1);
        else {
            sample = (j1 + 1 - d1) * BS_AT(j1);
            for (j = j1 + 1; j < j2; j++) sample += BS_AT(j);
            sample += (d2 - j2) * BS_AT(j2);
        }
#undef BS_AT
        sample = sample * scale;
        out[i] = sample <= 1E-10 ? -200.0 : 20.0 * log10(sample);
    }
    if (adc_level) *adc_level = b->the_max;
    b->the_max = 0; b->count = 0;
    for (i = 0; i < L; i++) b->avg[i] = 0;
    return count;
}

/* ---- input conditioning ---------------------------------------------------------------- */
void qko_dcremove_init(qko_dcstate *d) { memset(d, 0, sizeof *d); d->alpha = 0.95; }
void qko_dcremove(qko_dcstate *d, qko_cd *x, int n, int sample_rate, int dc_remove_bw, int key_state)
{   /* DCremove, sound.c:165-230 (quisk_is_vna == 0) */
    int i;
    double omega, Qsin, Qcos, H0, xx;
    qko_cd c;
    if (sample_rate != d->old_sample_rate || dc_remove_bw != d->old_bandwidth) {
        d->old_sample_rate = sample_rate;
        d->old_bandwidth = dc_remove_bw;
        if (d->old_bandwidth > 0) {
            omega = M_PI * d->old_bandwidth / (d->old_sample_rate / 2.0);
            Qsin = sin(omega);
            Qcos = cos(omega);
            H0 = 1.0 / sqrt(2.0);
            xx = ((Qcos - 1) * (Qcos - 1) + Qsin * Qsin) / (H0 * H0) - Qsin * Qsin;
            xx = sqrt(xx);
            d->alpha = Qcos - xx;
        }
    }
    if (d->old_bandwidth == 0) {
    } else if (d->old_bandwidth == 1) {
        if (key_state) {
            d->dc_key_delay = 0; d->dc_sum = 0; d->dc_count = 0;
        } else if (d->dc_key_delay < d->old_sample_rate) {
            d->dc_key_delay += n;
        } else {
            d->dc_count += n;
            for (i = 0; i < n; i++) d->dc_sum += x[i];
            if (d->dc_count > d->old_sample_rate * 2) {
                d->dc_average = d->dc_sum / d->dc_count;
                d->dc_sum = 0;
                d->dc_count = 0;
            }
        }
        for (i = 0; i < n; i++) x[i] -= d->dc_average;
    } else {
        for (i = 0; i < n; i++) {
            c = x[i] + d->dc_remove * d->alpha;
            x[i] = c - d->dc_remove;
            d->dc_remove = c;
        }
    }
}
void qko_invert_spectrum(qko_cd *x, int n)
{   /* quisk.c:2315-2320 */
    int i;
    for (i = 0; i < n; i++) x[i] = conj(x[i]);
}

/* NoiseBlanker, quisk.c:615-720.  Every sample goes through a ring of 3 * hwindow samples (hwindow = 500 us); the
 * magnitudes in the ring are kept in a running sum.  A sample whose magnitude exceeds limit * (sum / size) is a pulse. */
void qko_nb_init(qko_nbstate *b) { memset(b, 0, sizeof *b); b->sample_rate = -1; }
void qko_nb_free(qko_nbstate *b) { free(b->ring); free(b->mags); b->ring = NULL; b->mags = NULL; }
void qko_noise_blanker(qko_nbstate *b, qko_cd *x, int n, int sample_rate, int level)
{
    int i, j, k;
    double limit;
    if (level <= 0) return;                                         /* quisk.c:631-632 */
    if (sample_rate != b->sample_rate) {                            /* quisk.c:633-651 */
        b->sample_rate = sample_rate;
        b->state = b->index = b->win_index = 0;
        b->sum = 0.0;
        b->hwindow = (int)(sample_rate * 500.E-6 + 0.5);
        b->size = b->hwindow * 3;
        b->mags = (double *)realloc(b->mags, b->size * sizeof(double));
        memset(b->mags, 0, b->size * sizeof(double));
        b->ring = (qko_cd *)realloc(b->ring, b->size * sizeof(qko_cd));
        memset(b->ring, 0, b->size * sizeof(qko_cd));
    }
    limit = level == 2 ? 4.0 : level == 3 ? 2.5 : 6.0;              /* quisk.c:652-663 */
    for (i = 0; i < n; i++) {                                       /* quisk.c:664-718 */
        const qko_cd newest = x[i];
        const double mag = cabs(newest);
        int pulse;
        x[i] = b->ring[b->index];                                   /* the oldest sample leaves */
        b->ring[b->index] = newest;
        b->sum -= b->mags[b->index];
        b->mags[b->index] = mag;
        b->sum += mag;
        pulse = !(mag <= b->sum / b->size * limit);
        if (b->state == 0) {
            if (pulse) {                                            /* ramp down into the pulse: this sample times 0, ... */
                b->state = 1;
                k = b->index;
                for (j = 0; j < b->hwindow; j++) {
                    b->ring[k--] *= (double)j / b->hwindow;
                    if (k < 0) k = b->size - 1;
                }
            } else if (b->win_index) {                              /* ramp up behind the last pulse */
                b->ring[b->index] *= (double)b->win_index / b->hwindow;
                if (++b->win_index >= b->hwindow) b->win_index = 0;
            }
        } else {
            b->ring[b->index] = 0;                                  /* zero until the pulses stop */
            if (!pulse) { b->state = 0; b->win_index = 1; }
        }
        if (++b->index >= b->size) b->index = 0;
    }
}
