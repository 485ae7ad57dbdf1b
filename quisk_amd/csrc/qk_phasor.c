/* qk_phasor.c -- host side of the tune mixer: the per-sample phase step and its power tables.
 *
 * The reference runs the mixer as a scalar recurrence, v *= phase once per sample, never
 * renormalised (quisk.c:2356-2360).  After n samples v = phase^n in floating point: since the
 * rounded phase has |phase| = 1 + d with |d| ~ 1e-16, the vector's magnitude and angle drift
 * SYSTEMATICALLY with n (the rounding of each product only adds a random walk ~ sqrt(n) 1e-16).
 * A closed form exp(-i 2 pi f n / fs) does not follow that drift.  The GPU needs v at arbitrary n
 * in parallel, so the host computes exact powers of the ROUNDED phase in double-double arithmetic
 * and hands the device three small tables per receiver:
 *     C[k] = phase^k          k = 0..63
 *     B[j] = phase^(64 j)     j = 0..63
 *     A[i] = phase^(4096 i)   i = 0..nA-1
 * so that phase^n = A[n >> 12] B[(n >> 6) & 63] C[n & 63], each entry correctly rounded.
 *
 * The phase itself must be the same double pair the reference gets from
 *     cexp((I * -2.0 * M_PI * tune) / sample_rate)        (quisk.c:2356)
 * so it is evaluated here with that C99 expression and the C library's cexp.
 */
#include <complex.h>
#include <math.h>

typedef struct { double hi, lo; } dd;
typedef struct { dd re, im; } ddc;

static inline dd two_sum(double a, double b)
{
    dd r; double bb;
    r.hi = a + b; bb = r.hi - a;
    r.lo = (a - (r.hi - bb)) + (b - bb);
    return r;
}
static inline dd quick_two_sum(double a, double b)
{
    dd r; r.hi = a + b; r.lo = b - (r.hi - a); return r;
}
static inline dd two_prod(double a, double b)
{
    dd r; r.hi = a * b; r.lo = fma(a, b, -r.hi); return r;
}
static inline dd dd_add(dd a, dd b)
{
    dd s = two_sum(a.hi, b.hi), t = two_sum(a.lo, b.lo);
    s.lo += t.hi; s = quick_two_sum(s.hi, s.lo);
    s.lo += t.lo; return quick_two_sum(s.hi, s.lo);
}
static inline dd dd_neg(dd a) { dd r; r.hi = -a.hi; r.lo = -a.lo; return r; }
static inline dd dd_mul(dd a, dd b)
{
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return quick_two_sum(p.hi, p.lo);
}
static inline ddc ddc_mul(ddc a, ddc b)
{
    ddc r;
    r.re = dd_add(dd_mul(a.re, b.re), dd_neg(dd_mul(a.im, b.im)));
    r.im = dd_add(dd_mul(a.re, b.im), dd_mul(a.im, b.re));
    return r;
}

/* the reference's per-sample phase step for an integer tune offset */
void qk_phase_step(int tune_hz, int sample_rate, double *re, double *im)
{
    double tune = tune_hz;
    double complex phase = cexp((I * -2.0 * M_PI * tune) / sample_rate);
    *re = creal(phase);
    *im = cimag(phase);
}

/* Fill C (64), B (64) and A (nA) with correctly rounded powers of (pre + i pim); interleaved re,im. */
void qk_phase_tables(double pre, double pim, double *C, double *B, double *A, int nA)
{
    ddc p, acc, p64, p4096;
    int k;
    p.re.hi = pre; p.re.lo = 0; p.im.hi = pim; p.im.lo = 0;
    acc.re.hi = 1; acc.re.lo = 0; acc.im.hi = 0; acc.im.lo = 0;
    for (k = 0; k < 64; k++) {
        C[2 * k] = acc.re.hi; C[2 * k + 1] = acc.im.hi;
        acc = ddc_mul(acc, p);
    }
    p64 = acc;
    acc.re.hi = 1; acc.re.lo = 0; acc.im.hi = 0; acc.im.lo = 0;
    for (k = 0; k < 64; k++) {
        B[2 * k] = acc.re.hi; B[2 * k + 1] = acc.im.hi;
        acc = ddc_mul(acc, p64);
    }
    p4096 = acc;
    acc.re.hi = 1; acc.re.lo = 0; acc.im.hi = 0; acc.im.lo = 0;
    for (k = 0; k < nA; k++) {
        A[2 * k] = acc.re.hi; A[2 * k + 1] = acc.im.hi;
        acc = ddc_mul(acc, p4096);
    }
}
