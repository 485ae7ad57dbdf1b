// qk_fft.h -- display spectrum: window, forward complex FFT, magnitude average (get_graph, quisk.c:4602-4759;
// window record_app quisk.c:5088-5096).  The reference calls FFTW3 (fftw_execute_dft, quisk.c:4653); this is a
// hand-written radix-4 Stockham autosort FFT in double, staged in LDS.
//
//   N <= 4096      one workgroup per frame: window on load, all passes in LDS, epilogue accumulates |X|.
//   N  > 4096      four-step, N = N1 x 4096:
//                  pass A  (k_fft_cols)  N1-point FFTs down the columns, 4096/N1 columns per workgroup (a 64 KiB
//                          LDS tile), window on load, twiddle W_N^(n2 k1) on store, Y[k1][n2] to HBM;
//                  pass B  (k_fft_rows)  4096-point FFT of each row Y[k1][.], epilogue accumulates |X[k1 + N1 k2]|
//                          into the average, kept in the same [k1][k2] layout so every HBM access is contiguous.
//                          The release kernel undoes the layout (and the fftshift) when it bins pixels.
// Algorithmic HBM bytes per frame: read 16 N (samples); the four-step adds 32 N for Y and 16 N for the
// read-modify-write of the average.
#pragma once
#include <hip/hip_runtime.h>

#define FFT_THREADS 256
#define FFT_BIG_THREADS 1024      // four-step passes: one butterfly per lane and pass, 16 waves hide the LDS and HBM latency

__device__ __forceinline__ double2 cmul_d(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Stockham autosort passes over an LDS array of TOTAL = n * s complex values holding s interleaved sequences of
// length n (element (p, q) at q + s p).  tw4096[j] = exp(-2 pi i j / 4096).  In place: every pass reads all its
// operands into registers, synchronises, then writes.  Result: natural order, same interleaving.
template <int TOTAL, int NT = FFT_THREADS>
__device__ __forceinline__ void lds_fft(double2 *buf, int n, int s, const double2 *__restrict__ tw4096)
{
    constexpr int PER4 = (TOTAL / 4 + NT - 1) / NT;
    const int tid = threadIdx.x;
    while (n >= 4) {
        const int n1 = n >> 2;
        const int tws = 4096 / n;                      // table stride for exp(-2 pi i p / n)
        double2 a[PER4], b[PER4], c[PER4], d[PER4];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER4; u++) {
            const int t = tid + u * NT;
            if (t < TOTAL / 4) {
                const int p = t / s, q = t - p * s;
                a[u] = buf[q + s * p];
                b[u] = buf[q + s * (p + n1)];
                c[u] = buf[q + s * (p + 2 * n1)];
                d[u] = buf[q + s * (p + 3 * n1)];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER4; u++) {
            const int t = tid + u * NT;
            if (t < TOTAL / 4) {
                const int p = t / s, q = t - p * s;
                const double2 apc = make_double2(a[u].x + c[u].x, a[u].y + c[u].y);
                const double2 amc = make_double2(a[u].x - c[u].x, a[u].y - c[u].y);
                const double2 bpd = make_double2(b[u].x + d[u].x, b[u].y + d[u].y);
                const double2 bmd = make_double2(b[u].x - d[u].x, b[u].y - d[u].y);
                // forward transform: -i (b - d) = (bmd.y, -bmd.x)
                const double2 y0 = make_double2(apc.x + bpd.x, apc.y + bpd.y);
                const double2 y1 = make_double2(amc.x + bmd.y, amc.y - bmd.x);
                const double2 y2 = make_double2(apc.x - bpd.x, apc.y - bpd.y);
                const double2 y3 = make_double2(amc.x - bmd.y, amc.y + bmd.x);
                const int o = q + s * 4 * p;
                buf[o] = y0;
                if (p == 0) {
                    buf[o + s] = y1; buf[o + 2 * s] = y2; buf[o + 3 * s] = y3;
                } else {
                    buf[o + s] = cmul_d(y1, tw4096[p * tws]);
                    buf[o + 2 * s] = cmul_d(y2, tw4096[2 * p * tws]);
                    buf[o + 3 * s] = cmul_d(y3, tw4096[3 * p * tws]);
                }
            }
        }
        n >>= 2; s <<= 2;
    }
    if (n == 2) {       // one radix-2 pass finishes sizes 2 * 4^m: p = 0 only, no twiddle
        constexpr int PER2 = (TOTAL / 2 + NT - 1) / NT;
        double2 a[PER2], b[PER2];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER2; u++) {
            const int q = tid + u * NT;
            if (q < TOTAL / 2) { a[u] = buf[q]; b[u] = buf[q + s]; }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER2; u++) {
            const int q = tid + u * NT;
            if (q < TOTAL / 2) {
                buf[q] = make_double2(a[u].x + b[u].x, a[u].y + b[u].y);
                buf[q + s] = make_double2(a[u].x - b[u].x, a[u].y - b[u].y);
            }
        }
    }
    __syncthreads();
}

struct FftArgs {
    const double2 *x;          // frame input (N complex samples)
    const double *window;      // N doubles (quisk.c:5091-5096)
    const double2 *tw4096;     // exp(-2 pi i j / 4096)
    const double2 *tw_lo, *tw_hi;   // four-step twiddles: W_N^j (j < 1024) and W_N^(1024 j)
    double2 *Y;                // four-step intermediate [N1][4096]
    double *avg;               // magnitude sum, layout [k1][k2] with k = k1 + N1 k2  (N1 = 1: natural order)
    double *band;              // |X|^2 of the S-meter pass-band bins of this frame (quisk.c:4659-4680)
    int N, N1;
    int band_start, band_count;   // first bin (signed, relative to DC) and number of bins kept (whole + fractional)
};

// epilogue shared by the single-workgroup FFT and pass B: element e of the row is X[k1 + N1 e]
__device__ __forceinline__ void fft_epilogue(const FftArgs &a, int k1, int e, double2 X)
{
    a.avg[(long long)k1 * (a.N / a.N1) + e] += hypot(X.x, X.y);          // cabs, quisk.c:4707-4709
    const int k = k1 + a.N1 * e;
    const int sk = k < a.N / 2 ? k : k - a.N;                             // signed bin number
    const int j = sk - a.band_start;
    if (j >= 0 && j < a.band_count) a.band[j] = X.x * X.x + X.y * X.y;   // c * conj(c), quisk.c:4673
}

template <int TOTAL>
__global__ __launch_bounds__(FFT_THREADS) void k_fft_small(FftArgs a)
{
    __shared__ __attribute__((aligned(16))) double2 buf[TOTAL];
    for (int e = threadIdx.x; e < TOTAL; e += FFT_THREADS) {
        const double2 v = a.x[e];
        const double w = a.window[e];
        buf[e] = make_double2(v.x * w, v.y * w);
    }
    lds_fft<TOTAL>(buf, TOTAL, 1, a.tw4096);
    for (int e = threadIdx.x; e < TOTAL; e += FFT_THREADS) fft_epilogue(a, 0, e, buf[e]);
}

// pass A: blockIdx.x = column tile; TW = 4096 / N1 columns per tile
__global__ __launch_bounds__(FFT_BIG_THREADS) void k_fft_cols(FftArgs a)
{
    __shared__ __attribute__((aligned(16))) double2 buf[4096];
    const int N1 = a.N1, TW = 4096 / N1;
    const int c0 = blockIdx.x * TW;
    for (int e = threadIdx.x; e < 4096; e += FFT_BIG_THREADS) {
        const int n1 = e / TW, col = e - n1 * TW;
        const long long idx = (long long)n1 * 4096 + c0 + col;          // x[N2 n1 + n2]
        const double2 v = a.x[idx];
        const double w = a.window[idx];
        buf[e] = make_double2(v.x * w, v.y * w);
    }
    lds_fft<4096, FFT_BIG_THREADS>(buf, N1, TW, a.tw4096);
    for (int e = threadIdx.x; e < 4096; e += FFT_BIG_THREADS) {
        const int k1 = e / TW, col = e - k1 * TW;
        const int n2 = c0 + col;
        const long long t = (long long)n2 * k1;                          // < N
        const double2 w = cmul_d(a.tw_hi[t >> 10], a.tw_lo[t & 1023]);
        a.Y[(long long)k1 * 4096 + n2] = cmul_d(buf[e], w);
    }
}

// pass B: blockIdx.x = k1
__global__ __launch_bounds__(FFT_BIG_THREADS) void k_fft_rows(FftArgs a)
{
    __shared__ __attribute__((aligned(16))) double2 buf[4096];
    const int k1 = blockIdx.x;
    const double2 *row = a.Y + (long long)k1 * 4096;
    for (int e = threadIdx.x; e < 4096; e += FFT_BIG_THREADS) buf[e] = row[e];
    lds_fft<4096, FFT_BIG_THREADS>(buf, 4096, 1, a.tw4096);
    for (int e = threadIdx.x; e < 4096; e += FFT_BIG_THREADS) fft_epilogue(a, k1, e, buf[e]);
}

// S-meter accumulation of one frame, in the reference's order (quisk.c:4668-4679): whole bins, then the
// fractional bin.  One thread.
__global__ void k_meter_accum(const double *band, int n_whole, double frac, double *meter)
{
#pragma clang fp contract(off)
    if (threadIdx.x || blockIdx.x) return;
    double m = *meter;
    for (int j = 0; j < n_whole; j++) m = m + band[j];
    m = m + band[n_whole] * frac;
    *meter = m;
}

// Release (quisk.c:4713-4753): pixel i sums n average bins starting at shifted index k0[i]; then dB scaling.
// One wave per pixel.  `shifted` index kk counts from -fs/2: X index k = (kk + N/2) mod N.
struct ReleaseArgs {
    const double *avg; double *out; const int *k0;
    int N, N1, W, n;
    double scale;              // 20 (log10 count + log10 N + 31 log10 2)
};
__device__ __forceinline__ double avg_at(const ReleaseArgs &a, int kk)
{
    const int k = (kk + a.N / 2) & (a.N - 1);
    const int k1 = k % a.N1, k2 = k / a.N1;
    return a.avg[(long long)k1 * (a.N / a.N1) + k2];
}
__global__ __launch_bounds__(256) void k_graph_release(ReleaseArgs a)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= a.W) return;
    const int k0 = a.k0[i];
    double sum = 0.0;
    for (int j = lane; j < a.n; j += 64) {
        const int kk = k0 + j;
        if (kk >= 0 && kk < a.N) sum += avg_at(a, kk);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) {
        double d = 20.0 * log10(sum) - a.scale;
        if (d < -200) d = -200;
        a.out[i] = d;
    }
}
// In-place variant for the zoom settings where the reference's in-place binning reads entries it has already
// overwritten (fft_avg[i] = sum, quisk.c:4726): pixels are done in order by one workgroup, reading pixel
// results for indices already written.
__global__ __launch_bounds__(256) void k_graph_release_inplace(ReleaseArgs a, double *pix)
{
    __shared__ double red[256];
    for (int i = 0; i < a.W; i++) {
        const int k0 = a.k0[i];
        double sum = 0.0;
        for (int j = threadIdx.x; j < a.n; j += 256) {
            const int kk = k0 + j;
            if (kk >= 0 && kk < a.N) sum += (kk < i) ? pix[kk] : avg_at(a, kk);
        }
        red[threadIdx.x] = sum;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) pix[i] = red[0];
        __syncthreads();
    }
    for (int i = threadIdx.x; i < a.W; i += 256) {
        double d = 20.0 * log10(pix[i]) - a.scale;
        if (d < -200) d = -200;
        a.out[i] = d;
    }
}
