// qk_fft.h -- display spectrum: window, forward complex FFT, magnitude average (get_graph, quisk.c:4602-4759;
// window record_app quisk.c:5088-5096).  The reference calls FFTW3 (fftw_execute_dft, quisk.c:4653); these are
// hand-written Stockham autosort FFTs staged in LDS, in double (or float in the engine's fp32 mode).
//
//   N <= 4096      one workgroup per frame (k_fft_small, radix-4 passes of lds_fft): window on load, all passes in
//                  LDS, epilogue accumulates |X|.  lds_fft also serves the audio-rate kernels at the end of this file.
//   N  > 4096      four-step, N = N1 x 4096, eight frames per launch pair, compile-time radix-16 passes (ft_pass16,
//                  ft_final, ft_run: 16 points per lane in registers, twiddles from four table entries and products):
//                  pass A  (k_fft_cols)  N1-point FFTs down the columns, window on load (one table product per lane
//                          times compile-time W16^j), twiddle W_N^(n2 k1) on store, Y[frame][k1][n2] to HBM;
//                  pass B  (k_fft_rows)  4096-point FFT of each row Y[k1][.] in three LDS exchanges; a workgroup walks
//                          the frames of its group with the next frame's row prefetched in registers, accumulates
//                          |X[k1 + N1 k2]| in registers and writes ONE partial average per group and launch, in the
//                          same [k1][k2] layout so every HBM access is contiguous.  The release kernel adds the
//                          parts and undoes the layout (and the fftshift) when it bins pixels.
// Algorithmic HBM bytes per frame: read 16 N (samples); the four-step adds 32 N for Y and 2 N for the partial
// averages (measured 56 MB per 2^20-point frame with tables and the release, profiles/r02_pmc_fft.json).
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
    void *Y;                   // four-step intermediate [frame][N1][4096], complex in the kernels' compute type
    double *avg;               // magnitude sum, layout [k1][k2] with k = k1 + N1 k2  (N1 = 1: natural order); four-step: [group][k1][k2]
    double *band;              // |X|^2 of the S-meter pass-band bins of this frame (quisk.c:4659-4680)
    int N, N1;
    int band_start, band_count;   // first bin (signed, relative to DC) and number of bins kept (whole + fractional)
    int frames, band_stride;      // four-step passes: frames per launch (consecutive in x and Y), doubles between the frames' band arrays
    int acc_parts;                // four-step pass B: group g adds to its partial average if g < acc_parts, else writes it (starts it)
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

// ------------------------------------------------------------------------------------------------
// The 4096-element tile of the four-step passes: 256 lanes x 16 elements, radix-16 butterflies in registers.
// A tile holds s interleaved sequences of length n (n s = 4096; element (p, q) at q + s p) and is transformed by
// Stockham passes of radix 16 while n >= 16 and one last pass of radix n in {2, 4, 8}.  The first pass takes its
// operands straight from global memory (`load`), the last one hands its results to `store`: a 4096-point row needs
// two trips through LDS, a 256-point column tile one, a 16-point one none.  LDS indices are padded by one element
// in sixteen, which makes the stride-16 stores of a radix-16 pass conflict-free (272 B apart instead of 256 B).
// `slot` (0..15) is the lane's compile-time register slot of the element, for callers that keep per-element state.
// ------------------------------------------------------------------------------------------------
#define FT16_THREADS 256
#define FT16_LDS (4096 + 256)
// workgroup barrier that waits for the wave's LDS traffic only: __syncthreads() also drains the vector-memory queue, which
// would make every pass wait for prefetched rows and for stores nobody reads
#define FT_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__device__ __forceinline__ int ft16_pad(int i) { return i + (i >> 4); }
// The tile code is written once for complex double (double2: the reference's arithmetic) and complex float (float2: the
// optional fp32 mode).  V = the vector type, FtReal<V> its scalar.
template <class V> struct FtRealOf;
template <> struct FtRealOf<double2> { typedef double type; };
template <> struct FtRealOf<float2> { typedef float type; };
template <class V> __device__ __forceinline__ V ft_mk(typename FtRealOf<V>::type x, typename FtRealOf<V>::type y) { V v; v.x = x; v.y = y; return v; }
template <class V> __device__ __forceinline__ V ft_cmul(V a, V b) { return ft_mk<V>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <class V> __device__ __forceinline__ V ft_tw(double2 w) { typedef typename FtRealOf<V>::type T; return ft_mk<V>((T)w.x, (T)w.y); }   // table entry in the compute type
template <class V>
__device__ __forceinline__ void ft_bfly2(V &a, V &b)
{
    const V t = ft_mk<V>(a.x - b.x, a.y - b.y);
    a = ft_mk<V>(a.x + b.x, a.y + b.y); b = t;
}
// forward 4-point DFT in place: (a, b, c, d) <- (y0, y1, y2, y3)
template <class V>
__device__ __forceinline__ void ft_bfly4(V &a, V &b, V &c, V &d)
{
    const V apc = ft_mk<V>(a.x + c.x, a.y + c.y), amc = ft_mk<V>(a.x - c.x, a.y - c.y);
    const V bpd = ft_mk<V>(b.x + d.x, b.y + d.y), bmd = ft_mk<V>(b.x - d.x, b.y - d.y);
    a = ft_mk<V>(apc.x + bpd.x, apc.y + bpd.y);
    c = ft_mk<V>(apc.x - bpd.x, apc.y - bpd.y);
    b = ft_mk<V>(amc.x + bmd.y, amc.y - bmd.x);         // amc - i bmd
    d = ft_mk<V>(amc.x - bmd.y, amc.y + bmd.x);         // amc + i bmd
}
#define FT_C8 0.92387953251128673848    /* cos(pi/8) */
#define FT_S8 0.38268343236508978178    /* sin(pi/8) */
#define FT_H  0.70710678118654752440    /* sqrt(1/2) */
template <class V>
__device__ __forceinline__ V ft_mul_h(V v, double cr_, double ci_)
{
    typedef typename FtRealOf<V>::type T;
    const T cr = (T)cr_, ci = (T)ci_;
    return ft_mk<V>(v.x * cr - v.y * ci, v.x * ci + v.y * cr);
}
// forward 16-point DFT of r[j], j = j0 + 4 j1; X[k] is left in r[(k >> 2) + 4 (k & 3)]
template <class V>
__device__ __forceinline__ void ft_radix16(V (&r)[16])
{
#pragma unroll
    for (int j0 = 0; j0 < 4; j0++) ft_bfly4(r[j0], r[j0 + 4], r[j0 + 8], r[j0 + 12]);     // r[j0 + 4 k1] = sum_j1 a W4^(j1 k1)
    // twiddles W16^(j0 k1)
    r[5] = ft_mul_h(r[5], FT_C8, -FT_S8);                   // 1
    r[9] = ft_mul_h(r[9], FT_H, -FT_H);                     // 2
    r[13] = ft_mul_h(r[13], FT_S8, -FT_C8);                 // 3
    r[6] = ft_mul_h(r[6], FT_H, -FT_H);                     // 2
    r[10] = ft_mk<V>(r[10].y, -r[10].x);                    // 4: -i
    r[14] = ft_mul_h(r[14], -FT_H, -FT_H);                  // 6
    r[7] = ft_mul_h(r[7], FT_S8, -FT_C8);                   // 3
    r[11] = ft_mul_h(r[11], -FT_H, -FT_H);                  // 6
    r[15] = ft_mul_h(r[15], -FT_C8, FT_S8);                 // 9
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) ft_bfly4(r[4 * k1], r[4 * k1 + 1], r[4 * k1 + 2], r[4 * k1 + 3]);   // r[k2 + 4 k1] = X[k1 + 4 k2]
}
template <int V> struct FtLog2 { static constexpr int value = 1 + FtLog2<V / 2>::value; };
template <> struct FtLog2<1> { static constexpr int value = 0; };
// W16^j = exp(-2 pi i j / 16) as compile-time constants (j is a constant after unrolling)
__device__ __forceinline__ constexpr double ft_cos16(int j)
{
    constexpr double c[16] = {1.0, FT_C8, FT_H, FT_S8, 0.0, -FT_S8, -FT_H, -FT_C8, -1.0, -FT_C8, -FT_H, -FT_S8, 0.0, FT_S8, FT_H, FT_C8};
    return c[j & 15];
}
__device__ __forceinline__ constexpr double ft_sin16(int j) { return ft_cos16(j + 12); }        // sin(2 pi j / 16)
// register slot of X[k] after ft_radix16
__device__ __forceinline__ constexpr int ft_slot16(int k) { return (k >> 2) + 4 * (k & 3); }

// One radix-16 Stockham pass with compile-time geometry: the tile holds S = 4096 / NN interleaved sequences of length NN.
// The lane's operands are the tile elements t + 256 j (always: NN / 16 * S = 256); X[k] goes to tile index
// q + S (16 p + k).  The fifteen twiddles W_NN^(p k) come from four table entries (k = 1, 2, 4, 8, loaded before the
// butterfly) and eleven products.  FROM_REGS: operands already in r; TO_REGS: X[k] stays in r[ft_slot16(k)].
template <int NN, bool FROM_REGS, bool TO_REGS, class V>
__device__ __forceinline__ void ft_pass16(V (&r)[16], V *buf, const double2 *__restrict__ tw4096)
{
    constexpr int S = 4096 / NN, SH = FtLog2<S>::value, SUB = NN / 16;
    const int t = threadIdx.x;
    const int p = t >> SH, q = t & (S - 1);
    V w1, w2, w4, w8;
    if constexpr (SUB > 1) {
        w1 = ft_tw<V>(tw4096[p * S]); w2 = ft_tw<V>(tw4096[2 * p * S]); w4 = ft_tw<V>(tw4096[4 * p * S]); w8 = ft_tw<V>(tw4096[8 * p * S]);
    }
    if constexpr (!FROM_REGS) {
        const int base = t + (t >> 4);
#pragma unroll
        for (int j = 0; j < 16; j++) r[j] = buf[base + 272 * j];
    }
    ft_radix16(r);
    if constexpr (SUB > 1) {
        const V w3 = ft_cmul(w1, w2), w5 = ft_cmul(w1, w4), w6 = ft_cmul(w2, w4), w7 = ft_cmul(w3, w4);
        r[ft_slot16(1)] = ft_cmul(r[ft_slot16(1)], w1);   r[ft_slot16(2)] = ft_cmul(r[ft_slot16(2)], w2);
        r[ft_slot16(3)] = ft_cmul(r[ft_slot16(3)], w3);   r[ft_slot16(4)] = ft_cmul(r[ft_slot16(4)], w4);
        r[ft_slot16(5)] = ft_cmul(r[ft_slot16(5)], w5);   r[ft_slot16(6)] = ft_cmul(r[ft_slot16(6)], w6);
        r[ft_slot16(7)] = ft_cmul(r[ft_slot16(7)], w7);   r[ft_slot16(8)] = ft_cmul(r[ft_slot16(8)], w8);
        r[ft_slot16(9)] = ft_cmul(r[ft_slot16(9)], ft_cmul(w1, w8));    r[ft_slot16(10)] = ft_cmul(r[ft_slot16(10)], ft_cmul(w2, w8));
        r[ft_slot16(11)] = ft_cmul(r[ft_slot16(11)], ft_cmul(w3, w8));  r[ft_slot16(12)] = ft_cmul(r[ft_slot16(12)], ft_cmul(w4, w8));
        r[ft_slot16(13)] = ft_cmul(r[ft_slot16(13)], ft_cmul(w5, w8));  r[ft_slot16(14)] = ft_cmul(r[ft_slot16(14)], ft_cmul(w6, w8));
        r[ft_slot16(15)] = ft_cmul(r[ft_slot16(15)], ft_cmul(w7, w8));
    }
    if constexpr (!TO_REGS) {
        if constexpr (!FROM_REGS) FT_BARRIER();       // every lane has read the tile before anyone overwrites it
        const int o0 = q + ((16 * p) << SH);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int o = o0 + (k << SH);
            buf[o + (o >> 4)] = r[ft_slot16(k)];
        }
        FT_BARRIER();
    }
}
// The last pass of a tile whose sequences are shorter than 16: radix NN in {2, 4, 8}, no twiddles.  The lane's elements are
// t + 256 m; the result for tile index t + 256 m is left in r[ft_final_slot<NN>(m)].
template <int NN> __device__ __forceinline__ constexpr int ft_final_slot(int m)
{
    if (NN == 8) { const int u = m & 1, k = m >> 1; return u + 4 * (k & 3) + 2 * (k >> 2); }
    return m;
}
template <int NN, bool FROM_REGS, class V>
__device__ __forceinline__ void ft_final(V (&r)[16], V *buf)
{
    if constexpr (!FROM_REGS) {
        const int t = threadIdx.x, base = t + (t >> 4);
#pragma unroll
        for (int m = 0; m < 16; m++) r[m] = buf[base + 272 * m];
    }
    if constexpr (NN == 2) {
#pragma unroll
        for (int u = 0; u < 8; u++) ft_bfly2(r[u], r[u + 8]);
    } else if constexpr (NN == 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) ft_bfly4(r[u], r[u + 4], r[u + 8], r[u + 12]);
    } else {    // 8 = 2 x 4: element j = m >> 1 of butterfly u = m & 1; j = j0 + 2 j1
#pragma unroll
        for (int u = 0; u < 2; u++) {
            ft_bfly4(r[u], r[u + 4], r[u + 8], r[u + 12]);              // j0 = 0: j = 0, 2, 4, 6 -> k1 at r[u + 4 k1]
            ft_bfly4(r[u + 2], r[u + 6], r[u + 10], r[u + 14]);         // j0 = 1: j = 1, 3, 5, 7 -> k1 at r[u + 2 + 4 k1]
            r[u + 6] = ft_mul_h(r[u + 6], FT_H, -FT_H);                 // W8^1
            r[u + 10] = ft_mk<V>(r[u + 10].y, -r[u + 10].x);            // W8^2 = -i
            r[u + 14] = ft_mul_h(r[u + 14], -FT_H, -FT_H);              // W8^3
#pragma unroll
            for (int k1 = 0; k1 < 4; k1++) ft_bfly2(r[u + 4 * k1], r[u + 2 + 4 * k1]);   // X[k1] at r[u + 4 k1], X[k1 + 4] at r[u + 2 + 4 k1]
        }
    }
}
// The whole tile: radix-16 passes while the sequences have at least 16 elements, then the short last pass.  In: r[j] = tile
// element t + 256 j.  Out: the transform's value at tile index t + 256 m (= sequence q, frequency pos, index q + S pos) is
// r[ft_out_slot<NN>(m)].  A 4096-point row makes two trips through LDS, a 256-point column tile one, a 16-point one none.
// The caller puts a barrier between two tiles that use the same buffer.
template <int NN, bool FIRST = true, class V>
__device__ __forceinline__ void ft_run(V (&r)[16], V *buf, const double2 *__restrict__ tw4096)
{
    if constexpr (NN >= 16) {
        ft_pass16<NN, FIRST, NN == 16, V>(r, buf, tw4096);
        if constexpr (NN > 16) ft_run<NN / 16, false, V>(r, buf, tw4096);
    } else if constexpr (NN > 1) {
        ft_final<NN, FIRST, V>(r, buf);
    }
}
template <int NN> __device__ __forceinline__ constexpr int ft_out_slot(int m)
{
    int n = NN;
    while (n > 16) n /= 16;
    return n == 16 ? ft_slot16(m) : n == 8 ? ft_final_slot<8>(m) : m;
}

// Four-step FFT of the display spectrum, N = N1 x 4096, F frames per launch:
//   pass A (k_fft_cols<N1>, blockIdx.y = frame)  N1-point FFTs down the columns, 4096/N1 columns per workgroup; periodic Hann
//        window on load, computed from the twiddle tables: w[i] = 0.5 + 0.5 cos(2 pi (i - N/2) / N) = 0.5 - 0.5 Re W_N^i
//        (quisk.c:5091-5096; no 8 B/sample window table is streamed).  The lane's sixteen samples are 4096 N1/16 apart, so
//        W_N^i = (one table product per lane) x W16^j with compile-time W16.  Twiddle W_N^(n2 k1) on store; Y[f][k1][n2] to HBM.
//   pass B (k_fft_rows, blockIdx.y = group)  4096-point FFT of each row Y[f][k1][.]; the workgroup walks the frames f = g,
//        g + G, ... of its row (the next one's loads in flight while this one is transformed) and adds up |X| in registers:
//        the average is read-modified-written once per launch, not per frame.  Group g keeps its own partial average
//        avg[g][k1][k2] (summed by the release kernel), so two workgroups per CU work on different frames of a row.
template <int N1, class V = double2>
__global__ __launch_bounds__(FT16_THREADS, 2) void k_fft_cols(FftArgs a)
{
    typedef typename FtRealOf<V>::type T;
    __shared__ __attribute__((aligned(16))) V buf[FT16_LDS];
    constexpr int TW = 4096 / N1, TSH = FtLog2<TW>::value;
    const int t = threadIdx.x, c0 = blockIdx.x * TW;
    const double2 *__restrict__ x = a.x + (long long)blockIdx.y * a.N;
    V *__restrict__ Y = static_cast<V *>(a.Y) + (long long)blockIdx.y * a.N;
    const double2 *__restrict__ tlo = a.tw_lo, *__restrict__ thi = a.tw_hi;
    V r[16];
    if constexpr (N1 >= 16) {
        const int gb = (t >> TSH) * 4096 + c0 + (t & (TW - 1));          // x[4096 n1 + n2], n1 = (t >> TSH) + (N1 / 16) j
        const double2 wb = cmul_d(thi[gb >> 10], tlo[gb & 1023]);        // W_N^gb
        double2 xin[16];
#pragma unroll
        for (int j = 0; j < 16; j++) xin[j] = x[gb + j * (N1 / 16) * 4096];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const double w = 0.5 - 0.5 * (wb.x * ft_cos16(j) + wb.y * ft_sin16(j));       // Re(wb W16^j)
            r[j] = ft_mk<V>((T)(xin[j].x * w), (T)(xin[j].y * w));
        }
    } else {
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int idx = t + 256 * m;
            const int g = (idx >> TSH) * 4096 + c0 + (idx & (TW - 1));
            const double2 v = x[g];
            const double2 wh = thi[g >> 10], wl = tlo[g & 1023];
            const double w = 0.5 - 0.5 * (wh.x * wl.x - wh.y * wl.y);
            r[m] = ft_mk<V>((T)(v.x * w), (T)(v.y * w));
        }
    }
    ft_run<N1, true, V>(r, buf, a.tw4096);
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const int o = t + 256 * m;
        const int k1 = o >> TSH, n2 = c0 + (o & (TW - 1));
        const int tt = n2 * k1;                                          // < N <= 2^24
        const double2 w = cmul_d(thi[tt >> 10], tlo[tt & 1023]);
        Y[k1 * 4096 + n2] = ft_cmul(r[ft_out_slot<N1>(m)], ft_tw<V>(w));
    }
}

// |X| = sqrt(s), s = re^2 + im^2 >= 0 far from the ends of the exponent range (samples are below 2^31, N <= 2^24): the
// reciprocal square root seed and two coupled Newton steps, without the scaling a general sqrt needs; s = 0 gives 0.
__device__ __forceinline__ double ft_sqrt_pos(double s)
{
    const double y = __builtin_amdgcn_rsq(fmax(s, 1e-280));
    double g = s * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double d = fma(-g, g, s);
    return fma(d, h, g);
}
// pass B: blockIdx.x = k1, blockIdx.y = frame group
template <class V = double2>
__global__ __launch_bounds__(FT16_THREADS, 2) void k_fft_rows(FftArgs a)
{
    __shared__ __attribute__((aligned(16))) V buf[FT16_LDS];
    const int t = threadIdx.x, k1 = blockIdx.x, g = blockIdx.y, G = gridDim.y;
    const V *__restrict__ Yrow = static_cast<const V *>(a.Y) + (long long)k1 * 4096 + t;
    double acc[16];
    V nxt[16], r[16];
#pragma unroll
    for (int m = 0; m < 16; m++) acc[m] = 0.0;
    if (g < a.frames) {
        const V *__restrict__ row = Yrow + (long long)g * a.N;
#pragma unroll
        for (int m = 0; m < 16; m++) nxt[m] = row[256 * m];
    }
    for (int f = g; f < a.frames; f += G) {
#pragma unroll
        for (int m = 0; m < 16; m++) r[m] = nxt[m];
        if (f + G < a.frames) {
            const V *__restrict__ row = Yrow + (long long)(f + G) * a.N;
#pragma unroll
            for (int m = 0; m < 16; m++) nxt[m] = row[256 * m];
        }
        if (f != g) FT_BARRIER();
        ft_run<4096, true, V>(r, buf, a.tw4096);
        // X[k1 + N1 e], e = t + 256 k, is r[ft_slot16(k)]
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const double xr = (double)r[ft_slot16(k)].x, xi = (double)r[ft_slot16(k)].y;
            acc[k] += ft_sqrt_pos(fma(xr, xr, xi * xi));                       // cabs, quisk.c:4707-4709; the group's frames in order
        }
        if (a.band_count > 0) {
            double *__restrict__ band = a.band + (long long)f * a.band_stride;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const double xr = (double)r[ft_slot16(k)].x, xi = (double)r[ft_slot16(k)].y;
                const int kk = k1 + a.N1 * (t + 256 * k);
                const int sk = kk < a.N / 2 ? kk : kk - a.N;             // signed bin number
                const int j = sk - a.band_start;
                if (j >= 0 && j < a.band_count) band[j] = xr * xr + xi * xi;   // c * conj(c), quisk.c:4673
            }
        }
    }
    if (g < a.frames) {
        double *__restrict__ avg = a.avg + (long long)g * a.N + (long long)k1 * 4096 + t;
        if (g < a.acc_parts) {
#pragma unroll
            for (int m = 0; m < 16; m++) avg[256 * m] += acc[m];
        } else {
#pragma unroll
            for (int m = 0; m < 16; m++) avg[256 * m] = acc[m];
        }
    }
}

// S-meter accumulation of `frames` frames, in the reference's order (quisk.c:4668-4679): whole bins, then the
// fractional bin, frame after frame.  One thread.
__global__ void k_meter_accum(const double *band, int n_whole, double frac, double *meter, int frames, int band_stride)
{
#pragma clang fp contract(off)
    if (threadIdx.x || blockIdx.x) return;
    double m = *meter;
    for (int f = 0; f < frames; f++, band += band_stride) {
        for (int j = 0; j < n_whole; j++) m = m + band[j];
        m = m + band[n_whole] * frac;
    }
    *meter = m;
}

// Release (quisk.c:4713-4753): pixel i sums n average bins starting at shifted index k0[i]; then dB scaling.
// One wave per pixel.  `shifted` index kk counts from -fs/2: X index k = (kk + N/2) mod N.
struct ReleaseArgs {
    const double *avg; double *out; const int *k0;
    int N, N1, W, n;
    double scale;              // 20 (log10 count + log10 N + 31 log10 2)
    int parts;                 // partial averages to add up (four-step pass B keeps one per frame group)
    double *meter;             // S-meter accumulator: copied to out[W] and cleared (quisk.c:4728-4737 uses it on the host)
    int shifted;               // 1: avg is the stitched scan average, already in order of frequency (index = kk)
    double *host;              // pinned host copy of the row and the meter sum, written by the kernel itself (no copy command)
};
__device__ __forceinline__ double avg_at(const ReleaseArgs &a, int kk)
{
    if (a.shifted) return a.avg[kk];
    int k = kk + a.N / 2;
    if (k >= a.N) k -= a.N;
    const int k1 = k % a.N1, k2 = k / a.N1;
    const double *p = a.avg + (long long)k1 * (a.N / a.N1) + k2;
    double v = p[0];
    for (int g = 1; g < a.parts; g++) v += p[(long long)g * a.N];
    return v;
}
// Scan stitching (change_scan; quisk.c:4682-4702): the frame of block b contributes its valid centre part, scan_blocks
// neighbouring bins summed into one, to positions m0 + b deltam ... of the stitched average (assigned, not added).
// `a` describes the magnitudes of this one frame (the average buffers hold nothing else in scan mode).
__global__ __launch_bounds__(256) void k_scan_stitch(ReleaseArgs a, double *scan_avg, int m0, int deltam, int block, int scan_blocks)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= deltam) return;
    double d = 0.0;
    for (int n = 0; n < scan_blocks; n++) d += avg_at(a, m0 + j * scan_blocks + n);
    scan_avg[m0 + block * deltam + j] = d;
}

// Sixteen pixels per workgroup: with zoom 1 they share the 128-byte lines of the [k1][k2] average (neighbouring k2 of one k1).
#define REL_PIX 16
__global__ __launch_bounds__(64 * REL_PIX) void k_graph_release(ReleaseArgs a)
{
    const int i = blockIdx.x * REL_PIX + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) { const double m = *a.meter; a.out[a.W] = m; a.host[a.W] = m; *a.meter = 0.0; }
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
        a.host[i] = d;
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
    if (threadIdx.x == 0) { const double m = *a.meter; a.out[a.W] = m; a.host[a.W] = m; *a.meter = 0.0; }
    for (int i = threadIdx.x; i < a.W; i += 256) {
        double d = 20.0 * log10(pix[i]) - a.scale;
        if (d < -200) d = -200;
        a.out[i] = d;
        a.host[i] = d;
    }
}

// ------------------------------------------------------------------------------------------------
// Generic power-of-two complex FFT in HBM, natural order in and out, with the element-wise steps of its
// callers fused into the load and the store.  It serves the spectra that are not the 2^k display FFT: any
// other fft_size (Quisk's fft_size = data_width * fft_mult has factors 3, 5, 7 ..., quisk.py:3858-3890), the
// sub-receiver graphs (get_multirx_graph, quisk.c:4468-4530: 8 * multirx_data_width points) and the bandscope
// (get_bandscope, quisk.c:4532-4600).  A length N that is not a power of two goes through Bluestein's
// identity  X[k] = c[k] * sum_n (x[n] c[n]) conj(c[k-n]),  c[n] = exp(-i pi n^2 / N):  one forward and one
// inverse FFT of M >= 2N-1 points; the chirp and the transformed kernel are made once per size on the host
// (n^2 reduced mod 2N in integers, so the phases are exact to the last bit of cos/sin).
//   load :  v = in[e] (zero for e >= n_in; a real input has zero imaginary part); v *= pre[e]; inverse: conj
//   store:  inverse: conj and scale; r *= post[k]; stored for k < n_out
// M <= 4096 runs in one workgroup; larger M is the four-step above with M = N1 x 4096 and a strided final store.
// ------------------------------------------------------------------------------------------------
struct GfftArgs {
    const double2 *in; const double *in_real; double2 *out; double2 *Y;
    const double2 *pre, *post;
    const double2 *tw4096, *tw_lo, *tw_hi;
    int M, N1, n_in, n_out, inverse;
    double scale;
};
__device__ __forceinline__ double2 gfft_load(const GfftArgs &a, int e)
{
    double2 v = make_double2(0.0, 0.0);
    if (e < a.n_in) {
        v = a.in_real ? make_double2(a.in_real[e], 0.0) : a.in[e];
        if (a.pre) v = cmul_d(v, a.pre[e]);
        if (a.inverse) v.y = -v.y;
    }
    return v;
}
__device__ __forceinline__ void gfft_store(const GfftArgs &a, int k, double2 r)
{
    if (k >= a.n_out) return;
    if (a.inverse) r = make_double2(r.x * a.scale, -r.y * a.scale);
    if (a.post) r = cmul_d(r, a.post[k]);
    a.out[k] = r;
}
template <int TOTAL>
__global__ __launch_bounds__(FFT_THREADS) void k_gfft_small(GfftArgs a)
{
    __shared__ __attribute__((aligned(16))) double2 buf[TOTAL];
    for (int e = threadIdx.x; e < TOTAL; e += FFT_THREADS) buf[e] = gfft_load(a, e);
    lds_fft<TOTAL>(buf, TOTAL, 1, a.tw4096);
    for (int e = threadIdx.x; e < TOTAL; e += FFT_THREADS) gfft_store(a, e, buf[e]);
}
__global__ __launch_bounds__(FFT_BIG_THREADS) void k_gfft_cols(GfftArgs a)
{
    __shared__ __attribute__((aligned(16))) double2 buf[4096];
    const int N1 = a.N1, TW = 4096 / N1;
    const int c0 = blockIdx.x * TW;
    for (int e = threadIdx.x; e < 4096; e += FFT_BIG_THREADS) {
        const int n1 = e / TW, col = e - n1 * TW;
        buf[e] = gfft_load(a, n1 * 4096 + c0 + col);
    }
    lds_fft<4096, FFT_BIG_THREADS>(buf, N1, TW, a.tw4096);
    for (int e = threadIdx.x; e < 4096; e += FFT_BIG_THREADS) {
        const int k1 = e / TW, col = e - k1 * TW;
        const int n2 = c0 + col;
        const long long t = (long long)n2 * k1;
        const double2 w = cmul_d(a.tw_hi[t >> 10], a.tw_lo[t & 1023]);
        a.Y[(long long)k1 * 4096 + n2] = cmul_d(buf[e], w);
    }
}
__global__ __launch_bounds__(FFT_BIG_THREADS) void k_gfft_rows(GfftArgs a)
{
    __shared__ __attribute__((aligned(16))) double2 buf[4096];
    const int k1 = blockIdx.x;
    const double2 *row = a.Y + (long long)k1 * 4096;
    for (int e = threadIdx.x; e < 4096; e += FFT_BIG_THREADS) buf[e] = row[e];
    lds_fft<4096, FFT_BIG_THREADS>(buf, 4096, 1, a.tw4096);
    for (int e = threadIdx.x; e < 4096; e += FFT_BIG_THREADS) gfft_store(a, k1 + a.N1 * e, buf[e]);
}

// get_graph's accumulation for a spectrum X[0..N) in natural order (quisk.c:4668-4680, 4703-4710).
__global__ __launch_bounds__(256) void k_graph_accum(const double2 *X, FftArgs a)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < a.N) fft_epilogue(a, 0, e, X[e]);
}

// get_multirx_graph (quisk.c:4491-4519): pixel k sums the magnitudes of MULT consecutive bins in order of
// frequency, starting at bin N/2 (the most negative frequency), then 20 log10 - scale, floor -200.
__global__ __launch_bounds__(256) void k_multirx_bins(const double2 *X, int N, int mult, int width, double scale, double *out)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= width) return;
    double d1 = 0.0;
    for (int t = 0; t < mult; t++) {
        int i = N / 2 + k * mult + t;
        if (i >= N) i -= N;
        const double2 v = X[i];
        d1 += hypot(v.x, v.y);
    }
    double d2 = 20.0 * log10(d1) - scale;
    if (d2 < -200) d2 = -200;
    out[k] = d2;
}

// Bandscope (get_bandscope, quisk.c:4545-4556): largest |sample| of the frame (before the window), as the
// ordered bit pattern of a non-negative double; and the magnitude accumulation of bins 0..L-1.
__global__ __launch_bounds__(256) void k_absmax(const double *x, int n, unsigned long long *mx)
{
    double m = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { const double v = fabs(x[i]); if (v > m) m = v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(m, o, 64); m = m < w ? w : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(mx, (unsigned long long)__double_as_longlong(m));
}
__global__ __launch_bounds__(256) void k_bscope_accum(const double2 *X, int L, double *avg)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < L) avg[i] += hypot(X[i].x, X[i].y);
}
// Release (quisk.c:4560-4588): pixel i integrates the average from d1 to d2 (fractional bins at both ends), scales
// by 1 / frac / count / size, and converts to dB with the reference's 1e-10 floor.  avg[L] is zero ("in case we
// run off the end").  One thread per pixel, the inner sum in the reference's order.
struct BscopeRelease { const double *avg; double *out; int L, width; double zoom, deltaf, rate, scale; };
__global__ __launch_bounds__(256) void k_bscope_release(BscopeRelease a)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.width) return;
    const double f1 = a.deltaf + a.rate / 2.0 * (1.0 - a.zoom);
    const double d1 = a.L / a.rate * (f1 + (double)i / a.width * a.zoom * a.rate);
    const double d2 = a.L / a.rate * (f1 + (double)(i + 1) / a.width * a.zoom * a.rate);
    const int j1 = (int)floor(d1), j2 = (int)floor(d2);
    auto at = [&](int j) { return (j >= 0 && j <= a.L) ? a.avg[j] : 0.0; };   // the reference reads out of bounds there
    double sample;
    if (j1 == j2) {
        sample = (d2 - d1) * at(j1);
    } else {
        sample = (j1 + 1 - d1) * at(j1);
        for (int j = j1 + 1; j < j2; j++) sample += at(j);
        sample += (d2 - j2) * at(j2);
    }
    sample = sample * a.scale;
    a.out[i] = sample <= 1E-10 ? -200.0 : 20.0 * log10(sample);
}

// ------------------------------------------------------------------------------------------------
// Audio-rate hooks between the detector and the audio interpolators (quisk.c:1792-1793, 1829-1834, 1879-1884, 1909-1910,
// 1940-1941): the SSB squelch and the automatic notch filter.  Both work on frames of the receiver's audio that are
// transformed one after the other with state carried from frame to frame (a hysteresis, a one-second timer): one
// workgroup per receiver walks the frames of a block in order; the transforms inside a frame are the LDS FFT above.
// ------------------------------------------------------------------------------------------------
#define SSQ_N 512                  /* SQUELCH_FFT_SIZE, quisk.c:36 */
struct SsbSquelchArgs {
    StreamView au;                 // the audio stream at the filter rate (one plane)
    int n, samp_rate, level;       // new samples of this block; quisk_filter_srate; ssb_squelch_level
    const int *rx_list; const int *bandwidth;     // [rx] filter bandwidth (the reference reads bank 0's, quisk.c:1056)
    double *carry;                 // [rx][SSQ_N] samples of the unfinished frame (MS->in_fft)
    int *state;                    // [rx][4]: index, sq_open, plan made, -
    int *flag;                     // [rx] squelch_active of this block
    const double2 *tw4096;
};
__device__ __forceinline__ double block_sum(double v, double *red)
{
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    return red[0];
}
__global__ __launch_bounds__(256) void k_ssb_squelch(SsbSquelchArgs a)
{
    __shared__ __attribute__((aligned(16))) double2 buf[SSQ_N];
    __shared__ double red[256];
    const int rx = a.rx_list ? a.rx_list[blockIdx.x] : blockIdx.x;
    int *st = a.state + 4 * rx;
    if (!st[2]) {                  // the first call only makes the plan (quisk.c:1040-1048): the block is not analysed and the
        __syncthreads();           // flag keeps the 0 that quisk.c:1777 gave it
        if (threadIdx.x == 0) { st[0] = 0; st[1] = 0; st[2] = 1; a.flag[rx] = 0; }
        return;
    }
    const int idx0 = st[0];
    const double *x = sv_ptr(a.au, rx, 0);
    double *carry = a.carry + (long long)rx * SSQ_N;
    const int frames = (idx0 + a.n) / SSQ_N;
    int bw = a.bandwidth[rx];
    if (bw > 3000) bw = 3000;
    const int bw1 = 300 * SSQ_N / a.samp_rate, bw2 = (bw + 300) * SSQ_N / a.samp_rate;
    bool opened = false;
    for (int f = 0; f < frames; f++) {
        __syncthreads();
        for (int j = threadIdx.x; j < SSQ_N; j += 256) {
            const int pos = f * SSQ_N + j;
            const double v = pos < idx0 ? carry[pos] : x[pos - idx0];
            buf[j] = make_double2(v * (0.50 - 0.50 * cos(2. * M_PI * j / SSQ_N)), 0.0);
        }
        lds_fft<SSQ_N>(buf, SSQ_N, 1, a.tw4096);
        double ar = 0.0, ge = 0.0;
        for (int i = bw1 + (int)threadIdx.x; i < bw2; i += 256) {
            const double re = buf[i].x / 32767, im = buf[i].y / 32767;          // CLIP16, quisk.c:1064
            const double d = re * re + im * im;
            if (d > 1E-4) { ar += d; ge += log(d); }
        }
        ar = block_sum(ar, red);
        ge = block_sum(ge, red);
        double ratio = 1.0;
        if (ar > 1E-4) { const int nb = bw2 - bw1; ratio = log(ar / nb) - ge / nb; }
        if (ratio > a.level * 0.005) opened = true;                            // MS->sq_open = samp_rate: a one second timer
    }
    __syncthreads();
    // what is left of (carry + block) behind the last whole frame becomes the new carry
    const int rest = idx0 + a.n - frames * SSQ_N;
    double keep[2];
    for (int u = 0; u < 2; u++) {
        const int j = threadIdx.x + 256 * u;
        const int pos = frames * SSQ_N + j;
        keep[u] = (j < rest) ? (pos < idx0 ? carry[pos] : x[pos - idx0]) : 0.0;
    }
    __syncthreads();
    for (int u = 0; u < 2; u++) { const int j = threadIdx.x + 256 * u; if (j < rest) carry[j] = keep[u]; }
    if (threadIdx.x == 0) {
        int open = opened ? a.samp_rate : st[1];
        open -= a.n;
        if (open < 0) open = 0;
        st[0] = rest; st[1] = open;
        a.flag[rx] = open == 0;
    }
}

#define AN_N 2048                  /* NOTCH_DATA_SIZE and the sizes quisk.c:723-729 derive from it */
#define AN_DESIGN (AN_N / 4)
#define AN_TAPS (AN_DESIGN - 1)
#define AN_FBINS (AN_TAPS / 2 + 1)
#define AN_START (AN_TAPS - 1)
#define AN_OUT (AN_N - AN_START)
#define AN_BINS (AN_N / 2 + 1)
struct NotchState {                // dAutoNotch's function statics (quisk.c:734-746), per receiver
    int index, old1, count1, old2, count2, sig, pad0, pad1;
    double data_in[AN_N], data_out[AN_N], average[AN_BINS + 7];
    double2 fltr_fft[AN_BINS + 1];
};
struct NotchArgs {
    StreamView au; int n, rate, sidetone;
    const int *rx_list; const int *on;      // [rx] auto-notch enabled
    const int *mode;                        // [rx] rx_mode_type: only CWL / CWU pass a sidetone (quisk.c:1793, 1812; 0 elsewhere)
    NotchState *state;                      // [rx]
    const double2 *tw4096;
};
// largest value of v over the workgroup with the lowest index among equals; entries with v <= 0 never win (the reference starts
// its search at d1 = 0, i1 = 0 with a strict comparison)
__device__ __forceinline__ int block_argmax(double v, int i, double *redv, int *redi)
{
    __syncthreads();
    redv[threadIdx.x] = v; redi[threadIdx.x] = i;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double w = redv[threadIdx.x + o]; const int wi = redi[threadIdx.x + o];
            if (w > redv[threadIdx.x] || (w == redv[threadIdx.x] && wi < redi[threadIdx.x])) { redv[threadIdx.x] = w; redi[threadIdx.x] = wi; }
        }
        __syncthreads();
    }
    return redv[0] > 0.0 ? redi[0] : 0;
}
__global__ __launch_bounds__(256) void k_auto_notch(NotchArgs a)
{
    __shared__ __attribute__((aligned(16))) double2 buf[AN_N];
    __shared__ __attribute__((aligned(16))) double2 spec[AN_BINS + 1];
    __shared__ double redv[256];
    __shared__ int redi[256];
    __shared__ int sh[4];
    const int rx = a.rx_list ? a.rx_list[blockIdx.x] : blockIdx.x;
    if (!a.on[rx]) return;
    NotchState *s = a.state + rx;
    double *x = sv_ptr(a.au, rx, 0);
    const int t = threadIdx.x;
    int index = s->index, inp = 0;
    const int rate = a.rate;
    const int delta_sig = (300 * 2 * AN_BINS + rate / 2) / rate, delta_i1 = (400 * 2 * AN_BINS + rate / 2) / rate;
    const int sidetone = a.mode[rx] <= 1 ? a.sidetone : 0;
    const int signal = sidetone != 0 ? (abs(sidetone) * 2 * AN_BINS + rate / 2) / rate : -999;
    while (inp < a.n) {
        const int chunk = min(a.n - inp, AN_N - index);
        for (int j = t; j < chunk; j += 256) {           // quisk.c:776-777: the sample goes in, the filtered one of a frame ago comes out
            const double v = x[inp + j];
            s->data_in[index + j] = v;
            x[inp + j] = s->data_out[index + j];
        }
        index += chunk; inp += chunk;
        __syncthreads();
        if (index < AN_N) break;
        index = AN_START;
        // ---- a full frame: forward transform (quisk.c:780) ----
        for (int i = t; i < AN_N; i += 256) buf[i] = make_double2(s->data_in[i], 0.0);
        lds_fft<AN_N>(buf, AN_N, 1, a.tw4096);
        for (int i = t; i < AN_BINS; i += 256) spec[i] = buf[i];
        // ---- the two strongest lines of the smoothed spectrum, away from the CW signal and from each other (quisk.c:792-835) ----
        double bv = 0.0; int bi = 0;
        for (int i = t; i < AN_BINS; i += 256) {
            const double d = hypot(buf[i].x, buf[i].y);
            const double av = 0.5 * s->average[i] + 0.5 * d;
            s->average[i] = av;
            if (abs(i - signal) > delta_sig && av > bv) { bv = av; bi = i; }
        }
        const int i1 = block_argmax(bv, bi, redv, redi);
        bv = 0.0; bi = 0;
        for (int i = t; i < AN_BINS; i += 256) {
            const double av = s->average[i];
            if (abs(i - signal) > delta_sig && abs(i - i1) > delta_i1 && av > bv) { bv = av; bi = i; }
        }
        const int i2 = block_argmax(bv, bi, redv, redi);
        if (t == 0) {
            int c1 = s->count1, c2 = s->count2;
            if (abs(i1 - s->old1) < 3) c1++; else c1--;
            if (c1 > 4) c1 = 4; else if (c1 < -1) c1 = -1;
            if (c1 < 0) s->old1 = i1;
            if (abs(i2 - s->old2) < 3) c2++; else c2--;
            if (c2 > 4) c2 = 4; else if (c2 < -2) c2 = -2;
            if (c2 < 0) s->old2 = i2;
            s->count1 = c1; s->count2 = c2;
            int k = 0;
            if (c1 > 0 && c2 > 0) k = i1 + 10000 * i2; else if (c1 > 0) k = i1;
            sh[0] = s->sig != k; sh[1] = c1 > 0; sh[2] = c1 > 0 && c2 > 0;
            s->sig = k;
        }
        __syncthreads();
        if (sh[0]) {               // ---- make the filter (quisk.c:844-886) ----
            int half_width = (100 * 2 * AN_FBINS + rate / 2) / rate;
            if (half_width < 3) half_width = 3;
            const int k1 = (i1 + 2) / 4, k2 = (i2 + 2) / 4;
            // the design spectrum: bins 0 .. 255 ones with the notches cut; bin 256 keeps the real part that the last
            // 2048-point filter spectrum left in fltr_fft[256] (the reference designs in that array); Hermitian, real: its inverse
            // real transform is the forward transform's real part
            for (int k = t; k <= AN_DESIGN / 2; k += 256) {
                double v = 1.0;
                if (k == AN_DESIGN / 2) v = s->fltr_fft[AN_DESIGN / 2].x;
                else if ((sh[1] && abs(k - k1) <= half_width) || (sh[2] && abs(k - k2) <= half_width)) v = 0.0;
                buf[k] = make_double2(v, 0.0);
                if (k > 0 && k < AN_DESIGN / 2) buf[AN_DESIGN - k] = make_double2(v, 0.0);
            }
            lds_fft<AN_DESIGN>(buf, AN_DESIGN, 1, a.tw4096);
            // centre the response, mirror it, drop one tap (quisk.c:878-880, with what its overlapping copies really leave)
            double tap[2];
            for (int u = 0; u < 2; u++) {
                const int i = t + 256 * u;
                double v = 0.0;
                if (i < AN_TAPS) {
                    int src;
                    if (i >= AN_DESIGN / 2 - 1) src = (i <= AN_DESIGN - 4) ? i - (AN_DESIGN / 2 - 1) : i;     // moved block / untouched tail
                    else { const int j = AN_DESIGN - 2 - i; src = (j <= AN_DESIGN - 4) ? j - (AN_DESIGN / 2 - 1) : j; }
                    v = buf[src].x * (0.50 - 0.50 * cos(2. * M_PI * i / (AN_TAPS))) / AN_DESIGN;
                }
                tap[u] = v;
            }
            __syncthreads();
            for (int i = t; i < AN_N; i += 256) buf[i] = make_double2(0.0, 0.0);
            __syncthreads();
            for (int u = 0; u < 2; u++) { const int i = t + 256 * u; if (i < AN_TAPS) buf[i] = make_double2(tap[u], 0.0); }
            lds_fft<AN_N>(buf, AN_N, 1, a.tw4096);
            for (int i = t; i < AN_BINS; i += 256) s->fltr_fft[i] = buf[i];
            __syncthreads();
        }
        // ---- apply the filter and transform back (quisk.c:890-892): conj(FFT(conj X)) of the Hermitian spectrum ----
        for (int i = t; i < AN_BINS; i += 256) {
            const double2 y = cmul_d(spec[i], s->fltr_fft[i]);
            if (i == 0 || i == AN_N / 2) buf[i] = make_double2(y.x, 0.0);
            else { buf[i] = make_double2(y.x, -y.y); buf[AN_N - i] = y; }
        }
        lds_fft<AN_N>(buf, AN_N, 1, a.tw4096);
        // ---- slide the input, scale the output (quisk.c:893-895) ----
        double keep[2];
        for (int u = 0; u < 2; u++) { const int i = t + 256 * u; keep[u] = i < AN_START ? s->data_in[AN_OUT + i] : 0.0; }
        __syncthreads();
        for (int u = 0; u < 2; u++) { const int i = t + 256 * u; if (i < AN_START) s->data_in[i] = keep[u]; }
        for (int i = AN_START + t; i < AN_N; i += 256) s->data_out[i] = buf[i].x / (AN_N / 20);
        __syncthreads();
    }
    if (t == 0) s->index = index;
}
