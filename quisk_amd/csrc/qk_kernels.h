// qk_kernels.h -- gfx950 device code of the receive-DSP engine.
//
// Data layout in HBM ("streams"): every stage of the chain writes a PLANAR stream
//     double d[receiver][plane][hist + cap]
// plane 0 = real part, plane 1 = imaginary part (real-valued audio streams have one plane).  New samples of
// a block start at offset `hist`; the `hist` doubles in front of them hold the tail of the previous
// block, so a consumer reads x[i - k] for k up to its tap count with plain negative offsets and no
// ring arithmetic.  All the FIRs of the reference have REAL coefficients (filter.c), so a complex
// stream is two independent real streams and is processed plane by plane.
//
// Every kernel cites the reference lines whose arithmetic it reproduces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define QK_CLIP32 2147483647.0

struct StreamView {            // passed by value to kernels
    double *d;
    long long rx_stride;       // doubles between receivers
    int plane_stride;          // doubles between planes (= hist + cap)
    int hist;
};
__device__ __forceinline__ double *sv_ptr(const StreamView &s, int rx, int plane)
{
    return s.d + (long long)rx * s.rx_stride + (long long)plane * s.plane_stride + s.hist;
}

// a stream whose elements are X (float in the fp32 mode's front chain): the same element indices on the same allocation
template <class X>
__device__ __forceinline__ X *sv_ptr_t(const StreamView &s, int rx, int plane)
{
    return reinterpret_cast<X *>(s.d) + (long long)rx * s.rx_stride + (long long)plane * s.plane_stride + s.hist;
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Compute type of the FIR kernels: double (the reference's arithmetic) or float (the optional fp32 mode of BASELINE
// configs[4]: fp32 samples, taps and accumulators inside the front end and the tiled FIRs; the mixer phasor, the streams
// in HBM and everything behind the filters stay double).
template <class T> struct QkVec2;
template <> struct QkVec2<double> { typedef double2 type; };
template <> struct QkVec2<float> { typedef float2 type; };
// One complex float sample as a 64-bit register pair (re, im): the packed FP32 operations (v_pk_add_f32, v_pk_mul_f32,
// v_pk_fma_f32) then filter both planes with one instruction (k_frontend_pk).
typedef float qk_f2 __attribute__((ext_vector_type(2)));
typedef float qk_f4 __attribute__((ext_vector_type(4)));
template <> struct QkVec2<qk_f2> { typedef qk_f4 type; };
__device__ __forceinline__ qk_f2 fma(qk_f2 a, qk_f2 b, qk_f2 c) { return __builtin_elementwise_fma(a, b, c); }
template <class T> __device__ __forceinline__ T hb_c(double c) { return T(c); }
template <> __device__ __forceinline__ qk_f2 hb_c<qk_f2>(double c) { const float f = (float)c; return qk_f2{f, f}; }
template <class V, class T> __device__ __forceinline__ void qk_unpack2(const V &v, T &a, T &b) { a = v.x; b = v.y; }
template <> __device__ __forceinline__ void qk_unpack2<qk_f4, qk_f2>(const qk_f4 &v, qk_f2 &a, qk_f2 &b) { a = v.lo; b = v.hi; }

// The 12 half-band constants (filter.c:359-362).
#define HB_C0 0.000018566625444266
#define HB_C1 -0.000118469698701817
#define HB_C2 0.000457318798253456
#define HB_C3 -0.001347840471412094
#define HB_C4 0.003321838571445455
#define HB_C5 -0.007198422696929033
#define HB_C6 0.014211106939802483
#define HB_C7 -0.026424776824073383
#define HB_C8 0.048414810444971007
#define HB_C9 -0.096214669073304823
#define HB_C10 0.314881034738348550
#define HB_C11 0.500000000000000000
__device__ __constant__ double c_hb45[12] = { HB_C0, HB_C1, HB_C2, HB_C3, HB_C4, HB_C5, HB_C6, HB_C7, HB_C8,
                                              HB_C9, HB_C10, HB_C11 };

// ------------------------------------------------------------------------------------------------
// Tune mixer (quisk.c:2350-2361).  The reference's running vector after k samples of this block is
// v0 * phase^k; phase^k = A[k>>12] B[(k>>6)&63] C[k&63] with correctly rounded table entries made by the
// host (qk_phasor.c), so the systematic drift of the never-renormalised recurrence is reproduced.
// ------------------------------------------------------------------------------------------------
struct MixTables {
    const double2 *tabA, *tabB, *tabC;   // [rx][nA], [rx][64], [rx][64]
    const double2 *v0;                   // [rx] vector at the first sample of this block
    const int *mix_on;                   // [rx] 0 when tune == 0: the reference skips the multiply
    int nA;
};
__device__ __forceinline__ double2 mix_base(const MixTables &m, int rx, long long j)   // j = k >> 6
{
    double2 t = cmul(m.v0[rx], m.tabA[(long long)rx * m.nA + (j >> 6)]);
    return cmul(t, m.tabB[rx * 64 + (int)(j & 63)]);
}
__device__ __forceinline__ double2 mix_w(const MixTables &m, int rx, long long k)
{
    return cmul(mix_base(m, rx, k >> 6), m.tabC[rx * 64 + (int)(k & 63)]);
}

struct FrontArgs {
    const double2 *x;          // this block's input, n interleaved complex samples
    int n;
    long long N0;              // global index of x[0] (samples consumed before this block)
    const double2 *hist;       // [rx][H] mixed samples of global indices [N0-H, N0)
    int H;
    MixTables mt;
    StreamView out;            // stage-S output stream (2 planes)
    long long J0;              // global index of the first stage-S output of this block
    long long g_lo0, g_end;    // global input range [g_lo0, g_end) consumed by this block's outputs (multiples of 2^S)
    int span_len;              // input samples per span (multiple of 2^S)
    int nspans;                // spans per receiver; a workgroup walks spans blockIdx.x, blockIdx.x + gridDim.x, ...
    double2 *hist_out;         // the front-end kernels also leave the next block's history and tune vector (what k_mixhist_update
    double2 *v0_next;          // does; null: a launch of that kernel follows)
    int out_f32;               // k_frontend_pk: the output stream holds floats (its readers are the float FIRs)
    int helpers;               // short blocks: the last `helpers` workgroups of a receiver's row do the two copies below and no span, so
                               // that they run beside the span walk instead of in front of it
    const double *head_src;    // the output stream is double-buffered: the last head_hist elements of the previous block (head_n of them,
    int head_hist, head_n;     // in the twin head_src) go in front of this block's first output (null: a k_save_tails launch does it)
};

// Mixer alone (no half-band in front of the FIR chain: rates < 96 ksps).
__global__ __launch_bounds__(256) void k_mix_only(FrontArgs a)
{
    const int rx = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    double2 v = a.x[i];
    if (a.mt.mix_on[rx]) v = cmul(v, mix_w(a.mt, rx, i));
    sv_ptr(a.out, rx, 0)[i] = v.x;
    sv_ptr(a.out, rx, 1)[i] = v.y;
}

// After a block: the last H mixed samples become the next block's history, and the tune vector moves on
// by phase^n.  hist_in/hist_out and v0/v0_next are ping-pong pairs.
__device__ __forceinline__ void mixhist_update(const FrontArgs &a, double2 *hist_out, double2 *v0_next, const int rx, const int first, const int stride)
{
    const int on = a.mt.mix_on[rx];
    for (int i = first; i < a.H; i += stride) {
        long long k = (long long)a.n - a.H + i;           // block-relative index
        double2 v;
        if (k >= 0) {
            v = a.x[k];
            if (on) v = cmul(v, mix_w(a.mt, rx, k));
        } else {
            v = a.hist[(long long)rx * a.H + i + a.n];
        }
        hist_out[(long long)rx * a.H + i] = v;
    }
    if (first == 0)
        v0_next[rx] = on ? mix_w(a.mt, rx, a.n) : a.mt.v0[rx];
}
// the head copy of the front end's double-buffered output stream, spread over the grid like mixhist_update (elements are floats
// when the stream holds floats): dst[plane][i] = src[plane][i + n], i < hist
__device__ __forceinline__ void front_head_copy(const FrontArgs &a, const int rx, const int first, const int stride)
{
    const long long base = (long long)rx * a.out.rx_stride;
    for (int i = first; i < 2 * a.head_hist; i += stride) {
        const int pl = i >= a.head_hist, k = i - pl * a.head_hist;
        const long long o = base + (long long)pl * a.out.plane_stride + k;
        if (a.out_f32) reinterpret_cast<float *>(a.out.d)[o] = reinterpret_cast<const float *>(a.head_src)[o + a.head_n];
        else a.out.d[o] = a.head_src[o + a.head_n];
    }
}
__global__ __launch_bounds__(256) void k_mixhist_update(FrontArgs a, double2 *hist_out, double2 *v0_next)
{
    mixhist_update(a, hist_out, v0_next, blockIdx.y, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// ------------------------------------------------------------------------------------------------
// Front end: mixer + S cascaded 45-tap half-band decimators (quisk_cDecim2HB45, filter.c:354-394),
// fused.  One workgroup = one receiver x one span of the input.  The span is walked in chunks of CHUNK
// samples; every level keeps its samples in LDS as two polyphase arrays per plane:
//     E[j] = x[2j]   (feeds the centre tap:  center[] in the reference)
//     O[j] = x[2j+1] (feeds the symmetric taps: samples[] in the reference)
// so output m = sum_k c[k] (O[m-k] + O[m-21+k]) + c[11] E[m-10] is a contiguous dot product, and each
// lane computes R = 6 consecutive outputs from 28 + 6 LDS doubles read as 16-byte vectors
// (lane stride 48 B: conflict-free for ds_read_b128).  22 samples of history per array carry the filter
// state from chunk to chunk; a span warms its histories up by running 44*2^S input samples ahead of its
// first output (those come from the previous block's mixed-history buffer at a block start).
// Global index convention: stage-s output j is emitted at input index 2^s j + 2^s - 1 (the reference
// emits on every second input, toggle starting at 0).
// ------------------------------------------------------------------------------------------------
#define FE_HIST 22

// ------------------------------------------------------------------------------------------------
// Front end kernel: ONE WAVE owns a receiver x span and walks it in chunks of 384 samples (6 per lane)
// with all its levels in its own 16 KiB of LDS: no workgroup barrier ever waits for another wave, so a
// SIMD always has a runnable wave.  (A 256-lane workgroup form with barriers between the stages came first
// and was 2.5x slower.)  The deeper (smaller) stages use fewer outputs per lane,
// R = 6, 3, 2, 1, 1, so that the 2 x 32 lanes (plane x item) stay busy; an idle lane costs a wave as much as
// a busy one.
// ------------------------------------------------------------------------------------------------
#define FW_CHUNK 384
#ifndef FE_SKIP
#define FE_SKIP 0     // instrumented builds: bit 0 no mixer stores, 1..5 no stage 0..4, 6 no carries (outputs wrong, counters matter)
#endif
#ifndef FE_WAVES_PER_EU
#define FE_WAVES_PER_EU 2
#endif
#ifndef FE_PRIO
#define FE_PRIO 0
#endif
struct FwLayout {
    static __host__ __device__ constexpr int sub(int s) { return (FW_CHUNK >> (s + 1)) + FE_HIST + (s == 2 ? 16 : s == 4 ? 18 : 14); }   // (pads chosen so that a stage's stores into the next level's E and O arrays fall on disjoint banks)
    static __host__ __device__ constexpr int level_off(int s) { int o = 0; for (int i = 0; i < s; i++) o += 4 * sub(i); return o; }
    static __host__ __device__ constexpr int total(int S) { return level_off(S) + 16; }
#ifndef FE_EVEN_R
#define FE_EVEN_R 0
#endif
    // outputs per lane at level s.  (FE_EVEN_R, double only: even counts everywhere, so that every window starts on a 16-byte
    // boundary and is read with compiler-visible ds_read_b128 -- no spelled-out reads, the compiler counts the waits itself)
    static __host__ __device__ constexpr int R(int s, int esz = 8)
    { return (FE_EVEN_R && esz == 8) ? (s == 0 ? 6 : s == 1 ? 4 : 2) : (s == 0 ? 6 : s == 1 ? 3 : s == 2 ? 2 : 1); }
    static __host__ __device__ constexpr int warm(int S) { return ((44 << S) + 63) / 64 * 64; }
};
template <class T>
__device__ __forceinline__ T *fw_arr(T *smem, int s, int plane, int phase)
{
    return smem + FwLayout::level_off(s) + (plane * 2 + phase) * FwLayout::sub(s);
}

template <class T, int R, bool LAST = false>
__device__ __forceinline__ void hb45_out(const T *__restrict__ O, const T *__restrict__ E, int m0, T (&y)[R])
{
    // o[j] = O[m0 + 2 + j] (odd-phase samples m0-21 .. m0+R-1), e[r] = E[m0 + 12 + r] (even-phase sample m0+r-10)
    typedef typename QkVec2<T>::type V2;
    T o[R + 22];
    T e[R + 1];
    if constexpr (sizeof(T) == 8 && R == 2 && LAST && !FE_EVEN_R) {
        // (23 + 2 single reads: measured faster than the 13 ds_read_b128 of the vector form where this is the last level and
        // all 64 lanes read -- 398 against 405 us at S = 3 -- and slower where it is not: 248 against 233 us at S = 4)
        const unsigned ao = (unsigned)reinterpret_cast<size_t>(O + m0 + 2), ae = (unsigned)reinterpret_cast<size_t>(E + m0 + 12);
        asm volatile(
            "ds_read_b64 %0, %25\n\tds_read_b64 %1, %25 offset:8\n\tds_read_b64 %2, %25 offset:16\n\tds_read_b64 %3, %25 offset:24\n\t"
            "ds_read_b64 %4, %25 offset:32\n\tds_read_b64 %5, %25 offset:40\n\tds_read_b64 %6, %25 offset:48\n\tds_read_b64 %7, %25 offset:56\n\t"
            "ds_read_b64 %8, %25 offset:64\n\tds_read_b64 %9, %25 offset:72\n\tds_read_b64 %10, %25 offset:80\n\tds_read_b64 %11, %25 offset:88\n\t"
            "ds_read_b64 %12, %25 offset:96\n\tds_read_b64 %13, %25 offset:104\n\tds_read_b64 %14, %25 offset:112\n\tds_read_b64 %15, %25 offset:120\n\t"
            "ds_read_b64 %16, %25 offset:128\n\tds_read_b64 %17, %25 offset:136\n\tds_read_b64 %18, %25 offset:144\n\tds_read_b64 %19, %25 offset:152\n\t"
            "ds_read_b64 %20, %25 offset:160\n\tds_read_b64 %21, %25 offset:168\n\tds_read_b64 %22, %25 offset:176\n\t"
            "ds_read_b64 %23, %26\n\tds_read_b64 %24, %26 offset:8\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]),
              "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(o[16]), "=&v"(o[17]), "=&v"(o[18]),
              "=&v"(o[19]), "=&v"(o[20]), "=&v"(o[21]), "=&v"(o[22]), "=&v"(e[0]), "=&v"(e[1])
            : "v"(ao), "v"(ae)
            : "memory");
    } else if constexpr (sizeof(T) == 4 && (R == 6 || R == 2)) {
        // float: the windows are read as float pairs; the compiler would pair those 8-byte reads into ds_read2_b64 (see above)
        constexpr int NO = (R + 22) / 2, NE = R / 2;
        unsigned long long t[NO + NE];
        const unsigned ao = (unsigned)reinterpret_cast<size_t>(O + m0 + 2), ae = (unsigned)reinterpret_cast<size_t>(E + m0 + 12);
        if constexpr (R == 6)
            asm volatile(
            "ds_read_b64 %0, %17\n\tds_read_b64 %1, %17 offset:8\n\tds_read_b64 %2, %17 offset:16\n\tds_read_b64 %3, %17 offset:24\n\t"
            "ds_read_b64 %4, %17 offset:32\n\tds_read_b64 %5, %17 offset:40\n\tds_read_b64 %6, %17 offset:48\n\tds_read_b64 %7, %17 offset:56\n\t"
            "ds_read_b64 %8, %17 offset:64\n\tds_read_b64 %9, %17 offset:72\n\tds_read_b64 %10, %17 offset:80\n\tds_read_b64 %11, %17 offset:88\n\t"
            "ds_read_b64 %12, %17 offset:96\n\tds_read_b64 %13, %17 offset:104\n\tds_read_b64 %14, %18\n\tds_read_b64 %15, %18 offset:8\n\t"
            "ds_read_b64 %16, %18 offset:16\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), "=&v"(t[14]), "=&v"(t[15]), "=&v"(t[16])
            : "v"(ao), "v"(ae) : "memory");
        else
            asm volatile(
            "ds_read_b64 %0, %13\n\tds_read_b64 %1, %13 offset:8\n\tds_read_b64 %2, %13 offset:16\n\tds_read_b64 %3, %13 offset:24\n\t"
            "ds_read_b64 %4, %13 offset:32\n\tds_read_b64 %5, %13 offset:40\n\tds_read_b64 %6, %13 offset:48\n\tds_read_b64 %7, %13 offset:56\n\t"
            "ds_read_b64 %8, %13 offset:64\n\tds_read_b64 %9, %13 offset:72\n\tds_read_b64 %10, %13 offset:80\n\tds_read_b64 %11, %13 offset:88\n\t"
            "ds_read_b64 %12, %14\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12])
            : "v"(ao), "v"(ae) : "memory");
#pragma unroll
        for (int j = 0; j < NO; j++) { o[2 * j] = __uint_as_float((unsigned)t[j]); o[2 * j + 1] = __uint_as_float((unsigned)(t[j] >> 32)); }
#pragma unroll
        for (int j = 0; j < NE; j++) { e[2 * j] = __uint_as_float((unsigned)t[NO + j]); e[2 * j + 1] = __uint_as_float((unsigned)(t[NO + j] >> 32)); }
    } else if constexpr (R % 2 == 0) {         // m0 even: two-element vector reads (16 bytes in double, 8 in float)
        const V2 *po = reinterpret_cast<const V2 *>(__builtin_assume_aligned(O, sizeof(V2))) + (m0 / 2 + 1);
#pragma unroll
        for (int j = 0; j < (R + 22) / 2; j++) { const V2 v = po[j]; qk_unpack2(v, o[2 * j], o[2 * j + 1]); }
        const V2 *pe = reinterpret_cast<const V2 *>(__builtin_assume_aligned(E, sizeof(V2))) + (m0 / 2 + 6);
#pragma unroll
        for (int j = 0; j < R / 2; j++) { const V2 v = pe[j]; qk_unpack2(v, e[2 * j], e[2 * j + 1]); }
    } else if constexpr (sizeof(T) == 8 && R == 3) {
        // 24 + 3 single ds_read_b64: the compiler pairs neighbouring 8-byte reads into ds_read2_b64, which the LDS serves at
        // half the rate of ds_read_b64 (8 clk per 16 B against 2 clk per 8 B), so the loads are spelled out.  The low 32
        // bits of a generic pointer into LDS are the LDS byte address.
        const unsigned ao = (unsigned)reinterpret_cast<size_t>(O + m0 + 2), ae = (unsigned)reinterpret_cast<size_t>(E + m0 + 12);
        asm volatile(
            "ds_read_b64 %0, %27\n\tds_read_b64 %1, %27 offset:8\n\tds_read_b64 %2, %27 offset:16\n\tds_read_b64 %3, %27 offset:24\n\t"
            "ds_read_b64 %4, %27 offset:32\n\tds_read_b64 %5, %27 offset:40\n\tds_read_b64 %6, %27 offset:48\n\tds_read_b64 %7, %27 offset:56\n\t"
            "ds_read_b64 %8, %27 offset:64\n\tds_read_b64 %9, %27 offset:72\n\tds_read_b64 %10, %27 offset:80\n\tds_read_b64 %11, %27 offset:88\n\t"
            "ds_read_b64 %12, %27 offset:96\n\tds_read_b64 %13, %27 offset:104\n\tds_read_b64 %14, %27 offset:112\n\tds_read_b64 %15, %27 offset:120\n\t"
            "ds_read_b64 %16, %27 offset:128\n\tds_read_b64 %17, %27 offset:136\n\tds_read_b64 %18, %27 offset:144\n\tds_read_b64 %19, %27 offset:152\n\t"
            "ds_read_b64 %20, %27 offset:160\n\tds_read_b64 %21, %27 offset:168\n\tds_read_b64 %22, %27 offset:176\n\tds_read_b64 %23, %27 offset:184\n\t"
            "ds_read_b64 %24, %28\n\tds_read_b64 %25, %28 offset:8\n\tds_read_b64 %26, %28 offset:16\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]),
              "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(o[16]), "=&v"(o[17]), "=&v"(o[18]),
              "=&v"(o[19]), "=&v"(o[20]), "=&v"(o[21]), "=&v"(o[22]), "=&v"(o[23]), "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2])
            : "v"(ao), "v"(ae)
            : "memory");
    } else if constexpr (sizeof(T) == 8 && R == 1) {
        const unsigned ao = (unsigned)reinterpret_cast<size_t>(O + m0 + 2), ae = (unsigned)reinterpret_cast<size_t>(E + m0 + 12);
        asm volatile(
            "ds_read_b64 %0, %23\n\tds_read_b64 %1, %23 offset:8\n\tds_read_b64 %2, %23 offset:16\n\tds_read_b64 %3, %23 offset:24\n\t"
            "ds_read_b64 %4, %23 offset:32\n\tds_read_b64 %5, %23 offset:40\n\tds_read_b64 %6, %23 offset:48\n\tds_read_b64 %7, %23 offset:56\n\t"
            "ds_read_b64 %8, %23 offset:64\n\tds_read_b64 %9, %23 offset:72\n\tds_read_b64 %10, %23 offset:80\n\tds_read_b64 %11, %23 offset:88\n\t"
            "ds_read_b64 %12, %23 offset:96\n\tds_read_b64 %13, %23 offset:104\n\tds_read_b64 %14, %23 offset:112\n\tds_read_b64 %15, %23 offset:120\n\t"
            "ds_read_b64 %16, %23 offset:128\n\tds_read_b64 %17, %23 offset:136\n\tds_read_b64 %18, %23 offset:144\n\tds_read_b64 %19, %23 offset:152\n\t"
            "ds_read_b64 %20, %23 offset:160\n\tds_read_b64 %21, %23 offset:168\n\t"
            "ds_read_b64 %22, %24\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]),
              "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(o[16]), "=&v"(o[17]), "=&v"(o[18]),
              "=&v"(o[19]), "=&v"(o[20]), "=&v"(o[21]), "=&v"(e[0])
            : "v"(ao), "v"(ae)
            : "memory");
    } else {
#pragma unroll
        for (int j = 0; j < R + 21; j++) o[j] = O[m0 + 2 + j];
#pragma unroll
        for (int r = 0; r < R; r++) e[r] = E[m0 + 12 + r];
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        T acc = (o[r + 21] + o[r]) * hb_c<T>(HB_C0);
        acc = fma(o[r + 20] + o[r + 1], hb_c<T>(HB_C1), acc);
        acc = fma(o[r + 19] + o[r + 2], hb_c<T>(HB_C2), acc);
        acc = fma(o[r + 18] + o[r + 3], hb_c<T>(HB_C3), acc);
        acc = fma(o[r + 17] + o[r + 4], hb_c<T>(HB_C4), acc);
        acc = fma(o[r + 16] + o[r + 5], hb_c<T>(HB_C5), acc);
        acc = fma(o[r + 15] + o[r + 6], hb_c<T>(HB_C6), acc);
        acc = fma(o[r + 14] + o[r + 7], hb_c<T>(HB_C7), acc);
        acc = fma(o[r + 13] + o[r + 8], hb_c<T>(HB_C8), acc);
        acc = fma(o[r + 12] + o[r + 9], hb_c<T>(HB_C9), acc);
        acc = fma(o[r + 11] + o[r + 10], hb_c<T>(HB_C10), acc);
        y[r] = fma(e[r], hb_c<T>(HB_C11), acc);
    }
}

// carry the last 22 entries of the four polyphase arrays of level s to their front (88 doubles: lanes 0..43
// move two each).  n_new = samples added per array this chunk.  One wave: the LDS serves a wave's instructions in
// the order they were issued, so the reads of every lane come before the writes of any lane without a barrier, and
// the two halves can sit apart: the reads are issued in front of the next stage's window reads, the writes behind
// that stage's stores, and nobody waits for the carry alone.
template <class T>
__device__ __forceinline__ void fw_carry_load(const T *smem, int s, int n_new, int lane, T (&v)[2])
{
    const int l = lane < 44 ? lane : 43;
    const int sub = l / 11, i = (l % 11) * 2;               // 4 sub-arrays x 11 pairs
    const T *arr = smem + FwLayout::level_off(s) + sub * FwLayout::sub(s) + (sub & 1);
    if constexpr (sizeof(T) == 8) {
        // Two single 8-byte reads, not the ds_read2_b64 the compiler would make of a plain pair (see hb45_out): volatile
        // accesses in the LDS address space are not merged by the load/store optimizer, and -- unlike reads spelled out
        // in assembly -- the compiler counts them, so it places the wait for them itself wherever v[] is first used,
        // copied or spilled.
        typedef volatile __attribute__((address_space(3))) const double lds_cd;
        lds_cd *q = (lds_cd *)(arr + n_new + i);        // (generic -> LDS address space: the low 32 bits)
        v[0] = q[0]; v[1] = q[1];
    } else {
        v[0] = arr[n_new + i]; v[1] = arr[n_new + i + 1];
    }
}
template <class T>
__device__ __forceinline__ void fw_carry_store(T *smem, int s, int lane, const T (&v)[2])
{
    if (lane < 44) {
        const int sub = lane / 11, i = (lane % 11) * 2;
        T *arr = smem + FwLayout::level_off(s) + sub * FwLayout::sub(s) + (sub & 1);
        arr[i] = v[0]; arr[i + 1] = v[1];
    }
}
// Orders the wave's own LDS traffic for the compiler only: a one-wave workgroup needs neither s_barrier nor a drained
// LDS queue between a stage's stores and the next stage's reads.
#define FW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

template <class T, int S, int s>
__device__ __forceinline__ void fw_stage(T *smem, int c, int lane, bool emit, double *out0, double *out1, long long obase, long long dummy,
                                         T (&pend_y)[FwLayout::R(S - 1, sizeof(T))], long long (&pend_i)[FwLayout::R(S - 1, sizeof(T))])
{
    constexpr int R = FwLayout::R(s, sizeof(T));
    const int n_out = c >> (s + 1);
    const int plane = lane >> 5, q = lane & 31;
    const int m0 = q * R;
    if constexpr (s + 1 == S) {
        // last level: every lane computes (idle lanes read padding) and hands its outputs to the chunk loop, which stores
        // them one chunk later -- real outputs to the stream, the rest to the spare slot at its end: a fixed number of
        // stores per chunk, issued in front of the next prefetch
        T y[R];
        hb45_out<T, R, true>(fw_arr(smem, s, plane, 1), fw_arr(smem, s, plane, 0), m0 < FW_CHUNK ? m0 : 0, y);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool ok = emit && m0 + r < n_out;
            pend_y[r] = y[r];
            pend_i[r] = ok ? obase + m0 + r : dummy;
        }
    } else if (m0 < n_out) {
        T y[R];
        hb45_out<T, R>(fw_arr(smem, s, plane, 1), fw_arr(smem, s, plane, 0), m0, y);
        if constexpr (s + 1 < S) {
            T *En = fw_arr(smem, s + 1, plane, 0), *On = fw_arr(smem, s + 1, plane, 1);
#pragma unroll
            for (int r = 0; r < R; r++) {       // a partial last item spills into the arrays' padding
                const int m = m0 + r;
                T *d = (m & 1) ? On : En;
                d[FE_HIST + (m & 1) + (m >> 1)] = y[r];
            }
        } else {
            if (emit) {
                double *op = (plane ? out1 : out0) + obase + m0;
                if (m0 + R <= n_out) {
#pragma unroll
                    for (int r = 0; r < R; r++) op[r] = (double)y[r];
                } else {
#pragma unroll
                    for (int r = 0; r < R; r++)
                        if (m0 + r < n_out) op[r] = (double)y[r];
                }
            }
        }
    }
}

template <int S, class T = double>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FE_WAVES_PER_EU, FE_WAVES_PER_EU))) void k_frontend_wave(FrontArgs a)
{
    __shared__ __attribute__((aligned(16))) T smem[FwLayout::total(S)];
    const int rx = blockIdx.y, lane = threadIdx.x;
    constexpr int W = FwLayout::warm(S);
    constexpr int NIT = FW_CHUNK / 64;
    const int mix = a.mt.mix_on[rx];
    // the next block's history (its own buffer: nobody reads it in this launch), a sample or two per lane across the grid
    if (a.helpers > 0) {
        const int hb = (int)blockIdx.x - ((int)gridDim.x - a.helpers);
        if (hb >= 0) {
            if (a.hist_out) mixhist_update(a, a.hist_out, a.v0_next, rx, hb * 64 + lane, a.helpers * 64);
            if (a.head_src) front_head_copy(a, rx, hb * 64 + lane, a.helpers * 64);
            return;
        }
    } else {
        if (a.hist_out) mixhist_update(a, a.hist_out, a.v0_next, rx, blockIdx.x * 64 + lane, gridDim.x * 64);
        if (a.head_src) front_head_copy(a, rx, blockIdx.x * 64 + lane, gridDim.x * 64);
    }
    const double2 *histp = a.hist + (long long)rx * a.H;
    double *out0 = sv_ptr(a.out, rx, 0), *out1 = sv_ptr(a.out, rx, 1);
    const long long dummy = (long long)a.out.plane_stride - a.out.hist - 1;     // last slot of the plane: never a sample
#if FE_PRIO
    {   // The two waves that share a SIMD get different issue priorities (from their wave slot), so that one of them
        // fills the other's LDS waits instead of both marching in step from read block to FMA block.  (-DFE_PRIO=1, an
        // experiment: three waves in four then also stand above the chain kernels' priority 0 -- round 3: the launch shortens
        // from 0.307 to 0.276 ms at config 2 and the block period stays, profiles/ub/README.md "block period")
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        switch (hwid & 3u) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(3); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(1); break;
        }
    }
#endif
  for (int span = blockIdx.x; span < a.nspans; span += gridDim.x - a.helpers) {
    const long long g_lo = a.g_lo0 + (long long)span * a.span_len;
    if (g_lo >= a.g_end) break;
    const long long g_hi = (g_lo + a.span_len < a.g_end) ? g_lo + a.span_len : a.g_end;
    FW_SYNC();
    for (int i = lane; i < FwLayout::level_off(S); i += 64) smem[i] = T(0);
    long long pos = g_lo - W;
    T *E0 = fw_arr(smem, 0, 0, 0), *O0 = fw_arr(smem, 0, 0, 1);
    T *E1 = fw_arr(smem, 0, 1, 0), *O1 = fw_arr(smem, 0, 1, 1);
    // Lanes 0..31 take the even samples of each 64, lanes 32..63 the odd ones (sample li + 64 it): a 16-lane store
    // group then writes 16 consecutive doubles of ONE polyphase array -- all 32 store banks once.  (Lane = sample
    // put even and odd lanes of a group on two arrays whose bank ranges overlapped: 2-way on every mixer store.)
    const int li = ((lane & 31) << 1) | (lane >> 5);
    T *d0 = (lane >> 5) ? O0 : E0, *d1 = (lane >> 5) ? O1 : E1;
    const int dofs = FE_HIST + (lane >> 5) + (lane & 31);
    // The lane's mixer phasor: exact (from the tables) at its first in-block sample, then advanced by phase^64
    // from sample to sample of the lane (64 apart).  The recurrence runs over one span only (a few hundred
    // steps, ~1e-14 of rounding), then the next span starts from the tables again.
    const double2 P64 = a.mt.tabB[rx * 64 + 1];
    double2 w = make_double2(1.0, 0.0);
    {
        long long kf = pos - a.N0 + li;                // block-relative index of the lane's first sample
        if (kf < 0) kf += ((-kf + 63) >> 6) << 6;        // first one inside this block (same lane)
        if (mix) w = mix_w(a.mt, rx, kf);
    }
    // prefetch of the first chunk
    double2 xv[NIT];
    // Every global access of the chunk loop is unconditional (clamped index, validity kept as a bit; stores of lanes
    // with nothing to emit go to a spare slot at the end of the stream): the compiler can then count loads and
    // stores in flight (s_waitcnt vmcnt(n)) instead of draining the queue -- a drained queue made every chunk wait
    // for the previous chunk's stores to be acknowledged by memory.
    unsigned okbits = 0;
    auto fetch = [&](long long p, int c) {
        const long long k0 = p - a.N0;
        if (k0 >= 0 && c == FW_CHUNK) {          // whole chunk inside this block: no clamps, no validity bits
            const double2 *src = a.x + k0 + li;
#pragma unroll
            for (int it = 0; it < NIT; it++) xv[it] = src[64 * it];
            okbits = (1u << NIT) - 1;
            return;
        }
        okbits = 0;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = li + 64 * it;
            const long long k = k0 + i;
            const long long hi = k + a.H;
            const bool in_blk = k >= 0;
            const double2 *src = in_blk ? a.x : histp;
            long long idx = in_blk ? (k < a.n ? k : (long long)a.n - 1) : (hi > 0 ? hi : 0);
            xv[it] = src[idx];
            if (i < c && (in_blk || hi >= 0)) okbits |= 1u << it;
        }
    };
    auto chunk_len = [&](long long p) -> int {
        const long long lim = (p >= g_lo) ? g_hi : g_lo;
        return (lim - p < FW_CHUNK) ? (int)(lim - p) : FW_CHUNK;
    };
    int c = chunk_len(pos);
    fetch(pos, c);
    FW_SYNC();
#ifdef FE_PROF
    long long fp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fp_last = clock64(), fp_n = 0;
#define FE_PROBE(k) do { const long long t_ = clock64(); fp[k] += t_ - fp_last; fp_last = t_; } while (0)
#else
#define FE_PROBE(k) do { } while (0)
#endif
    constexpr int RL = FwLayout::R(S - 1, sizeof(T));
    T pend_y[RL];
    long long pend_i[RL];
#pragma unroll
    for (int r = 0; r < RL; r++) { pend_y[r] = T(0); pend_i[r] = dummy; }
    double *obase_p = (lane >> 5) ? out1 : out0;
    while (pos < g_hi) {
        FE_PROBE(7);
        const bool emit = pos >= g_lo;
        const long long k0 = pos - a.N0;
        // ---- mix the prefetched samples into level 0 ----
        if (k0 >= 0 && c == FW_CHUNK && mix) {      // the common case, free of per-sample conditions
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const double2 v = cmul(xv[it], w);
                w = cmul(w, P64);
                if (!(FE_SKIP & 1)) { d0[dofs + 32 * it] = (T)v.x; d1[dofs + 32 * it] = (T)v.y; } else { pend_y[0] += (T)(v.x + v.y); }
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int i = li + 64 * it;
                if (i < c) {
                    double2 v = (okbits >> it & 1u) ? xv[it] : make_double2(0.0, 0.0);
                    if (k0 + i >= 0 && mix) { v = cmul(v, w); w = cmul(w, P64); }
                    d0[dofs + 32 * it] = (T)v.x; d1[dofs + 32 * it] = (T)v.y;
                }
            }
        }
        // ---- the previous chunk's outputs leave, then the next chunk's samples start their trip ----
#pragma unroll
        for (int r = 0; r < RL; r++) obase_p[pend_i[r]] = (double)pend_y[r];

        const long long npos = pos + c;
        const int nc = npos < g_hi ? chunk_len(npos) : 0;
        if (nc > 0) fetch(npos, nc);
        FW_SYNC();
        FE_PROBE(0);
        const long long obase = (pos >> S) - a.J0;
        const bool full = c == FW_CHUNK;                 // partial chunks (warm-up tail, span end) take the safe path
        T cv[2];
        if (!(FE_SKIP & 2)) fw_stage<T, S, 0>(smem, c, lane, emit, out0, out1, obase, dummy, pend_y, pend_i);
        FW_SYNC();
        FE_PROBE(1);
        if (full && !(FE_SKIP & 64)) fw_carry_load(smem, 0, c >> 1, lane, cv);
#define FW_NEXT(s_) \
        if constexpr (S > s_) { \
            if (!(FE_SKIP & (2 << s_))) fw_stage<T, S, s_>(smem, c, lane, emit, out0, out1, obase, dummy, pend_y, pend_i); \
            FW_SYNC(); \
            FE_PROBE(s_ + 1); \
            if (full && !(FE_SKIP & 64)) { fw_carry_store(smem, s_ - 1, lane, cv); fw_carry_load(smem, s_, c >> (s_ + 1), lane, cv); } \
        }
        FW_NEXT(1) FW_NEXT(2) FW_NEXT(3) FW_NEXT(4)
#undef FW_NEXT
        if (full && !(FE_SKIP & 64)) fw_carry_store(smem, S - 1, lane, cv);
        if (!full) {
            // generic carry with a barrier between the reads and the writes (source and destination may overlap)
            constexpr int NC = S * 4 * FE_HIST;
            T keep[(NC + 63) / 64];
#pragma unroll
            for (int u = 0; u < (NC + 63) / 64; u++) {
                const int wi = lane + 64 * u;
                if (wi < NC) {
                    const int s = wi / (4 * FE_HIST), rem = wi % (4 * FE_HIST);
                    const int sub = rem / FE_HIST, i = rem % FE_HIST;
                    keep[u] = fw_arr(smem, s, sub >> 1, sub & 1)[(c >> (s + 1)) + (sub & 1) + i];
                }
            }
            FW_SYNC();
#pragma unroll
            for (int u = 0; u < (NC + 63) / 64; u++) {
                const int wi = lane + 64 * u;
                if (wi < NC) {
                    const int s = wi / (4 * FE_HIST), rem = wi % (4 * FE_HIST);
                    const int sub = rem / FE_HIST, i = rem % FE_HIST;
                    fw_arr(smem, s, sub >> 1, sub & 1)[(sub & 1) + i] = keep[u];
                }
            }
        }
        FW_SYNC();
        pos = npos; c = nc;
    }
#pragma unroll
    for (int r = 0; r < RL; r++) obase_p[pend_i[r]] = (double)pend_y[r];
#ifdef FE_PROF
    if (lane == 0 && rx == 0 && span == 0)
        printf("fe prof: mix %lld st0 %lld st1 %lld st2 %lld st3 %lld st4 %lld carry %lld looptail %lld\n", fp[0], fp[1], fp[2], fp[3], fp[4], fp[5], fp[6], fp[7]);
#endif
  }
}

// ------------------------------------------------------------------------------------------------
// Front end of the fp32 mode (BASELINE configs[4]) on packed FP32 operations: the same walk as k_frontend_wave -- one wave per
// receiver x span, every level in the wave's own LDS, no barriers -- with a complex sample kept as ONE 64-bit (re, im) pair.
// A lane then owns outputs, not a plane: 64 lanes x R outputs, chunks of 768 samples, and every half-band add and
// multiply-add is one v_pk_add_f32 / v_pk_fma_f32 for both planes -- half the vector instructions of the plane-per-lane
// form, whose v_fma_f32 runs at the FP64 rate.  Per level two polyphase arrays (even, odd) of (re, im) pairs, laid out and
// padded as in FwLayout; the windows are the same 8- and 16-byte reads.  The mixer: the lane's phasor is kept in double
// and advanced once per chunk (by p^768); inside the chunk the lane's phasor runs on in float, float(w) * float(p^64)^it, as does the
// product with the sample (a few 1e-7 relative over the twelve samples, against the mode's 1e-5 bar).
// ------------------------------------------------------------------------------------------------
#define PK_CHUNK 768
struct PkLayout {
    static __host__ __device__ constexpr int sub(int s) { return (PK_CHUNK >> (s + 1)) + FE_HIST + (s == 2 ? 16 : s == 4 ? 18 : 14); }
    static __host__ __device__ constexpr int level_off(int s) { int o = 0; for (int i = 0; i < s; i++) o += 2 * sub(i); return o; }
    static __host__ __device__ constexpr int total(int S) { return level_off(S) + 16; }
    static __host__ __device__ constexpr int R(int s) { return s == 0 ? 6 : s == 1 ? 3 : s == 2 ? 2 : 1; }
};
__device__ __forceinline__ qk_f2 *pk_arr(qk_f2 *smem, int s, int phase) { return smem + PkLayout::level_off(s) + phase * PkLayout::sub(s); }
__device__ __forceinline__ qk_f2 pk_cmul(qk_f2 a, qk_f2 b)
{
    // (a.x b.x - a.y b.y, a.x b.y + a.y b.x) in two packed operations: the operand selects pick (a.x, a.x) x (b.x, b.y), then
    // (a.y, a.y) x (-b.y, b.x) on top (the compiler spends two more instructions on the swap and the sign)
    qk_f2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
    return d;
}
// an 8-byte LDS store that the compiler does not pair with its neighbour into ds_write2_b64
__device__ __forceinline__ void pk_store(qk_f2 *p, qk_f2 v)
{
    typedef volatile __attribute__((address_space(3))) qk_f2 lds_f2;
    *(lds_f2 *)p = v;
}
// carry of level s: the last 22 pairs of its two arrays go to their front (lanes 0..43 move one pair each); reads and
// writes sit apart as in fw_carry_load / fw_carry_store
__device__ __forceinline__ qk_f2 pk_carry_load(const qk_f2 *smem, int s, int n_new, int lane)
{
    const int l = lane < 44 ? lane : 43;
    const int sub = l / 22, i = l % 22;
    return smem[PkLayout::level_off(s) + sub * PkLayout::sub(s) + sub + n_new + i];
}
__device__ __forceinline__ void pk_carry_store(qk_f2 *smem, int s, int lane, qk_f2 v)
{
    if (lane < 44) {
        const int sub = lane / 22, i = lane % 22;
        smem[PkLayout::level_off(s) + sub * PkLayout::sub(s) + sub + i] = v;
    }
}
template <int S, int s>
__device__ __forceinline__ void pk_stage(qk_f2 *smem, int c, int lane, bool emit, long long obase, long long dummy,
                                         qk_f2 (&pend_y)[PkLayout::R(S - 1)], long long (&pend_i)[PkLayout::R(S - 1)])
{
    constexpr int R = PkLayout::R(s);
    const int n_out = c >> (s + 1);
    const int m0 = lane * R;
    if constexpr (s + 1 == S) {
        // last level: every lane computes (idle lanes redo item 0) and hands its outputs to the chunk loop, which stores them
        // one chunk later -- real outputs to the stream, the rest to the spare slot at its end
        qk_f2 y[R];
        hb45_out<qk_f2, R, true>(pk_arr(smem, s, 1), pk_arr(smem, s, 0), m0 < n_out ? m0 : 0, y);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool ok = emit && m0 + r < n_out;
            pend_y[r] = y[r];
            pend_i[r] = ok ? obase + m0 + r : dummy;
        }
    } else if (m0 < n_out) {
        qk_f2 y[R];
        hb45_out<qk_f2, R>(pk_arr(smem, s, 1), pk_arr(smem, s, 0), m0, y);
        qk_f2 *En = pk_arr(smem, s + 1, 0), *On = pk_arr(smem, s + 1, 1);
#pragma unroll
        for (int r = 0; r < R; r++) {       // a partial last item spills into the arrays' padding
            const int m = m0 + r;
            qk_f2 *d = (m & 1) ? On : En;
            d[FE_HIST + (m & 1) + (m >> 1)] = y[r];
        }
    }
}

#ifndef PK_WAVES_PER_EU
#define PK_WAVES_PER_EU 2
#endif
#ifndef PK_LATE_FETCH
#define PK_LATE_FETCH 0
#endif
#ifndef PK_SKIP
#define PK_SKIP 0     // instrumented builds (outputs wrong, durations matter): 1 no prefetch, 2 no output stores, 4 no carries, 8 no mixer stores, 16 << s no level s
#endif
template <int S>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PK_WAVES_PER_EU, PK_WAVES_PER_EU))) void k_frontend_pk(FrontArgs a)
{
    __shared__ __attribute__((aligned(16))) qk_f2 smem[PkLayout::total(S)];
    const int rx = blockIdx.y, lane = threadIdx.x;
    constexpr int W = FwLayout::warm(S);
    constexpr int NIT = PK_CHUNK / 64;
    const int mix = a.mt.mix_on[rx];
    // the next block's history (its own buffer: nobody reads it in this launch), a sample or two per lane across the grid
    if (a.helpers > 0) {
        const int hb = (int)blockIdx.x - ((int)gridDim.x - a.helpers);
        if (hb >= 0) {
            if (a.hist_out) mixhist_update(a, a.hist_out, a.v0_next, rx, hb * 64 + lane, a.helpers * 64);
            if (a.head_src) front_head_copy(a, rx, hb * 64 + lane, a.helpers * 64);
            return;
        }
    } else {
        if (a.hist_out) mixhist_update(a, a.hist_out, a.v0_next, rx, blockIdx.x * 64 + lane, gridDim.x * 64);
        if (a.head_src) front_head_copy(a, rx, blockIdx.x * 64 + lane, gridDim.x * 64);
    }
    const double2 *histp = a.hist + (long long)rx * a.H;
    double *out0 = sv_ptr(a.out, rx, 0), *out1 = sv_ptr(a.out, rx, 1);
    float *outf0 = sv_ptr_t<float>(a.out, rx, 0), *outf1 = sv_ptr_t<float>(a.out, rx, 1);
    const bool of32 = a.out_f32 != 0;
    auto put = [&](const long long i, const qk_f2 y) {
        if (of32) { outf0[i] = y.x; outf1[i] = y.y; } else { out0[i] = (double)y.x; out1[i] = (double)y.y; }
    };
    const long long dummy = (long long)a.out.plane_stride - a.out.hist - 1;     // last slot of the plane: never a sample
    // p^(64 it) in float for the samples of a chunk, p^64 and p^768 in double for the lane's own phasor
    const double2 P64 = a.mt.tabB[rx * 64 + 1], PCH = a.mt.tabB[rx * 64 + NIT];
    const qk_f2 P64f = qk_f2{(float)P64.x, (float)P64.y};
  for (int span = blockIdx.x; span < a.nspans; span += gridDim.x - a.helpers) {
    const long long g_lo = a.g_lo0 + (long long)span * a.span_len;
    if (g_lo >= a.g_end) break;
    const long long g_hi = (g_lo + a.span_len < a.g_end) ? g_lo + a.span_len : a.g_end;
    FW_SYNC();
    for (int i = lane; i < PkLayout::level_off(S); i += 64) smem[i] = qk_f2{0.f, 0.f};
    long long pos = g_lo - W;
    // lanes 0..31 take the even samples of each 64, lanes 32..63 the odd ones (see k_frontend_wave)
    const int li = ((lane & 31) << 1) | (lane >> 5);
    qk_f2 *d0 = pk_arr(smem, 0, lane >> 5);
    const int dofs = FE_HIST + (lane >> 5) + (lane & 31);
    double2 w = make_double2(1.0, 0.0);
    {
        long long kf = pos - a.N0 + li;                // block-relative index of the lane's first sample
        if (kf < 0) kf += ((-kf + 63) >> 6) << 6;        // first one inside this block (same lane)
        if (mix) w = mix_w(a.mt, rx, kf);
    }
    double2 xv[NIT];
    unsigned okbits = 0;
    auto fetch = [&](long long p, int c) {
        const long long k0 = p - a.N0;
        if (k0 >= 0 && c == PK_CHUNK) {          // whole chunk inside this block: no clamps, no validity bits
            const double2 *src = a.x + k0 + li;
#pragma unroll
            for (int it = 0; it < NIT; it++) xv[it] = src[64 * it];
            okbits = (1u << NIT) - 1;
            return;
        }
        okbits = 0;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = li + 64 * it;
            const long long k = k0 + i;
            const long long hi = k + a.H;
            const bool in_blk = k >= 0;
            const double2 *src = in_blk ? a.x : histp;
            long long idx = in_blk ? (k < a.n ? k : (long long)a.n - 1) : (hi > 0 ? hi : 0);
            xv[it] = src[idx];
            if (i < c && (in_blk || hi >= 0)) okbits |= 1u << it;
        }
    };
    auto chunk_len = [&](long long p) -> int {
        const long long lim = (p >= g_lo) ? g_hi : g_lo;
        return (lim - p < PK_CHUNK) ? (int)(lim - p) : PK_CHUNK;
    };
    int c = chunk_len(pos);
    fetch(pos, c);
    FW_SYNC();
    constexpr int RL = PkLayout::R(S - 1);
    qk_f2 pend_y[RL];
    long long pend_i[RL];
#pragma unroll
    for (int r = 0; r < RL; r++) { pend_y[r] = qk_f2{0.f, 0.f}; pend_i[r] = dummy; }
    while (pos < g_hi) {
        const bool emit = pos >= g_lo;
        const long long k0 = pos - a.N0;
        // ---- mix the prefetched samples into level 0 ----
        if (k0 >= 0 && c == PK_CHUNK && mix) {      // the common case, free of per-sample conditions
            qk_f2 wf = qk_f2{(float)w.x, (float)w.y};
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const qk_f2 xf = qk_f2{(float)xv[it].x, (float)xv[it].y};
                if (!(PK_SKIP & 8)) pk_store(d0 + dofs + 32 * it, pk_cmul(xf, wf)); else pend_y[0] += pk_cmul(xf, wf);
                if (it + 1 < NIT) wf = pk_cmul(wf, P64f);
            }
            w = cmul(w, PCH);
        } else {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int i = li + 64 * it;
                if (i < c) {
                    double2 v = (okbits >> it & 1u) ? xv[it] : make_double2(0.0, 0.0);
                    if (k0 + i >= 0 && mix) { v = cmul(v, w); w = cmul(w, P64); }
                    d0[dofs + 32 * it] = qk_f2{(float)v.x, (float)v.y};
                }
            }
        }
        // ---- the previous chunk's outputs leave, then the next chunk's samples start their trip ----
#pragma unroll
        for (int r = 0; r < RL; r++) { if (!(PK_SKIP & 2) || pend_y[r].x == 1.2345f) put(pend_i[r], pend_y[r]); }

        const long long npos = pos + c;
        const int nc = npos < g_hi ? chunk_len(npos) : 0;
#if !PK_LATE_FETCH
        if (nc > 0 && !(PK_SKIP & 1)) fetch(npos, nc);
#endif
        FW_SYNC();
        const long long obase = (pos >> S) - a.J0;
        const bool full = c == PK_CHUNK;                 // partial chunks (warm-up tail, span end) take the safe path
        qk_f2 cv = qk_f2{0.f, 0.f};
        if (!(PK_SKIP & 16)) pk_stage<S, 0>(smem, c, lane, emit, obase, dummy, pend_y, pend_i);
        FW_SYNC();
#if PK_LATE_FETCH
        // the next chunk's samples start their trip behind level 0, whose window (70 registers) is dead by now: the prefetch
        // registers and the widest window are never live together, which is what lets a third wave onto the SIMD
        if (nc > 0) fetch(npos, nc);
#endif
        if (full && !(PK_SKIP & 4)) cv = pk_carry_load(smem, 0, c >> 1, lane);
#define PK_NEXT(s_) \
        if constexpr (S > s_) { \
            if (!(PK_SKIP & (16 << s_))) pk_stage<S, s_>(smem, c, lane, emit, obase, dummy, pend_y, pend_i); \
            FW_SYNC(); \
            if (full && !(PK_SKIP & 4)) { pk_carry_store(smem, s_ - 1, lane, cv); cv = pk_carry_load(smem, s_, c >> (s_ + 1), lane); } \
        }
        PK_NEXT(1) PK_NEXT(2) PK_NEXT(3) PK_NEXT(4)
#undef PK_NEXT
        if (full && !(PK_SKIP & 4)) pk_carry_store(smem, S - 1, lane, cv);
        if (!full) {
            // generic carry with a fence between the reads and the writes (source and destination may overlap)
            constexpr int NC = S * 2 * FE_HIST;
            qk_f2 keep[(NC + 63) / 64];
#pragma unroll
            for (int u = 0; u < (NC + 63) / 64; u++) {
                const int wi = lane + 64 * u;
                if (wi < NC) {
                    const int s = wi / (2 * FE_HIST), rem = wi % (2 * FE_HIST);
                    const int sub = rem / FE_HIST, i = rem % FE_HIST;
                    keep[u] = pk_arr(smem, s, sub)[(c >> (s + 1)) + sub + i];
                }
            }
            FW_SYNC();
#pragma unroll
            for (int u = 0; u < (NC + 63) / 64; u++) {
                const int wi = lane + 64 * u;
                if (wi < NC) {
                    const int s = wi / (2 * FE_HIST), rem = wi % (2 * FE_HIST);
                    const int sub = rem / FE_HIST, i = rem % FE_HIST;
                    pk_arr(smem, s, sub)[sub + i] = keep[u];
                }
            }
        }
        FW_SYNC();
        pos = npos; c = nc;
    }
#pragma unroll
    for (int r = 0; r < RL; r++) put(pend_i[r], pend_y[r]);
  }
}

// ------------------------------------------------------------------------------------------------
// Generic real-coefficient FIR with decimation on a planar stream: quisk_cDecimate / quisk_dDecimate /
// quisk_dFilter / quisk_cFilter (filter.c:180-206, 236-262, 324-352).  Output j uses the newest input at
// index first + D j and accumulates h[k] x[newest - k] for k = 0..T-1 in the reference's order.
//
// LDS-tiled.  With k = p + D q the sum of filter.c:193-199 is
//     out[j] = sum_{p<D} sum_{q<TQP} hp[p][q] X_p[j - q],   X_p[i] = x[first + D i - p],   hp[p][q] = h[p + D q]
// (hp zero-padded to TQP = a multiple of FT_QB taps per branch, built on the host): D unit-stride FIRs over
// the D polyphase branches of the input.  A workgroup of NTH lanes makes NO = NTH*R consecutive outputs of
// one (receiver, plane): it de-interleaves the input range into D LDS arrays, then every lane keeps R
// accumulators and walks the taps FT_QB at a time with a register window of R + FT_QB - 1 samples, so an LDS
// load feeds R*FT_QB/(R+FT_QB-1) FMAs.  R is odd: the lane stride of R doubles keeps b64 reads conflict-free.
// The taps are wave-uniform scalar loads.  The products are summed branch by branch instead of in the
// reference's k order (a different rounding order, inside the 1e-9 contract of DESIGN.md section 3).
// ------------------------------------------------------------------------------------------------
#define FT_QB 8
#define FT_LD 8
template <int D, int NTH, int R, class T = double, class XI = double>
__device__ __forceinline__ void fir_tile(const XI *__restrict__ x, const T *__restrict__ hp, int TQP,
                                         int first, int jb, int t_max, T *lds, T (&acc)[R])
{
    constexpr int NO = NTH * R;
    const int LEN = NO + TQP;
    const int t_lo = first + D * (jb - TQP + 1) - (D - 1);
    const int count = D * (LEN - 1);
    // FT_LD loads of a lane are in flight together (clamped index, so none is conditional): one load per trip left
    // the kernel waiting on memory latency with a few KB in flight per CU
    for (int u0 = threadIdx.x; u0 < count; u0 += FT_LD * NTH) {
        XI v[FT_LD];
#pragma unroll
        for (int i = 0; i < FT_LD; i++) {
            const int t = t_lo + u0 + i * NTH;
            v[i] = x[t < t_max ? t : t_max];
        }
#pragma unroll
        for (int i = 0; i < FT_LD; i++) {
            const int u = u0 + i * NTH;
            if (u < count) {
                const int m = u / D, s = u - m * D;
                lds[(D - 1 - s) * LEN + m] = t_lo + u <= t_max ? (T)v[i] : T(0);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = T(0);
    const int base0 = threadIdx.x * R + TQP - FT_QB;
#pragma unroll 1
    for (int p = 0; p < D; p++) {
        const T *xp = lds + p * LEN + base0;
        const T *__restrict__ h = hp + p * TQP;
#pragma unroll 2
        for (int qc = 0; qc < TQP; qc += FT_QB) {
            T w[R + FT_QB - 1];
#pragma unroll
            for (int i = 0; i < R + FT_QB - 1; i++) w[i] = xp[i - qc];
#pragma unroll
            for (int qq = 0; qq < FT_QB; qq++) {
                const T hq = h[qc + qq];
#pragma unroll
                for (int r = 0; r < R; r++) acc[r] = fma(w[r + FT_QB - 1 - qq], hq, acc[r]);
            }
        }
    }
}

struct FirTArgs {
    StreamView in, out;
    const void *hp;            // [D][TQP] permuted taps, in the kernel's compute type
    int TQP, first, n_out;
    const int *rx_list;
    int in_f32, out_f32;       // float kernels only: the input / output stream holds floats (front chain of the fp32 mode)
};
extern __shared__ __attribute__((aligned(16))) unsigned char ft_lds_raw[];
template <int D, int NTH, int R, class T = double>
__global__ __launch_bounds__(NTH) void k_fir_tiled(FirTArgs a)
{
    T *ft_lds = reinterpret_cast<T *>(ft_lds_raw);
    const int rx = a.rx_list ? a.rx_list[blockIdx.z] : blockIdx.z;
    const int plane = blockIdx.y;
    const int jb = blockIdx.x * (NTH * R);
    T acc[R];
    const int left = a.n_out - (jb + (int)threadIdx.x * R);
    if constexpr (sizeof(T) == 4) {
        if (a.in_f32) fir_tile<D, NTH, R, T, float>(sv_ptr_t<float>(a.in, rx, plane), static_cast<const T *>(a.hp), a.TQP, a.first, jb, a.first + D * (a.n_out - 1), ft_lds, acc);
        else fir_tile<D, NTH, R, T>(sv_ptr(a.in, rx, plane), static_cast<const T *>(a.hp), a.TQP, a.first, jb, a.first + D * (a.n_out - 1), ft_lds, acc);
        if (a.out_f32) {
            float *__restrict__ of = sv_ptr_t<float>(a.out, rx, plane) + jb + threadIdx.x * R;
#pragma unroll
            for (int r = 0; r < R; r++) if (r < left) of[r] = acc[r];
            return;
        }
    } else {
        fir_tile<D, NTH, R, T>(sv_ptr(a.in, rx, plane), static_cast<const T *>(a.hp), a.TQP, a.first, jb, a.first + D * (a.n_out - 1), ft_lds, acc);
    }
    double *__restrict__ o = sv_ptr(a.out, rx, plane) + jb + threadIdx.x * R;
#pragma unroll
    for (int r = 0; r < R; r++) if (r < left) o[r] = (double)acc[r];
}

// Polyphase interpolator x L (quisk_dInterpolate / quisk_cInterpolate, filter.c:108-178):
// out[L i + jj] = L * sum_{k < T/L} h[jj + k L] x[i - k].
struct InterpArgs { StreamView in, out; const double *h; int T, L, n_in; const int *rx_list; int delay; };   // delay: read the input this many samples back (d_delay behind the SSB squelch, quisk.c:1833)
__global__ __launch_bounds__(256) void k_interp(InterpArgs a)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= a.n_in * a.L) return;
    const int rx = a.rx_list ? a.rx_list[blockIdx.z] : blockIdx.z;
    const int plane = blockIdx.y;
    const int i = o / a.L, jj = o - i * a.L;
    const double *__restrict__ p = sv_ptr(a.in, rx, plane) + i - a.delay;
    double acc = 0.0;
    const int terms = a.T / a.L;
    // five terms' loads in flight together, the sum in the same order (one dependent trip to L2 per term was the kernel's time)
    int k = 0;
    for (; k + 5 <= terms; k += 5) {
        double xv[5], hv[5];
#pragma unroll
        for (int u = 0; u < 5; u++) { xv[u] = p[-(k + u)]; hv[u] = a.h[jj + (k + u) * a.L]; }
#pragma unroll
        for (int u = 0; u < 5; u++) acc = fma(xv[u], hv[u], acc);
    }
    for (; k < terms; k++) acc = fma(p[-k], a.h[jj + k * a.L], acc);
    sv_ptr(a.out, rx, plane)[o] = acc * a.L;
}

// Rational resampler x L / D (quisk_cInterpDecim, filter.c:264-301): output m sits at virtual position
// v = idx0 + m D of the L-times upsampled stream: input i = v / L, polyphase branch v % L.
struct ResampArgs { StreamView in, out; const double *h; int T, L, D, idx0, n_out; const int *rx_list; };
__global__ __launch_bounds__(256) void k_resample(ResampArgs a)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= a.n_out) return;
    const int rx = a.rx_list ? a.rx_list[blockIdx.z] : blockIdx.z;
    const int plane = blockIdx.y;
    const long long v = a.idx0 + (long long)m * a.D;
    const int i = (int)(v / a.L), ph = (int)(v - (long long)i * a.L);
    const double *__restrict__ p = sv_ptr(a.in, rx, plane) + i;
    double acc = 0.0;
    const int terms = a.T / a.L;
    for (int k = 0; k < terms; k++) acc = fma(p[-k], a.h[ph + k * a.L], acc);
    sv_ptr(a.out, rx, plane)[m] = acc * a.L;
}

// Half-band interpolator x 2 (quisk_dInterp2HB45 / quisk_cInterp2HB45, filter.c:397-465):
// out[2i] = x[i-11] c[11] 2;  out[2i+1] = 2 sum_{k<11} (x[i-k] + x[i-21+k]) c[k].
struct HbInterpArgs { StreamView in, out; int n_in; const int *rx_list; int delay; };
__global__ __launch_bounds__(256) void k_hb_interp(HbInterpArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_in) return;
    const int rx = a.rx_list ? a.rx_list[blockIdx.z] : blockIdx.z;
    const int plane = blockIdx.y;
    const double *__restrict__ p = sv_ptr(a.in, rx, plane) + i - a.delay;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 11; k++) acc = fma(p[-k] + p[-21 + k], c_hb45[k], acc);
    double *o = sv_ptr(a.out, rx, plane) + 2 * (long long)i;
    o[0] = p[-11] * HB_C11 * 2;
    o[1] = acc * 2;
}

// Fractional decimator (cFracDecim, quisk.c:570-613): 4-point Lagrange interpolation at position dindex
// between the last four samples.  The position recurrence does not depend on the data, so the host runs it
// (same operations, same order) and hands over, per output, the index of the newest sample and dindex.
// The positions come from pinned host memory that the device can read: a tiny kernel moves them, because an
// asynchronous host-to-device copy in the per-block path slows down every later launch on the host side
// (measured: ~65 us per call until the copy has run).
__global__ __launch_bounds__(256) void k_stage_frac(const int *h_idx, const double *h_pos, int *d_idx, double *d_pos, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { d_idx[i] = h_idx[i]; d_pos[i] = h_pos[i]; }
}
struct FracArgs { StreamView in, out; const int *idx; const double *pos; int n_out; const int *rx_list; };
__global__ __launch_bounds__(256) void k_fracdecim(FracArgs a)
{
#pragma clang fp contract(off)
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= a.n_out) return;
    const int rx = a.rx_list ? a.rx_list[blockIdx.z] : blockIdx.z;
    const double *__restrict__ p = sv_ptr(a.in, rx, blockIdx.y) + a.idx[m];
    const double d = a.pos[m];
    const double c0 = p[-3], c1 = p[-2], c2 = p[-1], c3 = p[0];
    const double xm0 = d - 0, xm1 = d - 1, xm2 = d - 2, xm3 = d - 3;
    sv_ptr(a.out, rx, blockIdx.y)[m] = (xm1 * xm2 * xm3 * c0 / -6.0 + xm0 * xm2 * xm3 * c1 / 2.0 +
                               xm0 * xm1 * xm3 * c2 / -2.0 + xm0 * xm1 * xm2 * c3 / 6.0);
}

// ------------------------------------------------------------------------------------------------
// Receive band-pass + detector.
//   cRxFilterOut (quisk.c:1154-1192): I taps on the real plane, Q taps on the imaginary plane.
//   dRxFilterOut (quisk.c:1118-1152): I taps on both planes.
// Tap order quirk of both: the newest sample meets tap 0, then tap k (k >= 1) meets x[n - (N - k)].
// N == 0 passes the sample through (quisk.c:1137, 1174).
// mode_kind: 0 = SSB/CW/DGT (audio = re -/+ im, quisk.c:1786,1805,1823,1847), 1 = AM magnitude (cabs,
// quisk.c:1868), 2 = FM: keep the filtered complex sample (2 planes out).
// ------------------------------------------------------------------------------------------------
struct RxfArgs {
    StreamView in, out;
    const void *taps;          // [rx][2][tap_cap]  (I taps, Q taps), in plain FIR order h[0] = f[0], h[m] = f[N - m], zero-padded; compute type
    const int *ntaps;          // [rx]
    const int *mode;           // [rx] rx_mode_type
    int tap_cap, n, kind;
    const int *rx_list;
    double *ma_acc;            // measure_audio (quisk.c:1787-1790): receiver 0's outputs squared are added here (null: not this stage)
};
template <int NTH, int R, class T = double>
__global__ __launch_bounds__(NTH) void k_rxf_tiled(RxfArgs a)
{
    T *ft_lds = reinterpret_cast<T *>(ft_lds_raw);
    const int rx = a.rx_list ? a.rx_list[blockIdx.z] : blockIdx.z;
    const int N = a.ntaps[rx];
    const int jb = blockIdx.x * (NTH * R), j0 = jb + threadIdx.x * R;
    const T *__restrict__ fi = static_cast<const T *>(a.taps) + (long long)rx * 2 * a.tap_cap;
    const T *__restrict__ fq = (a.kind == 0) ? fi + a.tap_cap : fi;
    T ar[R], ai[R];
    if (N == 0) {
        const double *pr = sv_ptr(a.in, rx, 0), *pi = sv_ptr(a.in, rx, 1);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool ok = j0 + r < a.n;
            ar[r] = ok ? (T)pr[j0 + r] : T(0); ai[r] = ok ? (T)pi[j0 + r] : T(0);
        }
    } else {
        const int TQP = (N + FT_QB - 1) / FT_QB * FT_QB;
        fir_tile<1, NTH, R, T>(sv_ptr(a.in, rx, 0), fi, TQP, 0, jb, a.n - 1, ft_lds, ar);
        __syncthreads();
        fir_tile<1, NTH, R, T>(sv_ptr(a.in, rx, 1), fq, TQP, 0, jb, a.n - 1, ft_lds, ai);
    }
    double *o0 = sv_ptr(a.out, rx, 0) + j0;
    const int left = a.n - j0;
    double msum = 0.0;          // sum of the squares of what this lane stores
    if (a.kind == 0) {
        const int m = a.mode[rx];
        const bool lower = (m == 0 /*CWL*/ || m == 2 /*LSB*/ || m == 8 /*DGT_L*/);
#pragma unroll
        for (int r = 0; r < R; r++) if (r < left) { const double v = (double)(lower ? ar[r] + ai[r] : ar[r] - ai[r]); o0[r] = v; msum += v * v; }
    } else if (a.kind == 1) {
#pragma unroll
        for (int r = 0; r < R; r++) if (r < left) { const double v = hypot((double)ar[r], (double)ai[r]); o0[r] = v; msum += v * v; }
    } else {
        double *o1 = sv_ptr(a.out, rx, 1) + j0;
#pragma unroll
        for (int r = 0; r < R; r++) if (r < left) { o0[r] = (double)ar[r]; o1[r] = (double)ai[r]; msum += (double)ar[r] * (double)ar[r] + (double)ai[r] * (double)ai[r]; }
    }
    if (a.ma_acc && rx == 0) {  // (wave-uniform: one atomic per wave; the order of the adds is free within the test's 1e-9)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) msum += __shfl_xor(msum, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.ma_acc, msum);
    }
}

// FM discriminator (quisk.c:1894-1896): d = fs * arg(c[i] conj(c[i-1])); c[-1] of the very first block is 10.
struct FmDiscArgs { StreamView in, out; int n; double fs; const int *rx_list; double *partial; int pstride; };   // partial: [rx][pstride] sums of |c| per workgroup, for the squelch
__global__ __launch_bounds__(256) void k_fm_disc(FmDiscArgs a)
{
    __shared__ double red[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int rx = a.rx_list ? a.rx_list[blockIdx.z] : blockIdx.z;
    double mag = 0.0;
    if (i < a.n) {
        const double *pr = sv_ptr(a.in, rx, 0) + i, *pi = sv_ptr(a.in, rx, 1) + i;
        const double cr = pr[0], ci = pi[0], qr = pr[-1], qi = pi[-1];
        const double re = cr * qr + ci * qi;       // c * conj(q)
        const double im = ci * qr - cr * qi;
        sv_ptr(a.out, rx, 0)[i] = a.fs * atan2(im, re);
        mag = hypot(cr, ci);                        // rf_sum of the FM squelch, quisk.c:1892
    }
    red[threadIdx.x] = mag;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.partial[(long long)rx * a.pstride + blockIdx.x] = red[0];
}

// FM squelch (quisk.c:1892-1893, 1911-1920): the mean magnitude of the filtered samples over stretches of at least 2400
// samples, closed at block ends as in the reference, in dB re full scale; the receiver is muted while it is below the
// level.  The block's sum is a fixed-order tree over the discriminator's workgroups (the reference adds sample by sample: the
// sums differ in the last bits, the decision only if the level is met to 1e-13 dB).
struct SquelchArgs { const double *partial; int pstride, np, n; const int *rx_list; double *state; int *flag; double level; };   // state: [rx][3] sum, count, dB
__global__ __launch_bounds__(256) void k_fm_squelch(SquelchArgs a)
{
    __shared__ double red[256];
    const int rx = a.rx_list ? a.rx_list[blockIdx.x] : blockIdx.x;
    double acc = 0.0;       // the discriminator's workgroups left the sums of |c| of their 256 samples
    for (int i = threadIdx.x; i < a.np; i += 256) acc += a.partial[(long long)rx * a.pstride + i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double *st = a.state + 3 * rx;
        double sum = st[0] + red[0], cnt = st[1] + a.n, db = st[2];
        if (cnt >= 2400.0) {
            db = sum / cnt / QK_CLIP32;
            db = db > 1.E-10 ? 20 * log10(db) : -200.0;
            sum = 0.0; cnt = 0.0;
        }
        st[0] = sum; st[1] = cnt; st[2] = db;
        a.flag[rx] = db < a.level;
    }
}

// One-pole recurrences of the demodulators, as a scan.
//   kind 0: AM DC removal (quisk.c:1869-1871): d = m + 0.99 dc; y = d - dc; dc = d.
//   kind 1: FM de-emphasis (quisk.c:1898-1900): y = x a0 + x1 a1 - y1 b1.
// Both are s[n] = u[n] + lambda s[n-1] (u = m, lambda = 0.99; u = x a0 + x1 a1, lambda = -b1).  One wave per receiver walks
// tiles of 64 x IIR_K samples: lane L runs the recurrence over its IIR_K consecutive samples from a zero state, the 64
// segment ends are chained (S_in(L+1) = end_L + lambda^len_L S_in(L), 64 wave-uniform steps), and every lane adds
// lambda^(k+1) S_in(L) to its local values.  The values differ from a sample-by-sample walk only by rounding order
// (|lambda| < 1: errors do not grow), inside the 1e-9 contract; a serial walk of 32768 samples per receiver took 0.65 ms.
#define IIR_K 16
struct IirArgs { StreamView in, out; int n, kind; double a0, a1, b1; double *state; const int *rx_list; int n_rx; };
__device__ __forceinline__ void serial_iir_wave(const IirArgs &a, const int rx, const int lane)
{
#pragma clang fp contract(off)
    const double *__restrict__ p = sv_ptr(a.in, rx, 0);
    double *__restrict__ o = sv_ptr(a.out, rx, 0);
    const double lam = a.kind == 0 ? 0.99 : -a.b1;
    double pw[IIR_K];                                   // lambda^(k+1)
    pw[0] = lam;
#pragma unroll
    for (int k = 1; k < IIR_K; k++) pw[k] = pw[k - 1] * lam;
    double xprev = a.state[2 * rx];                     // kind 1: x[n-1]; kind 0: unused
    double S = a.kind == 0 ? a.state[2 * rx] : a.state[2 * rx + 1];     // the recurrence state before the tile
    for (int t0 = 0; t0 < a.n; t0 += 64 * IIR_K) {
        const int rem = a.n - t0;
        const int K = rem >= 64 * IIR_K ? IIR_K : (rem + 63) / 64;       // samples per lane in this tile
        const int base = t0 + lane * K;
        int v = a.n - base;                              // valid samples of this lane
        v = v < 0 ? 0 : (v > K ? K : v);
        double x[IIR_K];
#pragma unroll
        for (int k = 0; k < IIR_K; k++) x[k] = k < v ? p[base + k] : 0.0;
        // u[k]
        double xlast = 0.0;
#pragma unroll
        for (int k = 0; k < IIR_K; k++) if (k < v) xlast = x[k];
        if (a.kind == 1) {
            // x[-1] of the lane: the last sample of the nearest lane below that has one, or the carried x1
            double xm1 = __shfl_up(xlast, 1, 64);
            if (lane == 0) xm1 = xprev;
            // (lanes are filled from 0 upwards, so every lane below a non-empty lane is full: its predecessor is lane - 1)
            double prev = xm1;
#pragma unroll
            for (int k = 0; k < IIR_K; k++) {
                const double cur = x[k];
                x[k] = cur * a.a0 + prev * a.a1;
                prev = cur;
            }
        }
        // local recurrence from a zero state
        double l = 0.0, lend = 0.0;
#pragma unroll
        for (int k = 0; k < IIR_K; k++) {
            l = x[k] + l * lam;
            x[k] = l;
            if (k < v) lend = l;
        }
        double fl = 1.0;                                  // lambda^v
#pragma unroll
        for (int k = 0; k < IIR_K; k++) if (k < v) fl = pw[k];
        // chain the segment ends: Sin = state in front of this lane's first sample
        double Sin = 0.0, run = S;
        for (int L = 0; L < 64; L++) {
            if (lane == L) Sin = run;
            run = readlane_f64(lend, L) + readlane_f64(fl, L) * run;
        }
        // fix up and store
        if (a.kind == 0) {
            double dprev = Sin;
#pragma unroll
            for (int k = 0; k < IIR_K; k++) {
                const double d = x[k] + pw[k] * Sin;
                if (k < v) o[base + k] = d - dprev;
                dprev = d;
            }
        } else {
#pragma unroll
            for (int k = 0; k < IIR_K; k++) if (k < v) o[base + k] = x[k] + pw[k] * Sin;
        }
        S = run;
        // the last valid sample of the tile sits in the highest non-empty lane
        const int nv = rem < 64 * K ? rem : 64 * K;
        xprev = readlane_f64(xlast, (nv - 1) / K);
    }
    if (lane == 0) {
        if (a.kind == 0) a.state[2 * rx] = S;
        else { a.state[2 * rx] = xprev; a.state[2 * rx + 1] = S; }
    }
}
__global__ __launch_bounds__(64) void k_serial_iir(IirArgs a)
{
    serial_iir_wave(a, a.rx_list ? a.rx_list[blockIdx.x] : blockIdx.x, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// AGC (process_agc, quisk.c:2022-2147), mono path (is_cpx == 0, quisk.c:2573-2575).
// The look-ahead ring of the reference is the stream's own history: the delayed sample is x[i - buf_size].
// Pass 1 (serial per receiver) walks the gain state machine and records the gain applied at each sample;
// pass 2 (parallel) applies it and the hard limiter.
// ------------------------------------------------------------------------------------------------
struct AgcState {
    double themax, gain, delta, target_gain;
    int index_read, index_start, is_clipping, pad;
};
#define QK_AGC_CLASSES 8
struct AgcArgs {
    // one launch serves every receiver: the receivers of chain class c read the play stream in_c[c] (mono audio at the
    // play rate, hist >= buf_size), n_c[c] new samples this block (0: nothing to do)
    StreamView in_c[QK_AGC_CLASSES];
    int n_c[QK_AGC_CLASSES];
    int cpx_c[QK_AGC_CLASSES];  // the class's play stream is complex (DGT-IQ): is_cpx of process_agc, magnitudes by cabs
    const int *mute;            // [rx] FM squelch closed for this block (quisk.c:2590-2593); may be null
    int kill;                   // kill_audio (quisk.c:2585-2589)
    StreamView mag_c[QK_AGC_CLASSES];   // what k_agc_gain takes magnitudes of: in_c, or for a complex class the cabs that
                                        // k_agc_cabs has left in a plane of its own (the gain kernel stays free of that code)
    const int *cls;            // [rx] chain class
    double *gains;             // [rx][cap]
    int gain_stride, buf_size, n_rx;
    double ceiling;            // max_out * CLIP32
    double time_release;
    const double *release_gain;   // [rx] agcReleaseGain
    const int *mode;           // [rx]
    AgcState *state;           // [rx]
    double2 *out; long long out_stride;   // [rx][out_stride] interleaved stereo
    const int *started;        // [rx] 0 until the receiver's first call: the reference then only initialises
                               // and returns the block untouched (quisk.c:2033-2050)
    int *abort_flag;           // host-visible word, set when a hand-over of k_agc_gain timed out (the gains are then NaN)
    int apply_here;            // short blocks: k_agc_gain applies the gains itself behind its last tile (no k_agc_apply launch)
};

// cabs for the complex AGC.  The reference's cabs is the C library's hypot; the oracle runs against glibc 2.35, whose
// hypot (C. F. Borges, "An improved algorithm for hypot(a,b)", 2019; the variant without fused multiply-add) is
// restated here operation for operation, so that the magnitudes, and with them every AGC decision, are the
// oracle's bit for bit (2 x 10^7 random and integer-valued pairs checked against the library on the host).  No
// scaling branch: the AGC's magnitudes stay far from the thresholds (2^511, 2^-459) where the library rescales.
__device__ __forceinline__ double cabs_cr(double x, double y)
{
#pragma clang fp contract(off)
    double ax = fabs(x), ay = fabs(y);
    if (ax < ay) { const double t = ax; ax = ay; ay = t; }
    if (ay <= ax * 0x1p-54) return ax + ay;
    double h = sqrt(ax * ax + ay * ay), t1, t2;
    if (h <= 2.0 * ay) {
        const double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        const double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h;
}

// One sample of the reference's state machine, verbatim (quisk.c:2076-2131).
struct AgcConst { double rel, tr, keep, ceil_, bs; int buf_size; bool fm; };
__device__ __forceinline__ void agc_step(AgcState &s, const double m, const AgcConst &c)
{
#pragma clang fp contract(off)
    if (!s.is_clipping) {
        if (m * s.gain > c.ceil_) {
            s.target_gain = c.ceil_ / m;
            s.delta = (s.gain - s.target_gain) / c.bs;
            s.is_clipping = 1;
            s.themax = m;
            s.gain -= s.delta;
        } else if (s.index_read == s.index_start) {
            const double clip_gain = c.ceil_ / s.themax;
            s.target_gain = c.fm ? clip_gain : (c.rel > clip_gain ? clip_gain : c.rel);
            s.themax = m;
            s.gain = s.gain * c.keep + s.target_gain * c.tr;
        } else {
            if (s.themax < m) s.themax = m;
            s.gain = s.gain * c.keep + s.target_gain * c.tr;
        }
    } else {
        if (m > s.themax) {
            s.themax = m;
            s.target_gain = c.ceil_ / m;
            const double d = (s.gain - s.target_gain) / c.bs;
            if (d > s.delta) s.delta = d;
        }
        s.gain -= s.delta;
        if (s.gain <= s.target_gain) {
            s.is_clipping = 0;
            s.gain = s.target_gain;
            s.themax = m;
            s.index_start = s.index_read;
        }
    }
    if (++s.index_read >= c.buf_size) s.index_read = 0;
}

// The serial part of a speculation pass: G_0 = g0, G_{k+1} = G_k * keep + add (MUL) or G_k + add (!MUL), and
// lane L >= start must end up holding G_{L - start}, the gain in force at its sample (lanes below `start` keep
// g0).  Every lane runs the same recurrence, and the EXEC mask retires one lane per step (exec <<= 1), so a lane
// simply stops at its own value: two dependent FP64 operations per sample and nothing else on the vector pipe --
// no selects, no LDS.  v_mul_f64 / v_add_f64 are the separately rounded operations of the reference
// (quisk.c:2112, 2117); the compiler cannot express the shifting EXEC mask, hence the assembly.  The mask starts
// shifted by `start`, and the chain leaves at a quarter mark once no lane is left.
#define AGC_S1(MULOP) "s_lshl_b64 exec, exec, 1\n\t" MULOP "v_add_f64 %[g], %[g], %[t]\n\t"
#define AGC_S4(M) AGC_S1(M) AGC_S1(M) AGC_S1(M) AGC_S1(M)
#define AGC_S16(M) AGC_S4(M) AGC_S4(M) AGC_S4(M) AGC_S4(M)
#define AGC_S15(M) AGC_S4(M) AGC_S4(M) AGC_S4(M) AGC_S1(M) AGC_S1(M) AGC_S1(M)
#define AGC_CHAIN(M) "s_mov_b64 %[sv], exec\n\t" "s_lshl_b64 exec, exec, %[st]\n\t" \
                     AGC_S16(M) "s_cbranch_execz 2f\n\t" AGC_S16(M) "s_cbranch_execz 2f\n\t" AGC_S16(M) "s_cbranch_execz 2f\n\t" \
                     AGC_S15(M) "2:\n\t" "s_mov_b64 exec, %[sv]\n\t"
template <bool MUL>
__device__ __forceinline__ double agc_chain(double g0, double keep, double add, int start)
{
    double gv = g0;
    unsigned long long save;
    if (MUL)
        asm volatile(AGC_CHAIN("v_mul_f64 %[g], %[g], %[k]\n\t")
                     : [g] "+v"(gv), [sv] "=&s"(save) : [k] "v"(keep), [t] "v"(add), [st] "s"(start) : "scc");
    else
        asm volatile(AGC_CHAIN("")
                     : [g] "+v"(gv), [sv] "=&s"(save) : [t] "v"(add), [st] "s"(start) : "scc");
    return gv;
}

// The record of a ring slot: {end, add} go out as one 16-byte store from lane 0, the tag behind them
struct __attribute__((aligned(16))) AgcSlot {
    double end, add;            // gain after the tile's last sample; step constant at its first
    double add1;                // \ only for a tile that is not PLAIN (not clipping, no window boundary in it):
    int bl, clip;               // / step constant behind the boundary, the boundary's lane (64: none), the path
    int tag;                    // plain << 31 | generation << 20 | tile, -1 nothing
    int pad[3];
};
// a slot of the ring: the tile's 64 gains and its record, 1 KiB apart (agc_tight_run makes slot addresses by mask)
struct __attribute__((aligned(1024))) AgcRingSlot { double gv[64]; AgcSlot rec; };
// Three WAVES per receiver.  The state machine is serial in time, but almost every sample takes one of two
// trivial paths: "not clipping, no new clip, not at the window boundary" (gain = gain*(1-tr) + target*tr,
// themax = max(themax, m)) or "clipping, no new maximum, ramp not finished" (gain -= delta).  A tile of 64
// samples is therefore run SPECULATIVELY down the current path (agc_chain: 63 dependent steps, about 350 ns,
// the floor of this kernel) and then tested for an event (clip start, window boundary, new maximum,
// end of ramp) in one lane-parallel pass; samples in front of the first event are committed, the event sample
// goes through the full state machine (agc_step), and speculation restarts behind it.
//   wave 0 (the runner) runs the chains, tile after tile, each continuing from the end of the one before as if
//     no CLIP event happened.  Window boundaries it foresees itself: their position is a counter, and the new
//     target needs only the maximum of |x| over the window, which it tracks like the walker does (max is exact:
//     no rounding to reproduce); the tile with the boundary in it is chained in two parts, with the step
//     constant before and after.  It leaves the 64 gains of a tile in a ring of AGC_RING slots in LDS together
//     with what it assumed (step constants, boundary lane, path) and the gain it ends with;
//   wave 1 (the walker) owns the state and never runs a chain.  It takes the runner's gains from sample `start`
//     of a tile on whenever its own state there is the one the runner's chain passed through, bit for bit (the
//     same instructions on the same operands: the same values), which also holds behind an event that left gain
//     and target where they would have been (a window boundary whose new target equals the old one).  Otherwise
//     it posts the true state and where it holds (a RESTART: tile, first sample) and waits for the runner to
//     make that tile again from there;
//   wave 2 (the fetcher) brings |x| into LDS a chunk of AGC_CHUNK tiles ahead: a trip to memory under the front
//     end's load is several tiles long, and nobody waits for it here.
// The waves hand over through tags and counters in LDS that the consumer polls (no barrier: the runner never
// waits for the walker unless the ring is full).  Every floating-point operation is the reference's own, in its
// order, without FMA contraction: the result is bit-identical to a sample-by-sample walk (tests check it against
// the oracle).  Every poll is bounded: a wave that waits AGC_SPIN_LIMIT polls gives up and the walker poisons
// the state (NaN gain) rather than hang the GPU.  Control flow is kept flat on purpose (one loop, one chain of
// ifs, no early exits): a taken branch costs this single wave some twenty cycles, and the compiler's loop-exit
// unification turns `continue` / `return` into many of them.
#ifndef AGC_RING
#define AGC_RING 16
#endif
#ifndef AGC_CHUNK
#define AGC_CHUNK 32
#endif
#define AGC_SPIN_LIMIT (1 << 22)
#define AGC_LDS_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// a 32-bit LDS store from lane 0 alone, without a branch the compiler can see
__device__ __forceinline__ void lds_store_lane0(const unsigned addr, const int v)
{
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t" "s_mov_b64 exec, 1\n\t" "ds_write_b32 %[a], %[v]\n\t" "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv) : [a] "v"(addr), [v] "v"(v) : "memory");
}
// The walker's view of a ring slot in ONE trip to the LDS: the tag first, then the record, the tile's gains and its |x| -- the
// LDS executes a wave's reads in the order they were issued and the runner wrote the tag last, so a tag that is the wanted one
// vouches for what was read behind it.  (A tag read and a dependent second read cost this wave two LDS latencies per tile,
// each several hundred cycles beside the front end's traffic.)
typedef double qk_d2 __attribute__((ext_vector_type(2)));
typedef int qk_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void agc_take(const unsigned a_rec, const unsigned a_gv, const unsigned a_mv, int &tag, qk_d2 &r0, qk_i4 &r1,
                                         double &gv, double &mv)
{
    asm volatile("ds_read_b32 %[tg], %[ar] offset:32\n\t" "ds_read_b128 %[r0], %[ar]\n\t" "ds_read_b128 %[r1], %[ar] offset:16\n\t"
                 "ds_read_b64 %[gv], %[ag]\n\t" "ds_read_b64 %[mv], %[am]\n\t" "s_waitcnt lgkmcnt(0)"
                 : [tg] "=&v"(tag), [r0] "=&v"(r0), [r1] "=&v"(r1), [gv] "=&v"(gv), [mv] "=&v"(mv)
                 : [ar] "v"(a_rec), [ag] "v"(a_gv), [am] "v"(a_mv) : "memory");
}
// The runner's tight loop -- `count` whole PLAIN tiles (no clip, no window boundary, room in the ring for all of them) -- as one
// block of assembly.  What bounds it: a lone wave is given ONE issue slot every four cycles whatever the instruction
// (profiles/ub/agc_chain_shadow.hip: every instruction added to the chain, scalar ones between two dependent FP64 operations
// included, costs 4.0 cycles; nothing rides in a shadow), so the loop is a count of instructions and is written as one.
//   - The recurrence without a shift of EXEC per step: lane L must make L steps, L = b5..b0, so the lanes with bit 5 set make
//     32 steps under one mask, then those with bit 4 set 16 steps, ... 1 step: every lane applies the same operation to its own
//     running value L times in a row, which is the sample-by-sample sequence G_0 .. G_L bit for bit, in 63 x 2 + 6 slots
//     instead of 63 x 3.
//   - A tile's hand-over is deferred to the head of the next tile's pass (its end gain is lane 0 of the next chain register,
//     which no mask ever touches): one 8-byte store of the gains, then from lane 0 one 16-byte store {end, step constant} and
//     the tag -- the tag last.  The two chain registers alternate and each is followed by a copy of the step constant, so that
//     {end, step} is one register quadruple.  The first pass hands over to a spare slot nobody reads.
//   - Ring slots are 1 KiB apart in a 16 KiB-aligned ring: slot address = (running tile count << 10 & mask) | lane base.
//   - No ring test inside: the caller looks at the walker's progress and asks for no more tiles than there is room for.
// Wait states of the gfx940 family (VALU-written VGPR -> v_readlane: 1; VALU-written SGPR -> VALU operand: 2) are kept by the
// order of the instructions.  v112-v131, s44-s45 and s[60:71] are the block's own (clobbers).
#define AGC_B1(X) "v_mul_f64 " X ", " X ", %[k]\n\t" "v_add_f64 " X ", " X ", %[t]\n\t"
#define AGC_B2(X) AGC_B1(X) AGC_B1(X)
#define AGC_B4(X) AGC_B2(X) AGC_B2(X)
#define AGC_B8(X) AGC_B4(X) AGC_B4(X)
#define AGC_B16(X) AGC_B8(X) AGC_B8(X)
#define AGC_B32(X) AGC_B16(X) AGC_B16(X)
#define AGC_BIN(X) "s_mov_b64 exec, s[60:61]\n\t" AGC_B32(X) "s_mov_b64 exec, s[62:63]\n\t" AGC_B16(X) "s_mov_b64 exec, s[64:65]\n\t" AGC_B8(X) \
                   "s_mov_b64 exec, s[66:67]\n\t" AGC_B4(X) "s_mov_b64 exec, s[68:69]\n\t" AGC_B2(X) "s_mov_b64 exec, s[70:71]\n\t" AGC_B1(X) \
                   "s_mov_b64 exec, -1\n\t"
#define AGC_MASKS "s_mov_b32 s60, 0\n\t" "s_mov_b32 s61, -1\n\t" "s_mov_b32 s62, 0xffff0000\n\t" "s_mov_b32 s63, 0xffff0000\n\t" \
                  "s_mov_b32 s64, 0xff00ff00\n\t" "s_mov_b32 s65, 0xff00ff00\n\t" "s_mov_b32 s66, 0xf0f0f0f0\n\t" "s_mov_b32 s67, 0xf0f0f0f0\n\t" \
                  "s_mov_b32 s68, 0xcccccccc\n\t" "s_mov_b32 s69, 0xcccccccc\n\t" "s_mov_b32 s70, 0xaaaaaaaa\n\t" "s_mov_b32 s71, 0xaaaaaaaa\n\t"
// hand-over of the tile whose gains are in Y; X (with the step constant behind it: XQ) is the chain register of the tile after it
#define AGC_HAND(XQ, Y) "ds_write_b64 v115, " Y "\n\t" "s_mov_b64 exec, 1\n\t" "ds_write_b128 v115, " XQ " offset:512\n\t" \
                        "ds_write_b32 v115, v117 offset:544\n\t"
// one tile: chain in X (low / high register XL / XH), the tile before it (gains in Y) handed over, the next tile's first gain into Y
#define AGC_TILE(X, XQ, XL, XH, Y) \
        "ds_read_b64 v[112:113], v118\n\t" AGC_HAND(XQ, Y) AGC_BIN(X) \
        "v_readlane_b32 s44, v" XL ", 63\n\t" "v_readlane_b32 s45, v" XH ", 63\n\t" \
        "v_add_u32 v119, 0x400, v119\n\t" "v_and_or_b32 v115, v119, v120, v121\n\t" \
        "v_mul_f64 " Y ", %[k], s[44:45]\n\t" "v_add_f64 " Y ", " Y ", %[t]\n\t" \
        "v_add_u32 v117, 1, v117\n\t" "v_add_u32 v118, 8, v118\n\t" "s_waitcnt lgkmcnt(0)\n\t" "v_max_f64 %[rm], %[rm], v[112:113]\n\t"
__device__ __forceinline__ void agc_tight_run(double &g0, const double keep, const double add, double &rmax, const int count, const int j,
                                              const int tag_j, const unsigned a_ring_lane, const unsigned a_spare_lane, const unsigned a_tmax)
{
    int left = count - 1;                   // tiles behind the one in hand
    asm volatile(
        AGC_MASKS
        "v_mov_b64 v[124:125], %[g0]\n\t" "v_mov_b64 v[128:129], %[g0]\n\t" "v_mov_b64 v[126:127], %[t]\n\t" "v_mov_b64 v[130:131], %[t]\n\t"
        "v_mov_b32 v115, %[aspare]\n\t" "v_mov_b32 v117, %[tagm1]\n\t" "v_mov_b32 v118, %[atm]\n\t" "v_mov_b32 v119, %[jm1]\n\t"
        "v_mov_b32 v120, %[rmask]\n\t" "v_mov_b32 v121, %[aring]\n\t"
        "1:\n\t"
        AGC_TILE("v[124:125]", "v[124:127]", "124", "125", "v[128:129]")
        "s_add_u32 %[left], %[left], -1\n\t" "s_cbranch_scc0 2f\n\t"
        AGC_TILE("v[128:129]", "v[128:131]", "128", "129", "v[124:125]")
        "s_add_u32 %[left], %[left], -1\n\t" "s_cbranch_scc1 1b\n\t"
        // left behind an even number of tiles: the last one's gains are in 128:129, the gain after it in 124:125
        AGC_HAND("v[124:127]", "v[128:129]") "s_mov_b64 exec, -1\n\t" "v_mov_b64 %[g0], v[124:125]\n\t" "s_branch 3f\n\t"
        "2:\n\t"
        AGC_HAND("v[128:131]", "v[124:125]") "s_mov_b64 exec, -1\n\t" "v_mov_b64 %[g0], v[128:129]\n\t"
        "3:\n\t"
        : [g0] "+v"(g0), [rm] "+v"(rmax), [left] "+s"(left)
        : [k] "v"(keep), [t] "v"(add), [aring] "v"(a_ring_lane), [aspare] "v"(a_spare_lane), [atm] "v"(a_tmax),
          [tagm1] "v"(tag_j - 1), [jm1] "v"((j - 1) << 10), [rmask] "i"((AGC_RING - 1) << 10)
        : "v112", "v113", "v115", "v117", "v118", "v119", "v120", "v121", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131",
          "s44", "s45", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "scc", "memory");
}
// a whole chain from lane 0 (start = 0) by the same masks, for the tiles made outside the tight loop
__device__ __forceinline__ double agc_chain_bin(const double g0, const double keep, const double add)
{
    double gv = g0;
    asm volatile(AGC_MASKS
                 "s_mov_b64 exec, s[60:61]\n\t" AGC_B32("%[g]") "s_mov_b64 exec, s[62:63]\n\t" AGC_B16("%[g]") "s_mov_b64 exec, s[64:65]\n\t" AGC_B8("%[g]")
                 "s_mov_b64 exec, s[66:67]\n\t" AGC_B4("%[g]") "s_mov_b64 exec, s[68:69]\n\t" AGC_B2("%[g]") "s_mov_b64 exec, s[70:71]\n\t" AGC_B1("%[g]")
                 "s_mov_b64 exec, -1\n\t"
                 : [g] "+v"(gv) : [k] "v"(keep), [t] "v"(add)
                 : "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "scc");
    return gv;
}

__device__ __forceinline__ void agc_apply_one(const AgcArgs &a, const int rx, const int i);
__global__ __launch_bounds__(192) void k_agc_gain(AgcArgs a)
{
#pragma clang fp contract(off)
    // runner -> walker: the gains of tile k, what the runner assumed and what it ends with, and the tag that says so, in slot
    // k % AGC_RING (one spare slot nobody reads)
    __shared__ __attribute__((aligned(16384))) AgcRingSlot ring[AGC_RING + 1];
    __shared__ double tmaxs[2][AGC_CHUNK];          // fetcher -> runner: max |x| over each tile of the chunk
    static_assert(sizeof(AgcRingSlot) == 1024 && (AGC_RING == 16 || AGC_RING == 8) && offsetof(AgcRingSlot, rec) == 512 && offsetof(AgcSlot, end) == 0 && offsetof(AgcSlot, add) == 8 &&
                  offsetof(AgcSlot, tag) == 32, "agc_tight_run writes these offsets");
    __shared__ int wstat;                           // walker -> runner, fetcher: restart generation << 20 | tiles finished
    __shared__ int fetched;                         // fetcher -> walker: chunks in LDS
    __shared__ double rs_g, rs_add, rs_max;         // walker -> runner, with a new generation: the true state at sample
                                                    // rs_start of tile rs_tile
    __shared__ int rs_clip, rs_tile, rs_start, rs_ir, rs_is;
    __shared__ int aborted;
    __shared__ double sm[2][AGC_CHUNK * 64];        // fetcher -> walker: |x| of chunk c in buffer c & 1
    const int rx = blockIdx.x;
    const int cl = a.cls[rx];
    const int n_new = a.n_c[cl];
    if (!a.started[rx] || n_new <= 0) {             // (the whole workgroup)
        if (a.apply_here) for (int i = threadIdx.x; i < n_new; i += 192) agc_apply_one(a, rx, i);    // (a first block goes out untouched)
        return;
    }
    if (threadIdx.x < AGC_RING) ring[threadIdx.x].rec.tag = -1;
    // (agc_tight_run makes slot addresses by OR-ing a masked count into the ring's base: a ring that the compiler did not put on
    // a 16 KiB boundary must not go unnoticed -- the block is then poisoned like a timed-out hand-over)
    if (threadIdx.x == 0) { wstat = 0; fetched = 0; aborted = ((unsigned)reinterpret_cast<size_t>(&ring[0]) & 0x3fffu) ? 1 : 0; }
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);      // latency-bound waves among the front end's throughput-bound ones: issue first
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const double *__restrict__ p = sv_ptr(a.mag_c[cl], rx, 0);
    double *__restrict__ g = a.gains + (long long)rx * a.gain_stride;
    AgcState s = a.state[rx];
    AgcConst c;
    c.rel = a.release_gain[rx]; c.tr = a.time_release; c.keep = 1.0 - a.time_release; c.ceil_ = a.ceiling;
    c.bs = (double)a.buf_size; c.buf_size = a.buf_size;
    const int md = a.mode[rx];
    c.fm = (md == 5 || md == 13);
    const int T = (n_new + 63) >> 6;                // tiles
    const int last = n_new - 1;
    // the constant added per step on the current path
    auto step_const = [&](const AgcState &st) { return st.is_clipping ? -st.delta : st.target_gain * c.tr; };
    auto one_more = [&](double glast, double add, int clip) { return clip ? glast + add : glast * c.keep + add; };
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto tag_of = [](int gen, int tile) { return ((gen & 0x7ff) << 20) | tile; };
    // the hand-over words: relaxed workgroup-scope atomics on LDS (read anew at every poll, plain ds_read / ds_write;
    // `volatile` would turn them into system-scope FLAT accesses, a trip to memory each)
    auto peek = [](int *w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto poke = [](int *w, int v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto unid = [](double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); };
    auto lds_addr = [](const void *q) { return (unsigned)reinterpret_cast<size_t>(q); };    // (the low word of an LDS pointer is its offset)

    if (wave == 0) {
        // ---------------- the runner ----------------
        // (j, st): the tile to make and its first sample; the state there: gain g0, step constant add, path clip,
        // maximum rmax, ring indices ir / is (the walker's index_read / index_start)
        int j = 0, st = 0, gen = 0, waits = 0, f_known = 0;
        double g0 = s.gain, add = step_const(s), rmax = s.themax;
        int clip = s.is_clipping, ir = s.index_read, is = s.index_start;
        bool alive = true;
#ifdef AGC_PROF
        long long ap[4] = {0, 0, 0, 0}, ap_last = clock64(), ap_t0 = ap_last;
        long long ac[6] = {0, 0, 0, 0, 0, 0};
#define AGC_COUNT(k) ac[k]++
#define AGC_PROBE(k) do { const long long t_ = clock64(); ap[k] += t_ - ap_last; ap_last = t_; } while (0)
#else
#define AGC_PROBE(k) do { } while (0)
#define AGC_COUNT(k) do { } while (0)
#endif
        // max of |x| over the lanes in `in`, on top of m: walk the successive record holders (usually none)
        auto walk_max = [&](double m, const double mv, const bool in) {
            unsigned long long over = __ballot(in && mv > m);
            while (over) {
                AGC_COUNT(0);
                m = readlane_f64(mv, (int)__ffsll((long long)over) - 1);
                over = __ballot(in && mv > m);
            }
            return m;
        };
        auto hand_over = [&](const int jj, const double gs, const double a0, const double pend, const bool plain,
                             const double a1, const int cl_, const int bl) {
            const int slot = jj & (AGC_RING - 1);
            // The tag goes last and nothing waits in between: the LDS executes one wave's instructions in the order
            // they were issued, so whoever sees the tag sees the tile.  (The compiler is told not to reorder them.)
            ring[slot].gv[lane] = gs;
            if (lane == 0) {
                ring[slot].rec.add = a0; ring[slot].rec.end = pend;
                if (!plain) { ring[slot].rec.add1 = a1; ring[slot].rec.bl = bl; ring[slot].rec.clip = cl_; }
                asm volatile("" ::: "memory");
                poke(&ring[slot].rec.tag, tag_of(gen, jj) | (plain ? (int)0x80000000 : 0));
            }
        };
        while (alive) {
            if (j < T && j / AGC_CHUNK >= f_known) f_known = uni(peek(&fetched));   // (one look per chunk, as a rule)
            bool can = j < T && j / AGC_CHUNK < f_known;
            if (can && !clip && st == 0) {
                // ---- the tight loop: the whole tiles in front of the window boundary, within the chunk in LDS.  A
                // lone wave issues an instruction every four or five cycles at best: whatever is not the chain
                // (850 cycles) is kept to a few dozen instructions here.
                int ahead = is - ir;
                if (ahead < 0) ahead += a.buf_size;
                int jend = (j / AGC_CHUNK + 1) * AGC_CHUNK;
                if (T < jend) jend = T;
                if (j + (ahead >> 6) < jend) jend = j + (ahead >> 6);
                const int j0 = j;
                bool room_out = false;              // the ring, not jend, cut the run short
                if (j < jend) {
                    // as many tiles as the ring has room for (the walker's progress, looked at once per run; a restart shows
                    // as another generation and sends the run to the general step)
                    const int ws = uni(peek(&wstat));
                    int room = (ws >> 20) == (gen & 0x7ff) ? AGC_RING - (j - (ws & 0xfffff)) : 0;
                    int count = jend - j;
                    if (room < count) { count = room; room_out = true; }
                    if (count > 0) {
                        AGC_PROBE(0);
                        agc_tight_run(g0, c.keep, add, rmax, uni(count), uni(j), uni((int)(tag_of(gen, j) | 0x80000000)), lds_addr(&ring[0].gv[lane]),
                                      lds_addr(&ring[AGC_RING].gv[lane]), lds_addr(&tmaxs[(j / AGC_CHUNK) & 1][j % AGC_CHUNK]));
                        j += count;
                        AGC_PROBE(1);
                    }
                }
                const bool ran_out = j >= jend && !room_out;     // (not: left early for the general step)
                ir += (j - j0) << 6;
                while (ir >= a.buf_size) ir -= a.buf_size;
                if (j < T && j / AGC_CHUNK >= f_known) f_known = uni(peek(&fetched));
                can = j < T && j / AGC_CHUNK < f_known;
                if (ran_out && can) {
                    // The loop stopped at the end of a chunk or in front of the tile with the window boundary in it.  That
                    // tile is made here, as lean as the loop above (the general step below does the same at twice the cost:
                    // one tile in eleven): chain up to the boundary with the step constant so far, the new target from the
                    // window's maximum (quisk.c:2095-2103), chain behind the boundary with the new step constant.
                    int left = is - ir;
                    if (left < 0) left += a.buf_size;
                    bool made = true;
                    if (left < 64) {
                        AGC_PROBE(0);
                        const int bl = left;
                        const int ws_v = peek(&wstat);
                        const double mv = sm[(j / AGC_CHUNK) & 1][(j % AGC_CHUNK) * 64 + lane];
                        double gs = agc_chain_bin(g0, c.keep, add);             // (right up to lane bl; the lanes behind are made again)
                        double nmax = walk_max(rmax, mv, lane < bl);
                        const double clip_gain = c.ceil_ / nmax;
                        const double tnew = c.fm ? clip_gain : (c.rel > clip_gain ? clip_gain : c.rel);
                        const double add1 = tnew * c.tr;
                        double pend = readlane_f64(gs, bl) * c.keep + add1;
                        if (bl < 63) {
                            const double g2 = agc_chain<true>(pend, c.keep, add1, bl + 1);
                            if (lane > bl) gs = g2;
                            pend = readlane_f64(g2, 63) * c.keep + add1;
                        }
                        nmax = walk_max(readlane_f64(mv, bl), mv, lane > bl);
                        made = (unsigned)(j - (uni(ws_v) - (gen << 20))) < (unsigned)AGC_RING;
                        if (made) {
                            hand_over(j, gs, add, pend, false, add1, 0, bl);
                            g0 = pend; add = add1; rmax = nmax; ++j;
                            ir += 64;
                            if (ir >= a.buf_size) ir -= a.buf_size;
                        }
                        AGC_PROBE(2);
                    }
                    if (made) continue;             // back into the loop above (or to whatever the next tile needs)
                }
            }
            // ---- the general step: one tile, whatever it holds
            const int ws_v = peek(&wstat);
            double gs = g0, pend = g0, add1 = add, nmax = rmax;
            int bl = 64;
            if (can) {
                const double mv = sm[(j / AGC_CHUNK) & 1][(j % AGC_CHUNK) * 64 + lane];
                int ahead = is - ir;                // samples from sample st to the window boundary
                if (ahead < 0) ahead += a.buf_size;
                if (clip) {
                    gs = agc_chain<false>(g0, 0.0, add, st);
                    pend = readlane_f64(gs, 63) + add;
                } else if (st + ahead > 63) {       // no boundary in this tile
                    gs = agc_chain<true>(g0, c.keep, add, st);
                    pend = readlane_f64(gs, 63) * c.keep + add;
                    nmax = walk_max(rmax, mv, lane >= st);
                } else {                            // the boundary is sample bl: quisk.c:2095-2103 there, on the maximum so far
                    bl = st + ahead;
                    gs = agc_chain<true>(g0, c.keep, add, st);      // (right up to lane bl; the lanes behind are made again)
                    nmax = walk_max(rmax, mv, lane >= st && lane < bl);
                    const double clip_gain = c.ceil_ / nmax;
                    const double tnew = c.fm ? clip_gain : (c.rel > clip_gain ? clip_gain : c.rel);
                    add1 = tnew * c.tr;
                    const double gnext = readlane_f64(gs, bl) * c.keep + add1;
                    pend = gnext;
                    if (bl < 63) {
                        const double g2 = agc_chain<true>(gnext, c.keep, add1, bl + 1);
                        if (lane > bl) gs = g2;
                        pend = readlane_f64(g2, 63) * c.keep + add1;
                    }
                    nmax = walk_max(readlane_f64(mv, bl), mv, lane > bl);
                }
            }
            const int ws = uni(ws_v), w = ws & 0xfffff, rs = ws >> 20;
            if (rs == gen && can && j - w < AGC_RING) {             // hand the tile over, go on behind it
                hand_over(j, gs, add, pend, false, add1, clip, bl);
                ir += 64 - st;
                if (ir >= a.buf_size) ir -= a.buf_size;
                g0 = pend; add = add1; rmax = nmax; st = 0; ++j;
                AGC_COUNT(4);
                waits = 0;
            } else if (rs != gen) {                 // a restart: continue from the truth (the tile just made is dropped)
                AGC_LDS_WAIT();
                j = uni(rs_tile); st = uni(rs_start); clip = uni(rs_clip); ir = uni(rs_ir); is = uni(rs_is);
                g0 = readlane_f64(rs_g, 0); add = readlane_f64(rs_add, 0); rmax = readlane_f64(rs_max, 0);
                gen = rs;
                AGC_COUNT(5);
                waits = 0;
            } else {                                // ring full, chunk not there yet, or all tiles made: the walker may still restart
                if (j >= T && w >= T) alive = false;
                AGC_PROBE(0);
                __builtin_amdgcn_s_sleep(2);
                AGC_PROBE(3);
                AGC_COUNT(can ? 1 : (j < T ? 2 : 3));
                if (++waits > AGC_SPIN_LIMIT || uni(peek(&aborted))) { poke(&aborted, 1); alive = false; }
            }
        }
#ifdef AGC_PROF
        if (lane == 0 && rx == 0) printf("agc runner: tiles %d, cycles: other %lld tight run %lld boundary tiles %lld sleeping %lld, whole wave %lld; waits: ring full %lld, chunk not fetched %lld, all made %lld; record-walk steps %lld; general-step tiles %lld, restarts %lld\n", T, ap[0], ap[1], ap[2], ap[3], clock64() - ap_t0, ac[1], ac[2], ac[3], ac[0], ac[4], ac[5]);
#endif
    } else if (wave == 1) {
        // ---------------- the walker ----------------
        int gen = 0;
        int dead = uni(peek(&aborted));             // (read through uni(): the tests on it are wave-uniform branches; set from the start
                                                    // by the ring-alignment check above)
#ifdef AGC_PROF
        long long wp_wait = 0, wp_waited = 0, wp_take = 0; const long long wp_t0 = clock64();
#endif
        double cur_add = step_const(s);             // its step constant: changes only on the general path
        // what the slot of the current tile holds
        double spec_gv = 0.0, sa = 0.0, sa1 = 0.0, se = 0.0, cur_mv = 0.0;
        int sc = 0, sb = 64;
        // wait for tile k of the current generation, take it
        auto acquire = [&](const int k) {
            const int slot = k & (AGC_RING - 1);
            const int want = tag_of(gen, k);
            const unsigned a_rec = lds_addr(&ring[slot].rec), a_gv = lds_addr(&ring[slot].gv[lane]);
            const unsigned a_mv = lds_addr(&sm[(k / AGC_CHUNK) & 1][(k % AGC_CHUNK) * 64 + lane]);
            int spins = 0, tagv;
            qk_d2 r0; qk_i4 r1;
#ifdef AGC_PROF
            const long long wq0 = clock64();
#endif
            agc_take(a_rec, a_gv, a_mv, tagv, r0, r1, spec_gv, cur_mv);
            int tag = uni(tagv);
#ifdef AGC_PROF
            const long long wq1 = clock64();
            wp_take += wq1 - wq0;
            if ((tag & 0x7fffffff) != want) wp_waited++;
#endif
            while (!uni(dead) && (tag & 0x7fffffff) != want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > AGC_SPIN_LIMIT || uni(peek(&aborted))) dead = 1;
                agc_take(a_rec, a_gv, a_mv, tagv, r0, r1, spec_gv, cur_mv);
                tag = uni(tagv);
            }
#ifdef AGC_PROF
            wp_wait += clock64() - wq1;
#endif
            // (what an asm block returns is per-lane to the compiler: the record's fields are made wave-uniform again, or every
            // test on them turns into EXEC-mask bookkeeping)
            se = unid(r0.x); sa = unid(r0.y);
            sa1 = sa; sb = 64; sc = 0;              // a plain tile
            if (tag >= 0) { sa1 = __hiloint2double(uni(r1.y), uni(r1.x)); sb = uni(r1.z); sc = uni(r1.w); }
        };
        // ask the runner for tile k from sample `start` on, in the walker's state
        auto restart = [&](const int k, const int start) {
            ++gen;
            if (lane == 0) {
                rs_g = s.gain; rs_add = cur_add; rs_max = s.themax; rs_clip = s.is_clipping; rs_tile = k; rs_start = start;
                rs_ir = s.index_read; rs_is = s.index_start;
            }
            AGC_LDS_WAIT();
            if (lane == 0) poke(&wstat, tag_of(gen, k));
        };
        auto tile = [&](const int k) {
            const double mv = cur_mv;               // (|x| of the tile: came with the slot)
            const int i0 = k << 6;
            const int cnt = (n_new - i0 < 64) ? n_new - i0 : 64;
            // The common tile, at the least cost: the runner's gains are valid from sample 0, not clipping, a whole
            // tile, the window boundary not in it and no clip: commit all 64 samples.  (Exactly what one pass of the
            // loop below does; the gain after the tile is the runner's: the same operation on the same operands.
            // The comparisons are lane-wise on wave-uniform values: every lane gets the same verdict.)
            if (!s.is_clipping && cnt == 64) {
                int ahead = s.index_start - s.index_read;
                if (ahead < 0) ahead += a.buf_size;
                const bool quiet = (sc == 0) & (sb == 64) & (sa == cur_add) & (readlane_f64(spec_gv, 0) == s.gain) &
                                   !(mv * spec_gv > c.ceil_);
                if (ahead >= 64 && __ballot(quiet) == ~0ull) {
                    unsigned long long over = __ballot(mv > s.themax);
                    while (over) {
                        s.themax = readlane_f64(mv, (int)__ffsll((long long)over) - 1);
                        over = __ballot(mv > s.themax);
                    }
                    s.pad++;
                    s.index_read += 64;
                    if (s.index_read >= a.buf_size) s.index_read -= a.buf_size;
                    s.gain = se;
                    g[i0 + lane] = spec_gv;
                    return;
                }
            }
            double gv = 0.0;                                            // gain in force at this lane's sample
            int start = 0;
            while (start < cnt && !uni(dead)) {
                s.pad++;                                                // statistics: speculation passes
                const bool mine = lane >= start && lane < cnt;
                // The runner's gains serve from `start` on when the state there is the one its chain passed through at
                // that sample (lane L holds the gain in force at sample L); if not, it makes the tile again from here.
                // (Values read from LDS are per-lane to the compiler: the verdict is made wave-uniform.)
                const bool spec_ok = uni((int)((sc == s.is_clipping) & ((start > sb ? sa1 : sa) == cur_add) &
                                               (readlane_f64(spec_gv, start) == s.gain))) != 0;
                if (!spec_ok) { restart(k, start); acquire(k); }
                const double gs = spec_gv;
                if (lane >= start) gv = gs;
                bool ev;
                int fmax = cnt;                                         // first sample that is an event for a wave-uniform reason
                if (!s.is_clipping) {
                    // the window boundary (index_read == index_start) falls on a lane known in advance
                    int ahead = s.index_start - s.index_read;
                    if (ahead < 0) ahead += a.buf_size;
                    if (start + ahead < cnt) fmax = start + ahead;
                    ev = mine && (mv * gv > c.ceil_);
                } else {
                    ev = mine && ((mv > s.themax) || (gv - s.delta <= s.target_gain));
                }
                // behind lane sb the runner's gains follow its second step constant: whatever the walker's own view of the
                // window, it stops there and checks again (sample sb goes through agc_step: always right, only slower)
                if (sb >= start && sb < fmax) fmax = sb;
                const unsigned long long mask = __ballot(ev);
                int f = mask ? (int)__ffsll((long long)mask) - 1 : cnt;       // first event sample
                if (fmax < f) f = fmax;
                if (!s.is_clipping) {
                    // themax = max(themax, |x| of the committed samples): walk the successive record holders instead of
                    // reducing over the wave -- there are about ln(64) of them at worst on noise, usually none
                    const bool inr = lane >= start && lane < f;
                    unsigned long long over = __ballot(inr && mv > s.themax);
                    while (over) {
                        s.themax = readlane_f64(mv, (int)__ffsll((long long)over) - 1);
                        over = __ballot(inr && mv > s.themax);
                    }
                }
                s.index_read += f - start;
                if (s.index_read >= a.buf_size) s.index_read -= a.buf_size;
                if (f >= cnt) {
                    const double glast = readlane_f64(gs, cnt - 1);    // gain in force at the last sample: one more step
                    s.gain = one_more(glast, cur_add, s.is_clipping);
                    start = cnt;
                } else {
                    s.gain = readlane_f64(gv, f);
                    agc_step(s, readlane_f64(mv, f), c);
                    cur_add = step_const(s);
                    start = f + 1;
                }
            }
            g[i0 + lane] = gv;
            // the runner went on from (se, sa1) on the same path: if that is not where the tile ended, it starts the next one
            // again.  (Its window maximum is not compared: it follows the walker's by construction, and a wrong one would
            // only make a wrong target at the next boundary, which is caught there like any other.)
            if (!uni((int)((sc == s.is_clipping) & (sa1 == cur_add) & (se == s.gain))) && k + 1 < T) restart(k + 1, 0);
        };
        const int T_whole = n_new >> 6;             // tiles of 64 samples
        int f_seen = 0;                             // chunks known to be in LDS
        for (int k = 0; k < T && !uni(dead); ++k) {
            // ---- the quiet run: PLAIN tiles (no clip, no window boundary, the runner's step constant and gain are the walker's)
            // one after the other in a loop of a few dozen instructions.  The kernel is tens of KiB of code and shares its
            // instruction cache with the front end's; a tile that walks through acquire() and tile() below touches a dozen cache
            // lines far apart, and measured 1100 cycles against the 400 its instructions take.  Whatever this loop does not
            // like -- a tile that is not there yet, not plain, a chunk not fetched, a clip -- it leaves to the general code.
            if (!s.is_clipping) {
                int ahead = s.index_start - s.index_read;
                if (ahead < 0) ahead += a.buf_size;
                int kend = k + (ahead >> 6);                    // whole tiles in front of the window boundary
                if (T_whole < kend) kend = T_whole;
                while (k < kend) {
                    if (k / AGC_CHUNK >= f_seen) { f_seen = uni(peek(&fetched)); if (k / AGC_CHUNK >= f_seen) break; }
                    const int slot = k & (AGC_RING - 1);
                    int tagv; qk_d2 r0; qk_i4 r1; double gvv, mv;
                    agc_take(lds_addr(&ring[slot].rec), lds_addr(&ring[slot].gv[lane]),
                             lds_addr(&sm[(k / AGC_CHUNK) & 1][(k % AGC_CHUNK) * 64 + lane]), tagv, r0, r1, gvv, mv);
                    if (uni(tagv) != (int)(tag_of(gen, k) | 0x80000000)) break;         // not there yet, or not a plain tile
                    const bool quiet = ((int)(unid(r0.y) == cur_add) & (int)(readlane_f64(gvv, 0) == s.gain) & (int)!(mv * gvv > c.ceil_)) != 0;
                    if (__ballot(quiet) != ~0ull) break;
                    unsigned long long over = __ballot(mv > s.themax);
                    while (over) {
                        s.themax = readlane_f64(mv, (int)__ffsll((long long)over) - 1);
                        over = __ballot(mv > s.themax);
                    }
                    s.pad++;
                    s.index_read += 64;
                    if (s.index_read >= a.buf_size) s.index_read -= a.buf_size;
                    s.gain = unid(r0.x);
                    g[(k << 6) + lane] = gvv;
                    ++k;
                    lds_store_lane0(lds_addr(&wstat), tag_of(gen, k));
                }
            }
            if (k >= T) break;
            if (k % AGC_CHUNK == 0) {
                int spins = 0;
                while (!uni(dead) && uni(peek(&fetched)) <= k / AGC_CHUNK) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > AGC_SPIN_LIMIT || uni(peek(&aborted))) dead = 1;
                }
            }
            acquire(k);
            if (!uni(dead)) tile(k);
            // (lane 0 alone, by EXEC mask inside the asm: as an `if (lane == 0)` the store is a divergent branch to the compiler,
            // which then duplicates k + 1 into its arms -- and with that one phi the whole loop, k and the state included,
            // turns per-lane: every test in the walker became EXEC-mask bookkeeping)
            lds_store_lane0(lds_addr(&wstat), tag_of(gen, k + 1));
        }
        if (uni(dead)) { poke(&aborted, 1); s.gain = __builtin_nan(""); if (lane == 0 && a.abort_flag) *a.abort_flag = 1; }
        if (lane == 0) { poke(&wstat, tag_of(gen, T)); a.state[rx] = s; }
#ifdef AGC_PROF
        if (lane == 0 && rx == 0) printf("agc walker: whole wave %lld cycles, of them first look at the slot %lld, waiting for it %lld (three clock reads per tile, ~100 cycles each, are in these figures); tiles it had to wait for %lld of %d; speculation passes %d\n", clock64() - wp_t0, wp_take, wp_wait, wp_waited, T, s.pad);
#endif
    } else {
        // ---------------- the fetcher ----------------
        // Loads are unconditional (indices clamped to the block; lanes past the end are never `mine`).
        const int nch = (T + AGC_CHUNK - 1) / AGC_CHUNK;
        double v[AGC_CHUNK];
        auto load_chunk = [&](const int ch) {
#pragma unroll
            for (int u = 0; u < AGC_CHUNK; ++u) v[u] = p[min((ch * AGC_CHUNK + u) * 64 + lane, last)];
        };
        load_chunk(0);
        bool alive = true;
        for (int ch = 0; ch < nch && alive; ++ch) {
            // buffer ch & 1 is free once chunk ch-2 is walked
            int spins = 0;
            while (alive && (uni(peek(&wstat)) & 0xfffff) < (ch - 1) * AGC_CHUNK) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > AGC_SPIN_LIMIT || uni(peek(&aborted))) { poke(&aborted, 1); alive = false; }
            }
            double tm = 0.0;                        // lane u: the maximum over tile u
            const int nt = T - ch * AGC_CHUNK;      // tiles of this chunk that exist (a short block has a handful)
#pragma unroll
            for (int u = 0; u < AGC_CHUNK; ++u) {
                if (u < nt) {
                    double m = fabs(v[u]);
                    sm[ch & 1][u * 64 + lane] = m;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(m, off); m = o > m ? o : m; }
                    if (lane == u) tm = m;
                }
            }
            if (lane < AGC_CHUNK) tmaxs[ch & 1][lane] = tm;
            AGC_LDS_WAIT();
            if (lane == 0) poke(&fetched, ch + 1);
            if (ch + 1 < nch) load_chunk(ch + 1);   // in flight while the walker works through chunk ch
        }
    }
    if (a.apply_here) {
        // short blocks: the gains are all there once the walker is through -- the workgroup applies them itself (k_agc_apply's
        // operation per sample) instead of a launch behind this one
        __syncthreads();
        for (int i = threadIdx.x; i < n_new; i += 192) agc_apply_one(a, rx, i);
    }
}
// magnitudes of the complex classes' new samples, for k_agc_gain (grid: samples, 1, receivers)
__global__ __launch_bounds__(256) void k_agc_cabs(AgcArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int rx = blockIdx.z;
    const int cl = a.cls[rx];
    if (!a.cpx_c[cl] || i >= a.n_c[cl]) return;
    sv_ptr(a.mag_c[cl], rx, 0)[i] = cabs_cr(sv_ptr(a.in_c[cl], rx, 0)[i], sv_ptr(a.in_c[cl], rx, 1)[i]);
}
__device__ __forceinline__ void agc_apply_one(const AgcArgs &a, const int rx, const int i)
{
#pragma clang fp contract(off)
    const int cl = a.cls[rx];
    if (i >= a.n_c[cl]) return;
    const double *p = sv_ptr(a.in_c[cl], rx, 0) + i;
    if (a.cpx_c[cl]) {                      // stereo: Re = left, Im = right (quisk.c:2556-2558), gain on the complex sample
        const double *q = sv_ptr(a.in_c[cl], rx, 1) + i;
        double2 d;
        if (!a.started[rx]) {
            d = make_double2(p[0], q[0]);
        } else {
            const double g = a.gains[(long long)rx * a.gain_stride + i];
            d = make_double2(p[-a.buf_size] * g, q[-a.buf_size] * g);
            const double m = cabs_cr(d.x, d.y);
            if (m > QK_CLIP32) d = make_double2(d.x / m, d.y / m);
        }
        if (a.kill || (a.mute && a.mute[rx])) d = make_double2(0.0, 0.0);
        a.out[(long long)rx * a.out_stride + i] = d;
        return;
    }
    double d;
    if (!a.started[rx]) {
        d = p[0];
    } else {
        d = p[-a.buf_size] * a.gains[(long long)rx * a.gain_stride + i];
        const double m = fabs(d);
        if (m > QK_CLIP32) d = d / m;      // sic: the reference divides by the magnitude (quisk.c:2065)
    }
    if (a.kill || (a.mute && a.mute[rx])) d = 0.0;      // silence behind the AGC, whose state moves on
    a.out[(long long)rx * a.out_stride + i] = make_double2(d, d);
}
__global__ __launch_bounds__(256) void k_agc_apply(AgcArgs a)
{
    agc_apply_one(a, blockIdx.z, blockIdx.x * 256 + threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// End of block: move the last `hist` samples of every stream to the front of its buffer.
// ------------------------------------------------------------------------------------------------
#define QK_MAX_TAILS 24
struct TailDesc { double *d; long long rx_stride; int plane_stride, planes, hist, n; double *dst; int f32; const int *rx_list; int n_rx; };   // dst: same layout (ping-pong twin) or d itself; f32: the elements are floats; rx_list / n_rx: this entry's own receivers (n_rx > 0) in a launch that serves several chains
struct TailArgs { TailDesc t[QK_MAX_TAILS]; int count; const int *rx_list; };
__global__ __launch_bounds__(256) void k_save_tails(TailArgs a)
{
    // grid (count, 2 planes, receivers)
    const TailDesc td = a.t[blockIdx.x];
    if ((int)blockIdx.y >= td.planes || td.n <= 0) return;
    if (td.n_rx > 0 && (int)blockIdx.z >= td.n_rx) return;
    const int *rl = td.n_rx > 0 ? td.rx_list : a.rx_list;
    const int rx = rl ? rl[blockIdx.z] : blockIdx.z;
    const long long off = (long long)rx * td.rx_stride + (long long)blockIdx.y * td.plane_stride;
    if (td.f32) {
        const float *fb = reinterpret_cast<const float *>(td.d) + off;
        float *fd = reinterpret_cast<float *>(td.dst) + off;
        for (int c0 = 0; c0 < td.hist; c0 += 256) {
            const int i = c0 + threadIdx.x;
            float v = 0.f;
            if (i < td.hist) v = fb[i + td.n];
            __syncthreads();
            if (i < td.hist) fd[i] = v;
            __syncthreads();
        }
        return;
    }
    const double *base = td.d + off;
    double *dbase = td.dst + off;
    // dest[i] = src[i + n], i in [0, hist); ascending chunks keep reads ahead of writes when n < hist
    for (int c0 = 0; c0 < td.hist; c0 += 256) {
        const int i = c0 + threadIdx.x;
        double v = 0.0;
        if (i < td.hist) v = base[i + td.n];
        __syncthreads();
        if (i < td.hist) dbase[i] = v;
        __syncthreads();
    }
}

// add_rx_samples (quisk.c:2804-2834): signed integers of `bytes` bytes, left-justified into int32.
__global__ __launch_bounds__(256) void k_unpack_int(const unsigned char *raw, double2 *out, int n, int bytes, int big_endian)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned char *p = raw + (size_t)i * 2 * bytes;
    int32_t v[2];
    for (int c = 0; c < 2; c++) {
        uint32_t u = 0;
        for (int b = 0; b < bytes; b++) {
            const uint32_t byte = p[c * bytes + b];
            const int sig = big_endian ? (bytes - 1 - b) : b;       // significance of this byte within the sample
            u |= byte << (8 * (4 - bytes + sig));
        }
        v[c] = (int32_t)u;
    }
    out[i] = make_double2((double)v[0], (double)v[1]);
}

// ------------------------------------------------------------------------------------------------
// Input conditioning on the full-rate stream, in front of the spectrum and the mixer:
//   DCremove (sound.c:165-230), bandwidth > 1: c = x + dc alpha; y = c - dc; dc = c  (R. G. Lyons' DC blocker),
//   bandwidth == 1: y = x - dc_average, the average taken over two-second stretches (sound.c:198-222);
//   spectrum inversion (quisk.c:2315-2320): conjugate, applied behind the DC removal.
// The recurrence is one serial chain over the whole stream in the reference.  Here it is the same scan as
// k_serial_iir, over segments of 1024 samples: pass 0 finds every segment's end value from a zero state, a
// single wave chains the segment ends (k_dc_chain), pass 1 redoes the segment from its true start value and writes.
// Rounding order differs from the serial walk; alpha < 1, so the differences stay at the 1e-16 level.
// ------------------------------------------------------------------------------------------------
#define DC_K 16
#define DC_SEG (64 * DC_K)
struct DcArgs {
    const double2 *x; double2 *y; int n;
    double alpha, alpha_seg;       // alpha^DC_SEG
    double2 *seg_end;              // [nseg] pass 0 -> chain
    const double2 *seg_in;         // [nseg] chain -> pass 1
    int conj_out;
};
template <int PASS>
__global__ __launch_bounds__(64) void k_dc_iir(DcArgs a)
{
#pragma clang fp contract(off)
    const int seg = blockIdx.x, lane = threadIdx.x;
    const int base = seg * DC_SEG + lane * DC_K;
    int v = a.n - base;
    v = v < 0 ? 0 : (v > DC_K ? DC_K : v);
    double pw[DC_K];
    pw[0] = a.alpha;
#pragma unroll
    for (int k = 1; k < DC_K; k++) pw[k] = pw[k - 1] * a.alpha;
    double2 x[DC_K];
#pragma unroll
    for (int k = 0; k < DC_K; k++) x[k] = k < v ? a.x[base + k] : make_double2(0.0, 0.0);
    double2 l = make_double2(0.0, 0.0), lend = l;
    double fl = 1.0;
#pragma unroll
    for (int k = 0; k < DC_K; k++) {
        l = make_double2(x[k].x + l.x * a.alpha, x[k].y + l.y * a.alpha);
        x[k] = l;
        if (k < v) { lend = l; fl = pw[k]; }
    }
    double2 Sin = make_double2(0.0, 0.0), run = PASS ? a.seg_in[seg] : make_double2(0.0, 0.0);
    for (int L = 0; L < 64; L++) {
        if (lane == L) Sin = run;
        const double f = readlane_f64(fl, L);
        run = make_double2(readlane_f64(lend.x, L) + f * run.x, readlane_f64(lend.y, L) + f * run.y);
    }
    if (PASS == 0) {
        if (lane == 0) a.seg_end[seg] = run;
    } else {
        double2 dprev = Sin;
#pragma unroll
        for (int k = 0; k < DC_K; k++) {
            const double2 d = make_double2(x[k].x + pw[k] * Sin.x, x[k].y + pw[k] * Sin.y);
            double2 y = make_double2(d.x - dprev.x, d.y - dprev.y);
            if (a.conj_out) y.y = -y.y;
            if (k < v) a.y[base + k] = y;
            dprev = d;
        }
    }
}
// seg_in[s] = the recurrence state in front of segment s; *state carries over to the next block.  alpha_last = alpha^(length
// of the last, possibly partial, segment).
__global__ __launch_bounds__(64) void k_dc_chain(const double2 *seg_end, double2 *seg_in, int nseg, double alpha_seg, double alpha_last,
                                                 double2 *state)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    double2 run = *state;
    for (int s0 = 0; s0 < nseg; s0 += 64) {
        const int s = s0 + lane;
        const double2 e = s < nseg ? seg_end[s] : make_double2(0.0, 0.0);
        double2 mine = run;
        const int cnt = nseg - s0 < 64 ? nseg - s0 : 64;
        for (int L = 0; L < cnt; L++) {
            if (lane == L) mine = run;
            const double f = (s0 + L == nseg - 1) ? alpha_last : alpha_seg;
            run = make_double2(readlane_f64(e.x, L) + f * run.x, readlane_f64(e.y, L) + f * run.y);
        }
        if (s < nseg) seg_in[s] = mine;
    }
    if (lane == 0) *state = run;
}
// y = (x - avg) [conj];  avg may be null (inversion only)
__global__ __launch_bounds__(256) void k_dc_sub(const double2 *x, double2 *y, int n, const double2 *avg, int conj_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double2 v = x[i];
    if (avg) { const double2 m = *avg; v.x -= m.x; v.y -= m.y; }
    if (conj_out) v.y = -v.y;
    y[i] = v;
}
// dc_sum += sum of the block (sound.c:209-210), two deterministic stages: 256 partial sums, then one wave adds them in order
__global__ __launch_bounds__(256) void k_dc_partial(const double2 *x, int n, double2 *partial)
{
    __shared__ double2 red[256];
    double2 acc = make_double2(0.0, 0.0);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { acc.x += x[i].x; acc.y += x[i].y; }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[threadIdx.x].x += red[threadIdx.x + o].x; red[threadIdx.x].y += red[threadIdx.x + o].y; }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// mode: 0 = add the partials to *sum; 1 = also close the stretch: *avg = *sum / count, *sum = 0 (sound.c:211-217)
__global__ void k_dc_final(const double2 *partial, int np, double2 *sum, double2 *avg, int close_stretch, double count)
{
    if (threadIdx.x || blockIdx.x) return;
    double2 s = *sum;
    for (int i = 0; i < np; i++) { s.x += partial[i].x; s.y += partial[i].y; }
    if (close_stretch) { *avg = make_double2(s.x / count, s.y / count); s = make_double2(0.0, 0.0); }
    *sum = s;
}

// ------------------------------------------------------------------------------------------------
// Noise blanker (NoiseBlanker, quisk.c:615-720) on the full-rate stream, behind DC removal and inversion.
// The reference pushes every sample through a ring of 3 hw samples (hw = 500 us) and edits the ring in place: a
// sample whose magnitude exceeds limit x (running mean of the ring's magnitudes) is a PULSE; when a pulse starts the
// hw samples before it are multiplied by a ramp j / hw (j = 0 at the pulse), samples are zeroed while pulses last,
// and the samples behind the last pulse are multiplied by a ramp 1/hw, 2/hw ... until it reaches one or the next
// pulse starts.  All edits of a sample happen within hw samples of its arrival, and the state of the machine in
// front of sample t is simply pulse[t-1]; so the output, the sample that arrived 3 hw earlier, is
//     y[i] = x[t] * M[t],  t = i - 3 hw,
//     M[t] = 0                                   if pulse[t-1] (zeroed) or pulse[t] (a pulse starts: ramp factor 0),
//          = (t - r - 1) / hw  [r = the nearest pulse before t, if within hw]        (the ramp up, applied first)
//            x  prod over pulse starts q in (t, t + hw) of (q - t) / hw, in time order    (the ramps down)
// and everything is a function of the pulse flags within hw of t: no serial walk.  The flags need the running sum
// of the last 3 hw magnitudes: here the previous sum plus a prefix sum of mag[k] - mag[k - 3 hw] (segments of 1024
// samples, chained like the DC blocker), not the reference's  sum -= old; sum += new  -- another rounding order, so a
// flag could differ from the reference's only where a magnitude meets the limit to within 1e-13; the multiplications
// are the reference's own, in its order.  Arrays carry H >= 4 hw samples of history in front of the block ("ext"
// coordinates: the block starts at H); pulse flags are kept as bytes, as 64-sample words and as an any-flag per 1024
// samples so that the searches skip the quiet stretches, which is nearly all of them.
// ------------------------------------------------------------------------------------------------
#define NB_SEG 1024
struct NbArgs {
    const double2 *x; int n;            // the block
    double2 *ext_x; double *ext_mag; unsigned char *ext_p;      // [H + n]
    int H, hw, save;                    // save = 3 hw
    double limit;
    double *loc;                        // [n] prefix of the magnitude differences inside a segment
    double *segsum, *segbase;           // [nseg]
    double *S;                          // the running sum in front of the block; behind it afterwards
    unsigned long long *words; unsigned char *any;   // [(H + n) / 64 + 2], [(H + n) / 1024 + 2]
    double2 *y;                         // [n]
};
// magnitudes, their differences against the ones leaving the ring, prefix sums inside segments of 1024
__global__ __launch_bounds__(64) void k_nb_mag(NbArgs a)
{
#pragma clang fp contract(off)
    const int seg = blockIdx.x, lane = threadIdx.x;
    const int base = seg * NB_SEG + lane * 16;
    double run = 0.0, pre[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int i = base + k;
        double d = 0.0;
        if (i < a.n) {
            const double2 v = a.x[i];
            const double m = hypot(v.x, v.y);
            a.ext_x[a.H + i] = v;
            a.ext_mag[a.H + i] = m;
            const int o = i - a.save;               // the sample leaving the ring: in this block, or in the history
            double mo;
            if (o >= 0) { const double2 w = a.x[o]; mo = hypot(w.x, w.y); } else mo = a.ext_mag[a.H + o];
            d = m - mo;
        }
        run += d;
        pre[k] = run;
    }
    // exclusive scan of the lane totals (a serial chain of 64 adds: one wave, 1024 samples)
    double off = 0.0, acc = 0.0;
    for (int L = 0; L < 64; L++) {
        if (lane == L) off = acc;
        acc += readlane_f64(run, L);
    }
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (base + k < a.n) a.loc[base + k] = off + pre[k];
    if (lane == 0) a.segsum[seg] = acc;
}
__global__ __launch_bounds__(64) void k_nb_chain(NbArgs a, int nseg)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    double run = *a.S;
    for (int s0 = 0; s0 < nseg; s0 += 64) {
        const int s = s0 + lane;
        const double e = s < nseg ? a.segsum[s] : 0.0;
        double mine = run;
        const int cnt = nseg - s0 < 64 ? nseg - s0 : 64;
        for (int L = 0; L < cnt; L++) {
            if (lane == L) mine = run;
            run += readlane_f64(e, L);
        }
        if (s < nseg) a.segbase[s] = mine;
    }
    if (lane == 0) *a.S = run;
}
// pulse[t] = !(mag <= sum / size * limit)   (quisk.c:674-677)
__global__ __launch_bounds__(256) void k_nb_flags(NbArgs a)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const double sum = a.segbase[i / NB_SEG] + a.loc[i];
    const double m = a.ext_mag[a.H + i];
    a.ext_p[a.H + i] = !(m <= sum / a.save * a.limit);
}
// the flags of all H + n samples as words and per-1024 summaries (the history has moved by a ragged count: made anew)
__global__ __launch_bounds__(256) void k_nb_words(NbArgs a)
{
    const int E = a.H + a.n;
    int got = 0;
    for (int it = 0; it < 4; it++) {
        const int t = blockIdx.x * NB_SEG + it * 256 + threadIdx.x;
        const int f = t < E ? a.ext_p[t] : 0;
        const unsigned long long w = __ballot(f);
        if ((threadIdx.x & 63) == 0) a.words[t >> 6] = w;
        got |= f;
    }
    got = __syncthreads_or(got);
    if (threadIdx.x == 0) a.any[blockIdx.x] = got != 0;
}
__global__ __launch_bounds__(256) void k_nb_apply(NbArgs a)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int t = a.H + i - a.save;                 // >= hw + (rounding of H): every index below stays inside the arrays
    double2 u = a.ext_x[t];
    const double hw = (double)a.hw;
    if (a.ext_p[t - 1]) {
        u = make_double2(0.0, 0.0);                 // state 1: "zero samples until the pulses stop" (quisk.c:696)
    } else {
        if (a.ext_p[t]) {
            u = make_double2(u.x * 0.0, u.y * 0.0); // a pulse starts here: first factor of its ramp, j = 0 (quisk.c:683-684)
        } else {
            // the ramp up behind the nearest pulse r (quisk.c:689-693): win_index = t - (r + 1), 1 .. hw - 1
            int q = t - 1, r = -1;
            const int lo = t - a.hw;
            while (q >= lo) {
                if (!a.any[q >> 10]) { q = ((q >> 10) << 10) - 1; continue; }
                const int wi = q >> 6;
                const unsigned long long w = a.words[wi] & (~0ull >> (63 - (q & 63)));
                if (w) { r = (wi << 6) + 63 - __clzll((long long)w); break; }
                q = (wi << 6) - 1;
            }
            if (r >= lo) { const double f = (double)(t - r - 1) / hw; u = make_double2(u.x * f, u.y * f); }
        }
        // the ramps down in front of every pulse that starts within hw - 1 samples (quisk.c:680-688), in time order
        const int hi = t + a.hw - 1;
        int q = t + 1;
        while (q <= hi) {
            if (!a.any[q >> 10]) { q = ((q >> 10) + 1) << 10; continue; }
            const int wi = q >> 6;
            const unsigned long long w = a.words[wi], prev = a.words[wi - 1];
            unsigned long long sw = w & ~((w << 1) | (prev >> 63));        // a pulse with no pulse right before it
            sw &= ~0ull << (q & 63);
            while (sw) {
                const int pos = (wi << 6) + __ffsll((long long)sw) - 1;
                sw &= sw - 1;
                if (pos > hi) break;
                const double f = (double)(pos - t) / hw;
                u = make_double2(u.x * f, u.y * f);
            }
            q = (wi + 1) << 6;
        }
    }
    a.y[i] = u;
}
// the last H samples of (ext_x, ext_mag, ext_p) become the history of the twin set
__global__ __launch_bounds__(256) void k_nb_carry(NbArgs a, double2 *nx, double *nmag, unsigned char *np)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.H) return;
    nx[j] = a.ext_x[a.n + j]; nmag[j] = a.ext_mag[a.n + j]; np[j] = a.ext_p[a.n + j];
}

// ------------------------------------------------------------------------------------------------
// Stand-alone FIR over interleaved complex samples for the filter.c drop-ins (quisk_cDecimate, filter.c:180-206;
// quisk_cCDecimate, filter.c:208-234).  x holds ntaps-1 history samples followed by the new block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cfir_interleaved(const double2 *x, double2 *y, const double *h, int T, int D,
                                                          int first, int n_out, int cpx)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_out) return;
    const double2 *p = x + (T - 1) + first + (long long)D * j;      // newest sample of output j
    double ar = 0.0, ai = 0.0;
    if (cpx) {
        const double2 *c = reinterpret_cast<const double2 *>(h);
        for (int k = 0; k < T; k++) {
            const double2 v = p[-k], w = c[k];
            ar += v.x * w.x - v.y * w.y;
            ai += v.x * w.y + v.y * w.x;
        }
    } else {
        for (int k = 0; k < T; k++) {
            const double2 v = p[-k];
            ar = fma(v.x, h[k], ar);
            ai = fma(v.y, h[k], ai);
        }
    }
    y[j] = make_double2(ar, ai);
}

// The same, LDS-tiled: a workgroup makes 256 R consecutive outputs from the D (256 R - 1) + T input samples they need, staged
// once in LDS as a real and an imaginary plane (so an input sample is fetched from memory once instead of T / D times); lane
// l makes outputs l, l + 256, ...: neighbouring lanes read LDS words D apart.  The taps are wave-uniform scalar loads shared by
// the lane's R outputs.  Every output's sum runs over k = 0 .. T-1 in the reference's order, with the operations of
// k_cfir_interleaved: the two kernels give identical bits.  Dynamic LDS: 16 (D (256 R - 1) + T) bytes.
template <int R>
__global__ __launch_bounds__(256) void k_cfir_lds(const double2 *__restrict__ x, double2 *__restrict__ y, const double *__restrict__ h, int T, int D,
                                                  int first, int n_out, int cpx)
{
    extern __shared__ __attribute__((aligned(16))) double cf_lds[];
    const int S = D * (256 * R - 1) + T;
    double *re = cf_lds, *im = cf_lds + S;
    const int j0 = blockIdx.x * (256 * R);
    const long long base = first + (long long)D * j0;                       // x index of the oldest sample of the tile
    const long long last = (long long)(T - 1) + first + (long long)D * (n_out - 1);   // newest sample any output needs
    for (int i = threadIdx.x; i < S; i += 256) {
        const long long idx = base + i;
        const double2 v = x[idx < last ? idx : last];
        re[i] = v.x; im[i] = v.y;
    }
    __syncthreads();
    double ar[R], ai[R];
    int nw[R];
#pragma unroll
    for (int r = 0; r < R; r++) { ar[r] = 0.0; ai[r] = 0.0; nw[r] = D * ((int)threadIdx.x + 256 * r) + T - 1; }
    if (cpx) {
        const double2 *c = reinterpret_cast<const double2 *>(h);
        for (int k = 0; k < T; k++) {
            const double2 w = c[k];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const double vx = re[nw[r] - k], vy = im[nw[r] - k];
                ar[r] += vx * w.x - vy * w.y;
                ai[r] += vx * w.y + vy * w.x;
            }
        }
    } else {
        for (int k = 0; k < T; k++) {
            const double hk = h[k];
#pragma unroll
            for (int r = 0; r < R; r++) {
                ar[r] = fma(re[nw[r] - k], hk, ar[r]);
                ai[r] = fma(im[nw[r] - k], hk, ai[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = j0 + (int)threadIdx.x + 256 * r;
        if (j < n_out) y[j] = make_double2(ar[r], ai[r]);
    }
}

// The allocation-free drop-in (quisk_cCplxSampleFilter_noalloc): the filter runs in place, so inputs are staged.  Scratch
// layout: cat[nh + count] = history followed by the block (what k_cfir_interleaved reads), then y[n_out].
__global__ __launch_bounds__(256) void k_na_gather(const double2 *hist, const double2 *samples, double2 *cat, int nh, int count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nh + count) cat[i] = i < nh ? hist[i] : samples[i - nh];
}
__global__ __launch_bounds__(256) void k_na_scatter(const double2 *cat, const double2 *y, double2 *samples, double2 *hist, int nh, int count, int n_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_out) samples[i] = y[i];
    if (i < nh) hist[i] = cat[count + i];           // the last nh samples of (history + block)
}

// cRxFilterOut (quisk.c:1154-1192) over a block of interleaved complex samples, for callers outside the receive chain
// (qk_rx_filter_block): x holds H history samples followed by the n new ones; hI / hQ are the receiver's taps in the plain
// FIR order qk_set_filters stores (h[0] = f[0], h[m] = f[N - m]): I taps on the real parts, Q taps on the imaginary parts.
__global__ __launch_bounds__(256) void k_rxf_block(const double2 *x, double2 *y, const double *hI, const double *hQ, int ntaps, int H, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double2 *p = x + H + i;
    double ar = 0.0, ai = 0.0;
    for (int m = 0; m < ntaps; m++) {
        const double2 v = p[-m];
        ar = fma(v.x, hI[m], ar);
        ai = fma(v.y, hQ[m], ai);
    }
    y[i] = make_double2(ar, ai);
}
// the first `w` samples of each display-spectrum frame, kept for the raw scope (get_graph with use_fft == 0, quisk.c:4642-4649)
__global__ __launch_bounds__(256) void k_copy_c(const double2 *src, double2 *dst, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// Second-receiver routing (quisk.c:2410-2494): the audio of receiver `rx` becomes stereo, one channel its own audio and
// the other the audio of a partner receiver (split RX/TX: the same stream tuned to the transmit frequency; or the
// sub-receiver chosen by set_multirx_play_channel).  In the reference's order: the two DETECTED streams (at the decimated
// rate, in front of cFracDecim and the play-rate interpolators) are count-equalised by the two FIFOs of Buffer2Chan
// (quisk.c:1452-1487) and become the real and imaginary part of one stereo stream (k_route_pair: rows 0 = left, 1 = right of
// the route's own tail chain), which then goes through cFracDecim, the interpolators and -- each channel by itself --
// Agc1 (left) and Agc2 (right) (quisk.c:2528-2572); k_route_combine puts the two channels into the receiver's output row
// with the per-channel squelch of quisk.c:2590-2601.
// method 1: left = rx, right = partner; 2: left = partner, right = rx; 3: both channels = partner; 4: both = rx.
// ------------------------------------------------------------------------------------------------
struct RouteArgs {
    StreamView in_a, in_b;              // the detected audio of receiver rx and of the partner (plane 0), rows rx / partner
    int rx, partner, method;
    const double *fifo_a, *fifo_b;      // samples waiting from earlier blocks (a: receiver rx, b: the partner)
    double *next_a, *next_b;            // what stays behind after this block
    int nbuf_a, nbuf_b, ca, cb, nout;
    StreamView out;                     // the pair: row 0 = left, row 1 = right
};
__global__ __launch_bounds__(256) void k_route_pair(RouteArgs r)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ta = r.nbuf_a + r.ca, tb = r.nbuf_b + r.cb;
    double a = 0.0, b = 0.0;
    if (i < ta) a = i < r.nbuf_a ? r.fifo_a[i] : sv_ptr(r.in_a, r.rx, 0)[i - r.nbuf_a];
    if (i < tb) b = i < r.nbuf_b ? r.fifo_b[i] : sv_ptr(r.in_b, r.partner, 0)[i - r.nbuf_b];
    if (i < r.nout) {
        sv_ptr(r.out, 0, 0)[i] = (r.method == 1 || r.method == 4) ? a : b;
        sv_ptr(r.out, 1, 0)[i] = (r.method == 2 || r.method == 4) ? a : b;
    } else {
        if (i < ta) r.next_a[i - r.nout] = a;
        if (i < tb) r.next_b[i - r.nout] = b;
    }
}
// left / right: the two channels' rows behind their AGCs (mono pairs: .x is the sample); mute_l / mute_r: the squelch flags of
// the receivers the channels came from (quisk.c:2421-2493)
__global__ __launch_bounds__(256) void k_route_combine(const double2 *left, const double2 *right, double2 *out, int n,
                                                       const int *mute_l, const int *mute_r, int kill)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool ml = kill || (mute_l && *mute_l), mr = kill || (mute_r && *mute_r);
    out[i] = make_double2(ml ? 0.0 : left[i].x, mr ? 0.0 : right[i].x);
}
// measure_audio (quisk.c:1787-1790, 2015-2018): the running sum of squares of receiver 0's detected audio.  One workgroup,
// (the reference adds sample by sample: the sums differ in the last bits; the test bound is 1e-9).  acc[0] += sum.
__global__ __launch_bounds__(256) void k_measure_sum(StreamView in, int planes, int n, double *acc)
{
    __shared__ double red[256];
    // (long blocks: a few workgroups, a slice each, joined by an FP64 atomic add -- the single workgroup was 15 us on the
    // chain's critical path at 32768 samples)
    const int per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = min(n, lo + per);
    double s = 0.0;
    for (int pl = 0; pl < planes; pl++) {
        const double *p = sv_ptr(in, 0, pl);
        for (int i = lo + threadIdx.x; i < hi; i += 256) s += p[i] * p[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) { if (gridDim.x == 1) acc[0] += red[0]; else atomicAdd(&acc[0], red[0]); }
}
// ... and its close: measured_audio = sqrt(sum / count) / CLIP32 * 1e6; the sum starts again
__global__ void k_measure_close(double *acc, double count)
{
    acc[1] = sqrt(acc[0] / count) / QK_CLIP32 * 1e6;
    acc[0] = 0.0;
}
// ------------------------------------------------------------------------------------------------
// A class chain in ONE launch, for blocks that leave a receiver a few thousand samples at most behind the decimators: the
// string of a dozen short kernels (FIR / 2, band-pass + detector, one-pole filters, interpolators, fractional decimator ...)
// is launch-latency-bound there -- 5 to 10 us of queue time per kernel for a few hundred samples each -- and sits on the
// block's critical path.  One workgroup per receiver walks the chain's stages; a stage's outputs go to the same streams in
// HBM as in the separate kernels (they stay in L2), a workgroup barrier stands where the kernel boundary was.  Every output
// is computed by the same operations in the same order as in k_fir_tiled / k_rxf_tiled / k_interp / k_hb_interp /
// k_fracdecim / k_fm_disc + k_fm_squelch / k_serial_iir / k_measure_sum, so that a stream does not depend on which path a
// block took.  (The host keeps the per-stage bookkeeping -- decimation phases, counts, the fractional decimator's
// positions -- exactly as for the separate launches: run_chain.)
// ------------------------------------------------------------------------------------------------
enum { FS_FIR = 0, FS_INTERP, FS_HBINT, FS_RXF, FS_FMDISC, FS_IIR, FS_FRAC, FS_MEASURE };
struct FStage {
    StreamView in, out;
    const void *h;             // FS_FIR: permuted taps in the compute type; FS_INTERP: taps (double)
    double *iir_state;
    double a0, a1, b1, fs;
    int kind, planes, n_in, n_out, D, L, T, TQP, first, delay, sub, pad;
};
#define QK_FUSED_MAX 14
#define FZ_LDS 6144          // elements of LDS a fused stage may stage its window and taps in (48 KiB of doubles)
struct FusedArgs {
    int nst, tap_cap, n_rx, sq_pstride;
    const int *rx_list;
    const void *taps; const int *ntaps; const int *mode;            // FS_RXF (see RxfArgs)
    double *sq_state; int *sq_flag; double sq_level;                // FM squelch (see SquelchArgs)
    const int *frac_idx; const double *frac_pos;                    // FS_FRAC: positions in host memory the device can read
    double *ma_acc;                                                 // FS_MEASURE
    FStage st[QK_FUSED_MAX];
};
template <class T>
__global__ __launch_bounds__(256) void k_chain_fused(FusedArgs a)
{
    __shared__ double red[256];
    __shared__ double part[256];
    __shared__ T fz_lds[FZ_LDS];
    const int rx = a.rx_list ? a.rx_list[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x;
    for (int s = 0; s < a.nst; s++) {
        const FStage &st = a.st[s];
        switch (st.kind) {
        case FS_FIR: {          // fir_tile: branch by branch, taps in order, one fma each
            const T *__restrict__ hp = static_cast<const T *>(st.h);
            const int D = st.D, TQP = st.TQP;
            const int W = D * (st.n_out + TQP - 1);             // input window of one plane: indices lo .. lo + W - 1
            const int lo = st.first - (D - 1) - D * (TQP - 1);
            if (st.planes * W + D * TQP <= FZ_LDS) {
                // window and taps through LDS: a chain of T dependent fma per output, its operands a few cycles away
                // instead of a trip to L2 each (the same operations in the same order)
                T *X = fz_lds, *H = fz_lds + st.planes * W;
                for (int pl = 0; pl < st.planes; pl++) {
                    const double *__restrict__ x = sv_ptr(st.in, rx, pl) + lo;
                    for (int i = tid; i < W; i += 256) X[pl * W + i] = (T)x[i];
                }
                for (int i = tid; i < D * TQP; i += 256) H[i] = hp[i];
                __syncthreads();
                for (int o = tid; o < st.planes * st.n_out; o += 256) {
                    const int plane = o / st.n_out, j = o - plane * st.n_out;
                    const T *__restrict__ x = X + plane * W + (st.first + D * j - lo);
                    T acc = T(0);
                    for (int p = 0; p < D; p++) {
                        const T *__restrict__ h = H + p * TQP;
                        const T *__restrict__ xp = x - p;
#pragma unroll 8
                        for (int q = 0; q < TQP; q++) acc = fma(xp[-D * q], h[q], acc);
                    }
                    sv_ptr(st.out, rx, plane)[j] = (double)acc;
                }
                break;
            }
            for (int o = tid; o < st.planes * st.n_out; o += 256) {
                const int plane = o / st.n_out, j = o - plane * st.n_out;
                const double *__restrict__ x = sv_ptr(st.in, rx, plane) + st.first + D * j;
                T acc = T(0);
                for (int p = 0; p < D; p++) {
                    const T *__restrict__ h = hp + p * TQP;
                    const double *__restrict__ xp = x - p;
                    for (int q = 0; q < TQP; q++) acc = fma((T)xp[-D * q], h[q], acc);
                }
                sv_ptr(st.out, rx, plane)[j] = (double)acc;
            }
            break;
        }
        case FS_RXF: {          // k_rxf_tiled
            const int N = a.ntaps[rx];
            const T *__restrict__ fi = static_cast<const T *>(a.taps) + (long long)rx * 2 * a.tap_cap;
            const T *__restrict__ fq = (st.sub == 0) ? fi + a.tap_cap : fi;
            const int TQP = (N + FT_QB - 1) / FT_QB * FT_QB;
            const int m = a.mode[rx];
            const bool lower = (m == 0 /*CWL*/ || m == 2 /*LSB*/ || m == 8 /*DGT_L*/);
            const double *__restrict__ pr = sv_ptr(st.in, rx, 0), *__restrict__ pi = sv_ptr(st.in, rx, 1);
            double *o0 = sv_ptr(st.out, rx, 0);
            const int W = st.n_in + TQP - 1;                    // indices -(TQP - 1) .. n_in - 1 of either plane
            if (N > 0 && 2 * W + 2 * TQP + 2 * st.n_in <= FZ_LDS) {
                // both planes' windows, both tap sets and the two chains' results through LDS; the I chain and the Q chain of an
                // output run on two lanes
                T *X = fz_lds, *H = X + 2 * W, *Rz = H + 2 * TQP;
                for (int i = tid; i < W; i += 256) { X[i] = (T)pr[i - (TQP - 1)]; X[W + i] = (T)pi[i - (TQP - 1)]; }
                for (int i = tid; i < TQP; i += 256) { H[i] = fi[i]; H[TQP + i] = fq[i]; }
                __syncthreads();
                for (int o = tid; o < 2 * st.n_in; o += 256) {
                    const int pl = o / st.n_in, j = o - pl * st.n_in;
                    const T *__restrict__ x = X + pl * W + (TQP - 1) + j;
                    const T *__restrict__ h = H + pl * TQP;
                    T acc = T(0);
#pragma unroll 8
                    for (int q = 0; q < TQP; q++) acc = fma(x[-q], h[q], acc);
                    Rz[pl * st.n_in + j] = acc;
                }
                __syncthreads();
                for (int j = tid; j < st.n_in; j += 256) {
                    const T ar = Rz[j], ai = Rz[st.n_in + j];
                    if (st.sub == 0) o0[j] = (double)(lower ? ar + ai : ar - ai);
                    else if (st.sub == 1) o0[j] = hypot((double)ar, (double)ai);
                    else { o0[j] = (double)ar; sv_ptr(st.out, rx, 1)[j] = (double)ai; }
                }
                break;
            }
            for (int j = tid; j < st.n_in; j += 256) {
                T ar, ai;
                if (N == 0) { ar = (T)pr[j]; ai = (T)pi[j]; }
                else {
                    ar = T(0); ai = T(0);
                    for (int q = 0; q < TQP; q++) ar = fma((T)pr[j - q], fi[q], ar);
                    for (int q = 0; q < TQP; q++) ai = fma((T)pi[j - q], fq[q], ai);
                }
                if (st.sub == 0) o0[j] = (double)(lower ? ar + ai : ar - ai);
                else if (st.sub == 1) o0[j] = hypot((double)ar, (double)ai);
                else { o0[j] = (double)ar; sv_ptr(st.out, rx, 1)[j] = (double)ai; }
            }
            break;
        }
        case FS_INTERP: {       // k_interp
            const double *__restrict__ h = static_cast<const double *>(st.h);
            const int terms = st.T / st.L;
            for (int o = tid; o < st.n_in * st.L; o += 256) {
                const int i = o / st.L, jj = o - i * st.L;
                const double *__restrict__ p = sv_ptr(st.in, rx, 0) + i - st.delay;
                double acc = 0.0;
                for (int k = 0; k < terms; k++) acc = fma(p[-k], h[jj + k * st.L], acc);
                sv_ptr(st.out, rx, 0)[o] = acc * st.L;
            }
            break;
        }
        case FS_HBINT: {        // k_hb_interp
            for (int o = tid; o < st.planes * st.n_in; o += 256) {
                const int plane = o / st.n_in, i = o - plane * st.n_in;
                const double *__restrict__ p = sv_ptr(st.in, rx, plane) + i - st.delay;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 11; k++) acc = fma(p[-k] + p[-21 + k], c_hb45[k], acc);
                double *o2 = sv_ptr(st.out, rx, plane) + 2 * (long long)i;
                o2[0] = p[-11] * HB_C11 * 2;
                o2[1] = acc * 2;
            }
            break;
        }
        case FS_FRAC: {         // k_fracdecim
#pragma clang fp contract(off)
            for (int o = tid; o < st.planes * st.n_out; o += 256) {
                const int plane = o / st.n_out, m = o - plane * st.n_out;
                const double *__restrict__ p = sv_ptr(st.in, rx, plane) + a.frac_idx[m];
                const double d = a.frac_pos[m];
                const double c0 = p[-3], c1 = p[-2], c2 = p[-1], c3 = p[0];
                const double xm0 = d - 0, xm1 = d - 1, xm2 = d - 2, xm3 = d - 3;
                sv_ptr(st.out, rx, plane)[m] = (xm1 * xm2 * xm3 * c0 / -6.0 + xm0 * xm2 * xm3 * c1 / 2.0 +
                                                xm0 * xm1 * xm3 * c2 / -2.0 + xm0 * xm1 * xm2 * c3 / 6.0);
            }
            break;
        }
        case FS_FMDISC: {       // k_fm_disc, then k_fm_squelch: sums of |c| over groups of 256 samples, trees in the same order
            const int np = (st.n_in + 255) / 256;
            for (int g = 0; g < np; g++) {
                const int i = g * 256 + tid;
                double mag = 0.0;
                if (i < st.n_in) {
                    const double *pr = sv_ptr(st.in, rx, 0) + i, *pi = sv_ptr(st.in, rx, 1) + i;
                    const double cr = pr[0], ci = pi[0], qr = pr[-1], qi = pi[-1];
                    const double re = cr * qr + ci * qi;
                    const double im = ci * qr - cr * qi;
                    sv_ptr(st.out, rx, 0)[i] = st.fs * atan2(im, re);
                    mag = hypot(cr, ci);
                }
                red[tid] = mag;
                __syncthreads();
                for (int o = 128; o > 0; o >>= 1) {
                    if (tid < o) red[tid] += red[tid + o];
                    __syncthreads();
                }
                if (tid == 0) part[g] = red[0];
                __syncthreads();
            }
            red[tid] = tid < np ? part[tid] : 0.0;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if (tid < o) red[tid] += red[tid + o];
                __syncthreads();
            }
            if (tid == 0) {
                double *sq = a.sq_state + 3 * rx;
                double sum = sq[0] + red[0], cnt = sq[1] + st.n_in, db = sq[2];
                if (cnt >= 2400.0) {
                    db = sum / cnt / QK_CLIP32;
                    db = db > 1.E-10 ? 20 * log10(db) : -200.0;
                    sum = 0.0; cnt = 0.0;
                }
                sq[0] = sum; sq[1] = cnt; sq[2] = db;
                a.sq_flag[rx] = db < a.sq_level;
            }
            break;
        }
        case FS_IIR: {          // k_serial_iir: one wave
            if (tid < 64) {
                const IirArgs ia{st.in, st.out, st.n_in, st.sub, st.a0, st.a1, st.b1, st.iir_state, nullptr, a.n_rx};
                serial_iir_wave(ia, rx, tid);
            }
            break;
        }
        case FS_MEASURE: {      // k_measure_sum: receiver 0's workgroup only
            if (rx == 0) {
                double sm = 0.0;
                for (int pl = 0; pl < st.planes; pl++) {
                    const double *p = sv_ptr(st.in, 0, pl);
                    for (int i = tid; i < st.n_in; i += 256) sm += p[i] * p[i];
                }
                red[tid] = sm;
                __syncthreads();
                for (int o = 128; o > 0; o >>= 1) {
                    if (tid < o) red[tid] += red[tid + o];
                    __syncthreads();
                }
                if (tid == 0) a.ma_acc[0] += red[0];
            }
            break;
        }
        }
        __syncthreads();
    }
}

// Frames for the audio graph (quisk_calc_audio_graph, quisk.c:903-964) and for measure_freq (quisk.c:4850-4920): `take` samples
// of receiver 0's stream (plane 1 may be absent: a real stream) go to frame[index ...], each divided by `scale`
__global__ __launch_bounds__(256) void k_frame_collect(StreamView in, int planes, int first, int take, double2 *frame, int index, double scale)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= take) return;
    const double re = sv_ptr(in, 0, 0)[first + i] / scale;
    const double im = planes > 1 ? sv_ptr(in, 0, 1)[first + i] / scale : 0.0;
    frame[index + i] = make_double2(re, im);
}
// magnitudes of a transformed frame added in order of frequency (negative frequencies first; quisk.c:953-957, 4889-4893)
__global__ __launch_bounds__(256) void k_frame_accum(const double2 *X, int N, double *avg)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const int h = N / 2, src = k < N - h ? k + h : k - (N - h);
    avg[k] += hypot(X[src].x, X[src].y);
}
// AddTestTone (quisk.c:1194-1229; add_tone, quisk.c:3022-3034): a -40 dBFS complex tone added to the full-rate stream in front
// of the inversion, the noise blanker and the FFT staging.  testtoneVector and audioVector are never-renormalised recurrences
// like the mixer's: their values come from the same power tables of the rounded phase steps (qk_phasor.c).
//   kind 0: x += V;  1 (AM): x += V (1 + Re A);  2 (FM, DGT-FM): x += V cexp(I Re A)       V = v0 p^i, A = a0 q^i
struct ToneArgs { const double2 *x; double2 *y; int n, kind, invert; MixTables vt, at; const double2 *st_in; double2 *st_out; };
__global__ __launch_bounds__(256) void k_add_tone(ToneArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    // row 0 of the tables; the vectors at the block's first sample are st_in[0] (tone) and st_in[1] (audio)
    MixTables vt = a.vt, at = a.at;
    vt.v0 = a.st_in; at.v0 = a.st_in + 1;
    if (i == 0) {
        a.st_out[0] = mix_w(vt, 0, a.n);
        a.st_out[1] = a.kind ? mix_w(at, 0, a.n) : a.st_in[1];
    }
    if (i >= a.n) return;
    double2 v = mix_w(vt, 0, i);
    if (a.kind == 1) {
        const double m = 1.0 + mix_w(at, 0, i).x;
        v = make_double2(v.x * m, v.y * m);
    } else if (a.kind == 2) {
        const double ph = mix_w(at, 0, i).x;
        v = cmul(v, make_double2(cos(ph), sin(ph)));
    }
    double2 s = a.x[i];
    s = make_double2(s.x + v.x, s.y + v.y);
    if (a.invert) s.y = -s.y;                      // the inversion comes behind the tone (quisk.c:2311-2320)
    a.y[i] = s;
}
// Mono again behind a stretch of stereo (quisk.c:2495-2501, 2573-2575), while the pair's two planes still differ: the
// reference's mono sample is complex, d + I d, and runs through the SAME cFracDecim / HalfBand7..9 / Agc1 objects as the pair
// did -- whose imaginary parts hold the right channel's last samples, and whose AGC ring holds real samples only (the stereo
// path stored creal, quisk.c:2564-2568).  Until those have passed, the two output channels differ: the route keeps running
// with both rows fed from the receiver, ONE AGC (Agc1: gains from |Re|, k_agc_gain on row 0), and this apply pass, which
// puts that gain on both rows and limits by |Re| (process_agc with is_cpx == 0 on a complex sample, quisk.c:2056-2066).
struct DrainArgs { StreamView in; const double *gains; int buf_size, n; const int *started; double2 *out; const int *mute; int kill; };
__global__ __launch_bounds__(256) void k_route_drain_apply(DrainArgs a)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const double *p0 = sv_ptr(a.in, 0, 0) + i, *p1 = sv_ptr(a.in, 1, 0) + i;
    double d0, d1;
    if (!a.started[0]) { d0 = p0[0]; d1 = p1[0]; }
    else {
        const double g = a.gains[i];
        d0 = p0[-a.buf_size] * g; d1 = p1[-a.buf_size] * g;
        const double m = fabs(d0);
        if (m > QK_CLIP32) { d0 = d0 / m; d1 = d1 / m; }
    }
    if (a.kill || (a.mute && *a.mute)) { d0 = 0.0; d1 = 0.0; }
    a.out[i] = make_double2(d0, d1);
}
// Key-up envelope (quisk.c:2602-2612): the first samples behind a key-up are scaled by a ramp that the host has made with
// the reference's own additions (env += 1 / (rate * 5 ms), stopping at 1); every receiver's output row gets it.
__global__ __launch_bounds__(256) void k_keyup_ramp(double2 *out, long long out_stride, const double *factor, int nf, const int *counts_host, int n_rx)
{
    const int i = blockIdx.x * 256 + threadIdx.x, rx = blockIdx.y;
    if (i >= nf || i >= counts_host[rx]) return;
    double2 v = out[(long long)rx * out_stride + i];
    const double f = factor[i];
    out[(long long)rx * out_stride + i] = make_double2(v.x * f, v.y * f);
}

// ------------------------------------------------------------------------------------------------
// The remaining filter.c kernels as stand-alone drop-ins (the transmit chain uses them, microphone.c:373-717):
//   quisk_cInterpolate (filter.c:108-142): polyphase x L of complex samples with real taps, out[L i + j] = L sum_k h[j + k L] x[i - k]
//   quisk_dC_out (filter.c:60-81): real samples through COMPLEX taps (quisk_filt_tune), one complex output per input
//   quisk_dD_out (filter.c:303-322): real samples through real taps
// x holds ntaps - 1 history samples followed by the new block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cinterp_interleaved(const double2 *x, double2 *y, const double *h, int T, int L, int n_in)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n_in * L) return;
    const int i = o / L, j = o - i * L;
    const double2 *p = x + (T - 1) + i;
    double ar = 0.0, ai = 0.0;
    const int terms = T / L;
    for (int k = 0; k < terms; k++) {
        const double2 v = p[-k]; const double c = h[j + k * L];
        ar = fma(v.x, c, ar); ai = fma(v.y, c, ai);
    }
    y[o] = make_double2(ar * L, ai * L);
}
__global__ __launch_bounds__(256) void k_dfir_out(const double *x, double *y, const double *h, int T, int n, int cpx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *p = x + (T - 1) + i;
    if (cpx) {
        const double2 *c = reinterpret_cast<const double2 *>(h);
        double ar = 0.0, ai = 0.0;
        for (int k = 0; k < T; k++) { const double v = p[-k]; ar = fma(v, c[k].x, ar); ai = fma(v, c[k].y, ai); }
        reinterpret_cast<double2 *>(y)[i] = make_double2(ar, ai);
    } else {
        double a = 0.0;
        for (int k = 0; k < T; k++) a = fma(p[-k], h[k], a);
        y[i] = a;
    }
}
// Hermes / Metis receive packets (read_rx_udp10, quisk.c:3341-3550): 1032 bytes = 8 header bytes and two 512-byte frames of
// 3 sync bytes, 5 control bytes and num_records records of (1 + multirx_count) x (3 bytes Q, 3 bytes I, big endian, left-
// justified into int32: quisk.c:3516-3518) followed by 2 microphone bytes.  Sample s of the packet stream goes to
// out[s] (the first receiver's) and, for sub-receiver j, to sub[j * sub_stride + s].
__global__ __launch_bounds__(256) void k_unpack_hermes(const unsigned char *pk, int n_packets, int nrx, int num_records, double2 *out,
                                                       double2 *sub, long long sub_stride)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int per_packet = 2 * num_records;
    if (s >= n_packets * per_packet) return;
    const int p = s / per_packet, r = s - p * per_packet;
    const int frame = r / num_records, rec = r - frame * num_records;
    const unsigned char *b = pk + (long long)p * 1032 + 11 + 512 * frame + 5 + rec * (nrx * 6 + 2);
    for (int j = 0; j < nrx; j++, b += 6) {
        const int xi = (int)((unsigned)b[0] << 24 | (unsigned)b[1] << 16 | (unsigned)b[2] << 8);
        const int xr = (int)((unsigned)b[3] << 24 | (unsigned)b[4] << 16 | (unsigned)b[5] << 8);
        const double2 v = make_double2((double)xr, (double)xi);
        if (j == 0) out[s] = v;
        else if (sub) sub[(long long)(j - 1) * sub_stride + s] = v;
    }
}
