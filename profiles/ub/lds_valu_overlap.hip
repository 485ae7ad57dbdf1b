// Do a wave's LDS read blocks and FP64 FMA blocks overlap across the waves of a CU?  The front end's shape: one-wave workgroups,
// 8 per CU (two per SIMD), each alternating between a block of window reads (17 ds_read_b128 or 27 ds_read_b64, then a wait) and
// the FMAs that consume them (138 or 69 FP64 operations).  Modes: reads only, FMAs only, both in turn (the kernel's form), and
// both with the NEXT block's reads issued in front of the current block's FMAs (software-pipelined form).
//   hipcc --offload-arch=gfx950 -O2 -o profiles/tools/lds_valu_overlap.bin profiles/ub/lds_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void reads17(unsigned a, d2 (&v)[17])
{
    asm volatile(
        "ds_read_b128 %0, %17\n\tds_read_b128 %1, %17 offset:16\n\tds_read_b128 %2, %17 offset:32\n\tds_read_b128 %3, %17 offset:48\n\t"
        "ds_read_b128 %4, %17 offset:64\n\tds_read_b128 %5, %17 offset:80\n\tds_read_b128 %6, %17 offset:96\n\tds_read_b128 %7, %17 offset:112\n\t"
        "ds_read_b128 %8, %17 offset:128\n\tds_read_b128 %9, %17 offset:144\n\tds_read_b128 %10, %17 offset:160\n\tds_read_b128 %11, %17 offset:176\n\t"
        "ds_read_b128 %12, %17 offset:192\n\tds_read_b128 %13, %17 offset:208\n\tds_read_b128 %14, %17 offset:4096\n\tds_read_b128 %15, %17 offset:4112\n\t"
        "ds_read_b128 %16, %17 offset:4128"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]),
          "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15]), "=&v"(v[16])
        : "v"(a) : "memory");
}
__device__ __forceinline__ void pin(double (&y)[6])
{   // keeps the FMAs that made y in front of whatever volatile statement follows
    asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]));
}
__device__ __forceinline__ void wait0(d2 (&v)[17])
{
    // the wait, tied to the registers so that the compiler keeps the uses behind it
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                   "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]) :: "memory");
}
__device__ __forceinline__ void fmas(const d2 (&v)[17], double (&y)[6])
{
    double o[28];
#pragma unroll
    for (int j = 0; j < 14; j++) { o[2 * j] = v[j].x; o[2 * j + 1] = v[j].y; }
    const double e[6] = {v[14].x, v[14].y, v[15].x, v[15].y, v[16].x, v[16].y};
#pragma unroll
    for (int r = 0; r < 6; r++) {
        double acc = (o[r + 21] + o[r]) * 0.001;
#pragma unroll
        for (int k = 1; k < 11; k++) acc = fma(o[r + 21 - k] + o[r + k], 0.001 * (k + 1), acc);
        y[r] += fma(e[r], 0.5, acc);
    }
}

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(int iters, double *out, long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double lds[2304];      // 18 KiB: eight workgroups per CU
    for (int i = threadIdx.x; i < 2304; i += 64) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const unsigned a = (unsigned)reinterpret_cast<size_t>(lds) + (threadIdx.x & 31) * 48 + (threadIdx.x >> 5) * 8192;
    double y[6] = {0, 0, 0, 0, 0, 0};
    d2 v[17], w[17];
#pragma unroll
    for (int j = 0; j < 17; j++) { v[j].x = 1.0 + threadIdx.x; v[j].y = 0.5; w[j] = v[j]; }
    const long long t0 = clock64();
    if (MODE == 3) { reads17(a, v); }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { reads17(a, v); wait0(v); }
        if (MODE == 1) { fmas(v, y); pin(y); for (int j = 0; j < 17; j++) asm volatile("" : "+v"(v[j])); }
        if (MODE == 2) { reads17(a, v); wait0(v); fmas(v, y); pin(y); }
        if (MODE == 3) {       // two blocks per trip: the other block's reads are in flight under this block's FMAs
            wait0(v); reads17(a, w); fmas(v, y); pin(y);
            wait0(w); reads17(a, v); fmas(w, y); pin(y);
        }
    }
    if (MODE == 3) wait0(v);
    const long long t1 = clock64();
    double s = 0;
    for (int r = 0; r < 6; r++) s += y[r];
    if (MODE == 0) for (int j = 0; j < 17; j++) s += v[j].x;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- level-1 shape: 27 ds_read_b64 (24 + 3), 69 FP64 operations (3 outputs per lane) ----
__device__ __forceinline__ void reads27(unsigned a, double (&v)[27])
{
    asm volatile(
        "ds_read_b64 %0, %27\n\tds_read_b64 %1, %27 offset:8\n\tds_read_b64 %2, %27 offset:16\n\tds_read_b64 %3, %27 offset:24\n\t"
        "ds_read_b64 %4, %27 offset:32\n\tds_read_b64 %5, %27 offset:40\n\tds_read_b64 %6, %27 offset:48\n\tds_read_b64 %7, %27 offset:56\n\t"
        "ds_read_b64 %8, %27 offset:64\n\tds_read_b64 %9, %27 offset:72\n\tds_read_b64 %10, %27 offset:80\n\tds_read_b64 %11, %27 offset:88\n\t"
        "ds_read_b64 %12, %27 offset:96\n\tds_read_b64 %13, %27 offset:104\n\tds_read_b64 %14, %27 offset:112\n\tds_read_b64 %15, %27 offset:120\n\t"
        "ds_read_b64 %16, %27 offset:128\n\tds_read_b64 %17, %27 offset:136\n\tds_read_b64 %18, %27 offset:144\n\tds_read_b64 %19, %27 offset:152\n\t"
        "ds_read_b64 %20, %27 offset:160\n\tds_read_b64 %21, %27 offset:168\n\tds_read_b64 %22, %27 offset:176\n\tds_read_b64 %23, %27 offset:184\n\t"
        "ds_read_b64 %24, %27 offset:4096\n\tds_read_b64 %25, %27 offset:4104\n\tds_read_b64 %26, %27 offset:4112"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]),
          "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15]), "=&v"(v[16]), "=&v"(v[17]), "=&v"(v[18]),
          "=&v"(v[19]), "=&v"(v[20]), "=&v"(v[21]), "=&v"(v[22]), "=&v"(v[23]), "=&v"(v[24]), "=&v"(v[25]), "=&v"(v[26])
        : "v"(a) : "memory");
}
__device__ __forceinline__ void wait27(double (&v)[27])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                   "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]),
                   "+v"(v[19]), "+v"(v[20]), "+v"(v[21]), "+v"(v[22]), "+v"(v[23]), "+v"(v[24]), "+v"(v[25]), "+v"(v[26]) :: "memory");
}
__device__ __forceinline__ void fmas3(const double (&o)[27], double (&y)[6])
{
#pragma unroll
    for (int r = 0; r < 3; r++) {
        double acc = (o[r + 21] + o[r]) * 0.001;
#pragma unroll
        for (int k = 1; k < 11; k++) acc = fma(o[r + 21 - k] + o[r + k], 0.001 * (k + 1), acc);
        y[r] += fma(o[24 + r], 0.5, acc);
    }
}
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k1(int iters, double *out, long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double lds[2304];
    for (int i = threadIdx.x; i < 2304; i += 64) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const unsigned a = (unsigned)reinterpret_cast<size_t>(lds) + (threadIdx.x & 31) * 24 + (threadIdx.x >> 5) * 8192;
    double y[6] = {0, 0, 0, 0, 0, 0};
    double v[27], w[27];
#pragma unroll
    for (int j = 0; j < 27; j++) { v[j] = 1.0 + threadIdx.x; w[j] = v[j]; }
    if (MODE == 3) { reads27(a, v); }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { reads27(a, v); wait27(v); }
        if (MODE == 1) { fmas3(v, y); pin(y); for (int j = 0; j < 27; j++) asm volatile("" : "+v"(v[j])); }
        if (MODE == 2) { reads27(a, v); wait27(v); fmas3(v, y); pin(y); }
        if (MODE == 3) {
            wait27(v); reads27(a, w); fmas3(v, y); pin(y);
            wait27(w); reads27(a, v); fmas3(w, y); pin(y);
        }
    }
    if (MODE == 3) wait27(v);
    double s = 0;
    for (int r = 0; r < 6; r++) s += y[r];
    if (MODE == 0) for (int j = 0; j < 27; j++) s += v[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main()
{
    long long *cyc; double *out;
    CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&out, 8 * 64 * 256 * 8));
    const int iters = 2000;
    const char *names[8] = {"L0: reads only (17 x b128, wait)      ", "L0: FMAs only (138 FP64 ops)          ", "L0: reads, wait, FMAs (kernel's form) ",
                            "L0: next reads under this block's FMAs", "L1: reads only (27 x b64, wait)       ", "L1: FMAs only (69 FP64 ops)           ",
                            "L1: reads, wait, FMAs (kernel's form) ", "L1: next reads under this block's FMAs"};
    for (int per_cu : {4, 8}) {
        for (int mode = 0; mode < 8; mode++) {
            const dim3 grid(256 * per_cu);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(64), 0, 0, iters, out, cyc);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, grid, dim3(64), 0, 0, iters, out, cyc);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, grid, dim3(64), 0, 0, iters, out, cyc);
                else if (mode == 3) hipLaunchKernelGGL(k<3>, grid, dim3(64), 0, 0, iters / 2, out, cyc);
                else if (mode == 4) hipLaunchKernelGGL(k1<0>, grid, dim3(64), 0, 0, iters, out, cyc);
                else if (mode == 5) hipLaunchKernelGGL(k1<1>, grid, dim3(64), 0, 0, iters, out, cyc);
                else if (mode == 6) hipLaunchKernelGGL(k1<2>, grid, dim3(64), 0, 0, iters, out, cyc);
                else hipLaunchKernelGGL(k1<3>, grid, dim3(64), 0, 0, iters / 2, out, cyc);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%d waves/CU  %s: %7.1f ns per block per wave  (%6.1f ns per block per SIMD-slot)\n", per_cu, names[mode],
                   ms * 1e6 / iters, ms * 1e6 / iters / (per_cu / 4.0));
        }
    }
    return 0;
}
