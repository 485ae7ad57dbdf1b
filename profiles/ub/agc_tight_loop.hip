#include <hip/hip_runtime.h>
#include <cstdio>
// what does each piece of the runner's tight loop cost a lone wave, on top of the chain?
#define S1 "s_lshl_b64 exec, exec, 1\n\tv_mul_f64 %[g], %[g], %[k]\n\tv_add_f64 %[g], %[g], %[t]\n\t"
#define S4 S1 S1 S1 S1
#define S16 S4 S4 S4 S4
#define S15 S4 S4 S4 S1 S1 S1
#define CHAIN "s_mov_b64 %[sv], exec\n\ts_lshl_b64 exec, exec, %[st]\n\t" S16 "s_cbranch_execz 2f\n\t" S16 "s_cbranch_execz 2f\n\t" S16 "s_cbranch_execz 2f\n\t" S15 "2:\n\ts_mov_b64 exec, %[sv]\n\t"
__device__ __forceinline__ double chain(double g0, double keep, double add, int start)
{
    double gv = g0; unsigned long long save;
    asm volatile(CHAIN : [g] "+v"(gv), [sv] "=&s"(save) : [k] "v"(keep), [t] "v"(add), [st] "s"(start) : "scc");
    return gv;
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int MODE> __global__ void k(double *o, long long *t, double keep, double add, int n, int zero)
{
#pragma clang fp contract(off)
    __shared__ double sp_gv[4][64];
    __shared__ double sp_add[4], sp_add1[4], sp_end[4], sp_max[4];
    __shared__ int sp_clip[4], sp_b[4], sp_tag[4];
    __shared__ int walked, rs_seq;
    __shared__ double sm[2][32 * 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) (&sm[0][0])[i] = 1e-3 * (i & 63);
    if (lane == 0) { walked = 1 << 30; rs_seq = 0; }
    __syncthreads();
    auto peek = [](int *w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto poke = [](int *w, int v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    double g0 = o[lane], rmax = 100.0;
    int gen = zero;
    const double *mp = &sm[0][lane];
    long long w0 = wall_clock64();
    for (int j = 0; j < n; j++) {
        int w_v = 0, rs_v = 0; double mv = 0;
        if (MODE >= 3) { w_v = peek(&walked); rs_v = peek(&rs_seq); }
        if (MODE >= 4) mv = mp[(j & 31) * 64];
        const double gs = chain(g0, keep, add, zero);
        double pend = gs;
        if (MODE >= 1) pend = readlane_f64(gs, 63) * keep + add;
        double nmax = rmax;
        if (MODE >= 4) {
            unsigned long long over = __ballot(mv > nmax);
            while (over) { nmax = readlane_f64(mv, (int)__ffsll((long long)over) - 1); over = __ballot(mv > nmax); }
        }
        if (MODE >= 3) { if ((uni(rs_v) != gen) | (uni(w_v) - j < 4)) break; }
        if (MODE >= 2) {
            const int slot = j & 3;
            sp_gv[slot][lane] = gs;
            if (lane == 0) {
                sp_add[slot] = add; sp_add1[slot] = add; sp_end[slot] = pend; sp_max[slot] = nmax; sp_clip[slot] = 0; sp_b[slot] = 64;
                if (MODE == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("" ::: "memory");
                poke(&sp_tag[slot], (gen << 20) | j);
            }
        }
        g0 = pend; rmax = nmax;
    }
    long long w1 = wall_clock64();
    o[lane] = g0 + rmax;
    if (lane == 0) t[0] = w1 - w0;
}
int main()
{
    double *d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    long long *t; hipMalloc(&t, 16); long long h[2];
    const int n = 4000;
    const char *names[] = {"chain only", "+ pend (readlane, mul, add)", "+ hand-over LDS writes", "+ peeks and exit test", "+ |x| read and max test", "same, waiting before the tag"};
    for (int mode = 0; mode < 6; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (mode) {
            case 0: k<0><<<1, 64>>>(d, t, 0.999, 1e-3, n, 0); break;
            case 1: k<1><<<1, 64>>>(d, t, 0.999, 1e-3, n, 0); break;
            case 2: k<2><<<1, 64>>>(d, t, 0.999, 1e-3, n, 0); break;
            case 3: k<3><<<1, 64>>>(d, t, 0.999, 1e-3, n, 0); break;
            case 4: k<4><<<1, 64>>>(d, t, 0.999, 1e-3, n, 0); break;
            case 5: k<5><<<1, 64>>>(d, t, 0.999, 1e-3, n, 0); break;
            }
            hipDeviceSynchronize();
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            if (rep) printf("%-34s: %.1f ns/tile\n", names[mode], h[0] * 10.0 / n);
        }
    }
    return 0;
}
