#include <hip/hip_runtime.h>
#include <cstdio>
// dependent-op latency of FP64 VALU ops, one wave
template <int MODE> __global__ void k(double *o, double a, double b, int n)
{
    double g = o[threadIdx.x];
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 64; u++) {
            if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(g) : "v"(b));
            if (MODE == 1) asm volatile("v_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %2" : "+v"(g) : "v"(a), "v"(b));
            if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(g) : "v"(a), "v"(b));
            if (MODE == 3) asm volatile("s_lshl_b64 exec, exec, 1\n\tv_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %2\n\ts_mov_b64 exec, -1" : "+v"(g) : "v"(a), "v"(b) : "scc");
            if (MODE == 6) { double t; asm volatile("v_mul_f64 %1, %0, %2\n\ts_mov_b64 exec, -1\n\tv_add_f64 %0, %1, %3" : "+v"(g), "=&v"(t) : "v"(a), "v"(b)); }
            if (MODE == 7) { double t; asm volatile("v_mul_f64 %1, %0, %2\n\ts_lshl_b64 exec, exec, 0\n\tv_add_f64 %0, %1, %3" : "+v"(g), "=&v"(t) : "v"(a), "v"(b) : "scc"); }
            if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(*(float*)&g) : "v"((float)b));
            if (MODE == 5) { int t = u; asm volatile("v_mul_f64 %0, %0, %2\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %1, %1\n\tv_add_f64 %0, %0, %3" : "+v"(g), "+v"(t) : "v"(a), "v"(b)); }
        }
    }
    o[threadIdx.x] = g;
}
int main()
{
    double *d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    for (int mode = 0; mode < 8; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            switch (mode) {
            case 0: k<0><<<1, 64>>>(d, 0.999, 1e-3, n); break;
            case 1: k<1><<<1, 64>>>(d, 0.999, 1e-3, n); break;
            case 2: k<2><<<1, 64>>>(d, 0.999, 1e-3, n); break;
            case 3: k<3><<<1, 64>>>(d, 0.999, 1e-3, n); break;
            case 4: k<4><<<1, 64>>>(d, 0.999, 1e-3, n); break;
            case 5: k<5><<<1, 64>>>(d, 0.999, 1e-3, n); break;
            case 6: k<6><<<1, 64>>>(d, 0.999, 1e-3, n); break;
            case 7: k<7><<<1, 64>>>(d, 0.999, 1e-3, n); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.2f ns per iteration (%.1f clk at 2.4 GHz)\n", mode, ms * 1e6 / (n * 64.0), ms * 1e6 / (n * 64.0) * 2.4);
        }
    }
    return 0;
}
