#include <hip/hip_runtime.h>
#include <cstdio>
// does activity of a second wave (another SIMD of the same CU) slow a dependent FP64 chain in the first?
#define S1 "s_lshl_b64 exec, exec, 1\n\tv_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %2\n\t"
#define S8 S1 S1 S1 S1 S1 S1 S1 S1
#define S63 S8 S8 S8 S8 S8 S8 S8 S1 S1 S1 S1 S1 S1 S1
template <int MODE> __global__ void k(double *o, long long *t, double a, double b, int n)
{
    __shared__ volatile int done;
    __shared__ volatile int poll[64];
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) {
        double g = o[threadIdx.x];
        long long c0 = clock64(), w0 = wall_clock64();
        for (int i = 0; i < n; i++) {
            unsigned long long sv;
            asm volatile("s_mov_b64 %3, exec\n\t" S63 "s_mov_b64 exec, %3" : "+v"(g), "+v"(a), "+v"(b), "=&s"(sv) : : "scc");
        }
        long long c1 = clock64(), w1 = wall_clock64();
        o[threadIdx.x] = g;
        if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; done = 1; }
    } else {
        double x = a; float f = (float)b; int acc = 0;
        if (MODE == 0) return;
        while (!__builtin_amdgcn_readfirstlane(done)) {
            if (MODE == 1) { acc += poll[threadIdx.x & 63]; __builtin_amdgcn_s_sleep(1); }
            if (MODE == 2) { for (int u = 0; u < 64; u++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f)); }
            if (MODE == 3) { for (int u = 0; u < 64; u++) asm volatile("v_add_f64 %0, %0, %0" : "+v"(x)); }
            if (MODE == 4) { for (int u = 0; u < 64; u++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(acc) : : "scc"); }
            if (MODE == 5) { acc += poll[threadIdx.x & 63]; }
            if (MODE == 6) { __builtin_amdgcn_s_sleep(8); }
            if (MODE == 7) { for (int u = 0; u < 16; u++) { asm volatile("s_cbranch_scc0 1f\n1:" ::: "scc"); } }
        }
        o[threadIdx.x] = x + f + acc;
    }
}
int main()
{
    double *d; hipMalloc(&d, 256 * 8); hipMemset(d, 0, 256 * 8);
    long long *t; hipMalloc(&t, 16); long long h[2];
    const int n = 2000;
    const char *names[] = {"idle (exited)", "LDS poll + s_sleep 1", "VALU f32 loop", "VALU f64 loop", "SALU loop", "LDS poll, no sleep", "s_sleep 8 loop", "branch loop"};
    for (int mode = 0; mode < 8; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (mode) {
            case 0: k<0><<<1, 128>>>(d, t, 0.999, 1e-3, n); break;
            case 1: k<1><<<1, 128>>>(d, t, 0.999, 1e-3, n); break;
            case 2: k<2><<<1, 128>>>(d, t, 0.999, 1e-3, n); break;
            case 3: k<3><<<1, 128>>>(d, t, 0.999, 1e-3, n); break;
            case 4: k<4><<<1, 128>>>(d, t, 0.999, 1e-3, n); break;
            case 5: k<5><<<1, 128>>>(d, t, 0.999, 1e-3, n); break;
            case 6: k<6><<<1, 128>>>(d, t, 0.999, 1e-3, n); break;
            case 7: k<7><<<1, 128>>>(d, t, 0.999, 1e-3, n); break;
            }
            hipDeviceSynchronize();
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            if (rep) printf("other wave %-22s: chain %.2f clk/step, %.1f ns/tile, clock %.2f GHz\n", names[mode], h[0] / (n * 63.0), h[1] * 10.0 / n, h[0] / (h[1] * 10.0));
        }
    }
    return 0;
}
