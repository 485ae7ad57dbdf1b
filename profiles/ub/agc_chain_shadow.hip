// What fits in the shadow of the AGC gain chain?  A lone wave runs s_lshl_b64 exec / v_mul_f64 / v_add_f64 x 63 (the runner's
// recurrence); variants add independent instructions between the dependent ones.  One wave per workgroup, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O2 -o profiles/tools/agc_chain_shadow.bin profiles/ub/agc_chain_shadow.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define STEP(EXTRA1, EXTRA2) "s_lshl_b64 exec, exec, 1\n\t" "v_mul_f64 %[g], %[g], %[k]\n\t" EXTRA1 "v_add_f64 %[g], %[g], %[t]\n\t" EXTRA2
#define S4(A, B) STEP(A, B) STEP(A, B) STEP(A, B) STEP(A, B)
#define S16(A, B) S4(A, B) S4(A, B) S4(A, B) S4(A, B)
#define S63(A, B) S16(A, B) S16(A, B) S16(A, B) S4(A, B) S4(A, B) S4(A, B) STEP(A, B) STEP(A, B) STEP(A, B)
#define SALU "s_add_i32 %[s], %[s], 1\n\t"
#define VALU "v_add_u32 %[v], 1, %[v]\n\t"
#define TOGV "s_mov_b64 %[m], exec\n\t" "s_mov_b64 exec, -1\n\t" "v_add_u32 %[v], 1, %[v]\n\t" "s_mov_b64 exec, %[m]\n\t"
#define LDSW "ds_write_b32 %[a], %[v]\n\t"
template <int MODE>
__global__ __launch_bounds__(64) void k(int iters, double *out, long long *cyc)
{
    __shared__ int lds[256];
    double g = 1.0 + threadIdx.x * 1e-9;
    const double keep = 0.999, add = 0.001;
    int s = 0, v = threadIdx.x;
    unsigned long long m = 0;
    const unsigned a = (unsigned)reinterpret_cast<size_t>(&lds[threadIdx.x]);
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        unsigned long long sv;
#define RUN(A, B) asm volatile("s_mov_b64 %[sv], exec\n\t" S63(A, B) "s_mov_b64 exec, %[sv]\n\t" \
                     : [g] "+v"(g), [sv] "=&s"(sv), [s] "+s"(s), [v] "+v"(v), [m] "+s"(m) : [k] "v"(keep), [t] "v"(add), [a] "v"(a) : "scc", "memory")
        if (MODE == 0) RUN("", "");
        if (MODE == 1) RUN(SALU, "");
        if (MODE == 2) RUN(SALU, SALU);
        if (MODE == 3) RUN(SALU SALU, SALU SALU);
        if (MODE == 4) RUN(VALU, "");
        if (MODE == 5) RUN(VALU, VALU);
        if (MODE == 6) RUN(TOGV, "");
        if (MODE == 7) RUN(LDSW, "");
        if (MODE == 8) RUN("", SALU SALU SALU);
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = g + s + v;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    long long *cyc; double *out;
    CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&out, 256 * 64 * 8));
    const int iters = 2000;
    const char *names[9] = {"chain only", "+1 SALU between mul and add", "+1 SALU in each gap", "+2 SALU in each gap", "+1 VALU (int) between mul and add",
                            "+1 VALU in each gap", "+ exec toggle, VALU at full exec, restore (4 instr) between mul and add", "+1 ds_write_b32 between mul and add",
                            "+3 SALU behind the add"};
    for (int mode = 0; mode < 9; mode++) {
        for (int rep = 0; rep < 2; rep++) {
#define L(M) if (mode == M) hipLaunchKernelGGL(k<M>, dim3(256), dim3(64), 0, 0, iters, out, cyc)
            L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8);
            CK(hipDeviceSynchronize());
        }
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        printf("%-75s: %6.2f clk per step, %7.1f clk per 63-step tile\n", names[mode], (double)c / iters / 63, (double)c / iters);
    }
    return 0;
}
