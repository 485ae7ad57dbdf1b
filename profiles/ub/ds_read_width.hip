// LDS read rate by instruction: 16 bytes per lane as one ds_read_b128, one ds_read2_b64 or two ds_read_b64, at lane strides
// of 16, 24 and 48 bytes (one-wave workgroups, 8 and 16 per CU).  Inline assembly so that the compiler neither pairs nor
// drops the reads.  Prints LDS clocks per 16 bytes per lane per CU.
//   hipcc --offload-arch=gfx950 -O2 -o profiles/tools/ds_read_width.bin profiles/ub/ds_read_width.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define R8(OP) OP OP OP OP OP OP OP OP
template <int KIND>
__global__ __launch_bounds__(64) void k_rd(int stride_bytes, int iters, long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = i;
    __syncthreads();
    const unsigned a = (unsigned)reinterpret_cast<size_t>(lds) + (threadIdx.x & 31) * stride_bytes + (threadIdx.x >> 5) * 8192;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0)      asm volatile(R8("ds_read_b128 v[10:13], %0\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(a) : "v10", "v11", "v12", "v13", "memory");
        else if (KIND == 1) asm volatile(R8("ds_read2_b64 v[10:13], %0 offset1:1\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(a) : "v10", "v11", "v12", "v13", "memory");
        else                asm volatile(R8("ds_read_b64 v[10:11], %0\n\tds_read_b64 v[12:13], %0 offset:8\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(a) : "v10", "v11", "v12", "v13", "memory");
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    long long *cyc; CK(hipMalloc(&cyc, 8));
    const int iters = 4000;
    const char *names[3] = {"ds_read_b128      ", "ds_read2_b64      ", "2 x ds_read_b64   "};
    for (int per_cu : {8, 16}) {
        for (int stride : {16, 24, 48}) {
            for (int kind = 0; kind < 3; kind++) {
                if (kind == 0 && stride % 16) continue;
                const dim3 grid(256 * per_cu);
                if (kind == 0) hipLaunchKernelGGL(k_rd<0>, grid, dim3(64), 0, 0, stride, iters, cyc);
                else if (kind == 1) hipLaunchKernelGGL(k_rd<1>, grid, dim3(64), 0, 0, stride, iters, cyc);
                else hipLaunchKernelGGL(k_rd<2>, grid, dim3(64), 0, 0, stride, iters, cyc);
                CK(hipDeviceSynchronize());
                long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
                printf("%2d waves/CU, lane stride %2d B, %s: %5.2f clk per 16 B per lane (wave-level), %5.2f per CU\n", per_cu, stride, names[kind],
                       (double)c / (iters * 8), (double)c / (iters * 8) / per_cu);
            }
        }
    }
    return 0;
}
