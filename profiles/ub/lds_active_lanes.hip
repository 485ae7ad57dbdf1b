#include <hip/hip_runtime.h>
#include <cstdio>
// LDS read cost against the number of active lanes: does a ds_read_b128 / b64 issued by 32 (or 16) lanes cost the LDS
// pipe half (a quarter) of a full wave's?  One wave, 16 independent reads in flight per iteration, lane stride as in the
// front end (48 B for b128, 24 B for b64).
template <int WIDE> __global__ void k(double *o, long long *t, unsigned long long mask, int n)
{
    __shared__ __attribute__((aligned(16))) double sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = i;
    __syncthreads();
    const int lane = threadIdx.x;
    double acc[16] = {0};           // independent accumulators: the LDS pipe, not an add chain, is what is timed
    long long w0 = 0, w1 = 0;
    if ((mask >> lane) & 1) {
        w0 = wall_clock64();
        for (int it = 0; it < n; it++) {
            if (WIDE) {
                const double2 *p = reinterpret_cast<const double2 *>(sm) + lane * 3 + (it & 7);
#pragma unroll
                for (int j = 0; j < 16; j++) { const double2 v = p[j]; acc[j] += v.x; acc[(j + 8) & 15] += v.y; }
            } else {
                const double *p = sm + lane * 3 + (it & 7);
#pragma unroll
                for (int j = 0; j < 16; j++) acc[j] += p[j];
            }
        }
        w1 = wall_clock64();
    }
    double tot = 0.0;
    for (int j = 0; j < 16; j++) tot += acc[j];
    o[lane] = tot;
    if (lane == 0) t[0] = w1 - w0;
}
int main()
{
    double *d; hipMalloc(&d, 64 * 8);
    long long *t; hipMalloc(&t, 16); long long h[2];
    const int n = 20000;
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0x5555555555555555ull, 0xffffff00ffffffull};
    const char *names[] = {"64 lanes", "lanes 0-31", "lanes 0-15", "every other lane", "24 + 24 lanes (levels 2, 3)"};
    for (int wide = 0; wide < 2; wide++)
        for (int m = 0; m < 5; m++) {
            for (int rep = 0; rep < 2; rep++) {
                if (wide) k<1><<<1, 64>>>(d, t, masks[m], n); else k<0><<<1, 64>>>(d, t, masks[m], n);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("%s %-28s: %.2f ns per read instruction\n", wide ? "b128" : "b64 ", names[m], h[0] * 10.0 / (n * 16.0));
        }
    return 0;
}
