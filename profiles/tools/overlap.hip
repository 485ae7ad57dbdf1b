// Scheduling experiment: does a long kernel on one stream run beside a chain of short kernels on another, and what
// do the event dependencies of the engine's block pipeline do to that?  Kernels only spin on the wall clock.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/overlap profiles/tools/overlap.hip && gpurun_out/overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spin(long long ticks, int *sink)
{
    extern __shared__ int lds[];
    const long long t0 = wall_clock64();
    int v = 0;
    while (wall_clock64() - t0 < ticks) v++;
    if (v < 0) { lds[threadIdx.x] = v; sink[0] = lds[0]; }
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int blocks = 10, chainlen = 30;
    int *sink; CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const long long us = 100;         // wall_clock64 ticks per microsecond (100 MHz)
    auto A = [&](int wgs, int spin_us) { hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 14 * 1024 + 256, sa, spin_us * us, sink); };
    auto B = [&](hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(512), dim3(128), 8 * 1024, s, 10 * us, sink); };
    hipEvent_t evA[2], evB[2];
    for (int i = 0; i < 2; i++) { CK(hipEventCreateWithFlags(&evA[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evB[i], hipEventDisableTiming)); }
    auto Bbig = [&](hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(8192), dim3(128), 16 * 1024 + 512, s, 10 * us, sink); };
    auto Asmall = [&]() { hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, sa, 4 * us, sink); };
    struct Case { const char *name; int a_wgs, a_us; bool runA, runB, deps, same; int big = 0, atail = 0; };
    const Case cases[] = {
        {"A alone, 2 rounds of 2304", 4608, 200, true, false, false, false},
        {"B chain alone", 0, 0, false, true, false, false},
        {"A(2 rounds) | B independent streams", 4608, 200, true, true, false, false},
        {"A(2 rounds) | B pipeline deps", 4608, 200, true, true, true, false},
        {"A(1 round of 1800, resident) | B independent", 1800, 400, true, true, false, false},
        {"A(1 round of 1800, resident) | B pipeline deps", 1800, 400, true, true, true, false},
        {"A(2 rounds) ; B same stream", 4608, 200, true, true, false, true},
        {"A(resident 1800) | Bbig + B, deps", 1800, 400, true, true, true, false, 1, 0},
        {"A(resident 1800)+2 small | B, deps", 1800, 400, true, true, true, false, 0, 2},
        {"A(resident 1800)+2 small | Bbig + B, deps", 1800, 400, true, true, true, false, 1, 2},
        {"A(resident 2500 of 2816) | Bbig + B, deps", 2500, 400, true, true, true, false, 1, 2},
        {"A(2 rounds 5632)+2 small | Bbig + B, deps", 5632, 200, true, true, true, false, 1, 2},
    };
    for (const Case &c : cases) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipDeviceSynchronize());
            const double t0 = now_ms();
            for (int b = 0; b < blocks; b++) {
                hipStream_t schain = c.same ? sa : sb;
                if (c.deps && b >= 2) CK(hipStreamWaitEvent(sa, evB[b & 1], 0));        // A(b) writes the buffer chain(b-2) read
                if (c.runA) A(c.a_wgs, c.a_us);
                for (int k = 0; k < c.atail; k++) Asmall();
                if (c.deps) { CK(hipEventRecord(evA[b & 1], sa)); CK(hipStreamWaitEvent(schain, evA[b & 1], 0)); }
                if (c.big) Bbig(schain);
                if (c.runB) for (int k = 0; k < chainlen; k++) B(schain);
                if (c.deps) CK(hipEventRecord(evB[b & 1], schain));
            }
            const double t1 = now_ms();
            CK(hipDeviceSynchronize());
            const double t2 = now_ms();
            if (rep) printf("%-52s host enqueue %6.2f ms, total %6.2f ms (%.3f ms per block)\n", c.name, t1 - t0, t2 - t0, (t2 - t0) / blocks);
        }
    }
    // ---- the engine's four-stream pipeline, with spinning kernels ----
    {
        hipStream_t st, agc, apply, fe;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&agc, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&apply, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&fe, hipStreamNonBlocking));
        hipEvent_t ev_fe[2], ev_read[2], ev_chain, ev_agc[2];
        for (int i = 0; i < 2; i++) { CK(hipEventCreateWithFlags(&ev_fe[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_read[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_agc[i], hipEventDisableTiming)); }
        CK(hipEventCreateWithFlags(&ev_chain, hipEventDisableTiming));
        for (int variant = 0; variant < 4; variant++) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipDeviceSynchronize());
                const double t0 = now_ms();
                for (int b = 0; b < blocks; b++) {
                    const int q = b & 1;
                    if (b >= 2 && !(variant & 1)) CK(hipStreamWaitEvent(fe, ev_read[q], 0));
                    hipLaunchKernelGGL(k_spin, dim3(1800), dim3(64), 14 * 1024 + 256, fe, 400 * us, sink);
                    hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, fe, 4 * us, sink);
                    hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, fe, 4 * us, sink);
                    CK(hipEventRecord(ev_fe[q], fe)); CK(hipStreamWaitEvent(st, ev_fe[q], 0));
                    for (int k = 0; k < chainlen; k++) B(st);
                    CK(hipEventRecord(ev_chain, st));
                    if (!(variant & 2)) CK(hipStreamWaitEvent(agc, ev_chain, 0));
                    CK(hipEventRecord(ev_read[q], st));
                    if (!(variant & 2)) {
                        hipLaunchKernelGGL(k_spin, dim3(128), dim3(192), 34 * 1024, agc, 20 * us, sink);
                        hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, agc, 4 * us, sink);
                        CK(hipEventRecord(ev_agc[q], agc));
                    }
                    B(st);
                    if (b >= 1 && !(variant & 2)) CK(hipStreamWaitEvent(st, ev_agc[q ^ 1], 0));
                    for (int k = 0; k < 3; k++) B(st);
                }
                const double t1 = now_ms();
                CK(hipDeviceSynchronize());
                const double t2 = now_ms();
                if (rep) printf("engine pipeline, variant %d (1: no ev_read wait, 2: no agc stream)      host %6.2f ms, total %6.2f ms (%.3f ms per block)\n", variant, t1 - t0, t2 - t0, (t2 - t0) / blocks);
            }
        }
    }
    // ---- bisect the pipeline: bit 0 = two small kernels behind the long one, bit 1 = two event records behind the chain,
    //      bit 2 = four more chain kernels behind those, bit 3 = long kernel's stream waits for the chain two blocks back
    {
        hipStream_t st, fe;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&fe, hipStreamNonBlocking));
        hipEvent_t ev_fe[2], ev_read[2], ev_chain;
        for (int i = 0; i < 2; i++) { CK(hipEventCreateWithFlags(&ev_fe[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_read[i], hipEventDisableTiming)); }
        CK(hipEventCreateWithFlags(&ev_chain, hipEventDisableTiming));
        for (int feat = 0; feat < 16; feat++) {
            double best = 1e9;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipDeviceSynchronize());
                const double t0 = now_ms();
                for (int b = 0; b < blocks; b++) {
                    const int q = b & 1;
                    if ((feat & 8) && b >= 2) CK(hipStreamWaitEvent(fe, ev_read[q], 0));
                    hipLaunchKernelGGL(k_spin, dim3(1800), dim3(64), 14 * 1024 + 256, fe, 400 * us, sink);
                    if (feat & 1) {
                        hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, fe, 4 * us, sink);
                        hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, fe, 4 * us, sink);
                    }
                    CK(hipEventRecord(ev_fe[q], fe)); CK(hipStreamWaitEvent(st, ev_fe[q], 0));
                    for (int k = 0; k < chainlen; k++) B(st);
                    if (feat & 2) { CK(hipEventRecord(ev_chain, st)); CK(hipEventRecord(ev_read[q], st)); }
                    else if (feat & 8) CK(hipEventRecord(ev_read[q], st));
                    if (feat & 4) for (int k = 0; k < 4; k++) B(st);
                }
                CK(hipDeviceSynchronize());
                best = std::min(best, (now_ms() - t0) / blocks);
            }
            printf("features %2d: %.3f ms per block\n", feat, best);
        }
    }
    // ---- which pairs of streams overlap? (variant 3 of the pipeline on every ordered pair of six streams) ----
    {
        hipStream_t S[6];
        for (auto &x : S) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        hipEvent_t ev_fe[2];
        for (int i = 0; i < 2; i++) CK(hipEventCreateWithFlags(&ev_fe[i], hipEventDisableTiming));
        for (int fi = 0; fi < 6; fi++) {
            for (int si = 0; si < 6; si++) {
                if (fi == si) { printf("   -   "); continue; }
                double best = 1e9;
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipDeviceSynchronize());
                    const double t0 = now_ms();
                    for (int b = 0; b < blocks; b++) {
                        const int q = b & 1;
                        hipLaunchKernelGGL(k_spin, dim3(1800), dim3(64), 14 * 1024 + 256, S[fi], 400 * us, sink);
                        CK(hipEventRecord(ev_fe[q], S[fi])); CK(hipStreamWaitEvent(S[si], ev_fe[q], 0));
                        for (int k = 0; k < chainlen; k++) B(S[si]);
                    }
                    CK(hipDeviceSynchronize());
                    best = std::min(best, (now_ms() - t0) / blocks);
                }
                printf(" %5.3f ", best);
            }
            printf("   <- long kernel on stream %d, chain on stream (column)\n", fi);
        }
    }
    return 0;
}
