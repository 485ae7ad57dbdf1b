// Does a CU-masked stream confine its kernels, and how are the mask bits numbered?  Each workgroup records the XCC and
// CU it ran on (hardware registers) and spins briefly.
//   hipcc --offload-arch=gfx950 -O2 -o profiles/tools/cumask.bin profiles/tools/cumask.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_where(unsigned *out, long long ticks)
{
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);        // HW_ID, all 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);       // XCC_ID[3:0]
        out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
    }
    while (wall_clock64() - t0 < ticks) { }
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int G = 4096;
    unsigned *d; CK(hipMalloc(&d, G * 4));
    std::vector<unsigned> h(G);
    auto run = [&](hipStream_t s, const char *name) {
        CK(hipMemset(d, 0xff, G * 4));
        CK(hipDeviceSynchronize());
        const double t0 = now_ms();
        hipLaunchKernelGGL(k_where, dim3(G), dim3(256), 0, s, d, 2000LL);      // 20 us each
        CK(hipStreamSynchronize(s));
        const double t1 = now_ms();
        CK(hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, std::set<unsigned>> cus;       // xcc -> set of (se, sh, cu)
        for (unsigned v : h) { const unsigned xcc = v >> 16, hw = v & 0xffff; cus[xcc].insert((hw >> 8) & 0xff); }
        size_t total = 0;
        printf("%-34s %.3f ms;", name, t1 - t0);
        for (auto &kv : cus) { total += kv.second.size(); printf(" xcc%u:%zu", kv.first, kv.second.size()); }
        printf("  = %zu CUs\n", total);
    };
    hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    run(s0, "plain stream");
    struct M { const char *name; std::vector<uint32_t> m; };
    std::vector<M> masks;
    { std::vector<uint32_t> m(8, 0); m[0] = 0xffffffffu; m[1] = 0xffffffffu; masks.push_back({"bits 0..63", m}); }
    { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 8; i++) m[i] = 0xffu; masks.push_back({"bits 32i..32i+7, i<8", m}); }
    { std::vector<uint32_t> m(8, 0xffffffffu); for (int i = 0; i < 8; i++) m[i] = 0xffffff00u; masks.push_back({"all but bits 32i..32i+7", m}); }
    { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; i += 4) m[i / 32] |= 1u << (i % 32); masks.push_back({"every fourth bit", m}); }
    for (auto &mk : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mk.m.size(), mk.m.data());
        if (e != hipSuccess) { printf("%s: create failed: %s\n", mk.name, hipGetErrorString(e)); continue; }
        run(s, mk.name);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
