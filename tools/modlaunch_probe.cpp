// Does a kernel loaded with hipModuleLoadData (hiprtc) cost more per launch than the same kernel registered statically?
// Two kernels (without / with private scratch), each built both ways; event-timed and wall-timed launches.
//   hipcc --offload-arch=gfx950 -O3 tools/modlaunch_probe.cpp -lhiprtc -o /tmp/modlaunch_probe
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>
#define SRC(...) #__VA_ARGS__
#define KERNELS(PFX)                                                                                                  \
    extern "C" __global__ void __launch_bounds__(128) PFX##plain(unsigned *o, unsigned n) {                            \
        unsigned i = blockIdx.x * blockDim.x + threadIdx.x, x = i;                                                      \
        for (unsigned j = 0; j < n; ++j) x = x * 1664525u + 1013904223u;                                                \
        o[i] = x;                                                                                                       \
    }                                                                                                                   \
    extern "C" __global__ void __launch_bounds__(128) PFX##big(unsigned *o, unsigned n) {                              \
        unsigned i = blockIdx.x * blockDim.x + threadIdx.x, x = i, y = n;                                               \
        for (unsigned j = 0; j < n; ++j) {                                                                              \
            _Pragma("unroll 8192") for (unsigned k = 0; k < 8192; ++k) { x = x * 1664525u + y; y = (y ^ x) + k; }       \
        }                                                                                                               \
        o[i] = x + y;                                                                                                   \
    }                                                                                                                   \
    extern "C" __global__ void __launch_bounds__(128) PFX##scratch(unsigned *o, unsigned n) {                          \
        unsigned i = blockIdx.x * blockDim.x + threadIdx.x, x = i;                                                      \
        unsigned buf[64];                                                                                               \
        for (unsigned j = 0; j < 64; ++j) buf[j] = x + j;                                                               \
        for (unsigned j = 0; j < n; ++j) { x = x * 1664525u + 1013904223u; buf[x & 63] += x; x ^= buf[(x >> 8) & 63]; } \
        o[i] = x;                                                                                                       \
    }
KERNELS(st_)
static const char *kSrc = "#define KERNELS(PFX) " SRC(
    extern "C" __global__ void __launch_bounds__(128) PFX##plain(unsigned *o, unsigned n) {
        unsigned i = blockIdx.x * blockDim.x + threadIdx.x, x = i;
        for (unsigned j = 0; j < n; ++j) x = x * 1664525u + 1013904223u;
        o[i] = x;
    }
    extern "C" __global__ void __launch_bounds__(128) PFX##big(unsigned *o, unsigned n) {
        unsigned i = blockIdx.x * blockDim.x + threadIdx.x, x = i, y = n;
        for (unsigned j = 0; j < n; ++j) {
            _Pragma("unroll 8192") for (unsigned k = 0; k < 8192; ++k) { x = x * 1664525u + y; y = (y ^ x) + k; }
        }
        o[i] = x + y;
    }
    extern "C" __global__ void __launch_bounds__(128) PFX##scratch(unsigned *o, unsigned n) {
        unsigned i = blockIdx.x * blockDim.x + threadIdx.x, x = i;
        unsigned buf[64];
        for (unsigned j = 0; j < 64; ++j) buf[j] = x + j;
        for (unsigned j = 0; j < n; ++j) { x = x * 1664525u + 1013904223u; buf[x & 63] += x; x ^= buf[(x >> 8) & 63]; }
        o[i] = x;
    }) "\nKERNELS(rt_)\n";
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const unsigned blocks = 1024, threads = 128;
    unsigned *d; CK(hipMalloc(&d, blocks * threads * 4));
    hiprtcProgram prog;
    hiprtcCreateProgram(&prog, kSrc, "p.hip", 0, nullptr, nullptr);
    const char *opts[] = {"--offload-arch=gfx950", "-O3"};
    if (hiprtcCompileProgram(prog, 2, opts) != HIPRTC_SUCCESS) { size_t n; hiprtcGetProgramLogSize(prog, &n); std::string l(n, 0); hiprtcGetProgramLog(prog, &l[0]); printf("%s\n", l.c_str()); return 1; }
    size_t cs; hiprtcGetCodeSize(prog, &cs); std::vector<char> code(cs); hiprtcGetCode(prog, code.data());
    hipModule_t mod; CK(hipModuleLoadData(&mod, code.data()));
    hipFunction_t f_plain, f_scratch, f_big; CK(hipModuleGetFunction(&f_big, mod, "rt_big")); CK(hipModuleGetFunction(&f_plain, mod, "rt_plain")); CK(hipModuleGetFunction(&f_scratch, mod, "rt_scratch"));
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void *)st_scratch)); printf("static scratch kernel: localSizeBytes %zu\n", (size_t)fa.localSizeBytes);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (unsigned n : {7u, 20000u}) for (int which = (n == 7u ? 4 : 0); which < (n == 7u ? 6 : 4); ++which) {
        const char *names[] = {"static plain  ", "module plain  ", "static scratch", "module scratch", "static big    ", "module big    "};
        auto launch = [&]() -> hipError_t {
            unsigned nn = n; void *args[] = {&d, &nn};
            switch (which) {
            case 0: hipLaunchKernelGGL(st_plain, blocks, threads, 0, st, d, nn); return hipGetLastError();
            case 1: return hipModuleLaunchKernel(f_plain, blocks, 1, 1, threads, 1, 1, 0, st, args, nullptr);
            case 4: hipLaunchKernelGGL(st_big, blocks, threads, 0, st, d, nn); return hipGetLastError();
            case 5: return hipModuleLaunchKernel(f_big, blocks, 1, 1, threads, 1, 1, 0, st, args, nullptr);
            case 2: hipLaunchKernelGGL(st_scratch, blocks, threads, 0, st, d, nn); return hipGetLastError();
            default: return hipModuleLaunchKernel(f_scratch, blocks, 1, 1, threads, 1, 1, 0, st, args, nullptr);
            }
        };
        for (int i = 0; i < 5; ++i) CK(launch());
        CK(hipStreamSynchronize(st));
        double ev = 0, wall = 0; const int R = 50;
        for (int i = 0; i < R; ++i) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, st)); CK(launch()); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            wall += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ev += ms;
        }
        printf("n=%-6u %s  event %.4f ms   wall %.4f ms\n", n, names[which], ev / R, wall / R);
    }
    return 0;
}
