// pcie_probe.cpp -- host-to-device copy rates on the GPU box: pageable vs page-locked source, by chunk size, one or two
// streams, and beside a running kernel.  hipcc --offload-arch=gfx950 -O2 tools/pcie_probe.cpp -o tools/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_spin(uint64_t *o, int iters) {
    uint64_t a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 6364136223846793005ull + 1442695040888963407ull;
    o[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    const size_t total = 403ull << 20;     // 12 * 2^20 field elements
    void *d = nullptr, *pinned = nullptr;
    CHECK(hipMalloc(&d, total));
    char *pageable = (char *)malloc(total);
    memset(pageable, 1, total);
    CHECK(hipHostMalloc(&pinned, total, hipHostMallocDefault));
    memset(pinned, 2, total);
    hipStream_t s0, s1, sk;
    CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    auto run = [&](const char *name, const void *src, size_t chunk, int nstreams, bool with_kernel) {
        uint64_t *spin = nullptr;
        if (with_kernel) { hipMalloc(&spin, 2048 * 256 * 8); hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, sk, spin, 4000000); }
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(s0); hipStreamSynchronize(s1);
            double t0 = now();
            size_t off = 0; int i = 0;
            while (off < total) {
                size_t c = total - off < chunk ? total - off : chunk;
                hipMemcpyAsync((char *)d + off, (const char *)src + off, c, hipMemcpyHostToDevice, (i++ % nstreams) ? s1 : s0);
                off += c;
            }
            hipStreamSynchronize(s0); hipStreamSynchronize(s1);
            double dt = now() - t0;
            if (dt < best) best = dt;
        }
        if (with_kernel) { hipStreamSynchronize(sk); hipFree(spin); }
        printf("%-44s chunk %8zu KiB streams %d : %7.2f ms  %6.2f GB/s\n", name, chunk >> 10, nstreams, best * 1e3, total / best / 1e9);
        fflush(stdout);
        return 0;
    };
    for (size_t chunk : {total, (size_t)64 << 20, (size_t)16 << 20, (size_t)4 << 20, (size_t)1 << 20}) {
        run("H2D pageable", pageable, chunk, 1, false);
        run("H2D page-locked (hipHostMalloc)", pinned, chunk, 1, false);
    }
    run("H2D page-locked, 2 streams", pinned, (size_t)16 << 20, 2, false);
    run("H2D page-locked beside a busy kernel", pinned, (size_t)16 << 20, 1, true);
    // hipHostRegister of an existing allocation (what a shim could do with a Rust Vec)
    double t0 = now();
    CHECK(hipHostRegister(pageable, total, hipHostRegisterDefault));
    printf("hipHostRegister(403 MiB): %.2f ms\n", (now() - t0) * 1e3);
    run("H2D registered (hipHostRegister)", pageable, (size_t)16 << 20, 1, false);
    run("H2D registered, whole", pageable, total, 1, false);
    t0 = now();
    CHECK(hipHostUnregister(pageable));
    printf("hipHostUnregister: %.2f ms\n", (now() - t0) * 1e3);
    // D2H
    {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) { double t = now(); hipMemcpy(pinned, d, total, hipMemcpyDeviceToHost); double dt = now() - t; if (dt < best) best = dt; }
        printf("D2H page-locked whole: %.2f ms %.2f GB/s\n", best * 1e3, total / best / 1e9);
    }
    // CPU memcpy into the pinned buffer (the staging copy a shim would pay), single thread
    {
        double t = now(); memcpy(pinned, pageable, total); double dt = now() - t;
        printf("host memcpy pageable -> page-locked, 1 thread: %.2f ms %.2f GB/s\n", dt * 1e3, total / dt / 1e9);
    }
    printf("done\n");
    return 0;
}
