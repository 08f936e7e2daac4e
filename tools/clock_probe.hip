// clock_probe.hip -- is the shader clock of an MI355X held down while the multiplier is saturated?  (r06: chunks of a streamed commit
// accumulate 8 % faster when the device idled 50-180 us before them, profiles/r06_ab_commit_schedules.txt.)
// Each kernel runs a dependent-free stream of one instruction kind on every SIMD for a given number of rounds; lane 0 of every 64th
// workgroup reads the shader cycle counter (s_memtime / clock64) and the constant 100 MHz counter (s_memrealtime / wall_clock64) at the
// start and the end: MHz = d(clock64) / d(wall_clock64) * wall rate.  Short (0.2 ms) and long (20 ms) launches, back to back and after idling.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/clock_probe.hip -o tools/clock_probe && tools/clock_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k_spin(unsigned long long *out, unsigned long long *stamp, int rounds, unsigned a, unsigned b) {
    unsigned long long acc[8];
    unsigned x = a + threadIdx.x, y = b | 1u;
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
                else if (KIND == 1) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
                else asm volatile("v_add_u32 %0, %1, %0" : "+v"(reinterpret_cast<unsigned *>(&acc[i])[0]) : "v"(x));
            }
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned long long s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) {
        stamp[(blockIdx.x >> 6) * 2] = c1 - c0;
        stamp[(blockIdx.x >> 6) * 2 + 1] = w1 - w0;
    }
}

int main() {
    int wall_khz = 0, sclk_khz = 0, cus = 0;
    CHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    CHECK(hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d, wall clock %d kHz, advertised shader clock %d kHz\n", cus, wall_khz, sclk_khz);
    const int blocks = cus * 4;            // 4 x 256 threads per CU: 4 waves per SIMD
    unsigned long long *out, *stamp;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    CHECK(hipMalloc(&stamp, (size_t)(blocks / 64 + 1) * 16));
    std::vector<unsigned long long> h(blocks / 64 * 2);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](int kind, int rounds, const char *what) -> int {
        CHECK(hipEventRecord(e0));
        if (kind == 0) hipLaunchKernelGGL(k_spin<0>, dim3(blocks), dim3(256), 0, 0, out, stamp, rounds, 12345u, 777u);
        else if (kind == 1) hipLaunchKernelGGL(k_spin<1>, dim3(blocks), dim3(256), 0, 0, out, stamp, rounds, 12345u, 777u);
        else hipLaunchKernelGGL(k_spin<2>, dim3(blocks), dim3(256), 0, 0, out, stamp, rounds, 12345u, 777u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), stamp, h.size() * 8, hipMemcpyDeviceToHost));
        double lo = 1e30, hi = 0, sum = 0;
        for (size_t i = 0; i < h.size() / 2; ++i) {
            const double mhz = (double)h[2 * i] / (double)h[2 * i + 1] * wall_khz / 1000.0;
            lo = mhz < lo ? mhz : lo; hi = mhz > hi ? mhz : hi; sum += mhz;
        }
        const double ops = (double)blocks * 256 * rounds * 64;
        printf("%-44s %8.3f ms  shader clock %6.0f MHz (min %6.0f max %6.0f)  %7.2f lane-ops/ns = %5.1f per CU per shader cycle\n", what, ms,
               sum / (h.size() / 2), lo, hi, ops / (ms * 1e6), ops / (ms * 1e6) / cus / (sum / (h.size() / 2) / 1000.0));
        return 0;
    };
    const char *names[3] = {"v_mad_u64_u32", "v_lshl_add_u64", "v_add_u32"};
    for (int kind = 0; kind < 3; ++kind) {
        char buf[96];
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        snprintf(buf, sizeof buf, "%s 0.2 ms after idling", names[kind]); if (run(kind, 60, buf)) return 1;
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        snprintf(buf, sizeof buf, "%s 1 ms after idling", names[kind]); if (run(kind, 300, buf)) return 1;
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        snprintf(buf, sizeof buf, "%s 20 ms after idling", names[kind]); if (run(kind, 6000, buf)) return 1;
        snprintf(buf, sizeof buf, "%s 20 ms again, back to back", names[kind]); if (run(kind, 6000, buf)) return 1;
        snprintf(buf, sizeof buf, "%s 1 ms right after", names[kind]); if (run(kind, 300, buf)) return 1;
        for (int rep = 0; rep < 3; ++rep) {     // a duty cycle like the commit's: 0.6 ms of work, 0.1 ms idle
            std::this_thread::sleep_for(std::chrono::microseconds(100));
            snprintf(buf, sizeof buf, "%s 0.6 ms after a 0.1 ms pause", names[kind]); if (run(kind, 180, buf)) return 1;
        }
    }
    return 0;
}
