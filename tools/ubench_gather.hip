// ubench_gather.hip -- random gathers of 64-byte and 128-byte entries from a table far larger than the caches (gfx950): gathers per second and
// useful bytes per second at saturation, to tell whether a 64-byte gather (one affine point of the MSM window table) costs the memory system a
// 64-byte or a 128-byte fetch ("put a number on the 128-byte-request waste of the 64-byte gathers").
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip      run: tools/ubench_gather [log2 table bytes = 33]
// Every thread runs ITER rounds of K independent gathers (indices from a hash of (thread, round, k): no two alike), XORs what it loaded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// WORDS16 = 16-byte words per entry: 4 -> 64-byte entries, 8 -> 128-byte entries (entry-size aligned); HALF: 64-byte entries that are the
// FIRST half of a 128-byte-aligned block only (the other half is never touched: a 128-byte fetch would be half wasted by construction)
// k_gather_sliced: the MSM's pattern -- 16 window slices of (mask + 1) 64-byte entries each, `stride` entries apart (T[w * len + i]: a chunk of
// the witness touches the same index range of all 16 windows); stride == mask + 1: one contiguous range (what a base-major table T[i][w] gives)
template <int K>
__global__ void __launch_bounds__(256) k_gather_sliced(const uint4 *__restrict__ table, uint64_t mask, uint64_t stride, uint32_t iters, uint4 *__restrict__ sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[K][4];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t h = mix(t * 0x9e3779b97f4a7c15ull + (uint64_t)it * K + k);
            const uint64_t e = (h >> 60) * stride + ((h >> 8) & mask);
            const uint4 *p = table + e * 4;
#pragma unroll
            for (int w = 0; w < 4; ++w) v[k][w] = p[w];
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int w = 0; w < 4; ++w) { acc.x ^= v[k][w].x; acc.y ^= v[k][w].y; acc.z ^= v[k][w].z; acc.w ^= v[k][w].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[t & 1023] = acc;
}

template <int WORDS16, int K, bool HALF>
__global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ table, uint64_t mask, uint32_t iters, uint4 *__restrict__ sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[K][WORDS16];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uint64_t e = mix(t * 0x9e3779b97f4a7c15ull + (uint64_t)it * K + k) & mask;
            if (HALF) e &= ~1ull;
            const uint4 *p = table + e * (HALF ? 4 : WORDS16);
#pragma unroll
            for (int w = 0; w < WORDS16; ++w) v[k][w] = p[w];
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int w = 0; w < WORDS16; ++w) { acc.x ^= v[k][w].x; acc.y ^= v[k][w].y; acc.z ^= v[k][w].z; acc.w ^= v[k][w].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[t & 1023] = acc;      // never true in practice: keeps the loads alive
}

template <int WORDS16, int K, bool HALF>
static void run(const char *name, const uint4 *table, size_t table_bytes, uint4 *sink) {
    const uint64_t entries = table_bytes / (HALF ? 64 : WORDS16 * 16);
    const uint32_t blocks = 256 * 32, iters = 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<WORDS16, K, HALF>), dim3(blocks), dim3(256), 0, 0, table, entries - 1, 4u, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_gather<WORDS16, K, HALF>), dim3(blocks), dim3(256), 0, 0, table, entries - 1, iters, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double gathers = (double)blocks * 256 * iters * K;
    std::printf("%-44s %7.2f G gathers/s  %7.1f GB/s useful  (%.3f ms, %d in flight per thread)\n", name, gathers / ms / 1e6,
                gathers * WORDS16 * 16 / ms / 1e6, ms, K);
}

static void run_sliced(const char *name, const uint4 *table, uint64_t slice_entries, uint64_t stride, uint4 *sink) {
    const uint32_t blocks = 256 * 32, iters = 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather_sliced<4>), dim3(blocks), dim3(256), 0, 0, table, slice_entries - 1, stride, 4u, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_gather_sliced<4>), dim3(blocks), dim3(256), 0, 0, table, slice_entries - 1, stride, iters, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double gathers = (double)blocks * 256 * iters * 4;
    std::printf("%-60s %7.2f G gathers/s  %7.1f GB/s useful\n", name, gathers / ms / 1e6, gathers * 64 / ms / 1e6);
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? std::atoi(argv[1]) : 33;
    const size_t bytes = (size_t)1 << lg;
    uint4 *table = nullptr, *sink = nullptr;
    CHECK(hipMalloc((void **)&table, bytes));
    CHECK(hipMalloc((void **)&sink, 1024 * sizeof(uint4)));
    CHECK(hipMemset(table, 0x5a, bytes));
    std::printf("random gathers from a %.0f GiB table (gfx950)\n", bytes / 1073741824.0);
    run<4, 2, false>("64-byte entries, 2 in flight", table, bytes, sink);
    run<4, 4, false>("64-byte entries, 4 in flight", table, bytes, sink);
    run<4, 8, false>("64-byte entries, 8 in flight", table, bytes, sink);
    run<8, 2, false>("128-byte entries (aligned), 2 in flight", table, bytes, sink);
    run<8, 4, false>("128-byte entries (aligned), 4 in flight", table, bytes, sink);
    run<4, 4, true>("64-byte entries, first halves of 128 B only", table, bytes, sink);
    run<4, 8, true>("64-byte entries, first halves only, 8", table, bytes, sink);
    if (lg >= 34) {      // the MSM's access pattern inside a 16 GiB table: 16 window slices against one contiguous range of the same total size
        const uint64_t GiB = 1ull << 24;      // entries of 64 B per GiB
        run_sliced("16 slices of 64 MiB, 1 GiB apart (a 2^20-scalar chunk, T[w][i])", table, 1ull << 20, GiB, sink);
        run_sliced("one contiguous 1 GiB (the same chunk, T[i][w])", table, 1ull << 20, 1ull << 20, sink);
        run_sliced("16 slices of 128 MiB, 1 GiB apart (2^21-scalar chunk)", table, 1ull << 21, GiB, sink);
        run_sliced("one contiguous 2 GiB", table, 1ull << 21, 1ull << 21, sink);
        run_sliced("16 slices of 16 MiB, 1 GiB apart (2^18-scalar chunk)", table, 1ull << 18, GiB, sink);
        run_sliced("one contiguous 256 MiB", table, 1ull << 18, 1ull << 18, sink);
        run_sliced("16 slices of 1 GiB = the whole 16 GiB table", table, GiB, GiB, sink);
    }
    return 0;
}
