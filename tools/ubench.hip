// ubench.hip -- gfx950 instruction-rate probes for the 256-bit modular arithmetic design
// (SURVEY.md section 7 step 0): which integer-multiply flavour is fastest on CDNA4, and what the
// field / group primitives built on it sustain.  Standalone:  hipcc --offload-arch=gfx950 -O3
// -I sirius_amd/csrc tools/ubench.hip -o tools/ubench && tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "curve.cuh"
using namespace srs;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 16;   // instructions per chain per loop trip, 4 independent chains

#define FOUR(op) op(0) op(1) op(2) op(3)

__global__ void k_mad_u64_u32(uint64_t *out, uint32_t a, uint32_t b) {
    uint64_t acc[4];
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (int i = 0; i < 4; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
            FOUR(OP)
#undef OP
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}
#define GEN32(name, asmtext)                                                              \
    __global__ void name(uint64_t *out, uint32_t a, uint32_t b) {                         \
        uint32_t acc[4];                                                                  \
        uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;                                \
        (void)y;                                                                          \
        for (int i = 0; i < 4; ++i) acc[i] = threadIdx.x + i;                             \
        for (int it = 0; it < ITERS; ++it) {                                              \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                          \
                asm volatile(asmtext : "+v"(acc[0]) : "v"(x), "v"(y));                    \
                asm volatile(asmtext : "+v"(acc[1]) : "v"(x), "v"(y));                    \
                asm volatile(asmtext : "+v"(acc[2]) : "v"(x), "v"(y));                    \
                asm volatile(asmtext : "+v"(acc[3]) : "v"(x), "v"(y));                    \
            }                                                                             \
        }                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];   \
    }
GEN32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
GEN32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
GEN32(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
GEN32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
GEN32(k_add_u32, "v_add_u32 %0, %0, %1")
GEN32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
GEN32(k_mov_b32, "v_mov_b32 %0, %1")

__global__ void k_fma_f64(uint64_t *out, uint32_t a, uint32_t b) {
    double acc[4];
    double x = 1.0 + a * 1e-9 + threadIdx.x * 1e-12, y = 1e-9 * b;
    for (int i = 0; i < 4; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#define OP(i) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
            FOUR(OP)
#undef OP
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)(acc[0] + acc[1] + acc[2] + acc[3]);
}
__global__ void k_lshl_add_u64(uint64_t *out, uint32_t a, uint32_t b) {
    uint64_t acc[4];
    uint64_t x = ((uint64_t)a << 32) | b;
    for (int i = 0; i < 4; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#define OP(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(x));
            FOUR(OP)
#undef OP
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}
__global__ void k_addc_chain(uint64_t *out, uint32_t a, uint32_t b) {
    uint32_t lo[4], hi[4];
    uint32_t x = a + threadIdx.x, y = b;
    for (int i = 0; i < 4; ++i) { lo[i] = threadIdx.x + i; hi[i] = i; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL / 2; ++u) {
#define OP(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo[i]), "+v"(hi[i]) : "v"(x), "v"(y) : "vcc");
            FOUR(OP)
#undef OP
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = lo[0] ^ lo[1] ^ lo[2] ^ lo[3] ^ hi[0] ^ hi[1] ^ hi[2] ^ hi[3];
}

// ---- field / curve level ----
constexpr int FITERS = 256;
template <class F, int CHAINS>
__global__ void k_fmul(fe_t *out, const fe_t *in) {
    fe_t a[CHAINS];
    fe_t b = in[threadIdx.x & 63];
    for (int i = 0; i < CHAINS; ++i) a[i] = in[(threadIdx.x + i + 1) & 63];
    for (int it = 0; it < FITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) a[i] = F::mul(a[i], b);
    }
    fe_t r = a[0];
    for (int i = 1; i < CHAINS; ++i) r = F::add(r, a[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class F>
__global__ void k_fadd(fe_t *out, const fe_t *in) {
    fe_t a = in[threadIdx.x & 63], b = in[(threadIdx.x + 7) & 63];
    for (int it = 0; it < FITERS * 8; ++it) { a = F::add(a, b); b = F::sub(b, a); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = F::add(a, b);
}
template <class C>
__global__ void k_madd(xyzz_t *out, const affine_t *pts) {
    xyzz_t acc = Ec<C>::identity();
    for (int it = 0; it < 64; ++it) acc = Ec<C>::madd(acc, pts[(threadIdx.x * 7 + it * 13 + blockIdx.x) & 1023]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <class C>
__global__ void k_padd(xyzz_t *out, const affine_t *pts) {
    xyzz_t acc = Ec<C>::from_affine(pts[threadIdx.x & 1023]);
    xyzz_t b = Ec<C>::dbl(Ec<C>::from_affine(pts[(threadIdx.x + 5) & 1023]));
    for (int it = 0; it < 64; ++it) { acc = Ec<C>::add(acc, b); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_fill_pts(affine_t *pts) {
    // points [i+1]G on bn256 for i < 1024 (slow, one thread each)
    uint32_t k[8] = {threadIdx.x + blockIdx.x * blockDim.x + 1, 0x9e3779b9u, 0x7f4a7c15u, 0x1234567u, 0, 0, 0, 0};
    affine_t g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one());
    pts[threadIdx.x + blockIdx.x * blockDim.x] = EcBn::to_affine(EcBn::mul_canon(k, g));
}

template <class K, class... A>
static double time_kernel(const char *name, double ops_per_thread, int blocks, int threads, K k, A... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, args...);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int reps = 3;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, args...);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    double total = ops_per_thread * blocks * threads;
    double rate = total / (ms * 1e-3);
    // lanes/clk/CU at 2.4 GHz x 256 CUs
    printf("%-28s %9.3f ms  %10.3f Gops/s  %7.2f lane-ops/clk/CU  (blocks=%d thr=%d)\n", name, ms, rate * 1e-9,
           rate / (2.4e9 * 256), blocks, threads);
    fflush(stdout);
    return rate;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s), CUs=%d, clock=%d MHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const int blocks = 256 * 8, threads = 256;
    uint64_t *out; CHECK(hipMalloc(&out, sizeof(uint64_t) * blocks * threads * 16));
    double ops = (double)ITERS * UNROLL * 4;
    time_kernel("v_mad_u64_u32", ops, blocks, threads, k_mad_u64_u32, out, 123u, 456u);
    time_kernel("v_mul_lo_u32", ops, blocks, threads, k_mul_lo_u32, out, 123u, 456u);
    time_kernel("v_mul_hi_u32", ops, blocks, threads, k_mul_hi_u32, out, 123u, 456u);
    time_kernel("v_mad_u32_u24", ops, blocks, threads, k_mad_u32_u24, out, 123u, 456u);
    time_kernel("v_mul_hi_u32_u24", ops, blocks, threads, k_mul_hi_u32_u24, out, 123u, 456u);
    time_kernel("v_add_u32", ops, blocks, threads, k_add_u32, out, 123u, 456u);
    time_kernel("v_add3_u32", ops, blocks, threads, k_add3_u32, out, 123u, 456u);
    time_kernel("v_mov_b32", ops, blocks, threads, k_mov_b32, out, 123u, 456u);
    time_kernel("v_fma_f64", ops, blocks, threads, k_fma_f64, out, 123u, 456u);
    time_kernel("v_lshl_add_u64", ops, blocks, threads, k_lshl_add_u64, out, 123u, 456u);
    time_kernel("v_add_co+v_addc (pairs)", ops / 2, blocks, threads, k_addc_chain, out, 123u, 456u);

    // field level
    std::vector<fe_t> h(64);
    for (int i = 0; i < 64; ++i) h[i] = Fr::from_u64(0x9e3779b97f4a7c15ull * (i + 3));
    fe_t *din; CHECK(hipMalloc(&din, 64 * sizeof(fe_t)));
    CHECK(hipMemcpy(din, h.data(), 64 * sizeof(fe_t), hipMemcpyHostToDevice));
    fe_t *fout = (fe_t *)out;
    for (int occ : {1, 2, 4, 8}) {
        char nm[64];
        snprintf(nm, 64, "Fr::mul x1chain blk/CU=%d", occ);
        time_kernel(nm, FITERS, 256 * occ, 256, k_fmul<Fr, 1>, fout, (const fe_t *)din);
    }
    time_kernel("Fr::mul x2chain blk/CU=4", FITERS * 2, 256 * 4, 256, k_fmul<Fr, 2>, fout, (const fe_t *)din);
    time_kernel("Fq::mul x1chain blk/CU=4", FITERS, 256 * 4, 256, k_fmul<Fq, 1>, fout, (const fe_t *)din);
    time_kernel("Fr::add+sub blk/CU=4", FITERS * 16, 256 * 4, 256, k_fadd<Fr>, fout, (const fe_t *)din);
    // single-wave latency probe: 1 block of 64 threads
    time_kernel("Fr::mul latency (1 wave)", FITERS, 1, 64, k_fmul<Fr, 1>, fout, (const fe_t *)din);

    affine_t *pts; CHECK(hipMalloc(&pts, 1024 * sizeof(affine_t)));
    hipLaunchKernelGGL(k_fill_pts, dim3(16), dim3(64), 0, 0, pts);
    CHECK(hipDeviceSynchronize());
    xyzz_t *pout; CHECK(hipMalloc(&pout, sizeof(xyzz_t) * 256 * 8 * 256));
    for (int occ : {1, 2, 4}) {
        char nm[64];
        snprintf(nm, 64, "bn256 madd blk/CU=%d", occ);
        time_kernel(nm, 64, 256 * occ, 256, k_madd<Bn256>, pout, (const affine_t *)pts);
    }
    time_kernel("bn256 add(xyzz) blk/CU=4", 64, 256 * 4, 256, k_padd<Bn256>, pout, (const affine_t *)pts);
    time_kernel("bn256 madd latency (1 wave)", 64, 1, 64, k_madd<Bn256>, pout, (const affine_t *)pts);
    time_kernel("bn256 add latency (1 wave)", 64, 1, 64, k_padd<Bn256>, pout, (const affine_t *)pts);
    printf("done\n");
    return 0;
}
