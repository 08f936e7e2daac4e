// ubench29.hip -- gfx950: the 9 x 29-bit carry-free Montgomery product (field29.cuh) against the 8 x 32-bit FIPS form
// (field.cuh), at field and at mixed-addition level, plus the issue rates of the instructions the new form leans on.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sirius_amd/csrc tools/ubench29.hip -o tools/ubench29 && tools/ubench29
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include "curve29.cuh"
using namespace srs;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048, UNROLL = 16;
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
__global__ void k_mad_u64_u32_sgpr(uint64_t *out, uint32_t a, uint32_t b) {   // one SGPR operand (the m * p half)
    uint64_t acc[4];
    uint32_t x = a + threadIdx.x;
    for (int i = 0; i < 4; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "s"(b) : "vcc");
            FOUR(OP)
#undef OP
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}
__global__ void k_lshrrev_b64(uint64_t *out, uint32_t a, uint32_t b) {
    uint64_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = ((uint64_t)a << 32 | b) + threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#define OP(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
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
GEN32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 29")
GEN32(k_and_b32, "v_and_b32 %0, 0x1fffffff, %0")
GEN32(k_lshrrev_b32, "v_lshrrev_b32 %0, 1, %0")
GEN32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
GEN32(k_add_u32, "v_add_u32 %0, %0, %1")

constexpr int FITERS = 256;
template <class F, int CHAINS>
__global__ void k_fmul32(fe_t *out, const fe_t *in) {
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
template <class F, int CHAINS, bool SQR>
__global__ void k_fmul29(fe_t *out, const fe_t *in) {
    f29_t a[CHAINS];
    f29_t b = F::unpack(in[threadIdx.x & 63]);
    for (int i = 0; i < CHAINS; ++i) a[i] = F::unpack(in[(threadIdx.x + i + 1) & 63]);
    for (int it = 0; it < FITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) a[i] = SQR ? F::sqr(a[i]) : F::mul(a[i], b);
    }
    f29_t r = a[0];
    for (int i = 1; i < CHAINS; ++i) r = F::normalize(F::add_lazy(r, a[i]));
    out[blockIdx.x * blockDim.x + threadIdx.x] = F::to_canonical_fe(r);
}
template <class C>
__global__ void k_madd32(xyzz_t *out, const affine_t *pts) {
    xyzz_t acc = Ec<C>::identity();
    for (int it = 0; it < 64; ++it) acc = Ec<C>::madd(acc, pts[(threadIdx.x * 7 + it * 13 + blockIdx.x) & 1023]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <class C>
__global__ void k_madd29(xyzz_t *out, const affine_t *pts29) {
    xyzz29_t acc = Ec29<C>::identity();
    for (int it = 0; it < 64; ++it) acc = Ec29<C>::madd(acc, Ec29<C>::load(pts29[(threadIdx.x * 7 + it * 13 + blockIdx.x) & 1023], false));
    out[blockIdx.x * blockDim.x + threadIdx.x] = Ec29<C>::to_xyzz(acc);
}
__global__ void k_fill_pts(affine_t *pts, affine_t *pts29) {
    uint32_t k[8] = {threadIdx.x + blockIdx.x * blockDim.x + 1, 0x9e3779b9u, 0x7f4a7c15u, 0x1234567u, 0, 0, 0, 0};
    affine_t g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one());
    affine_t p = EcBn::to_affine(EcBn::mul_canon(k, g));
    pts[threadIdx.x + blockIdx.x * blockDim.x] = p;
    pts29[threadIdx.x + blockIdx.x * blockDim.x] = Ec29<Bn256>::table_form(p);
}

template <class K, class... A>
static double time_kernel(const char *name, double ops_per_thread, int blocks, int threads, K k, A... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int reps = 3;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, args...);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    double rate = ops_per_thread * blocks * threads / (ms * 1e-3);
    printf("%-34s %9.3f ms  %10.3f Gops/s  %7.2f lane-ops/clk/CU  (blocks=%d thr=%d)\n", name, ms, rate * 1e-9, rate / (2.4e9 * 256), blocks, threads);
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
    time_kernel("v_mad_u64_u32 (vgpr,vgpr)", ops, blocks, threads, k_mad_u64_u32, out, 123u, 456u);
    time_kernel("v_mad_u64_u32 (vgpr,sgpr)", ops, blocks, threads, k_mad_u64_u32_sgpr, out, 123u, 456u);
    time_kernel("v_lshrrev_b64", ops, blocks, threads, k_lshrrev_b64, out, 123u, 456u);
    time_kernel("v_alignbit_b32", ops, blocks, threads, k_alignbit, out, 123u, 456u);
    time_kernel("v_and_b32 (literal)", ops, blocks, threads, k_and_b32, out, 123u, 456u);
    time_kernel("v_lshrrev_b32", ops, blocks, threads, k_lshrrev_b32, out, 123u, 456u);
    time_kernel("v_mul_lo_u32", ops, blocks, threads, k_mul_lo_u32, out, 123u, 456u);
    time_kernel("v_add_u32", ops, blocks, threads, k_add_u32, out, 123u, 456u);

    std::vector<fe_t> h(64);
    for (int i = 0; i < 64; ++i) h[i] = Fr::from_u64(0x9e3779b97f4a7c15ull * (i + 3));
    fe_t *din; CHECK(hipMalloc(&din, 64 * sizeof(fe_t)));
    CHECK(hipMemcpy(din, h.data(), 64 * sizeof(fe_t), hipMemcpyHostToDevice));
    fe_t *fout = (fe_t *)out;
    for (int occ : {2, 4, 8}) {
        char nm[64];
        snprintf(nm, 64, "Fr::mul (8x32 FIPS) blk/CU=%d", occ);
        time_kernel(nm, FITERS, 256 * occ, 256, k_fmul32<Fr, 1>, fout, (const fe_t *)din);
        snprintf(nm, 64, "Fr29::mul (9x29) blk/CU=%d", occ);
        time_kernel(nm, FITERS, 256 * occ, 256, k_fmul29<Fr29, 1, false>, fout, (const fe_t *)din);
        snprintf(nm, 64, "Fr29::sqr (9x29) blk/CU=%d", occ);
        time_kernel(nm, FITERS, 256 * occ, 256, k_fmul29<Fr29, 1, true>, fout, (const fe_t *)din);
    }
    time_kernel("Fr29::mul x2chain blk/CU=4", FITERS * 2, 256 * 4, 256, k_fmul29<Fr29, 2, false>, fout, (const fe_t *)din);
    time_kernel("Fq29::mul blk/CU=8", FITERS, 256 * 8, 256, k_fmul29<Fq29, 1, false>, fout, (const fe_t *)din);
    time_kernel("Fr29::mul latency (1 wave)", FITERS, 1, 64, k_fmul29<Fr29, 1, false>, fout, (const fe_t *)din);
    time_kernel("Fr::mul latency (1 wave)", FITERS, 1, 64, k_fmul32<Fr, 1>, fout, (const fe_t *)din);

    affine_t *pts, *pts29; CHECK(hipMalloc(&pts, 1024 * sizeof(affine_t))); CHECK(hipMalloc(&pts29, 1024 * sizeof(affine_t)));
    hipLaunchKernelGGL(k_fill_pts, dim3(16), dim3(64), 0, 0, pts, pts29);
    CHECK(hipDeviceSynchronize());
    xyzz_t *pout; CHECK(hipMalloc(&pout, sizeof(xyzz_t) * 256 * 8 * 256));
    xyzz_t *pout2; CHECK(hipMalloc(&pout2, sizeof(xyzz_t) * 256 * 8 * 256));
    for (int occ : {2, 4}) {
        char nm[64];
        snprintf(nm, 64, "bn256 madd (8x32) blk/CU=%d", occ);
        time_kernel(nm, 64, 256 * occ, 256, k_madd32<Bn256>, pout, (const affine_t *)pts);
        snprintf(nm, 64, "bn256 madd (9x29) blk/CU=%d", occ);
        time_kernel(nm, 64, 256 * occ, 256, k_madd29<Bn256>, pout2, (const affine_t *)pts29);
    }
    for (int thr : {128}) {
        time_kernel("bn256 madd (9x29) thr=128 blk=2048", 64, 2048, thr, k_madd29<Bn256>, pout2, (const affine_t *)pts29);
        time_kernel("bn256 madd (8x32) thr=128 blk=2048", 64, 2048, thr, k_madd32<Bn256>, pout, (const affine_t *)pts);
    }
    // the two forms must agree bit for bit
    std::vector<xyzz_t> r1(2048 * 128), r2(2048 * 128);
    CHECK(hipMemcpy(r1.data(), pout, r1.size() * sizeof(xyzz_t), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(r2.data(), pout2, r2.size() * sizeof(xyzz_t), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < r1.size(); ++i) if (memcmp(&r1[i], &r2[i], sizeof(xyzz_t))) ++bad;
    printf("madd 9x29 vs 8x32: %zu of %zu results differ\n", bad, r1.size());
    time_kernel("bn256 madd (9x29) latency 1 wave", 64, 1, 64, k_madd29<Bn256>, pout2, (const affine_t *)pts29);
    printf("done\n");
    return bad ? 2 : 0;
}
