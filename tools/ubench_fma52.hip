// ubench_fma52.hip -- gfx950 (r04, VERDICT r03 item 2 i): is a 256-bit Montgomery product on FP64 FMAs faster than the 9 x 29-bit integer
// form (field29.cuh)?  The double-precision route of Emmart / Zheng / Weems ("Faster modular exponentiation using double precision
// floating point arithmetic on the GPU", ARITH 2018): 5 limbs of 52 bits held exactly in doubles, R = 2^260; a limb product a_i * b_j
// < 2^104 is split into its high and low 52 bits by TWO fused multiply-adds in round-toward-zero mode
//     hi = fma(a, b, 2^104)              -> 2^104 + H * 2^52          (the mantissa field of hi IS H)
//     lo = fma(a, b, (2^104 + 2^52) - hi) -> 2^52 + L                   (the mantissa field of lo IS L)
// and the column sums are accumulated as 64-bit INTEGER additions of the raw bit patterns (the exponent fields add up to a constant
// that is subtracted per column).  Per limb product: 2 v_fma_f64 + 1 v_add_f64 + 2 v_lshl_add_u64 (all issue at the 64-bit rate) for
// 52 x 52 bits, against ONE v_mad_u64_u32 for 29 x 29 bits.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sirius_amd/csrc tools/ubench_fma52.hip -o tools/ubench_fma52 && tools/ubench_fma52
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "field29.cuh"
using namespace srs;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048, UNROLL = 16;
#define FOUR(op) op(0) op(1) op(2) op(3)

__global__ void k_fma_f64(double *out, double a, double b) {
    double acc[4];
    double x = a + threadIdx.x, y = b;
    for (int i = 0; i < 4; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#define OP(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
            FOUR(OP)
#undef OP
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void k_add_f64(double *out, double a, double b) {
    double acc[4];
    double x = a + threadIdx.x;
    for (int i = 0; i < 4; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#define OP(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
            FOUR(OP)
#undef OP
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void k_lshl_add_u64(uint64_t *out, uint64_t a) {
    uint64_t acc[4];
    uint64_t x = a + threadIdx.x;
    for (int i = 0; i < 4; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#define OP(i) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(acc[i]) : "v"(x));
            FOUR(OP)
#undef OP
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

// ---- the FP64 Montgomery product, bn256 Fr, R = 2^260 ---------------------------------------------------------------------------
struct f52_t { double v[5]; };      // limbs < 2^52, exact integers
struct Fr52 {
    static constexpr double C1 = 20282409603651670423947251286016.0;                 // 2^104
    static constexpr double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0;   // 2^104 + 2^52
    static constexpr uint64_t M52 = (1ull << 52) - 1;
    // p = bn256 Fr modulus in 52-bit limbs, np = -p^-1 mod 2^52 (filled by the host into constant memory)
};
__constant__ double c_p52[5];
__constant__ double c_np52;

__device__ __forceinline__ void set_rtz() {
    // MODE.FP_ROUND[3:2] = double/half rounding: 3 = toward zero
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}
// v_fma_f64 under the rounding mode set by set_rtz (hipcc has no __fma_rz; the asm also keeps the compiler from folding the constants)
__device__ __forceinline__ double fma_rz(double a, double b, double c) {
    double d;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint64_t bits(double x) { return (uint64_t)__double_as_longlong(x); }
__device__ __forceinline__ double dbl(uint64_t x) { return __longlong_as_double((long long)x); }

__device__ __forceinline__ void mulacc(double a, double b, uint64_t &c_lo, uint64_t &c_hi) {   // columns k (lo) and k + 1 (hi)
    const double hi = fma_rz(a, b, Fr52::C1);
    const double sub = Fr52::C2 - hi;
    const double lo = fma_rz(a, b, sub);
    c_hi += bits(hi);
    c_lo += bits(lo);
}

__device__ f52_t mul52(const f52_t &a, const f52_t &b) {
    // the exponent patterns that pile up per column: every `hi` brings E104 << 52, every `lo` brings E52 << 52
    constexpr uint64_t E104 = (uint64_t)(1023 + 104) << 52, E52 = (uint64_t)(1023 + 52) << 52;
    uint64_t col[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) mulacc(a.v[i], b.v[j], col[i + j], col[i + j + 1]);
#pragma unroll
    for (int k = 0; k < 10; ++k) {      // remove the exponent fields: column k holds (#lo terms) E52 + (#hi terms) E104
        const int nlo = (k < 5 ? k + 1 : 9 - k), nhi = (k >= 1 ? (k - 1 < 5 ? k : 10 - k) : 0);
        col[k] -= (uint64_t)(nlo > 0 ? nlo : 0) * E52 + (uint64_t)(nhi > 0 ? nhi : 0) * E104;
    }
    // Montgomery: five rounds, q = (col_i mod 2^52) * np mod 2^52, col += q * p * 2^(52 i)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double ci = dbl((col[i] & Fr52::M52) | E52) - 4503599627370496.0;        // low 52 bits as a double
        const double qh = fma_rz(ci, c_np52, Fr52::C1);
        const double ql = fma_rz(ci, c_np52, Fr52::C2 - qh);
        const double q = dbl((bits(ql) & Fr52::M52) | E52) - 4503599627370496.0;
        uint64_t lo_acc = 0, hi_acc = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            uint64_t l = 0, h = 0;
            mulacc(q, c_p52[j], l, h);
            col[i + j] += l - E52;
            col[i + j + 1] += h - E104;
        }
        (void)lo_acc; (void)hi_acc;
        col[i + 1] += col[i] >> 52;                                                       // the low 52 bits are zero now
    }
    f52_t o;
#pragma unroll
    for (int k = 5; k < 10; ++k) {
        if (k + 1 < 11 && k < 9) col[k + 1] += col[k] >> 52;
        o.v[k - 5] = dbl((col[k] & Fr52::M52) | E52) - 4503599627370496.0;
    }
    return o;
}

constexpr int FITERS = 256;
template <int CHAINS>
__global__ void k_fmul52(f52_t *out, const f52_t *in) {
    set_rtz();
    f52_t a[CHAINS];
    const f52_t b = in[threadIdx.x & 63];
    for (int i = 0; i < CHAINS; ++i) a[i] = in[(threadIdx.x + i + 1) & 63];
    for (int it = 0; it < FITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) a[i] = mul52(a[i], b);
    }
    f52_t r = a[0];
    for (int i = 1; i < CHAINS; ++i)
        for (int l = 0; l < 5; ++l) r.v[l] += a[i].v[l];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_mul52_once(f52_t *out, const f52_t *a, const f52_t *b) {
    set_rtz();
    out[threadIdx.x] = mul52(a[threadIdx.x], b[threadIdx.x]);
}
template <class F, int CHAINS, int MODE>       // MODE 0 mul, 1 sqr, 2 mul2
__global__ void k_fmul29(fe_t *out, const fe_t *in) {
    f29_t a[CHAINS];
    f29_t b = F::unpack(in[threadIdx.x & 63]);
    for (int i = 0; i < CHAINS; ++i) a[i] = F::unpack(in[(threadIdx.x + i + 1) & 63]);
    for (int it = 0; it < FITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) a[i] = MODE == 1 ? F::sqr(a[i]) : (MODE == 2 ? F::mul2(a[i], b, b, a[i]) : F::mul(a[i], b));
    }
    f29_t r = a[0];
    for (int i = 1; i < CHAINS; ++i) r = F::normalize(F::add_lazy(r, a[i]));
    out[blockIdx.x * blockDim.x + threadIdx.x] = F::to_canonical_fe(r);
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
    printf("%-40s %9.3f ms  %10.3f Gops/s  %7.2f lane-ops/clk/CU  (blocks=%d thr=%d)\n", name, ms, rate * 1e-9, rate / (2.4e9 * 256), blocks, threads);
    fflush(stdout);
    return rate;
}

// ---- host reference: 256-bit integers as 4 x u64, Montgomery product with R = 2^260 via plain big-integer arithmetic ------------
typedef unsigned __int128 u128;
struct Big { uint64_t w[9]; };     // up to 576 bits
static Big big_mul(const uint64_t *a, int na, const uint64_t *b, int nb) {
    Big r; memset(&r, 0, sizeof r);
    for (int i = 0; i < na; ++i) {
        u128 c = 0;
        for (int j = 0; j < nb; ++j) { u128 t = (u128)a[i] * b[j] + r.w[i + j] + c; r.w[i + j] = (uint64_t)t; c = t >> 64; }
        r.w[i + nb] += (uint64_t)c;
    }
    return r;
}
static const uint64_t FR_P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static void to52(const uint64_t *x, double *o) {        // 4 x u64 (< 2^260) -> 5 x 52-bit
    for (int i = 0; i < 5; ++i) {
        int bit = 52 * i, wi = bit >> 6, sh = bit & 63;
        uint64_t v = x[wi] >> sh;
        if (sh > 12 && wi + 1 < 4) v |= x[wi + 1] << (64 - sh);
        o[i] = (double)(v & ((1ull << 52) - 1));
    }
}
static void from52(const double *d, uint64_t *x) {      // limbs may exceed 52 bits slightly? (they do not here) -> 5 x u64 words
    u128 acc = 0; int bitpos = 0; memset(x, 0, 5 * 8);
    for (int i = 0; i < 5; ++i) {
        u128 v = (u128)(uint64_t)d[i];
        int bit = 52 * i, wi = bit >> 6, sh = bit & 63;
        x[wi] += (uint64_t)(v << sh);
        if (sh + 52 > 64) x[wi + 1] += (uint64_t)(v >> (64 - sh));
    }
    (void)acc; (void)bitpos;
}
// x mod p for x < 2^520 (schoolbook: shift-subtract), result 4 words
static void mod_p(Big x, uint64_t *r) {
    // reduce from the top bit down
    uint64_t rem[5] = {0, 0, 0, 0, 0};
    for (int bit = 9 * 64 - 1; bit >= 0; --bit) {
        // rem = rem * 2 + bit
        uint64_t c = (x.w[bit >> 6] >> (bit & 63)) & 1;
        for (int i = 0; i < 5; ++i) { uint64_t n = (rem[i] << 1) | c; c = rem[i] >> 63; rem[i] = n; }
        // if rem >= p: rem -= p
        bool ge = rem[4] != 0;
        if (!ge) { ge = true; for (int i = 3; i >= 0; --i) { if (rem[i] != FR_P[i]) { ge = rem[i] > FR_P[i]; break; } } }
        if (ge) { uint64_t b = 0; for (int i = 0; i < 4; ++i) { u128 t = (u128)rem[i] - FR_P[i] - b; rem[i] = (uint64_t)t; b = (uint64_t)(t >> 64) & 1; } rem[4] -= b; }
    }
    memcpy(r, rem, 32);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s), CUs=%d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    const int blocks = 256 * 8, threads = 256;
    uint64_t *out; CHECK(hipMalloc(&out, sizeof(uint64_t) * blocks * threads * 16));
    double ops = (double)ITERS * UNROLL * 4;
    time_kernel("v_fma_f64", ops, blocks, threads, k_fma_f64, (double *)out, 1.5, 0.999);
    time_kernel("v_add_f64", ops, blocks, threads, k_add_f64, (double *)out, 1.5, 0.999);
    time_kernel("v_lshl_add_u64", ops, blocks, threads, k_lshl_add_u64, out, (uint64_t)12345);

    // constants: p in 52-bit limbs, np = -p^-1 mod 2^52
    double p52[5]; to52(FR_P, p52);
    uint64_t p0 = FR_P[0] & ((1ull << 52) - 1), inv = 1;
    for (int i = 0; i < 6; ++i) inv = inv * (2 - p0 * inv);        // p0^-1 mod 2^64
    const double np = (double)((0 - inv) & ((1ull << 52) - 1));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_p52), p52, sizeof p52));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_np52), &np, sizeof np));
    // correctness: 64 products against big-integer arithmetic.  mul52(a, b) = a b 2^-260 mod p up to a multiple of p (< 2p + small):
    // check  result * 2^260 == a * b  (mod p)
    std::vector<f52_t> ha(64), hb(64), ho(64);
    std::vector<uint64_t> ia(64 * 4), ib(64 * 4);
    uint64_t st = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int t = 0; t < 64; ++t) {
        for (int w = 0; w < 4; ++w) { ia[t * 4 + w] = rnd(); ib[t * 4 + w] = rnd(); }
        ia[t * 4 + 3] &= (1ull << 61) - 1; ib[t * 4 + 3] &= (1ull << 61) - 1;      // < 2^253 < p
        to52(&ia[t * 4], ha[t].v); to52(&ib[t * 4], hb[t].v);
    }
    f52_t *da, *db, *dout;
    CHECK(hipMalloc(&da, 64 * sizeof(f52_t))); CHECK(hipMalloc(&db, 64 * sizeof(f52_t))); CHECK(hipMalloc(&dout, sizeof(f52_t) * blocks * threads));
    CHECK(hipMemcpy(da, ha.data(), 64 * sizeof(f52_t), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, hb.data(), 64 * sizeof(f52_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mul52_once, dim3(1), dim3(64), 0, 0, dout, (const f52_t *)da, (const f52_t *)db);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ho.data(), dout, 64 * sizeof(f52_t), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        uint64_t r5[5]; from52(ho[t].v, r5);
        uint64_t R260[5] = {0, 0, 0, 0, 1ull << 4};                 // 2^260
        Big lhs = big_mul(r5, 5, R260, 5);                          // result * 2^260   (< 2^(261 + 261))
        Big rhs = big_mul(&ia[t * 4], 4, &ib[t * 4], 4);
        uint64_t l[4], r[4]; mod_p(lhs, l); mod_p(rhs, r);
        if (memcmp(l, r, 32)) ++bad;
    }
    printf("mul52 correctness: %d of 64 products wrong\n", bad);
    for (int occ : {2, 4, 8}) {
        char nm[64];
        snprintf(nm, 64, "Fr52 mul (5x52 FP64 FMA) blk/CU=%d", occ);
        time_kernel(nm, FITERS, 256 * occ, 256, k_fmul52<1>, dout, (const f52_t *)da);
    }
    time_kernel("Fr52 mul x2 chains blk/CU=4", FITERS * 2, 256 * 4, 256, k_fmul52<2>, dout, (const f52_t *)da);
    std::vector<fe_t> h(64);
    for (int i = 0; i < 64; ++i) h[i] = Fr::from_u64(0x9e3779b97f4a7c15ull * (i + 3));
    fe_t *din; CHECK(hipMalloc(&din, 64 * sizeof(fe_t)));
    CHECK(hipMemcpy(din, h.data(), 64 * sizeof(fe_t), hipMemcpyHostToDevice));
    fe_t *fout = (fe_t *)out;
#if defined(SRS_F29_CHAIN)
    const char *tag = "[carry-chained asm]";
#else
    const char *tag = "[hipcc schedule]";
#endif
    for (int occ : {3, 4, 8}) {
        char nm[80];
        snprintf(nm, 80, "Fr29::mul %s blk/CU=%d", tag, occ);
        time_kernel(nm, FITERS, 256 * occ, 256, k_fmul29<Fr29, 1, 0>, fout, (const fe_t *)din);
        snprintf(nm, 80, "Fr29::sqr %s blk/CU=%d", tag, occ);
        time_kernel(nm, FITERS, 256 * occ, 256, k_fmul29<Fr29, 1, 1>, fout, (const fe_t *)din);
        snprintf(nm, 80, "Fr29::mul2 %s blk/CU=%d", tag, occ);
        time_kernel(nm, FITERS, 256 * occ, 256, k_fmul29<Fr29, 1, 2>, fout, (const fe_t *)din);
    }
    {
        char nm[80];
        snprintf(nm, 80, "Fr29::mul x2 chains %s blk/CU=3", tag);
        time_kernel(nm, FITERS * 2, 256 * 3, 256, k_fmul29<Fr29, 2, 0>, fout, (const fe_t *)din);
    }
    printf("done\n");
    return bad ? 2 : 0;
}
