// ubench_batch_affine.hip -- gfx950 (r04, VERDICT r03 item 2 ii): batched-affine bucket additions against the XYZZ mixed addition.
//
// An affine addition with the slope's denominator inverted by Montgomery's trick costs 5 products + 1 square (3 for the shared
// inversion, lambda = dy / dx, x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1) against 8 + 2 (minus one shared reduction) for
// the XYZZ mixed addition of k_accum0 -- IF the one real inversion per batch is amortised.  This file measures the three numbers
// that decide it: (1) what an inversion costs (Fermat on both limb forms, a binary extended GCD), per lane and as the issue slots of a
// wavefront with ONE active lane; (2) the additions per second of a workgroup-batched affine kernel -- 1024 threads x B additions
// each, prefix products in registers, the workgroup's product tree through LDS, one GCD inversion by one lane per round -- fed from an
// L2-resident table like k_madd29 in ubench29.hip (no HBM in the way: an UPPER bound); (3) the XYZZ chain under the same feed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sirius_amd/csrc tools/ubench_batch_affine.hip -o tools/ubench_batch_affine
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "curve29.cuh"
using namespace srs;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

using P = FqP;                 // bn256 base field (coordinates of G1)
using F29 = Fp29<P>;
using F32 = Fp<P>;

// ---- inversions ---------------------------------------------------------------------------------------------------------------
// a^(p-2) on the 9 x 29-bit form: 253 squares + (popcount - 1) products
__device__ f29_t inv_fermat29(const f29_t &a) {
    uint32_t e[8];
    for (int i = 0; i < 8; ++i) e[i] = P::p(i);
    e[0] -= 2;                                     // p - 2 (p is odd and p(0) >= 2)
    f29_t r = a;
    bool started = false;
    for (int bit = 255; bit >= 0; --bit) {
        const uint32_t b = (e[bit >> 5] >> (bit & 31)) & 1u;
        if (started) {
            r = F29::sqr(r);
            if (b) r = F29::mul(r, a);
        } else if (b) {
            started = true;
        }
    }
    return r;
}
// binary extended GCD on 8 x 32-bit words: x^-1 mod p for 0 < x < p (plain integers)
struct u256 { uint32_t w[8]; };
__device__ __forceinline__ bool is_even(const u256 &a) { return (a.w[0] & 1u) == 0; }
__device__ __forceinline__ bool is_one(const u256 &a) { uint32_t t = a.w[0] ^ 1u; for (int i = 1; i < 8; ++i) t |= a.w[i]; return t == 0; }
__device__ __forceinline__ bool geq(const u256 &a, const u256 &b) {
    for (int i = 7; i >= 0; --i) if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    return true;
}
__device__ __forceinline__ uint32_t sub_(u256 &a, const u256 &b) {
    uint32_t br = 0;
    for (int i = 0; i < 8; ++i) { uint64_t t = (uint64_t)a.w[i] - b.w[i] - br; a.w[i] = (uint32_t)t; br = (uint32_t)(t >> 32) & 1u; }
    return br;
}
__device__ __forceinline__ uint32_t add_(u256 &a, const u256 &b) {
    uint32_t c = 0;
    for (int i = 0; i < 8; ++i) { uint64_t t = (uint64_t)a.w[i] + b.w[i] + c; a.w[i] = (uint32_t)t; c = (uint32_t)(t >> 32); }
    return c;
}
__device__ __forceinline__ void shr1(u256 &a, uint32_t top) {
    for (int i = 0; i < 7; ++i) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 31);
    a.w[7] = (a.w[7] >> 1) | (top << 31);
}
__device__ u256 inv_gcd(const u256 &x) {
    u256 pm; for (int i = 0; i < 8; ++i) pm.w[i] = P::p(i);
    u256 u = x, v = pm, x1, x2;
    for (int i = 0; i < 8; ++i) { x1.w[i] = 0; x2.w[i] = 0; }
    x1.w[0] = 1;
    while (!is_one(u) && !is_one(v)) {
        while (is_even(u)) {
            shr1(u, 0);
            if (is_even(x1)) shr1(x1, 0); else { uint32_t c = add_(x1, pm); shr1(x1, c); }
        }
        while (is_even(v)) {
            shr1(v, 0);
            if (is_even(x2)) shr1(x2, 0); else { uint32_t c = add_(x2, pm); shr1(x2, c); }
        }
        if (geq(u, v)) { sub_(u, v); if (sub_(x1, x2)) add_(x1, pm); }
        else { sub_(v, u); if (sub_(x2, x1)) add_(x2, pm); }
    }
    return is_one(u) ? x1 : x2;
}
// Montgomery-form inverse on the 29-bit form through the GCD: a R' -> a^-1 R'   (R' = 2^261): (a R')^-1 * R'^2 = a^-1 R'
__constant__ uint32_t c_r3[9];          // R'^3 mod p in 29-bit limbs: mul(x, R'^3) = x R'^2
__device__ f29_t inv_gcd29(const f29_t &a) {
    fe_t c = F29::to_canonical_fe(a);              // < p, plain integer value a R' mod p
    u256 x; for (int i = 0; i < 8; ++i) x.w[i] = c.v[i];
    u256 y = inv_gcd(x);
    fe_t yy; for (int i = 0; i < 8; ++i) yy.v[i] = y.w[i];
    f29_t r3; for (int i = 0; i < 9; ++i) r3.v[i] = c_r3[i];
    return F29::mul(F29::unpack(yy), r3);
}

__global__ void k_inv(fe_t *out, const fe_t *in, int mode, int active_lanes) {
    if ((int)(threadIdx.x & 63) >= active_lanes) return;
    f29_t a = F29::unpack(in[(threadIdx.x + blockIdx.x) & 1023]);
    f29_t r;
    if (mode == 0) r = inv_fermat29(a);
    else if (mode == 1) r = inv_gcd29(a);
    else { fe_t x = in[(threadIdx.x + blockIdx.x) & 1023]; out[blockIdx.x * blockDim.x + threadIdx.x] = F32::inv(x); return; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = F29::to_canonical_fe(r);
}
__global__ void k_inv_check(const fe_t *in, uint32_t *bad) {          // a * inv(a) == 1 for both routes
    f29_t a = F29::unpack(in[threadIdx.x]);
    f29_t one = Ec29<Bn256>::one();
    fe_t want = F29::to_canonical_fe(one);
    fe_t g1 = F29::to_canonical_fe(F29::mul(a, inv_fermat29(a))), g2 = F29::to_canonical_fe(F29::mul(a, inv_gcd29(a)));
    for (int i = 0; i < 8; ++i) if (g1.v[i] != want.v[i] || g2.v[i] != want.v[i]) { atomicAdd(bad, 1u); break; }
}

// ---- batched affine additions ----------------------------------------------------------------------------------------------------
// thread: B independent additions P_i + Q_i (table points, distinct x: the P = +-Q cases would take the fallback of the real kernel).
// forward: d_i = xQ - xP, prefix products; workgroup product tree in LDS; ONE inversion (thread 0); back down; backward: the B slopes.
template <int B, int THREADS>
__global__ void __launch_bounds__(THREADS) k_affine_batch(affine_t *out, const affine_t *tab, int rounds) {
    __shared__ uint32_t tree[2 * THREADS][9];          // node 1 = root, leaves THREADS .. 2 THREADS - 1
    const uint32_t t = threadIdx.x;
    affine_t acc_out;
    for (int i = 0; i < 8; ++i) { acc_out.x.v[i] = 0; acc_out.y.v[i] = 0; }
    for (int r = 0; r < rounds; ++r) {
        f29_t pre[B];
        f29_t run = Ec29<Bn256>::one();
        uint32_t ia[B], ib[B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            ia[i] = (t * 7 + i * 13 + r * 29 + blockIdx.x) & 1023;
            ib[i] = (ia[i] + 1 + ((t + i + r) & 511)) & 1023;                      // never the same point
            const f29_t xa = F29::unpack(tab[ia[i]].x), xb = F29::unpack(tab[ib[i]].x);
            const f29_t d = F29::normalize(F29::template sub_lazy<2, 0>(xb, xa));   // xb - xa + 2p
            pre[i] = run;
            run = F29::mul(run, d);
        }
        // up the tree
#pragma unroll
        for (int l = 0; l < 9; ++l) tree[THREADS + t][l] = run.v[l];
        __syncthreads();
        for (uint32_t width = THREADS / 2; width >= 1; width >>= 1) {
            if (t < width) {
                f29_t a, b;
                for (int l = 0; l < 9; ++l) { a.v[l] = tree[2 * (width + t)][l]; b.v[l] = tree[2 * (width + t) + 1][l]; }
                const f29_t c = F29::mul(a, b);
                for (int l = 0; l < 9; ++l) tree[width + t][l] = c.v[l];
            }
            __syncthreads();
        }
        if (t == 0) {
            f29_t root; for (int l = 0; l < 9; ++l) root.v[l] = tree[1][l];
            const f29_t inv = inv_gcd29(root);
            for (int l = 0; l < 9; ++l) tree[1][l] = inv.v[l];
        }
        __syncthreads();
        // down: node n holds inv(product under n); children: inv(left) = inv(n) * right, inv(right) = inv(n) * left
        for (uint32_t width = 1; width < THREADS; width <<= 1) {
            if (t < width) {
                f29_t inv_n, le, ri;
                for (int l = 0; l < 9; ++l) { inv_n.v[l] = tree[width + t][l]; le.v[l] = tree[2 * (width + t)][l]; ri.v[l] = tree[2 * (width + t) + 1][l]; }
                const f29_t il = F29::mul(inv_n, ri), ir = F29::mul(inv_n, le);
                for (int l = 0; l < 9; ++l) { tree[2 * (width + t)][l] = il.v[l]; tree[2 * (width + t) + 1][l] = ir.v[l]; }
            }
            __syncthreads();
        }
        f29_t inv_run; for (int l = 0; l < 9; ++l) inv_run.v[l] = tree[THREADS + t][l];
        __syncthreads();
        // backward: the slopes and the sums (the operands are gathered again: they do not fit registers next to the prefixes)
#pragma unroll
        for (int i = B - 1; i >= 0; --i) {
            const aff29_t a = Ec29<Bn256>::load_raw(tab[ia[i]]), b = Ec29<Bn256>::load_raw(tab[ib[i]]);
            const f29_t d = F29::normalize(F29::template sub_lazy<2, 0>(b.x, a.x));
            const f29_t inv_d = F29::mul(inv_run, pre[i]);
            inv_run = F29::mul(inv_run, d);
            const f29_t dy = F29::normalize(F29::template sub_lazy<2, 0>(b.y, a.y));
            const f29_t lam = F29::mul(dy, inv_d);
            const f29_t l2 = F29::sqr(lam);
            const f29_t x3 = F29::normalize(F29::template sub_lazy<4, 1>(l2, F29::add_lazy(a.x, b.x)));     // l2 - xa - xb + 4p
            const f29_t dx = F29::normalize(F29::template sub_lazy<8, 0>(a.x, x3));                          // xa - x3 + 8p
            const f29_t y3 = F29::template sub_lazy<2, 0>(F29::mul(lam, dx), a.y);
            // fold the result into the output so that nothing is dead code
            const fe_t cx = F29::to_canonical_fe(F29::mul(x3, Ec29<Bn256>::one())), cy = F29::to_canonical_fe(F29::mul(F29::normalize(y3), Ec29<Bn256>::one()));
            for (int w = 0; w < 8; ++w) { acc_out.x.v[w] ^= cx.v[w]; acc_out.y.v[w] ^= cy.v[w]; }
        }
    }
    out[blockIdx.x * blockDim.x + t] = acc_out;
}
// the same without the two canonicalisations per sum (a real kernel stores the lazy limbs): ALU upper bound of the addition itself
template <int B, int THREADS>
__global__ void __launch_bounds__(THREADS) k_affine_core(uint32_t *out, const affine_t *tab, int rounds) {
    const uint32_t t = threadIdx.x;
    uint32_t x = 0;
    f29_t inv_run = F29::unpack(tab[t & 1023].x);          // stands for the inverse handed down the tree
    for (int r = 0; r < rounds; ++r) {
        f29_t pre[B];
        f29_t run = Ec29<Bn256>::one();
        uint32_t ia[B], ib[B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            ia[i] = (t * 7 + i * 13 + r * 29 + blockIdx.x) & 1023;
            ib[i] = (ia[i] + 1 + ((t + i + r) & 511)) & 1023;
            const f29_t xa = F29::unpack(tab[ia[i]].x), xb = F29::unpack(tab[ib[i]].x);
            const f29_t d = F29::normalize(F29::template sub_lazy<2, 0>(xb, xa));
            pre[i] = run;
            run = F29::mul(run, d);
        }
        inv_run = F29::mul(inv_run, run);                  // (keeps `run` alive)
#pragma unroll
        for (int i = B - 1; i >= 0; --i) {
            const aff29_t a = Ec29<Bn256>::load_raw(tab[ia[i]]), b = Ec29<Bn256>::load_raw(tab[ib[i]]);
            const f29_t d = F29::normalize(F29::template sub_lazy<2, 0>(b.x, a.x));
            const f29_t inv_d = F29::mul(inv_run, pre[i]);
            inv_run = F29::mul(inv_run, d);
            const f29_t dy = F29::normalize(F29::template sub_lazy<2, 0>(b.y, a.y));
            const f29_t lam = F29::mul(dy, inv_d);
            const f29_t l2 = F29::sqr(lam);
            const f29_t x3 = F29::normalize(F29::template sub_lazy<4, 1>(l2, F29::add_lazy(a.x, b.x)));
            const f29_t dx = F29::normalize(F29::template sub_lazy<8, 0>(a.x, x3));
            const f29_t y3 = F29::template sub_lazy<2, 0>(F29::mul(lam, dx), a.y);
            for (int l = 0; l < 9; ++l) x ^= x3.v[l] ^ y3.v[l];
        }
    }
    out[blockIdx.x * blockDim.x + t] = x;
}
// XYZZ chain under the same feed (what k_accum0 does)
__global__ void k_madd_chain(xyzz_t *out, const affine_t *tab, int n) {
    xyzz29_t acc = Ec29<Bn256>::identity();
    for (int it = 0; it < n; ++it) {
        const uint32_t idx = (threadIdx.x * 7 + it * 13 + blockIdx.x) & 1023;
        acc = Ec29<Bn256>::madd_signed(acc, Ec29<Bn256>::load_raw(tab[idx]), (it & 1) != 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = Ec29<Bn256>::pack(acc);
}
__global__ void k_fill(affine_t *tab) {
    uint32_t k[8] = {threadIdx.x + blockIdx.x * blockDim.x + 1, 0x9e3779b9u, 0x7f4a7c15u, 0x1234567u, 0, 0, 0, 0};
    affine_t g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one());
    tab[threadIdx.x + blockIdx.x * blockDim.x] = Ec29<Bn256>::table_form(EcBn::to_affine(EcBn::mul_canon(k, g)));
}
// batched sum == XYZZ sum for a sample of pairs (the same index rule as k_affine_batch round 0, i = 0)
__global__ void k_affine_verify(const affine_t *tab, uint32_t *bad) {
    const uint32_t t = threadIdx.x;
    const uint32_t ia = (t * 7 + blockIdx.x) & 1023, ib = (ia + 1 + (t & 511)) & 1023;
    const aff29_t a = Ec29<Bn256>::load_raw(tab[ia]), b = Ec29<Bn256>::load_raw(tab[ib]);
    const f29_t d = F29::normalize(F29::template sub_lazy<2, 0>(b.x, a.x));
    const f29_t inv_d = inv_gcd29(d);
    const f29_t dy = F29::normalize(F29::template sub_lazy<2, 0>(b.y, a.y));
    const f29_t lam = F29::mul(dy, inv_d);
    const f29_t x3 = F29::normalize(F29::template sub_lazy<4, 1>(F29::sqr(lam), F29::add_lazy(a.x, b.x)));
    const f29_t y3 = F29::template sub_lazy<2, 0>(F29::mul(lam, F29::normalize(F29::template sub_lazy<8, 0>(a.x, x3))), a.y);
    const fe_t cx = F29::to_canonical_fe(F29::mul(x3, Ec29<Bn256>::one())), cy = F29::to_canonical_fe(F29::mul(F29::normalize(y3), Ec29<Bn256>::one()));
    // reference: XYZZ, then affine through the 8 x 32 code (table form -> ABI form and back is not needed: compare x3 * zz == X etc.)
    xyzz29_t s = Ec29<Bn256>::madd_signed(Ec29<Bn256>::madd_signed(Ec29<Bn256>::identity(), a, false), b, false);
    // X = x3 * ZZ, Y = y3 * ZZZ in the field
    const fe_t lx = F29::to_canonical_fe(F29::mul(F29::mul(F29::unpack(cx), s.zz), Ec29<Bn256>::one()));
    const fe_t rx = F29::to_canonical_fe(F29::mul(s.x, Ec29<Bn256>::one()));
    const fe_t ly = F29::to_canonical_fe(F29::mul(F29::mul(F29::unpack(cy), s.zzz), Ec29<Bn256>::one()));
    const fe_t ry = F29::to_canonical_fe(F29::mul(F29::normalize(s.y), Ec29<Bn256>::one()));
    for (int i = 0; i < 8; ++i) if (lx.v[i] != rx.v[i] || ly.v[i] != ry.v[i]) { atomicAdd(bad, 1u); break; }
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
    printf("%-58s %9.3f ms  %10.4f Gops/s  (blocks=%d thr=%d)\n", name, ms, rate * 1e-9, blocks, threads);
    fflush(stdout);
    return rate;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s), CUs=%d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    // R'^3 mod p (R' = 2^261) in 29-bit limbs, by repeated doubling on the host
    {
        typedef unsigned __int128 u128;
        uint64_t p[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
        uint64_t x[4] = {1, 0, 0, 0};
        for (int i = 0; i < 3 * 261; ++i) {
            uint64_t c = 0;
            for (int j = 0; j < 4; ++j) { uint64_t n = (x[j] << 1) | c; c = x[j] >> 63; x[j] = n; }
            bool ge = c != 0;
            if (!ge) { ge = true; for (int j = 3; j >= 0; --j) if (x[j] != p[j]) { ge = x[j] > p[j]; break; } }
            if (ge) { uint64_t b = 0; for (int j = 0; j < 4; ++j) { u128 t = (u128)x[j] - p[j] - b; x[j] = (uint64_t)t; b = (uint64_t)(t >> 64) & 1; } }
        }
        uint32_t w[8]; for (int j = 0; j < 4; ++j) { w[2 * j] = (uint32_t)x[j]; w[2 * j + 1] = (uint32_t)(x[j] >> 32); }
        uint32_t l[9];
        for (int i = 0; i < 9; ++i) {
            int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
            uint64_t two = (uint64_t)w[wi] | (wi + 1 < 8 ? (uint64_t)w[wi + 1] << 32 : 0);
            l[i] = (uint32_t)(two >> sh) & (i < 8 ? 0x1fffffffu : 0xffffffffu);
        }
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_r3), l, sizeof l));
    }
    affine_t *tab; CHECK(hipMalloc(&tab, 1024 * sizeof(affine_t)));
    hipLaunchKernelGGL(k_fill, dim3(16), dim3(64), 0, 0, tab);
    CHECK(hipDeviceSynchronize());
    uint32_t *bad; CHECK(hipMalloc(&bad, 8)); CHECK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_inv_check, dim3(1), dim3(64), 0, 0, (const fe_t *)tab, bad);
    hipLaunchKernelGGL(k_affine_verify, dim3(4), dim3(64), 0, 0, (const affine_t *)tab, bad + 1);
    CHECK(hipDeviceSynchronize());
    uint32_t hb[2]; CHECK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
    printf("inverse check: %u of 64 wrong;  affine sum vs XYZZ sum: %u of 256 differ\n", hb[0], hb[1]);
    fe_t *fout; CHECK(hipMalloc(&fout, sizeof(fe_t) * 2048 * 1024));
    const fe_t *fin = (const fe_t *)tab;
    // (1) inversions: all 64 lanes active = cost per lane; 1 lane active = the same issue slots for ONE inversion
    time_kernel("inv Fermat 9x29, 64 lanes, 4 blk/CU", 1, 1024, 256, k_inv, fout, fin, 0, 64);
    time_kernel("inv GCD (8x32 words), 64 lanes, 4 blk/CU", 1, 1024, 256, k_inv, fout, fin, 1, 64);
    time_kernel("inv Fermat 8x32 (Fp::inv), 64 lanes, 4 blk/CU", 1, 1024, 256, k_inv, fout, fin, 2, 64);
    time_kernel("inv Fermat 9x29, ONE lane per wave, 4 blk/CU (x64 = slots)", 1.0 / 64, 1024, 256, k_inv, fout, fin, 0, 1);
    time_kernel("inv GCD, ONE lane per wave, 4 blk/CU", 1.0 / 64, 1024, 256, k_inv, fout, fin, 1, 1);
    time_kernel("inv Fermat 9x29 latency (1 wave, 1 lane)", 1.0 / 64, 1, 64, k_inv, fout, fin, 0, 1);
    time_kernel("inv GCD latency (1 wave, 1 lane)", 1.0 / 64, 1, 64, k_inv, fout, fin, 1, 1);
    // (2) batched affine: 1024 threads x B, `rounds` rounds
    affine_t *aout = (affine_t *)fout;
    time_kernel("affine batch B=8  x 1024 thr, 1 blk/CU, 8 rounds", 8.0 * 8, 256, 1024, k_affine_batch<8, 1024>, aout, (const affine_t *)tab, 8);
    time_kernel("affine batch B=8  x 1024 thr, 2 blk/CU, 8 rounds", 8.0 * 8, 512, 1024, k_affine_batch<8, 1024>, aout, (const affine_t *)tab, 8);
    time_kernel("affine batch B=4  x 1024 thr, 2 blk/CU, 8 rounds", 4.0 * 8, 512, 1024, k_affine_batch<4, 1024>, aout, (const affine_t *)tab, 8);
    time_kernel("affine batch B=8  x 256 thr, 4 blk/CU, 8 rounds", 8.0 * 8, 1024, 256, k_affine_batch<8, 256>, aout, (const affine_t *)tab, 8);
    time_kernel("affine core only (no tree / inversion) B=8, 256 thr, 4 blk/CU", 8.0 * 8, 1024, 256, k_affine_core<8, 256>, (uint32_t *)fout, (const affine_t *)tab, 8);
    time_kernel("affine core only B=8, 256 thr, 8 blk/CU", 8.0 * 8, 2048, 256, k_affine_core<8, 256>, (uint32_t *)fout, (const affine_t *)tab, 8);
    // (3) the XYZZ chain
    xyzz_t *pout; CHECK(hipMalloc(&pout, sizeof(xyzz_t) * 2048 * 256));
    time_kernel("XYZZ madd_signed chain of 64, 256 thr, 4 blk/CU", 64, 1024, 256, k_madd_chain, pout, (const affine_t *)tab, 64);
    time_kernel("XYZZ madd_signed chain of 64, 256 thr, 8 blk/CU", 64, 2048, 256, k_madd_chain, pout, (const affine_t *)tab, 64);
    printf("done\n");
    return (hb[0] || hb[1]) ? 2 : 0;
}
