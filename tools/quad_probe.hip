// latency probe: chain of dependent XYZZ additions, one lane per addition vs one quad per addition
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../sirius_amd/csrc/curve.cuh"
using namespace srs;
__device__ __forceinline__ fe_t qsel(uint32_t q, fe_t a0, fe_t a1, fe_t a2, fe_t a3) {
    fe_t o;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint32_t x = a0.v[i]; x = q == 1 ? a1.v[i] : x; x = q == 2 ? a2.v[i] : x; x = q == 3 ? a3.v[i] : x; o.v[i] = x; }
    return o;
}
template <class C>
__device__ __forceinline__ xyzz_t add_quad_core(const xyzz_t &a, const xyzz_t &b, uint32_t q) {
    using F = typename C::F;
    fe_t m1 = F::mul(qsel(q, a.x, b.x, a.y, b.y), qsel(q, b.zz, a.zz, b.zzz, a.zzz));
    fe_t u1 = quad_bcast<0>(m1), u2 = quad_bcast<1>(m1), s1 = quad_bcast<2>(m1), s2 = quad_bcast<3>(m1);
    fe_t p = F::sub(u2, u1), r = F::sub(s2, s1);
    fe_t m2 = F::mul(qsel(q, p, r, a.zz, a.zzz), qsel(q, p, r, b.zz, b.zzz));
    fe_t pp = quad_bcast<0>(m2), rr = quad_bcast<1>(m2), zzz12 = quad_bcast<3>(m2);
    fe_t m3 = F::mul(qsel(q, p, u1, m2, p), pp);
    fe_t ppp = quad_bcast<0>(m3), qv = quad_bcast<1>(m3), zz3 = quad_bcast<2>(m3);
    fe_t x3 = F::sub(F::sub(rr, ppp), F::dbl(qv));
    fe_t m4 = F::mul(qsel(q, r, s1, zzz12, s1), qsel(q, F::sub(qv, x3), ppp, ppp, ppp));
    fe_t t2 = quad_bcast<0>(m4), t1 = quad_bcast<1>(m4), zzz3 = quad_bcast<2>(m4);
    xyzz_t o;
    o.x = x3; o.y = F::sub(t2, t1); o.zz = zz3; o.zzz = zzz3;
    return o;
}
__global__ void __launch_bounds__(64, 1) k_quad(const xyzz_t *in, xyzz_t *out, int n) {
    uint32_t t = threadIdx.x >> 2, q = threadIdx.x & 3u;
    xyzz_t acc = in[t];
    for (int j = 1; j < n; ++j) acc = add_quad_core<Bn256>(acc, in[16 + j], q);
    if (q == 0) out[t] = acc;
}
__global__ void __launch_bounds__(64, 1) k_lane(const xyzz_t *in, xyzz_t *out, int n) {
    uint32_t t = threadIdx.x;
    xyzz_t acc = in[t & 15];
    for (int j = 1; j < n; ++j) acc = Ec<Bn256>::add(acc, in[16 + j]);
    out[t] = acc;
}
int main() {
    const int n = 256;
    std::vector<xyzz_t> h(16 + n);
    // distinct multiples of the generator (1, 2) in XYZZ with zz = zzz = 1: build by repeated host additions
    affine_t g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one());
    xyzz_t cur = Ec<Bn256>::from_affine(g);
    for (size_t i = 0; i < h.size(); ++i) { h[i] = cur; cur = Ec<Bn256>::add(Ec<Bn256>::dbl(cur), Ec<Bn256>::from_affine(g)); }
    xyzz_t *d_in, *d_out;
    hipMalloc(&d_in, h.size() * sizeof(xyzz_t)); hipMalloc(&d_out, 64 * sizeof(xyzz_t));
    hipMemcpy(d_in, h.data(), h.size() * sizeof(xyzz_t), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<xyzz_t> oq(64), ol(64);
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k_quad, dim3(1), dim3(64), 0, 0, (const xyzz_t *)d_in, d_out, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(oq.data(), d_out, 64 * sizeof(xyzz_t), hipMemcpyDeviceToHost);
        printf("quad : %8.3f us per add (%d dependent adds)\n", ms * 1e3 / (n - 1), n - 1);
        hipEventRecord(e0); hipLaunchKernelGGL(k_lane, dim3(1), dim3(64), 0, 0, (const xyzz_t *)d_in, d_out, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(ol.data(), d_out, 64 * sizeof(xyzz_t), hipMemcpyDeviceToHost);
        printf("lane : %8.3f us per add\n", ms * 1e3 / (n - 1));
    }
    int same = 1;
    for (int t = 0; t < 16; ++t) same &= memcmp(&oq[t], &ol[t], sizeof(xyzz_t)) == 0;
    printf("results identical: %d\n", same);
    return 0;
}
