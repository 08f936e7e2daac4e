// field29_check.cpp -- TEST INFRASTRUCTURE (host build of field29.cuh / curve29.cuh through the emulator headers).
// A line-oriented calculator: tests/test_field29_host.py feeds operands chosen at the stated bounds of the lazy 9 x 29-bit
// arithmetic and checks every result with Python integers.
//   mul F a[9] b[9] | sqr F a[9] | norm F a[9] | add F a[9] b[9] | sub F CP E a[9] b[9] | neg F CP E b[9] | canon F a[9]
//   unpack F w[8] | pack F a[9]
//   mul2 F a[9] b[9] c[9] d[9]   : (a b + c d) / 2^261 with one reduction
//   chain C n (w[16] neg){n}     : identity + n mixed additions of table-form points, lazy in between -> packed XYZZ (32 words)
//   chains C n (w[16] neg){n}    : the same through madd_signed / load_raw (what k_accum0 runs)
//   addp C a[32] b[32]           : Ec29::add of two packed partial sums -> packed
//   dblp C a[32]                 : Ec29::dbl -> packed
//   tform C w[16]                : table_form of an ABI affine point -> 16 words
//   xyzz C a[32]                 : packed R'-form -> ABI XYZZ (to_xyzz)
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

#include "hipemu.h"
#include "curve29.cuh"

using namespace srs;

template <class P>
static f29_t rd9(std::istringstream &in) {
    f29_t a;
    for (int i = 0; i < 9; ++i) { unsigned long long v; in >> std::hex >> v; a.v[i] = (uint32_t)v; }
    return a;
}
static fe_t rd8(std::istringstream &in) {
    fe_t a;
    for (int i = 0; i < 8; ++i) { unsigned long long v; in >> std::hex >> v; a.v[i] = (uint32_t)v; }
    return a;
}
static void pr(const uint32_t *v, int n) {
    for (int i = 0; i < n; ++i) std::printf("%x%c", v[i], i + 1 < n ? ' ' : '\n');
}

template <class P, uint32_t CP, uint32_t E>
static bool try_sub(uint32_t cp, uint32_t e, bool neg, const f29_t &a, const f29_t &b) {
    if (cp != CP || e != E) return false;
    f29_t o = neg ? Fp29<P>::template neg_lazy<CP, E>(b) : Fp29<P>::template sub_lazy<CP, E>(a, b);
    pr(o.v, 9);
    return true;
}

template <class P>
static void field_op(const std::string &op, std::istringstream &in) {
    using G = Fp29<P>;
    if (op == "mul") { f29_t a = rd9<P>(in), b = rd9<P>(in); pr(G::mul(a, b).v, 9); }
    else if (op == "mul2") { f29_t a = rd9<P>(in), b = rd9<P>(in), c = rd9<P>(in), d = rd9<P>(in); pr(G::mul2(a, b, c, d).v, 9); }
    else if (op == "sqr") { f29_t a = rd9<P>(in); pr(G::sqr(a).v, 9); }
    else if (op == "norm") { f29_t a = rd9<P>(in); pr(G::normalize(a).v, 9); }
    else if (op == "add") { f29_t a = rd9<P>(in), b = rd9<P>(in); pr(G::add_lazy(a, b).v, 9); }
    else if (op == "canon") { f29_t a = rd9<P>(in); pr(G::to_canonical_fe(a).v, 8); }
    else if (op == "redlazy") { f29_t a = rd9<P>(in); pr(G::reduce_lazy(a).v, 9); }
    else if (op == "unpack") { fe_t a = rd8(in); pr(G::unpack(a).v, 9); }
    else if (op == "pack") { f29_t a = rd9<P>(in); pr(G::pack(a).v, 8); }
    else if (op == "sub" || op == "neg") {
        unsigned cp, e;
        in >> std::dec >> cp >> e;
        const bool neg = op == "neg";
        f29_t a = neg ? G::zero() : rd9<P>(in), b = rd9<P>(in);
        bool ok = try_sub<P, 1, 0>(cp, e, neg, a, b) || try_sub<P, 2, 0>(cp, e, neg, a, b) || try_sub<P, 3, 0>(cp, e, neg, a, b) ||
                  try_sub<P, 5, 1>(cp, e, neg, a, b) || try_sub<P, 6, 2>(cp, e, neg, a, b) || try_sub<P, 7, 2>(cp, e, neg, a, b) ||
                  try_sub<P, 8, 0>(cp, e, neg, a, b) || try_sub<P, 10, 0>(cp, e, neg, a, b) || try_sub<P, 13, 0>(cp, e, neg, a, b) ||
                  try_sub<P, 31, 0>(cp, e, neg, a, b) || try_sub<P, 3, 2>(cp, e, neg, a, b) || try_sub<P, 12, 2>(cp, e, neg, a, b) ||
                  try_sub<P, 6, 0>(cp, e, neg, a, b) || try_sub<P, 8, 1>(cp, e, neg, a, b);
        if (!ok) std::printf("unsupported\n");
    } else std::printf("unsupported\n");
}

template <class C>
static void curve_op(const std::string &op, std::istringstream &in) {
    using E = Ec29<C>;
    auto rdp = [&](xyzz_t &p) { p.x = rd8(in); p.y = rd8(in); p.zz = rd8(in); p.zzz = rd8(in); };
    auto prp = [&](const xyzz_t &p) { uint32_t w[32]; std::memcpy(w, &p, sizeof p); pr(w, 32); };
    if (op == "chain" || op == "chains") {
        unsigned n;
        in >> std::dec >> n;
        xyzz29_t acc = E::identity();
        for (unsigned i = 0; i < n; ++i) {
            affine_t q;
            q.x = rd8(in);
            q.y = rd8(in);
            unsigned neg;
            in >> std::dec >> neg;
            acc = op == "chain" ? E::madd(acc, E::load(q, neg != 0)) : E::madd_signed(acc, E::load_raw(q), neg != 0);
        }
        prp(E::pack(acc));
    } else if (op == "addp") {
        xyzz_t a, b;
        rdp(a); rdp(b);
        prp(E::pack(E::add(E::unpack(a), E::unpack(b))));
    } else if (op == "dblp") {
        xyzz_t a;
        rdp(a);
        prp(E::pack(E::dbl(E::unpack(a))));
    } else if (op == "tform") {
        affine_t q;
        q.x = rd8(in);
        q.y = rd8(in);
        affine_t t = E::table_form(q);
        uint32_t w[16];
        std::memcpy(w, &t, sizeof t);
        pr(w, 16);
    } else if (op == "xyzz") {
        xyzz_t a;
        rdp(a);
        prp(E::to_xyzz(E::unpack(a)));
    } else std::printf("unsupported\n");
}

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream in(line);
        std::string op, which;
        in >> op >> which;
        if (which == "Fr") field_op<FrP>(op, in);
        else if (which == "Fq") field_op<FqP>(op, in);
        else if (which == "Bn256") curve_op<Bn256>(op, in);
        else if (which == "Grumpkin") curve_op<Grumpkin>(op, in);
        else std::printf("unsupported\n");
        std::fflush(stdout);
    }
    return 0;
}
