// rowprog.hip -- per-row polynomial work of the Sangria NIFS on gfx950:
//   * cross terms  T_k[row], k = 1..d   (reference VanillaFS::commit_cross_terms, src/nifs/sangria/mod.rs:102-158)
//   * plain gate evaluation per row       (deciders, src/nifs/sangria/mod.rs:334-383, src/plonk/mod.rs:304-361)
//   * witness / error-vector folds        (RelaxedPlonkWitness::fold, src/nifs/sangria/accumulator.rs:364-404)
//
// What the reference computes: with P the compressed + homogenised gate polynomial of the structure
// (src/plonk/util.rs:34-56, src/polynomial/expression.rs:356-429, src/plonk/mod.rs:68-121),
//   T_k[row] = coefficient of X^k in  P(fixed[row], W1[row] + X*W2[row], ch1 + X*ch2)
// obtained symbolically (GroupedPoly::new, src/polynomial/grouped_poly.rs:88-138) and evaluated by a
// per-row interpreter, one pass over all columns PER TERM (src/polynomial/graph_evaluator.rs:361-388).
//
// MI355X-first formulation: the coefficients are mathematically determined, and field arithmetic
// is exact, so they are recovered numerically instead -- evaluate P at the d+1 points X = 0..d and
// apply the (constant) inverse Vandermonde matrix.  One pass over the rows yields ALL d terms;
// the program is the small homogeneous expression (no symbolic blow-up), compiled on the host to an
// SSA register program with CSE; sub-expressions that do not depend on the row (constants,
// challenges, powers of u) are evaluated once per point on the host ("uniform table").
// Row registers live in LDS as [slot][thread] (conflict-free 32-byte lanes); the d accumulators
// T_k live in VGPRs.  Columns are read coalesced (column-major, consecutive rows per lane).
#include "rowprog.h"
#include "prof.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <tuple>

namespace srs {
namespace rowprog {

// ---------------------------------------------------------------------------------------------
// device program
// ---------------------------------------------------------------------------------------------
enum : uint32_t { I_LD_SEL = 0, I_LD_FIX, I_LD_ADV, I_ADD, I_SUB, I_MUL, I_SQR, I_DBL, I_NEG };
constexpr uint32_t UNIFORM_BIT = 0x80000000u;
constexpr uint32_t RP_THREADS = 128;
constexpr uint32_t DMAX = 8;   // cross terms kept in VGPRs; larger degrees are rejected at create time

struct DevArgs {
    const Insn *prog;
    uint32_t n_insn;
    uint32_t result;          // operand code of the expression value
    uint32_t rows, log_rows;
    const uint8_t *const *sel;
    const fe_t *const *fix;
    const fe_t *W1, *W2;      // column-major [num_advice][rows]
    const fe_t *utab;         // [npts][n_uniform]
    uint32_t n_uniform;
    uint32_t npts;            // d + 1 (interpolate) or 1 (plain evaluation)
    uint32_t d;               // number of outputs in interpolate mode
    const fe_t *vinv;         // [d][npts]: T_k = sum_j vinv[(k-1)*npts + j] * P(j)
    fe_t *const *out;         // d (interpolate) or 1 (plain) output vectors of `rows`
};

template <class F>
__device__ __forceinline__ fe_t small_times(const fe_t &x, uint32_t j) {   // j * x for a tiny j
    fe_t acc = F::zero();
    bool any = false;
    for (int b = 4; b >= 0; --b) {
        if (any) acc = F::dbl(acc);
        if ((j >> b) & 1u) {
            acc = any ? F::add(acc, x) : x;
            any = true;
        }
    }
    return acc;
}

template <class F, uint32_t NSLOT>
__global__ void SRS_KERNEL_BOUNDS(RP_THREADS, 1) k_rowprog(DevArgs A) {
    __shared__ fe_t slots[NSLOT * RP_THREADS];
    const uint32_t tid = threadIdx.x;
    uint32_t row = blockIdx.x * RP_THREADS + tid;
    const bool live = row < A.rows;
    if (!live) row = A.rows - 1;
    const uint32_t mask = A.rows - 1;
    fe_t T[DMAX];
#pragma unroll
    for (uint32_t k = 0; k < DMAX; ++k) T[k] = F::zero();
    for (uint32_t pt = 0; pt < A.npts; ++pt) {
        const fe_t *U = A.utab + (size_t)pt * A.n_uniform;
        for (uint32_t ip = 0; ip < A.n_insn; ++ip) {
            const Insn in = A.prog[ip];
            fe_t r;
            if (in.op <= I_LD_ADV) {
                uint32_t rr = (row + (uint32_t)(int32_t)in.b) & mask;     // (row + rot) rem_euclid 2^k
                if (in.op == I_LD_SEL) {
                    r = A.sel[in.a][rr] ? F::one() : F::zero();
                } else if (in.op == I_LD_FIX) {
                    r = A.fix[in.a][rr];
                } else {
                    size_t idx = (size_t)in.a * A.rows + rr;
                    r = A.W1[idx];
                    if (pt) r = F::add(r, small_times<F>(A.W2[idx], pt));
                }
            } else {
                fe_t a = (in.a & UNIFORM_BIT) ? U[in.a & ~UNIFORM_BIT] : slots[in.a * RP_THREADS + tid];
                if (in.op <= I_MUL) {
                    fe_t b = (in.b & UNIFORM_BIT) ? U[in.b & ~UNIFORM_BIT] : slots[in.b * RP_THREADS + tid];
                    r = in.op == I_ADD ? F::add(a, b) : (in.op == I_SUB ? F::sub(a, b) : F::mul(a, b));
                } else if (in.op == I_SQR) {
                    r = F::sqr(a);
                } else if (in.op == I_DBL) {
                    r = F::dbl(a);
                } else {
                    r = F::neg(a);
                }
            }
            slots[in.dst * RP_THREADS + tid] = r;
        }
        fe_t P = (A.result & UNIFORM_BIT) ? U[A.result & ~UNIFORM_BIT] : slots[A.result * RP_THREADS + tid];
        if (A.d == 0) {
            if (live) A.out[0][row] = P;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < DMAX; ++k) {
                if (k < A.d) T[k] = F::add(T[k], F::mul(A.vinv[k * A.npts + pt], P));
            }
        }
    }
    if (A.d && live) {
#pragma unroll
        for (uint32_t k = 0; k < DMAX; ++k)
            if (k < A.d) A.out[k][row] = T[k];
    }
}

// ---- folds ----
template <class F>
__global__ void k_fold_w(fe_t *__restrict__ out, const fe_t *__restrict__ w1, const fe_t *__restrict__ w2, fe_t r, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = F::add(w1[i], F::mul(r, w2[i]));
}
struct FoldEArgs {
    const fe_t *t[DMAX];
    fe_t rpow[DMAX];
    uint32_t n_terms;
};
template <class F>
__global__ void k_fold_e(fe_t *__restrict__ out, const fe_t *__restrict__ e, FoldEArgs fa, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        fe_t acc = e[i];
#pragma unroll
        for (uint32_t k = 0; k < DMAX; ++k)
            if (k < fa.n_terms) acc = F::add(acc, F::mul(fa.rpow[k], fa.t[k][i]));
        out[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// host: Expression AST (reference src/polynomial/expression.rs:112-120)
// ---------------------------------------------------------------------------------------------
enum NodeKind { N_CONST, N_POLY, N_CHAL, N_NEG, N_SUM, N_PROD, N_SCALED };
struct Node {
    int kind;
    fe_t c;           // N_CONST / N_SCALED
    int64_t index;    // N_POLY / N_CHAL
    int32_t rot;      // N_POLY
    int a, b;
};
struct Ast {
    std::vector<Node> n;
    int add(int kind, int a = -1, int b = -1, int64_t index = 0, int32_t rot = 0, const fe_t *c = nullptr) {
        Node x;
        x.kind = kind;
        x.a = a;
        x.b = b;
        x.index = index;
        x.rot = rot;
        std::memset(&x.c, 0, sizeof(x.c));
        if (c) x.c = *c;
        n.push_back(x);
        return (int)n.size() - 1;
    }
};

template <class F>
struct HostField {
    static fe_t add(const fe_t &a, const fe_t &b) { return F::add(a, b); }
};

struct FieldOps {   // runtime-dispatched host field arithmetic
    int field;
    fe_t zero() const { return Fr::zero(); }
    fe_t one() const { return field == 0 ? Fr::one() : Fq::one(); }
    fe_t add(const fe_t &a, const fe_t &b) const { return field == 0 ? Fr::add(a, b) : Fq::add(a, b); }
    fe_t sub(const fe_t &a, const fe_t &b) const { return field == 0 ? Fr::sub(a, b) : Fq::sub(a, b); }
    fe_t mul(const fe_t &a, const fe_t &b) const { return field == 0 ? Fr::mul(a, b) : Fq::mul(a, b); }
    fe_t neg(const fe_t &a) const { return field == 0 ? Fr::neg(a) : Fq::neg(a); }
    fe_t inv(const fe_t &a) const { return field == 0 ? Fr::inv(a) : Fq::inv(a); }
    fe_t from_u64(uint64_t v) const { return field == 0 ? Fr::from_u64(v) : Fq::from_u64(v); }
    bool is_zero(const fe_t &a) const { return Fr::is_zero(a); }
    bool eq(const fe_t &a, const fe_t &b) const { return Fr::eq(a, b); }
};

// gate stream: postfix words, see include/sirius_amd.h (SRS_EX_*)
static bool parse_gates(const uint64_t *w, size_t nw, size_t num_gates, Ast &ast, std::vector<int> &roots, std::string &err) {
    std::vector<int> st;
    size_t i = 0;
    while (i < nw) {
        uint64_t op = w[i++];
        switch (op) {
        case 0: {   // CONST c[4]
            if (i + 4 > nw) { err = "truncated constant"; return false; }
            fe_t c;
            std::memcpy(&c, &w[i], 32);
            i += 4;
            st.push_back(ast.add(N_CONST, -1, -1, 0, 0, &c));
            break;
        }
        case 1: {   // POLY index rot
            if (i + 2 > nw) { err = "truncated query"; return false; }
            st.push_back(ast.add(N_POLY, -1, -1, (int64_t)w[i], (int32_t)(int64_t)w[i + 1]));
            i += 2;
            break;
        }
        case 2:
            if (i + 1 > nw) { err = "truncated challenge"; return false; }
            st.push_back(ast.add(N_CHAL, -1, -1, (int64_t)w[i]));
            i += 1;
            break;
        case 3:
            if (st.empty()) { err = "stack underflow"; return false; }
            st.back() = ast.add(N_NEG, st.back());
            break;
        case 4:
        case 5: {
            if (st.size() < 2) { err = "stack underflow"; return false; }
            int b = st.back();
            st.pop_back();
            int a = st.back();
            st.back() = ast.add(op == 4 ? N_SUM : N_PROD, a, b);
            break;
        }
        case 6: {
            if (st.empty() || i + 4 > nw) { err = "bad scaled"; return false; }
            fe_t c;
            std::memcpy(&c, &w[i], 32);
            i += 4;
            st.back() = ast.add(N_SCALED, st.back(), -1, 0, 0, &c);
            break;
        }
        case 7:
            if (st.size() != 1) { err = "gate expression does not reduce to one value"; return false; }
            roots.push_back(st.back());
            st.clear();
            break;
        default:
            err = "unknown expression opcode";
            return false;
        }
    }
    if (!st.empty() || roots.size() != num_gates) { err = "gate count mismatch"; return false; }
    return true;
}

static void collect_challenges(const Ast &ast, int r, std::vector<int64_t> &set) {
    const Node &x = ast.n[r];
    if (x.kind == N_CHAL) {
        if (std::find(set.begin(), set.end(), x.index) == set.end()) set.push_back(x.index);
    }
    if (x.a >= 0) collect_challenges(ast, x.a, set);
    if (x.b >= 0) collect_challenges(ast, x.b, set);
}
static size_t num_challenges(const Ast &ast, int r) {   // Expression::num_challenges, expression.rs:163-167
    std::vector<int64_t> set;
    collect_challenges(ast, r, set);
    return set.size();
}

// compress_expression (src/plonk/util.rs:34-56)
static int compress(Ast &ast, const std::vector<int> &gates, size_t challenge_index, const FieldOps &f) {
    fe_t z = f.zero();
    if (gates.size() > 1) {
        int acc = ast.add(N_CONST, -1, -1, 0, 0, &z);
        for (int g : gates) {
            int y = ast.add(N_CHAL, -1, -1, (int64_t)challenge_index);
            acc = ast.add(N_SUM, g, ast.add(N_PROD, acc, y));
        }
        return acc;
    }
    if (gates.size() == 1) return gates[0];
    return ast.add(N_CONST, -1, -1, 0, 0, &z);
}

static int challenge_in_degree(Ast &ast, size_t idx, size_t degree) {   // expression.rs:501-513
    int r = ast.add(N_CHAL, -1, -1, (int64_t)idx);
    for (size_t i = 2; i <= degree; ++i) r = ast.add(N_PROD, r, ast.add(N_CHAL, -1, -1, (int64_t)idx));
    return r;
}

// Expression::homogeneous (src/polynomial/expression.rs:356-429)
struct Ctx {
    size_t num_selectors, num_fixed, num_advice, num_challenges;
};
static bool homogeneous(Ast &ast, int r, const Ctx &ctx, int &out, size_t &degree, std::string &err) {
    const Node x = ast.n[r];
    switch (x.kind) {
    case N_CONST: out = r; degree = 0; return true;
    case N_POLY: {
        size_t i = (size_t)x.index;
        if (i < ctx.num_selectors + ctx.num_fixed) degree = 0;
        else if (i < ctx.num_selectors + ctx.num_fixed + ctx.num_advice) degree = 1;
        else { err = "unknown query index " + std::to_string(i) + " (lookups are not supported)"; return false; }
        out = r;
        return true;
    }
    case N_CHAL: out = r; degree = 1; return true;
    case N_NEG: {
        int a; if (!homogeneous(ast, x.a, ctx, a, degree, err)) return false;
        out = ast.add(N_NEG, a);
        return true;
    }
    case N_SCALED: {
        int a; if (!homogeneous(ast, x.a, ctx, a, degree, err)) return false;
        out = ast.add(N_SCALED, a, -1, 0, 0, &x.c);
        return true;
    }
    case N_PROD: {
        int a, b; size_t da, db;
        if (!homogeneous(ast, x.a, ctx, a, da, err) || !homogeneous(ast, x.b, ctx, b, db, err)) return false;
        out = ast.add(N_PROD, a, b);
        degree = da + db;
        return true;
    }
    default: {   // N_SUM
        int a, b; size_t da, db;
        if (!homogeneous(ast, x.a, ctx, a, da, err) || !homogeneous(ast, x.b, ctx, b, db, err)) return false;
        if (da > db) {
            out = ast.add(N_SUM, a, ast.add(N_PROD, b, challenge_in_degree(ast, ctx.num_challenges, da - db)));
            degree = da;
        } else if (da < db) {
            out = ast.add(N_SUM, ast.add(N_PROD, a, challenge_in_degree(ast, ctx.num_challenges, db - da)), b);
            degree = db;
        } else {
            out = ast.add(N_SUM, a, b);
            degree = da;
        }
        return true;
    }
    }
}

// ---------------------------------------------------------------------------------------------
// host: compile an expression into (uniform program, row program)
// ---------------------------------------------------------------------------------------------
// A value is either
//   KNOWN   : compile-time constant (folded)                 -> becomes a uniform-table entry
//   UNIFORM : depends on challenges only (evaluated per call, per point on the host)
//   ROW     : depends on the row (virtual register)
struct Val {
    int cls;      // 0 known, 1 uniform, 2 row
    int id;       // uniform index / virtual register
    fe_t k;       // known value
};
struct UOp {      // uniform program: u[dst] = op(u[a], u[b]);  leaves: constant / challenge
    int op;       // 0 const, 1 challenge(index), 2 add, 3 sub, 4 mul, 5 neg
    int a, b;
    fe_t c;
    int64_t chal;
};
struct VInsn {
    uint32_t op;
    int dst;      // virtual register
    int a, b;     // operands: >= 0 virtual reg; < 0: uniform index = -(x)-1 ; loads: a = column, b = rotation
};

struct Compiler {
    const Ast &ast;
    FieldOps f;
    Ctx ctx;
    bool fold_mode;                 // true: advice/challenges are W1 + X*W2 (cross terms); false: plain
    std::vector<UOp> uops;
    std::vector<VInsn> vins;
    int nvreg = 0;
    std::map<int, Val> memo;                                     // AST node -> value
    std::map<std::tuple<int, int, int>, int> u_cse;              // (op, a, b) -> uniform index
    std::map<std::tuple<uint32_t, int, int>, int> r_cse;         // (op, a, b) -> vreg
    std::map<std::tuple<int64_t, int>, int> chal_cse;
    std::string err;

    Compiler(const Ast &a, FieldOps fo, Ctx c, bool fm) : ast(a), f(fo), ctx(c), fold_mode(fm) {}

    int u_const(const fe_t &c) {
        for (size_t i = 0; i < uops.size(); ++i)
            if (uops[i].op == 0 && f.eq(uops[i].c, c)) return (int)i;
        UOp u{};
        u.op = 0;
        u.c = c;
        uops.push_back(u);
        return (int)uops.size() - 1;
    }
    int u_chal(int64_t idx) {
        auto key = std::make_tuple(idx, 0);
        auto it = chal_cse.find(key);
        if (it != chal_cse.end()) return it->second;
        UOp u{};
        u.op = 1;
        u.chal = idx;
        uops.push_back(u);
        return chal_cse[key] = (int)uops.size() - 1;
    }
    int u_op(int op, int a, int b) {
        if ((op == 2 || op == 4) && a > b) std::swap(a, b);
        auto key = std::make_tuple(op, a, b);
        auto it = u_cse.find(key);
        if (it != u_cse.end()) return it->second;
        UOp u{};
        u.op = op;
        u.a = a;
        u.b = b;
        uops.push_back(u);
        return u_cse[key] = (int)uops.size() - 1;
    }
    Val known(const fe_t &k) { Val v; v.cls = 0; v.id = -1; v.k = k; return v; }
    Val uniform(int id) { Val v{}; v.cls = 1; v.id = id; return v; }
    Val rowv(int id) { Val v{}; v.cls = 2; v.id = id; return v; }
    int as_uniform(const Val &v) { return v.cls == 0 ? u_const(v.k) : v.id; }
    int operand(const Val &v) { return v.cls == 2 ? v.id : -(as_uniform(v)) - 1; }
    Val r_op(uint32_t op, int a, int b) {
        if ((op == I_ADD || op == I_MUL) && a > b) std::swap(a, b);
        auto key = std::make_tuple(op, a, b);
        auto it = r_cse.find(key);
        if (it != r_cse.end()) return rowv(it->second);
        VInsn in{op, nvreg++, a, b};
        vins.push_back(in);
        r_cse[key] = in.dst;
        return rowv(in.dst);
    }

    Val v_add(const Val &a, const Val &b) {
        if (a.cls == 0 && b.cls == 0) return known(f.add(a.k, b.k));
        if (a.cls == 0 && f.is_zero(a.k)) return b;
        if (b.cls == 0 && f.is_zero(b.k)) return a;
        if (a.cls < 2 && b.cls < 2) return uniform(u_op(2, as_uniform(a), as_uniform(b)));
        return r_op(I_ADD, operand(a), operand(b));
    }
    Val v_neg(const Val &a) {
        if (a.cls == 0) return known(f.neg(a.k));
        if (a.cls == 1) return uniform(u_op(5, a.id, -1));
        return r_op(I_NEG, a.id, 0);
    }
    Val v_mul(const Val &a, const Val &b) {
        if (a.cls == 0 && b.cls == 0) return known(f.mul(a.k, b.k));
        if ((a.cls == 0 && f.is_zero(a.k)) || (b.cls == 0 && f.is_zero(b.k))) return known(f.zero());
        if (a.cls == 0 && f.eq(a.k, f.one())) return b;
        if (b.cls == 0 && f.eq(b.k, f.one())) return a;
        if (a.cls < 2 && b.cls < 2) return uniform(u_op(4, as_uniform(a), as_uniform(b)));
        if (a.cls == 2 && b.cls == 2 && a.id == b.id) return r_op(I_SQR, a.id, 0);
        return r_op(I_MUL, operand(a), operand(b));
    }

    Val walk(int r) {
        auto it = memo.find(r);
        if (it != memo.end()) return it->second;
        const Node &x = ast.n[r];
        Val v;
        switch (x.kind) {
        case N_CONST: v = known(x.c); break;
        case N_CHAL: v = uniform(u_chal(x.index)); break;
        case N_POLY: {
            size_t i = (size_t)x.index;
            uint32_t op;
            int col;
            if (i < ctx.num_selectors) { op = I_LD_SEL; col = (int)i; }
            else if (i < ctx.num_selectors + ctx.num_fixed) { op = I_LD_FIX; col = (int)(i - ctx.num_selectors); }
            else if (i < ctx.num_selectors + ctx.num_fixed + ctx.num_advice) { op = I_LD_ADV; col = (int)(i - ctx.num_selectors - ctx.num_fixed); }
            else { err = "column index out of range"; v = known(f.zero()); break; }
            auto key = std::make_tuple(op, col, (int)x.rot);
            auto c = r_cse.find(key);
            if (c != r_cse.end()) { v = rowv(c->second); break; }
            VInsn in{op, nvreg++, col, (int)x.rot};
            vins.push_back(in);
            r_cse[key] = in.dst;
            v = rowv(in.dst);
            break;
        }
        case N_NEG: v = v_neg(walk(x.a)); break;
        case N_SUM: { Val a = walk(x.a); Val b = walk(x.b); v = v_add(a, b); break; }
        case N_PROD: { Val a = walk(x.a); Val b = walk(x.b); v = v_mul(a, b); break; }
        default: { Val a = walk(x.a); v = v_mul(a, known(x.c)); break; }
        }
        memo[r] = v;
        return v;
    }
};

// linear-scan allocation of virtual registers to LDS slots
static bool allocate(const std::vector<VInsn> &vins, int nvreg, int result_vreg, std::vector<Insn> &out,
                     uint32_t &result_code, uint32_t &nslots) {
    std::vector<int> last(nvreg, -1);
    for (size_t i = 0; i < vins.size(); ++i) {
        const VInsn &in = vins[i];
        if (in.op > I_LD_ADV) {
            if (in.a >= 0) last[in.a] = (int)i;
            if (in.op <= I_MUL && in.b >= 0) last[in.b] = (int)i;
        }
    }
    if (result_vreg >= 0) last[result_vreg] = (int)vins.size();
    std::vector<int> slot(nvreg, -1);
    std::vector<int> free_list;
    uint32_t n = 0;
    auto enc = [&](int x) -> uint32_t { return x >= 0 ? (uint32_t)slot[x] : (UNIFORM_BIT | (uint32_t)(-x - 1)); };
    for (size_t i = 0; i < vins.size(); ++i) {
        const VInsn &in = vins[i];
        Insn o;
        o.op = in.op;
        if (in.op <= I_LD_ADV) {
            o.a = (uint32_t)in.a;
            o.b = (uint32_t)in.b;
        } else {
            o.a = enc(in.a);
            o.b = in.op <= I_MUL ? enc(in.b) : 0;
            // operands dying here free their slots before the destination is chosen
            if (in.a >= 0 && last[in.a] == (int)i) free_list.push_back(slot[in.a]);
            if (in.op <= I_MUL && in.b >= 0 && in.b != in.a && last[in.b] == (int)i) free_list.push_back(slot[in.b]);
        }
        int s;
        if (!free_list.empty()) { s = free_list.back(); free_list.pop_back(); } else s = (int)n++;
        slot[in.dst] = s;
        o.dst = (uint32_t)s;
        out.push_back(o);
        if (last[in.dst] < 0) free_list.push_back(s);   // dead value (cannot happen after CSE, but stay safe)
    }
    nslots = n ? n : 1;
    result_code = result_vreg >= 0 ? (uint32_t)slot[result_vreg] : 0;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Structure
// ---------------------------------------------------------------------------------------------
struct Program {
    std::vector<UOp> uops;
    std::vector<Insn> insns;
    uint32_t result = 0, nslots = 1;
    Insn *d_insns = nullptr;
};

struct Structure {
    int field = 0;
    uint32_t k = 0;
    size_t rows = 0, num_selectors = 0, num_fixed = 0, num_advice = 0;
    size_t s_num_challenges = 0;   // PlonkStructure::num_challenges (compressed().num_challenges())
    size_t h_num_challenges = 0;   // homogeneous().num_challenges()  (challenge i folds with i + this)
    size_t degree = 0;             // homogeneous degree = number of cross terms
    Program cross;                 // homogeneous expression, fold mode
    Program plain_compressed;      // compressed expression, single witness (decider, plonk/mod.rs:328)
    Program plain_homogeneous;     // homogeneous expression, single witness (decider, sangria/mod.rs:351)
    std::vector<fe_t> vinv;        // [degree][degree+1]
    // device data
    uint8_t **d_sel_ptrs = nullptr;
    fe_t **d_fix_ptrs = nullptr;
    std::vector<void *> owned;
    fe_t *d_vinv = nullptr;
    Arena arena;
};

static bool build_program(const Ast &ast, int root, const FieldOps &f, const Ctx &ctx, bool fold_mode, Program &p,
                          std::string &err) {
    Compiler c(ast, f, ctx, fold_mode);
    Val v = c.walk(root);
    if (!c.err.empty()) { err = c.err; return false; }
    int result_vreg = -1;
    uint32_t result_uniform = 0;
    if (v.cls == 2) result_vreg = v.id; else result_uniform = UNIFORM_BIT | (uint32_t)c.as_uniform(v);
    p.uops = c.uops;
    allocate(c.vins, c.nvreg, result_vreg, p.insns, p.result, p.nslots);
    if (result_vreg < 0) p.result = result_uniform;
    return true;
}

// inverse Vandermonde for the points 0..d: vinv[(k-1)*(d+1) + j] = coefficient of X^k in L_j(X)
static std::vector<fe_t> inverse_vandermonde(const FieldOps &f, size_t d) {
    size_t m = d + 1;
    std::vector<fe_t> out(d * m);
    for (size_t j = 0; j < m; ++j) {
        std::vector<fe_t> poly(1, f.one());        // prod_{t != j} (X - t)
        fe_t denom = f.one();
        for (size_t t = 0; t < m; ++t) {
            if (t == j) continue;
            fe_t ft = f.from_u64(t);
            std::vector<fe_t> nx(poly.size() + 1, f.zero());
            for (size_t i = 0; i < poly.size(); ++i) {
                nx[i + 1] = f.add(nx[i + 1], poly[i]);
                nx[i] = f.sub(nx[i], f.mul(poly[i], ft));
            }
            poly.swap(nx);
            denom = f.mul(denom, f.sub(f.from_u64(j), ft));
        }
        fe_t di = f.inv(denom);
        for (size_t k = 1; k <= d; ++k) out[(k - 1) * m + j] = f.mul(poly[k], di);
    }
    return out;
}

static void upload_program(Program &p, Structure &S) {
    if (p.insns.empty()) return;
    SRS_HIP_CHECK(hipMalloc((void **)&p.d_insns, p.insns.size() * sizeof(Insn)));
    S.owned.push_back(p.d_insns);
    SRS_HIP_CHECK(hipMemcpy(p.d_insns, p.insns.data(), p.insns.size() * sizeof(Insn), hipMemcpyHostToDevice));
}

Structure *create(int field, uint32_t k, size_t num_selectors, size_t num_fixed, size_t num_advice,
                  const uint8_t *const *selectors, const fe_t *const *fixed, int space_device,
                  const uint64_t *gates, size_t gates_words, size_t num_gates, int &rc, std::string &err) {
    rc = 4;
    FieldOps f{field};
    Ast ast;
    std::vector<int> roots;
    if (!parse_gates(gates, gates_words, num_gates, ast, roots, err)) return nullptr;
    if (num_selectors + num_fixed == 0) { err = "Fixed & Selectors can't be empty in one time"; return nullptr; }   // eval.rs:47-54
    std::unique_ptr<Structure> S(new Structure());
    S->field = field;
    S->k = k;
    S->rows = (size_t)1 << k;
    S->num_selectors = num_selectors;
    S->num_fixed = num_fixed;
    S->num_advice = num_advice;
    // ConstraintSystemMetainfo::build with no lookups: ctx.num_challenges starts at 0
    // (src/table/constraint_system_metainfo.rs:81-97) -> CompressedGates::new (src/plonk/mod.rs:84-107)
    Ctx ctx{num_selectors, num_fixed, num_advice, 0};
    int compressed = compress(ast, roots, ctx.num_challenges, f);
    ctx.num_challenges = num_challenges(ast, compressed);
    S->s_num_challenges = ctx.num_challenges;
    int homog;
    size_t degree;
    if (!homogeneous(ast, compressed, ctx, homog, degree, err)) { rc = 7; return nullptr; }
    S->h_num_challenges = num_challenges(ast, homog);
    S->degree = degree;
    if (degree > DMAX) { err = "gate degree " + std::to_string(degree) + " exceeds the supported maximum"; return nullptr; }
    if (!build_program(ast, homog, f, ctx, true, S->cross, err) ||
        !build_program(ast, compressed, f, ctx, false, S->plain_compressed, err) ||
        !build_program(ast, homog, f, ctx, false, S->plain_homogeneous, err)) {
        rc = 7;
        return nullptr;
    }
    if (degree) S->vinv = inverse_vandermonde(f, degree);
    // ---- device residency: programs, fixed columns, selectors
    rc = 5;
    upload_program(S->cross, *S);
    upload_program(S->plain_compressed, *S);
    upload_program(S->plain_homogeneous, *S);
    if (!S->vinv.empty()) {
        SRS_HIP_CHECK(hipMalloc((void **)&S->d_vinv, S->vinv.size() * sizeof(fe_t)));
        S->owned.push_back(S->d_vinv);
        SRS_HIP_CHECK(hipMemcpy(S->d_vinv, S->vinv.data(), S->vinv.size() * sizeof(fe_t), hipMemcpyHostToDevice));
    }
    const hipMemcpyKind kind = space_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    std::vector<uint8_t *> selp(num_selectors);
    std::vector<fe_t *> fixp(num_fixed);
    for (size_t i = 0; i < num_selectors; ++i) {
        SRS_HIP_CHECK(hipMalloc((void **)&selp[i], S->rows));
        S->owned.push_back(selp[i]);
        SRS_HIP_CHECK(hipMemcpy(selp[i], selectors[i], S->rows, kind));
    }
    for (size_t i = 0; i < num_fixed; ++i) {
        SRS_HIP_CHECK(hipMalloc((void **)&fixp[i], S->rows * sizeof(fe_t)));
        S->owned.push_back(fixp[i]);
        SRS_HIP_CHECK(hipMemcpy(fixp[i], fixed[i], S->rows * sizeof(fe_t), kind));
    }
    SRS_HIP_CHECK(hipMalloc((void **)&S->d_sel_ptrs, (num_selectors + 1) * sizeof(void *)));
    S->owned.push_back(S->d_sel_ptrs);
    SRS_HIP_CHECK(hipMalloc((void **)&S->d_fix_ptrs, (num_fixed + 1) * sizeof(void *)));
    S->owned.push_back(S->d_fix_ptrs);
    if (num_selectors) SRS_HIP_CHECK(hipMemcpy(S->d_sel_ptrs, selp.data(), num_selectors * sizeof(void *), hipMemcpyHostToDevice));
    if (num_fixed) SRS_HIP_CHECK(hipMemcpy(S->d_fix_ptrs, fixp.data(), num_fixed * sizeof(void *), hipMemcpyHostToDevice));
    rc = 0;
    return S.release();
}

void destroy(Structure *S) {
    if (!S) return;
    for (void *p : S->owned) (void)hipFree(p);
    S->arena.release();
    delete S;
}

size_t degree(const Structure *S) { return S->degree; }
size_t num_challenges(const Structure *S) { return S->s_num_challenges; }
size_t num_advice(const Structure *S) { return S->num_advice; }
size_t rows(const Structure *S) { return S->rows; }
int field(const Structure *S) { return S->field; }

// evaluate the uniform program for one point: challenge i -> ch[i] + pt * ch[i + fold_offset]
static bool eval_uniform(const Program &p, const FieldOps &f, const fe_t *ch, size_t n_ch, size_t fold_offset, bool fold,
                         uint32_t pt, fe_t *out, std::string &err) {
    fe_t fpt = f.from_u64(pt);
    for (size_t i = 0; i < p.uops.size(); ++i) {
        const UOp &u = p.uops[i];
        switch (u.op) {
        case 0: out[i] = u.c; break;
        case 1: {
            size_t a = (size_t)u.chal;
            if (a >= n_ch) { err = "challenge index " + std::to_string(a) + " out of boundary " + std::to_string(n_ch); return false; }
            out[i] = ch[a];
            if (fold) {
                size_t b = a + fold_offset;
                if (b >= n_ch) { err = "challenge index " + std::to_string(b) + " out of boundary " + std::to_string(n_ch); return false; }
                if (pt) out[i] = f.add(out[i], f.mul(fpt, ch[b]));
            }
            break;
        }
        case 2: out[i] = f.add(out[u.a], out[u.b]); break;
        case 3: out[i] = f.sub(out[u.a], out[u.b]); break;
        case 4: out[i] = f.mul(out[u.a], out[u.b]); break;
        default: out[i] = f.neg(out[u.a]); break;
        }
    }
    return true;
}

template <class F>
static void launch_rowprog(const DevArgs &A, uint32_t nslots, hipStream_t st) {
    uint32_t blocks = (A.rows + RP_THREADS - 1) / RP_THREADS;
    if (nslots <= 8) SRS_LAUNCH((k_rowprog<F, 8>), (blocks), (RP_THREADS), 0, st, A);
    else if (nslots <= 16) SRS_LAUNCH((k_rowprog<F, 16>), (blocks), (RP_THREADS), 0, st, A);
    else if (nslots <= 24) SRS_LAUNCH((k_rowprog<F, 24>), (blocks), (RP_THREADS), 0, st, A);
    else SRS_LAUNCH((k_rowprog<F, 32>), (blocks), (RP_THREADS), 0, st, A);
}

// mode 0: cross terms (needs W2), outputs `degree` vectors; mode 1/2: plain evaluation of the
// compressed / homogeneous expression on W1, one output vector.
int evaluate(Structure *S, int mode, const fe_t *W1_dev, const fe_t *W2_dev, const fe_t *challenges_host, size_t n_ch,
             fe_t *const *out_dev_ptrs_host, hipStream_t st, std::string &err) {
    FieldOps f{S->field};
    Program &p = mode == 0 ? S->cross : (mode == 1 ? S->plain_compressed : S->plain_homogeneous);
    const uint32_t d = mode == 0 ? (uint32_t)S->degree : 0;
    const uint32_t npts = mode == 0 ? d + 1 : 1;
    const uint32_t nout = mode == 0 ? d : 1;
    if (mode == 0 && d == 0) return 0;
    if (p.nslots > 32) { err = "row program needs more than 32 live registers"; return 4; }
    const size_t nu = p.uops.size() ? p.uops.size() : 1;
    std::vector<fe_t> utab(nu * npts);
    for (uint32_t pt = 0; pt < npts; ++pt)
        if (!eval_uniform(p, f, challenges_host, n_ch, S->h_num_challenges, mode == 0, pt, utab.data() + (size_t)pt * nu, err)) return 7;
    Arena &A = S->arena;
    A.reserve(Arena::pad(utab.size() * sizeof(fe_t)) + Arena::pad(nout * sizeof(void *)) + 1024);
    A.reset();
    fe_t *d_utab = A.take<fe_t>(utab.size());
    fe_t **d_out = A.take<fe_t *>(nout);
    SRS_HIP_CHECK(hipMemcpyAsync(d_utab, utab.data(), utab.size() * sizeof(fe_t), hipMemcpyHostToDevice, st));
    SRS_HIP_CHECK(hipMemcpyAsync(d_out, out_dev_ptrs_host, nout * sizeof(void *), hipMemcpyHostToDevice, st));
    DevArgs a;
    a.prog = p.d_insns;
    a.n_insn = (uint32_t)p.insns.size();
    a.result = p.result;
    a.rows = (uint32_t)S->rows;
    a.log_rows = S->k;
    a.sel = S->d_sel_ptrs;
    a.fix = S->d_fix_ptrs;
    a.W1 = W1_dev;
    a.W2 = W2_dev ? W2_dev : W1_dev;
    a.utab = d_utab;
    a.n_uniform = (uint32_t)nu;
    a.npts = npts;
    a.d = d;
    a.vinv = S->d_vinv;
    a.out = d_out;
    {
        prof::Scope ps(mode == 0 ? "rowprog_cross_terms" : "rowprog_eval", st, S->rows);
        if (S->field == 0) launch_rowprog<Fr>(a, p.nslots, st); else launch_rowprog<Fq>(a, p.nslots, st);
    }
    SRS_HIP_CHECK(hipStreamSynchronize(st));   // utab / pointer staging lives in the arena
    SRS_HIP_CHECK(hipGetLastError());
    prof::collect();
    return 0;
}

void fold_w(int field, fe_t *out, const fe_t *w1, const fe_t *w2, const fe_t &r, size_t n, hipStream_t st) {
    if (!n) return;
    uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, 256 * 16);
    if (field == 0) SRS_LAUNCH((k_fold_w<Fr>), (blocks), (256), 0, st, out, w1, w2, r, n);
    else SRS_LAUNCH((k_fold_w<Fq>), (blocks), (256), 0, st, out, w1, w2, r, n);
}

int fold_e(int field, fe_t *out, const fe_t *e, const fe_t *const *t_dev_ptrs_host, size_t n_terms, const fe_t &r, size_t n,
           hipStream_t st, std::string &err) {
    if (n_terms > DMAX) { err = "more than 8 cross terms"; return 4; }
    FieldOps f{field};
    FoldEArgs fa;
    fa.n_terms = (uint32_t)n_terms;
    fe_t acc = r;   // r^1, r^2, ...  (accumulator.rs:380-383)
    for (uint32_t k = 0; k < DMAX; ++k) {
        fa.t[k] = k < n_terms ? t_dev_ptrs_host[k] : nullptr;
        fa.rpow[k] = acc;
        acc = f.mul(acc, r);
    }
    if (!n) return 0;
    uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, 256 * 16);
    if (field == 0) SRS_LAUNCH((k_fold_e<Fr>), (blocks), (256), 0, st, out, e, fa, n);
    else SRS_LAUNCH((k_fold_e<Fq>), (blocks), (256), 0, st, out, e, fa, n);
    return 0;
}

}  // namespace rowprog
}  // namespace srs
