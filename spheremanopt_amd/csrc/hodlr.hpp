// HODLR form of the per-wavenumber tau-solve operators of the Poiseuille path (host side: plan, factorisation, device layout).
//
// S_n = (rows of) A_n^-1 B_n, A_n the tau system of the LBVP (POIS:818-841): banded when unknowns and equations are interleaved by
// Chebyshev mode, apart from seven dense boundary rows.  The inverse of such a matrix is rank-structured: with rows and columns ordered
// mode-major (index 3*mode + variable) EVERY off-diagonal block of S_n has rank <= 10 (below the diagonal) / <= 16 (above) — exact ranks,
// set by the band widths and the boundary rows, independent of the block size (measured on the operators of the 384 x 192 problem, every
// wavenumber; n = 0: <= 9).  So S_n is stored as a hierarchically off-diagonal low-rank (HODLR) matrix:
//        S = [ S_11        U_12 V_12^H ]      recursively in S_11, S_22 down to dense leaves of <= LEAF_MAX rows,
//            [ U_21 V_21^H S_22        ]
// 77 k instead of 332 k complex entries at Nz = 192, applied as two rounds of short dot products (t_b = V_b^H x_cols(b); y = D x + sum_b
// U_b t_b) with no sequential dependence along the mode index — unlike the banded substitution the operator is the inverse of, which is a
// chain of 7*Nz dependent rows per wavenumber.  The factors come from a column-pivoted Gram-Schmidt sweep of each off-diagonal block of the
// dense S_n (truncated where the residual columns fall under tol); the conjugate transpose S_n^H (the reference's transposed solve,
// POIS:1417-1460) reuses the same factors with U and V exchanged, so the discrete adjoint stays the exact transpose of the forward apply.
//
// Everything here is host code without HIP types: tests/c/hodlr_host_test.cpp drives it on the CPU (emulate() is the device kernel's
// walk over the same descriptors).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <vector>

namespace smo {
namespace hodlr {

using cd = std::complex<double>;

constexpr int LEAF_MAX = 48;     // dense diagonal blocks of at most this many rows (3*192 = 576 -> 36-row leaves, 4 levels)
constexpr int PAD = 8;           // rows of the device layout are padded to a multiple of PAD complex entries (128 B, one lane group's load)

struct Block { int r0, r1, c0, c1, depth, pair; };        // off-diagonal block rows [r0,r1) x cols [c0,c1); pair = index of the transposed position
struct Node { int lo, hi, depth, left, right, up, blk; }; // tree node; blk = index of the block whose ROWS are this node (-1 for the root)
struct Plan {
    int n = 0;
    std::vector<Node> nodes;
    std::vector<Block> blocks;
    std::vector<int> leaves;                               // node indices, in row order
    std::vector<int> leaf_of_row;                          // index into `leaves`
};

inline int build_nodes(Plan& p, int lo, int hi, int depth, int up) {
    const int id = (int)p.nodes.size();
    p.nodes.push_back({lo, hi, depth, -1, -1, up, -1});
    if (hi - lo > LEAF_MAX) {
        const int mid = lo + (hi - lo) / 2;
        const int l = build_nodes(p, lo, mid, depth + 1, id);
        const int r = build_nodes(p, mid, hi, depth + 1, id);
        p.nodes[id].left = l; p.nodes[id].right = r;
        const int b = (int)p.blocks.size();
        p.blocks.push_back({lo, mid, mid, hi, depth + 1, b + 1});
        p.blocks.push_back({mid, hi, lo, mid, depth + 1, b});
        p.nodes[l].blk = b; p.nodes[r].blk = b + 1;
    } else {
        p.leaves.push_back(id);
    }
    return id;
}
inline Plan make_plan(int n) {
    Plan p; p.n = n;
    build_nodes(p, 0, n, 0, -1);
    std::sort(p.leaves.begin(), p.leaves.end(), [&](int x, int y) { return p.nodes[x].lo < p.nodes[y].lo; });
    p.leaf_of_row.assign(n, 0);
    for (size_t l = 0; l < p.leaves.size(); ++l)
        for (int r = p.nodes[p.leaves[l]].lo; r < p.nodes[p.leaves[l]].hi; ++r) p.leaf_of_row[r] = (int)l;
    return p;
}

// factors of ONE operator: per block U (m x k, row-major) and VH (k x nc, row-major), per leaf the dense diagonal block (row-major)
struct Factors {
    std::vector<int> rank;
    std::vector<std::vector<cd>> U, VH, D;
};

// B (m x nc, taken from S with leading dimension ld) ~= U VH by Gram-Schmidt with column pivoting: the sweep stops when every residual
// column has a 2-norm <= tol, so |B - U VH| <= tol column by column, whatever the orthogonality of U
inline int low_rank(const cd* S, int ld, int r0, int m, int c0, int nc, double tol, std::vector<cd>& U, std::vector<cd>& VH) {
    std::vector<cd> W((size_t)nc * m);                     // column-major working copy
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < nc; ++c) W[(size_t)c * m + r] = S[(size_t)(r0 + r) * ld + c0 + c];
    std::vector<char> done(nc, 0);
    std::vector<cd> Q, R;                                  // Q: k columns of m; R: k rows of nc
    int k = 0;
    const int kmax = std::min(m, nc);
    while (k < kmax) {
        int j = -1; double best = 0.0;
        for (int c = 0; c < nc; ++c) {
            if (done[c]) continue;
            double s = 0.0;
            const cd* w = &W[(size_t)c * m];
            for (int r = 0; r < m; ++r) s += std::norm(w[r]);
            if (s > best) { best = s; j = c; }
        }
        const double nrm = std::sqrt(best);
        if (j < 0 || !(nrm > tol)) break;
        Q.resize((size_t)(k + 1) * m); R.resize((size_t)(k + 1) * nc, cd(0));
        cd* q = &Q[(size_t)k * m];
        cd* wj = &W[(size_t)j * m];
        for (int r = 0; r < m; ++r) { q[r] = wj[r] / nrm; wj[r] = 0; }
        cd* rk = &R[(size_t)k * nc];
        std::fill(rk, rk + nc, cd(0));
        rk[j] = nrm; done[j] = 1;
        for (int c = 0; c < nc; ++c) {
            if (done[c]) continue;
            cd* w = &W[(size_t)c * m];
            cd d = 0;
            for (int r = 0; r < m; ++r) d += std::conj(q[r]) * w[r];
            rk[c] = d;
            for (int r = 0; r < m; ++r) w[r] -= d * q[r];
        }
        ++k;
    }
    U.assign((size_t)m * k, cd(0));
    for (int i = 0; i < k; ++i) for (int r = 0; r < m; ++r) U[(size_t)r * k + i] = Q[(size_t)i * m + r];
    VH.assign(R.begin(), R.begin() + (size_t)k * nc);
    return k;
}

inline void factor(const Plan& p, const cd* S, int ld, double tol, Factors& f) {
    const size_t nb = p.blocks.size(), nl = p.leaves.size();
    f.rank.assign(nb, 0); f.U.assign(nb, {}); f.VH.assign(nb, {}); f.D.assign(nl, {});
    for (size_t b = 0; b < nb; ++b) {
        const Block& B = p.blocks[b];
        f.rank[b] = low_rank(S, ld, B.r0, B.r1 - B.r0, B.c0, B.c1 - B.c0, tol, f.U[b], f.VH[b]);
    }
    for (size_t l = 0; l < nl; ++l) {
        const Node& L = p.nodes[p.leaves[l]];
        const int w = L.hi - L.lo;
        f.D[l].resize((size_t)w * w);
        for (int r = 0; r < w; ++r) for (int c = 0; c < w; ++c) f.D[l][(size_t)r * w + c] = S[(size_t)(L.lo + r) * ld + L.lo + c];
    }
}

// ---- device layout -------------------------------------------------------------------------------------------------------------------
// One operator = `stride` complex entries: the V^H rows of every block, then one row per output (leaf row | the U rows of the blocks over
// it, outermost first | xin dense extra columns), then xout dense extra rows; every row padded with zeros to a multiple of PAD.  The same
// descriptors serve every wavenumber (ranks are padded to the largest one found for the block).
// A workgroup (task) owns the rows of one tree node at depth `split` and walks two descriptor lists over an LDS array Z:
//     Z = [x (n, mode-major) | xin extras | PAD zeros | t_b of the blocks it needs | G = per leaf [x_leaf | t_b ... | extras], padded]
//     round 1: Z[row.out] = data[row.data .. +len) . Z[row.in .. +len)        (t = V^H x; the blocks over AND under the node)
//     copy   : G[i] = Z[lut[i]]
//     round 2: y[row.out] = data[row.data .. +len) . Z[row.in .. +len)        (leaf rows against their G; extra rows against x)
struct Row { uint32_t data; uint16_t len; uint16_t in; uint32_t out; uint32_t pad_; };
struct Task { uint32_t row1, n1, lut, nlut, row2, n2, zg, zend; };
struct Layout {
    int n = 0, xin = 0, xout = 0, W = 0;
    std::vector<int> K;                                    // uniform rank per block
    size_t stride = 0;
    std::vector<uint32_t> vh_off, row_off, row_len, extra_off;
    std::vector<Row> rows;
    std::vector<uint16_t> lut;
    std::vector<Task> tasks;
    uint32_t lds_entries = 0;                              // largest zend
};
inline uint32_t pad8(uint32_t v) { return (v + PAD - 1) / PAD * PAD; }

// blocks whose rows contain row range of `node` (outermost first)
inline std::vector<int> blocks_over(const Plan& p, int node) {
    std::vector<int> v;
    for (int id = node; id >= 0; id = p.nodes[id].up) if (p.nodes[id].blk >= 0) v.push_back(p.nodes[id].blk);
    std::reverse(v.begin(), v.end());
    return v;
}
inline void blocks_under(const Plan& p, int node, std::vector<int>& v) {      // strictly below
    const Node& N = p.nodes[node];
    if (N.left < 0) return;
    v.push_back(p.nodes[N.left].blk); v.push_back(p.nodes[N.right].blk);
    blocks_under(p, N.left, v); blocks_under(p, N.right, v);
}
inline void leaves_under(const Plan& p, int node, std::vector<int>& v) {
    const Node& N = p.nodes[node];
    if (N.left < 0) { v.push_back(node); return; }
    leaves_under(p, N.left, v); leaves_under(p, N.right, v);
}
inline void task_nodes(const Plan& p, int node, int split, std::vector<int>& v) {
    const Node& N = p.nodes[node];
    if (N.left < 0 || N.depth >= split) { v.push_back(node); return; }
    task_nodes(p, N.left, split, v); task_nodes(p, N.right, split, v);
}

inline Layout make_layout(const Plan& p, const std::vector<int>& K, int split, int xin, int xout) {
    Layout L; L.n = p.n; L.xin = xin; L.xout = xout; L.K = K;
    const int n = p.n;
    uint32_t off = 0;
    L.vh_off.resize(p.blocks.size());
    for (size_t b = 0; b < p.blocks.size(); ++b) { L.vh_off[b] = off; off += (uint32_t)K[b] * pad8(p.blocks[b].c1 - p.blocks[b].c0); }
    L.row_off.resize(n); L.row_len.resize(n);
    for (size_t l = 0; l < p.leaves.size(); ++l) {
        const Node& N = p.nodes[p.leaves[l]];
        uint32_t len = N.hi - N.lo;
        for (int b : blocks_over(p, p.leaves[l])) len += K[b];
        len = pad8(len + xin);
        for (int r = N.lo; r < N.hi; ++r) { L.row_off[r] = off; L.row_len[r] = len; off += len; }
    }
    L.extra_off.resize(xout);
    for (int e = 0; e < xout; ++e) { L.extra_off[e] = off; off += pad8(n); }
    L.stride = off;
    std::vector<int> tn;
    task_nodes(p, 0, split, tn);
    L.W = (int)tn.size();
    const uint32_t zero_slot = n + xin, zt0 = pad8(n + xin) + PAD;
    for (int w = 0; w < L.W; ++w) {
        Task T{};
        std::vector<int> need = blocks_over(p, tn[w]);
        blocks_under(p, tn[w], need);
        std::vector<uint32_t> tpos(p.blocks.size(), 0);
        uint32_t z = zt0;
        for (int b : need) { tpos[b] = z; z += K[b]; }
        // round 1, longest rows first (the lane groups take rows round-robin)
        std::vector<Row> r1;
        for (int b : need)
            for (int i = 0; i < K[b]; ++i) {
                const uint32_t lenp = pad8(p.blocks[b].c1 - p.blocks[b].c0);
                r1.push_back({L.vh_off[b] + (uint32_t)i * lenp, (uint16_t)lenp, (uint16_t)p.blocks[b].c0, tpos[b] + i, 0});
            }
        std::stable_sort(r1.begin(), r1.end(), [](const Row& x, const Row& y) { return x.len > y.len; });
        T.row1 = (uint32_t)L.rows.size(); T.n1 = (uint32_t)r1.size();
        L.rows.insert(L.rows.end(), r1.begin(), r1.end());
        // gathered inputs of the leaves, and round 2
        T.zg = pad8(z); T.lut = (uint32_t)L.lut.size();
        std::vector<int> lv;
        leaves_under(p, tn[w], lv);
        std::vector<Row> r2;
        uint32_t g = T.zg;
        for (int leaf : lv) {
            const Node& N = p.nodes[leaf];
            const uint32_t len = L.row_len[N.lo];
            uint32_t cnt = 0;
            for (int c = N.lo; c < N.hi; ++c) { L.lut.push_back((uint16_t)c); ++cnt; }
            for (int b : blocks_over(p, leaf)) for (int i = 0; i < K[b]; ++i) { L.lut.push_back((uint16_t)(tpos[b] + i)); ++cnt; }
            for (int e = 0; e < xin; ++e) { L.lut.push_back((uint16_t)(n + e)); ++cnt; }
            for (; cnt < len; ++cnt) L.lut.push_back((uint16_t)zero_slot);
            for (int r = N.lo; r < N.hi; ++r) r2.push_back({L.row_off[r], (uint16_t)len, (uint16_t)g, (uint32_t)r, 0});
            g += len;
        }
        T.nlut = (uint32_t)L.lut.size() - T.lut;
        for (int e = w; e < xout; e += L.W) r2.push_back({L.extra_off[e], (uint16_t)pad8(n), 0, (uint32_t)(n + e), 0});
        T.row2 = (uint32_t)L.rows.size(); T.n2 = (uint32_t)r2.size();
        L.rows.insert(L.rows.end(), r2.begin(), r2.end());
        T.zend = g;
        L.lds_entries = std::max(L.lds_entries, T.zend);
        L.tasks.push_back(T);
    }
    return L;
}

// Write one operator into dst[0 .. stride).  forward: the operator itself (extras = xout rows of n entries).  adjoint: its conjugate
// transpose from the SAME factors (block b of S^H = (block pair(b) of S)^H = V U^H; extras = the xin rows that become columns).
// L.K must be the ranks of the operator written (adjoint: K[b] = forward K[pair(b)]).
inline void pack(const Plan& p, const Layout& L, const Factors& f, const cd* extras, bool adjoint, cd* dst) {
    std::fill(dst, dst + L.stride, cd(0));
    const int n = p.n;
    for (size_t b = 0; b < p.blocks.size(); ++b) {
        const Block& B = p.blocks[b];
        const int nc = B.c1 - B.c0;
        const uint32_t lenp = pad8(nc);
        if (!adjoint) {
            const int k = f.rank[b];
            for (int i = 0; i < k; ++i) for (int c = 0; c < nc; ++c) dst[L.vh_off[b] + (size_t)i * lenp + c] = f.VH[b][(size_t)i * nc + c];
        } else {                                             // V'^H = U_pair^H : (k x nc), nc = rows of the partner
            const int pb = B.pair, k = f.rank[pb];
            for (int i = 0; i < k; ++i) for (int c = 0; c < nc; ++c) dst[L.vh_off[b] + (size_t)i * lenp + c] = std::conj(f.U[pb][(size_t)c * k + i]);
        }
    }
    for (size_t l = 0; l < p.leaves.size(); ++l) {
        const Node& N = p.nodes[p.leaves[l]];
        const int w = N.hi - N.lo;
        const std::vector<int> over = blocks_over(p, p.leaves[l]);
        for (int r = N.lo; r < N.hi; ++r) {
            cd* row = dst + L.row_off[r];
            for (int c = 0; c < w; ++c) row[c] = adjoint ? std::conj(f.D[l][(size_t)c * w + (r - N.lo)]) : f.D[l][(size_t)(r - N.lo) * w + c];
            int pos = w;
            for (int b : over) {
                const Block& B = p.blocks[b];
                if (!adjoint) {
                    const int k = f.rank[b];
                    for (int i = 0; i < k; ++i) row[pos + i] = f.U[b][(size_t)(r - B.r0) * k + i];
                } else {                                     // U' = VH_pair^H : row (r - r0) of the partner's column index
                    const int pb = B.pair, k = f.rank[pb], ncp = p.blocks[pb].c1 - p.blocks[pb].c0;
                    for (int i = 0; i < k; ++i) row[pos + i] = std::conj(f.VH[pb][(size_t)i * ncp + (r - B.r0)]);
                }
                pos += L.K[b];
            }
            if (adjoint) for (int e = 0; e < L.xin; ++e) row[pos + e] = std::conj(extras[(size_t)e * n + r]);
        }
    }
    if (!adjoint) for (int e = 0; e < L.xout; ++e) std::copy(extras + (size_t)e * n, extras + (size_t)(e + 1) * n, dst + L.extra_off[e]);
}

// the device kernel's walk, on the host (tests): x has n + xin entries, y gets n + xout
inline void emulate(const Layout& L, const cd* data, const cd* x, cd* y) {
    for (const Task& T : L.tasks) {
        std::vector<cd> Z(T.zend + PAD, cd(0));
        for (int i = 0; i < L.n + L.xin; ++i) Z[i] = x[i];
        auto dot = [&](const Row& r) { cd s = 0; for (int i = 0; i < r.len; ++i) s += data[r.data + i] * Z[r.in + i]; return s; };
        for (uint32_t i = 0; i < T.n1; ++i) { const Row& r = L.rows[T.row1 + i]; Z[r.out] = dot(r); }
        for (uint32_t i = 0; i < T.nlut; ++i) Z[T.zg + i] = Z[L.lut[T.lut + i]];
        for (uint32_t i = 0; i < T.n2; ++i) { const Row& r = L.rows[T.row2 + i]; y[r.out] = dot(r); }
    }
}

}  // namespace hodlr
}  // namespace smo
