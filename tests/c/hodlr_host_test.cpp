// CPU test of spheremanopt_amd/csrc/hodlr.hpp: factorise the inverse of a bordered banded matrix, pack it in the device layout and walk the
// descriptors the way the kernel does; compare with the dense products S x and S^H x.  Prints "ok <max rank> <compression>" or exits non-zero.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "hodlr.hpp"

using namespace smo::hodlr;

static std::vector<cd> inverse(std::vector<cd> A, int n) {
    std::vector<cd> X((size_t)n * n, cd(0));
    for (int i = 0; i < n; ++i) X[(size_t)i * n + i] = 1;
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int r = k + 1; r < n; ++r) if (std::abs(A[(size_t)r * n + k]) > std::abs(A[(size_t)p * n + k])) p = r;
        if (p != k) for (int c = 0; c < n; ++c) { std::swap(A[(size_t)k * n + c], A[(size_t)p * n + c]); std::swap(X[(size_t)k * n + c], X[(size_t)p * n + c]); }
        const cd inv = 1.0 / A[(size_t)k * n + k];
        for (int c = 0; c < n; ++c) { A[(size_t)k * n + c] *= inv; X[(size_t)k * n + c] *= inv; }
        for (int r = 0; r < n; ++r) {
            if (r == k) continue;
            const cd f = A[(size_t)r * n + k];
            if (f == cd(0)) continue;
            for (int c = 0; c < n; ++c) { A[(size_t)r * n + c] -= f * A[(size_t)k * n + c]; X[(size_t)r * n + c] -= f * X[(size_t)k * n + c]; }
        }
    }
    return X;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 300, split = argc > 2 ? atoi(argv[2]) : 2, nx = 3;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> g;
    auto rnd = [&]() { return cd(g(rng), g(rng)); };
    std::vector<cd> A((size_t)n * n, cd(0));
    for (int r = 0; r < n; ++r) {
        for (int c = std::max(0, r - 3); c < std::min(n, r + 6); ++c) A[(size_t)r * n + c] = rnd();
        A[(size_t)r * n + r] += 12.0;
    }
    for (int r = n - 4; r < n; ++r) for (int c = 0; c < n; ++c) A[(size_t)r * n + c] += 0.3 * rnd();     // dense boundary rows
    const std::vector<cd> S = inverse(A, n);
    std::vector<cd> E((size_t)nx * n);
    for (auto& e : E) e = rnd();
    double mx = 0;
    for (const cd& s : S) mx = std::max(mx, std::abs(s));

    const Plan p = make_plan(n);
    Factors f;
    factor(p, S.data(), n, 1e-14 * mx, f);
    int kmax = 0;
    for (int k : f.rank) kmax = std::max(kmax, k);
    std::vector<int> K = f.rank, KH(K.size());
    for (auto& k : K) k += 1;                                  // uniform ranks are >= the ranks found: exercise the padding
    for (size_t b = 0; b < K.size(); ++b) KH[b] = K[p.blocks[b].pair];
    const Layout Lf = make_layout(p, K, split, 0, nx), La = make_layout(p, KH, split, nx, 0);
    std::vector<cd> df(Lf.stride), da(La.stride);
    pack(p, Lf, f, E.data(), false, df.data());
    pack(p, La, f, E.data(), true, da.data());

    std::vector<cd> x(n + nx), y(n + nx, cd(0)), yh(n, cd(0));
    for (auto& v : x) v = rnd();
    emulate(Lf, df.data(), x.data(), y.data());
    emulate(La, da.data(), x.data(), yh.data());
    double e1 = 0, e2 = 0, s1 = 0, s2 = 0;
    for (int r = 0; r < n + nx; ++r) {
        cd ref = 0;
        for (int c = 0; c < n; ++c) ref += (r < n ? S[(size_t)r * n + c] : E[(size_t)(r - n) * n + c]) * x[c];
        e1 = std::max(e1, std::abs(ref - y[r])); s1 = std::max(s1, std::abs(ref));
    }
    for (int r = 0; r < n; ++r) {
        cd ref = 0;
        for (int c = 0; c < n; ++c) ref += std::conj(S[(size_t)c * n + r]) * x[c];
        for (int e = 0; e < nx; ++e) ref += std::conj(E[(size_t)e * n + r]) * x[n + e];
        e2 = std::max(e2, std::abs(ref - yh[r])); s2 = std::max(s2, std::abs(ref));
    }
    const double comp = (double)n * n / (double)Lf.stride;
    std::printf("%s rank %d compression %.2f tasks %d lds %u err %.2e %.2e\n", (e1 < 1e-12 * s1 && e2 < 1e-12 * s2) ? "ok" : "FAIL", kmax, comp, Lf.W,
                Lf.lds_entries, e1 / s1, e2 / s2);
    return (e1 < 1e-12 * s1 && e2 < 1e-12 * s2 && kmax < 24) ? 0 : 1;
}
