// Microbenchmark with real arithmetic: the forward fused x pass of the kinematic dynamo at G = 192 (128^3) with WAVE-OWNED transforms.
//
// csrc/kdyn.hip's x pass runs its 192-point transforms as Stockham stages through the LDS (a tile of 12 transforms per workgroup, a
// barrier per stage).  DESIGN.md section 5 names the alternative that was never built: one wave per transform, 3 points per lane, the
// 64-point part exchanged between lanes (six radix-2 stages of ds_bpermute / DPP) and a radix-3 stage inside the lane — no barrier
// between stages, the LDS only for the transpose between the coalesced tile (128-byte runs of 8 (y,z) points per kx plane) and the
// per-wave lines.  This program measures what that design reaches on the same bytes and the same arithmetic as the real kernel:
//     spectra of 3 components (64 kx modes, Hermitian) -> grid (two real lines packed in one complex transform) -> cross product with
//     a velocity field read from HBM -> forward transform -> Hermitian split -> 64 modes of 3 components stored
// and checks one tile against a direct O(G^2) evaluation on the host.  Bytes per launch: 113 MB spectra in + 170 MB velocity + 113 MB out.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro_wave_fft.hip -o gpurun_out/micro_wave_fft ;  run: gpurun_out/micro_wave_fft
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int G = 192, A = 64, RUN = 8;                 // grid length, modes kept, (y,z) points per tile (128-byte runs)
struct __attribute__((aligned(16))) c16 { double re, im; };

__device__ __forceinline__ c16 cmul(c16 a, c16 b) { return c16{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ c16 cadd(c16 a, c16 b) { return c16{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c16 csub(c16 a, c16 b) { return c16{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ c16 cconj(c16 a) { return c16{a.re, -a.im}; }
__device__ __forceinline__ c16 xch(c16 v, int d) { return c16{__shfl_xor(v.re, d), __shfl_xor(v.im, d)}; }       // ds_bpermute_b32 x 4
// the same exchange without the LDS crossbar: DPP inside a row of 16 lanes, v_permlane16/32_swap (gfx950) across rows
template <int D> __device__ __forceinline__ int xor_lane(int v, int lane) {
    if constexpr (D == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);            // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    else if constexpr (D == 4) {
        const int r = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);                    // row_shl:4 into banks 0, 2 (lane i <- i + 4)
        return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);                           // row_shr:4 into banks 1, 3 (lane i <- i - 4)
    } else if constexpr (D == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);    // row_ror:8
    else if constexpr (D == 16) { const auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16) ? s[0] : s[1]; }
    else { const auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32) ? s[0] : s[1]; }
}
template <int D> __device__ __forceinline__ double xor_d(double v, int lane) {
    return __hiloint2double(xor_lane<D>(__double2hiint(v), lane), xor_lane<D>(__double2loint(v), lane));
}
template <int D> __device__ __forceinline__ c16 xchd(c16 v, int lane) { return c16{xor_d<D>(v.re, lane), xor_d<D>(v.im, lane)}; }
template <int S, bool DPP> __device__ __forceinline__ c16 exchange(c16 v, int lane) {
    if constexpr (DPP) return xchd<(1 << S)>(v, lane); else return xch(v, 1 << S);
}
// one decimation-in-time stage (half size d = 2^S): twl = the stage's twiddle on the upper lane of a pair, 1 on the lower one
template <int S, bool DPP> __device__ __forceinline__ c16 dit_stage(c16 z, c16 twl, int lane) {
    const double sg = (lane & (1 << S)) ? -1.0 : 1.0;
    const c16 pre = S == 0 ? z : cmul(twl, z);
    const c16 o = exchange<S, DPP>(pre, lane);
    return c16{fma(sg, pre.re, o.re), fma(sg, pre.im, o.im)};          // lower lane: z + tw * upper; upper lane: lower - tw * z
}
template <int S, bool DPP> __device__ __forceinline__ c16 dif_stage(c16 z, c16 twl, int lane) {
    const double sg = (lane & (1 << S)) ? -1.0 : 1.0;
    const c16 o = exchange<S, DPP>(z, lane);
    const c16 t = c16{fma(sg, z.re, o.re), fma(sg, z.im, o.im)};
    return S == 0 ? t : cmul(t, cconj(twl));
}
__device__ __forceinline__ int bitrev6(int x) { return (int)(__brev((unsigned)x) >> 26); }

// tw[s][lane] = exp(+2 pi i (lane & (d-1)) / (2d)) on the upper lane of a pair (lane & d), 1 on the lower, d = 2^s, s = 0..5;  tw[6 + r - 1][lane] = exp(+2 pi i r lane / 192), r = 1, 2
#ifndef WAVES
#define WAVES 2
#endif
template <bool DPP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void wave_fft_xpass(const c16* __restrict__ spec, const double* __restrict__ U, c16* __restrict__ out,
                                                      const c16* __restrict__ tw, size_t plane_stride, size_t npair) {
    __shared__ c16 buf[3][RUN][A + 1];                   // in: [c][point][kx]; later reused per pair as [c][2w] = H[k], [c][2w+1] = H[G-k]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t p0 = (size_t)blockIdx.x * RUN;
#pragma unroll
    for (int i = 0; i < 6; ++i) {                        // coalesced: 8 lanes cover one 128-byte run
        const int e = tid + 256 * i, q = e & 7, kx = (e >> 3) & 63, c = e >> 9;
        buf[c][q][kx] = spec[((size_t)c * A + kx) * plane_stride + p0 + q];
    }
    c16 t6[6], t3[2];
#pragma unroll
    for (int s = 0; s < 6; ++s) t6[s] = tw[s * 64 + lane];
    t3[0] = tw[6 * 64 + lane]; t3[1] = tw[7 * 64 + lane];
    __syncthreads();
    const c16 w3 = c16{-0.5, 0.86602540378443864676};     // exp(+2 pi i / 3)
    c16 B[3][3];                                         // grid values of the three components at x = lane + 64 x1 (re: even point, im: odd point)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        c16 y[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int k = 3 * bitrev6(lane) + r;
            c16 z = c16{0.0, 0.0};
            if (k < A) { const c16 se = buf[c][2 * w][k], so = buf[c][2 * w + 1][k]; z = c16{se.re - so.im, se.im + so.re}; }
            else if (k > G - A) { const c16 se = buf[c][2 * w][G - k], so = buf[c][2 * w + 1][G - k]; z = c16{se.re + so.im, so.re - se.im}; }
            // decimation in time, bit-reversed input -> natural output
            z = dit_stage<0, DPP>(z, t6[0], lane); z = dit_stage<1, DPP>(z, t6[1], lane); z = dit_stage<2, DPP>(z, t6[2], lane);
            z = dit_stage<3, DPP>(z, t6[3], lane); z = dit_stage<4, DPP>(z, t6[4], lane); z = dit_stage<5, DPP>(z, t6[5], lane);
            y[r] = r == 0 ? z : cmul(z, t3[r - 1]);
        }
        // radix 3 inside the lane: z[lane + 64 x1] = sum_r y_r w3^(r x1)
        const c16 a1 = cmul(y[1], w3), a2 = cmul(y[2], cconj(w3)), b1 = cmul(y[1], cconj(w3)), b2 = cmul(y[2], w3);
        B[c][0] = cadd(y[0], cadd(y[1], y[2]));
        B[c][1] = cadd(y[0], cadd(a1, a2));
        B[c][2] = cadd(y[0], cadd(b1, b2));
    }
    // cross product with the velocity at the same points: U[c][pair][x][2]
    const size_t pp = p0 / 2 + w;
    c16 Wc[3][3];
#pragma unroll
    for (int x1 = 0; x1 < 3; ++x1) {
        c16 u[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) u[c] = *reinterpret_cast<const c16*>(U + (((size_t)c * npair + pp) * G + lane + 64 * x1) * 2);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int a = (c + 1) % 3, b = (c + 2) % 3;
            Wc[c][x1] = c16{u[a].re * B[b][x1].re - u[b].re * B[a][x1].re, u[a].im * B[b][x1].im - u[b].im * B[a][x1].im};
        }
    }
    // forward: radix 3 inside the lane, twiddle, decimation in frequency across the lanes (natural input -> bit-reversed output)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        c16 g[3];
        {
            const c16 h0 = Wc[c][0], h1 = Wc[c][1], h2 = Wc[c][2];
            g[0] = cadd(h0, cadd(h1, h2));
            g[1] = cadd(h0, cadd(cmul(h1, cconj(w3)), cmul(h2, w3)));
            g[2] = cadd(h0, cadd(cmul(h1, w3), cmul(h2, cconj(w3))));
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            c16 z = r == 0 ? g[0] : cmul(g[r], cconj(t3[r - 1]));
            z = dif_stage<5, DPP>(z, t6[5], lane); z = dif_stage<4, DPP>(z, t6[4], lane); z = dif_stage<3, DPP>(z, t6[3], lane);
            z = dif_stage<2, DPP>(z, t6[2], lane); z = dif_stage<1, DPP>(z, t6[1], lane); z = dif_stage<0, DPP>(z, t6[0], lane);
            const int k = 3 * bitrev6(lane) + r;        // this wave's rows of buf are its own: no workgroup barrier needed here
            if (k < A) buf[c][2 * w][k] = z;
            if (k > G - A) buf[c][2 * w + 1][G - k] = z;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) {                        // Hermitian split and coalesced store
        const int e = tid + 256 * i, q = e & 7, kx = (e >> 3) & 63, c = e >> 9;
        const c16 hl = buf[c][q & ~1][kx], hh = kx == 0 ? hl : buf[c][q | 1][kx];
        const c16 E = c16{0.5 * (hl.re + hh.re), 0.5 * (hl.im - hh.im)};
        const c16 O = c16{0.5 * (hl.im + hh.im), -0.5 * (hl.re - hh.re)};
        out[((size_t)c * A + kx) * plane_stride + p0 + q] = (q & 1) ? O : E;
    }
}

int main() {
    const size_t plane = (size_t)G * G, stride = plane + 8, nspec = (size_t)3 * A * stride, nu = (size_t)3 * plane * G;
    std::vector<c16> hs(nspec);
    std::vector<double> hu(nu);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
    for (auto& v : hs) v = c16{rnd(), rnd()};
    for (int c = 0; c < 3; ++c) for (size_t p = 0; p < stride; ++p) hs[((size_t)c * A) * stride + p].im = 0.0;     // kx = 0 of a real line
    for (auto& v : hu) v = rnd();
    std::vector<c16> htw(8 * 64);
    for (int s = 0; s < 6; ++s) for (int l = 0; l < 64; ++l) { const int d = 1 << s; const double ph = 2.0 * M_PI * (l & (d - 1)) / (2.0 * d); htw[s * 64 + l] = (l & d) ? c16{cos(ph), sin(ph)} : c16{1.0, 0.0}; }
    for (int r = 1; r <= 2; ++r) for (int l = 0; l < 64; ++l) { const double ph = 2.0 * M_PI * r * l / 192.0; htw[(5 + r) * 64 + l] = c16{cos(ph), sin(ph)}; }
    c16 *ds, *dout, *dtw; double* du;
    hipMalloc(&ds, nspec * 16); hipMalloc(&dout, nspec * 16); hipMalloc(&du, nu * 8); hipMalloc(&dtw, htw.size() * 16);
    hipMemcpy(ds, hs.data(), nspec * 16, hipMemcpyHostToDevice); hipMemcpy(du, hu.data(), nu * 8, hipMemcpyHostToDevice);
    hipMemcpy(dtw, htw.data(), htw.size() * 16, hipMemcpyHostToDevice);
    hipMemset(dout, 0, nspec * 16);
    const unsigned ntiles = (unsigned)(plane / RUN);
  int rc = 0;
  for (int variant = 0; variant < 2; ++variant) {
    hipMemset(dout, 0, nspec * 16);
    if (variant) hipLaunchKernelGGL(wave_fft_xpass<true>, dim3(ntiles), dim3(256), 0, 0, ds, du, dout, dtw, stride, plane / 2);
    else hipLaunchKernelGGL(wave_fft_xpass<false>, dim3(ntiles), dim3(256), 0, 0, ds, du, dout, dtw, stride, plane / 2);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    // ---- check tile 1234, all 8 points, against the direct sums -----------------------------------------------------
    std::vector<c16> ho(nspec);
    hipMemcpy(ho.data(), dout, nspec * 16, hipMemcpyDeviceToHost);
    using cd = std::complex<double>;
    double err = 0.0, scale = 0.0;
    const size_t tile = 1234;
    for (int q = 0; q < RUN; ++q) {
        const size_t p = tile * RUN + q;
        std::vector<double> f(3 * G), wv(3 * G);
        for (int c = 0; c < 3; ++c)
            for (int x = 0; x < G; ++x) {
                cd s = 0;
                for (int k = 0; k < A; ++k) {
                    const c16 v = hs[((size_t)c * A + k) * stride + p];
                    const cd F(v.re, v.im), e = std::polar(1.0, 2.0 * M_PI * k * x / G);
                    s += k == 0 ? F * e : F * e + std::conj(F * e);
                }
                f[c * G + x] = s.real();
            }
        auto Uat = [&](int c, int x) { return hu[(((size_t)c * (plane / 2) + p / 2) * G + x) * 2 + (p & 1)]; };
        for (int c = 0; c < 3; ++c) for (int x = 0; x < G; ++x) { const int a = (c + 1) % 3, b = (c + 2) % 3; wv[c * G + x] = Uat(a, x) * f[b * G + x] - Uat(b, x) * f[a * G + x]; }
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < A; ++k) {
                cd s = 0;
                for (int x = 0; x < G; ++x) s += wv[c * G + x] * std::polar(1.0, -2.0 * M_PI * k * x / G);
                const c16 v = ho[((size_t)c * A + k) * stride + p];
                err = std::max(err, std::abs(s - cd(v.re, v.im))); scale = std::max(scale, std::abs(s));
            }
    }
    printf("check (one tile, 8 points x 3 components x 64 modes): max error %.3e of %.3e\n", err, scale);
    // ---- timing -----------------------------------------------------------------------------------------------------
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) {
            if (variant) hipLaunchKernelGGL(wave_fft_xpass<true>, dim3(ntiles), dim3(256), 0, 0, ds, du, dout, dtw, stride, plane / 2);
            else hipLaunchKernelGGL(wave_fft_xpass<false>, dim3(ntiles), dim3(256), 0, 0, ds, du, dout, dtw, stride, plane / 2);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const double bytes = 2.0 * 3 * A * plane * 16 + 3.0 * plane * G * 8;
    printf("wave-owned forward x pass (%s), G = %d: %.1f us per launch, %.0f MB => %.2f TB/s  (csrc/kdyn.hip: 83-86 us, 4.6 TB/s; the access pattern alone: 6.2-6.45)\n",
           variant ? "DPP / permlane swaps" : "ds_bpermute", G, best * 1e3 / 20, bytes / 1e6, bytes / (best * 1e-3 / 20) / 1e12);
    if (!(err < 1e-9 * scale)) rc = 2;
  }
  return rc;
}
