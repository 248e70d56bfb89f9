// Candidate Ty layout "8x8 patches": Ty[c][y/8][z/8][kx][y%8][z%8].  Compare the copy rate of the x-pass pattern and the y-pass pattern in
// the current layout Ty[c][kx][y][z] (+8 pad per plane) and in the patch layout.  usage: patch [G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct __attribute__((aligned(16))) c16 { double re, im; };
struct Geo { int G, a; size_t stride; };
__device__ __forceinline__ size_t off_cur(int c, int kx, int y, int z, const Geo& g) { return ((size_t)c * g.a + kx) * g.stride + (size_t)y * g.G + z; }
__device__ __forceinline__ size_t off_patch(int c, int kx, int y, int z, const Geo& g) {
    const int gb = g.G >> 3;
    return ((((size_t)c * gb + (y >> 3)) * gb + (z >> 3)) * g.a + kx) * 64 + ((y & 7) << 3) + (z & 7);
}
// LAYOUT 2, "z-block major": Ty[c][z/8][kx][y][z%8] — a y-pass workgroup (c, kx, 8 z columns) owns ONE contiguous G*128-byte block; an x-pass
// tile gathers its runs from a planes G*128 bytes apart, all inside the 16*a*G*8-byte window of its (c, z block) instead of one run from each
// of 3a planes 16*G*G bytes apart
__device__ __forceinline__ size_t off_zb(int c, int kx, int y, int z, const Geo& g) {
    const int gb = g.G >> 3;
    return ((((size_t)c * gb + (z >> 3)) * g.a + kx) * g.G + y) * 8 + (z & 7);
}
// LAYOUT 3: z-block major with one 128-byte line of padding per kx block (stride G*128 + 128 B instead of 3 * 2^14: off one HBM channel)
__device__ __forceinline__ size_t off_zbp(int c, int kx, int y, int z, const Geo& g) {
    const int gb = g.G >> 3;
    return (((size_t)c * gb + (z >> 3)) * g.a + kx) * ((size_t)g.G * 8 + 8) + (size_t)y * 8 + (z & 7);
}
// LAYOUT 4: "row-block major" Ty[c][y][z/8][kx][z%8]: an x-pass tile of 8 points is ONE contiguous a*128-byte block per component (tile-major
// with T = 8; narrower tiles read half / quarter lines of it); a y-pass workgroup reads G lines (G/8)*a*128 bytes apart
__device__ __forceinline__ size_t off_rb(int c, int kx, int y, int z, const Geo& g) {
    const int gb = g.G >> 3;
    return ((((size_t)c * g.G + y) * gb + (z >> 3)) * g.a + kx) * 8 + (z & 7);
}
template <int LAYOUT> __device__ __forceinline__ size_t off_any(int c, int kx, int y, int z, const Geo& g) {
    return LAYOUT == 4 ? off_rb(c, kx, y, z, g) : (LAYOUT == 3 ? off_zbp(c, kx, y, z, g) : (LAYOUT == 2 ? off_zb(c, kx, y, z, g) : (LAYOUT == 1 ? off_patch(c, kx, y, z, g) : off_cur(c, kx, y, z, g))));
}
// x pattern: tile = RUN consecutive z at one y; NF field groups x 3 comps x a kx planes
template <int LAYOUT, int RUN, int NF, int PAIRED, int YFAST = 0>
__global__ __launch_bounds__(256) void xpat(const c16* __restrict__ in, c16* __restrict__ out, Geo g, unsigned ntiles, size_t fg_stride) {
    extern __shared__ c16 buf[];
    unsigned tile = blockIdx.x;
    if (PAIRED > 1 && blockIdx.x < (ntiles / (8 * PAIRED)) * (8 * PAIRED)) {
        const unsigned q = blockIdx.x / (8 * PAIRED), r = blockIdx.x % (8 * PAIRED);
        tile = q * (8 * PAIRED) + PAIRED * (r % 8) + r / 8;
    }
    size_t i0 = (size_t)tile * RUN;
    int y = (int)(i0 / g.G), z0 = (int)(i0 % g.G);
    if (YFAST) {                                    // tile order: the 8/RUN tiles of a line, then y, then the z block
        const unsigned per = 8 / RUN, line = tile / per, sub = tile % per;
        y = (int)(line % g.G); z0 = (int)(line / g.G) * 8 + (int)sub * RUN;
    }
    const int n = NF * 3 * g.a * RUN, tid = threadIdx.x;
    for (int t = tid; t < n; t += 256) {
        const int e = t % RUN, r = t / RUN, fc = r % (NF * 3), kx = r / (NF * 3), c = fc % 3, f = fc / 3;
        const size_t o = (size_t)f * fg_stride + off_any<LAYOUT>(c, kx, y, z0 + e, g);
        buf[t] = in[o];
    }
    __syncthreads();
    for (int t = tid; t < n; t += 256) {
        const int e = t % RUN, r = t / RUN, fc = r % (NF * 3), kx = r / (NF * 3), c = fc % 3, f = fc / 3;
        const size_t o = (size_t)f * fg_stride + off_any<LAYOUT>(c, kx, y, z0 + e, g);
        c16 v = buf[(t * 7 + 3) % n]; v.re += 1.0;
        out[o] = v;
    }
}
// y pattern: workgroup = (c, kx, 8 z columns), all y: reads Ty, writes Ty (the Tz side of the real pass is not modelled)
template <int LAYOUT, int KXFAST = 0>
__global__ __launch_bounds__(256) void ypat(const c16* __restrict__ in, c16* __restrict__ out, Geo g) {
    extern __shared__ c16 buf[];
    const int gb = g.G >> 3;
    int o_ = blockIdx.x / gb, zb = blockIdx.x % gb, c = o_ / g.a, kx = o_ % g.a;
    if (KXFAST) { kx = blockIdx.x % g.a; zb = (blockIdx.x / g.a) % gb; c = blockIdx.x / (g.a * gb); }      // consecutive workgroups: consecutive kx
    const int n = g.G * 8, tid = threadIdx.x;
    for (int t = tid; t < n; t += 256) {
        const int z = t & 7, y = t >> 3;
        buf[t] = in[off_any<LAYOUT>(c, kx, y, zb * 8 + z, g)];
    }
    __syncthreads();
    for (int t = tid; t < n; t += 256) {
        const int z = t & 7, y = t >> 3;
        c16 v = buf[(t * 7 + 3) % n]; v.re += 1.0;
        out[off_any<LAYOUT>(c, kx, y, zb * 8 + z, g)] = v;
    }
}
template <class F> void timeit(const char* what, double bytes, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-52s %8.1f us  %5.2f TB/s\n", what, best * 1e3 / 10, bytes * 10 / (best * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 384;
    Geo g{G, G / 3, (size_t)G * G + 8};
    const size_t fg = (size_t)3 * g.a * g.stride;
    c16 *in, *out;
    hipMalloc(&in, 2 * fg * 16 + (1 << 24)); hipMalloc(&out, 2 * fg * 16 + (1 << 24)); hipMemset(in, 0, 2 * fg * 16 + (1 << 24));
    const size_t plane = (size_t)G * G;
    const double b1 = 2.0 * 3 * g.a * plane * 16, b2 = 2 * b1;
    printf("G = %d\n", G);
#define X(L, RUN, NF, P, name) timeit(name, NF == 1 ? b1 : b2, [&]() { hipLaunchKernelGGL((xpat<L, RUN, NF, P>), dim3((unsigned)(plane / RUN)), dim3(256), (size_t)NF * 3 * g.a * RUN * 16, 0, in, out, g, (unsigned)(plane / RUN), fg); })
#define XY(L, RUN, NF, P, name) timeit(name, NF == 1 ? b1 : b2, [&]() { hipLaunchKernelGGL((xpat<L, RUN, NF, P, 1>), dim3((unsigned)(plane / RUN)), dim3(256), (size_t)NF * 3 * g.a * RUN * 16, 0, in, out, g, (unsigned)(plane / RUN), fg); })
    if (G <= 192) {
        X(0, 8, 1, 1, "x fwd  128-B runs            current layout");
        X(1, 8, 1, 1, "x fwd  128-B runs            8x8 patches");
        X(2, 8, 1, 1, "x fwd  128-B runs            z-block major, z-fastest tiles");
        XY(2, 8, 1, 1, "x fwd  128-B runs            z-block major, y-fastest tiles");
        XY(0, 8, 1, 1, "x fwd  128-B runs            current layout, y-fastest tiles");
        X(0, 4, 2, 2, "x adj  64-B runs, paired     current layout");
        X(1, 4, 2, 2, "x adj  64-B runs, paired     8x8 patches");
        X(1, 4, 2, 1, "x adj  64-B runs, unpaired   8x8 patches");
    } else {
        X(0, 4, 1, 2, "x fwd  64-B runs, paired     current layout");
        X(1, 4, 1, 2, "x fwd  64-B runs, paired     8x8 patches");
        X(1, 4, 1, 1, "x fwd  64-B runs, unpaired   8x8 patches");
        X(2, 4, 1, 2, "x fwd  64-B runs, paired     z-block major, z-fastest tiles");
        XY(2, 4, 1, 1, "x fwd  64-B runs            z-block major, y-fastest tiles (line halves adjacent)");
        XY(2, 4, 1, 2, "x fwd  64-B runs, paired    z-block major, y-fastest tiles");
        XY(0, 4, 1, 1, "x fwd  64-B runs            current layout, y-fastest tiles");
        XY(2, 8, 1, 1, "x fwd  128-B runs           z-block major, y-fastest tiles");
        X(3, 4, 1, 2, "x fwd  64-B runs, paired     z-block major + pad, z-fastest tiles");
        XY(3, 4, 1, 2, "x fwd  64-B runs, paired    z-block major + pad, y-fastest tiles");
        X(4, 4, 1, 2, "x fwd  64-B runs, paired     row-block major, z-fastest tiles");
        X(4, 4, 1, 1, "x fwd  64-B runs, unpaired   row-block major, z-fastest tiles");
        X(0, 2, 2, 4, "x adj  32-B runs, paired     current layout");
        X(1, 2, 2, 4, "x adj  32-B runs, paired     8x8 patches");
        X(3, 2, 2, 4, "x adj  32-B runs, paired     z-block major + pad");
        X(4, 2, 2, 4, "x adj  32-B runs, paired     row-block major");
        X(1, 2, 2, 1, "x adj  32-B runs, unpaired   8x8 patches");
    }
    const unsigned ny = 3u * g.a * (G / 8);
    timeit("y pass (Ty side: G runs of 128 B)  current layout", b1, [&]() { hipLaunchKernelGGL(ypat<0>, dim3(ny), dim3(256), (size_t)G * 8 * 16, 0, in, out, g); });
    timeit("y pass (Ty side: G/8 runs of 1 KB) 8x8 patches", b1, [&]() { hipLaunchKernelGGL(ypat<1>, dim3(ny), dim3(256), (size_t)G * 8 * 16, 0, in, out, g); });
    timeit("y pass (Ty side: one G*128-B block) z-block major", b1, [&]() { hipLaunchKernelGGL(ypat<2>, dim3(ny), dim3(256), (size_t)G * 8 * 16, 0, in, out, g); });
    timeit("y pass z-block major + pad", b1, [&]() { hipLaunchKernelGGL(ypat<3>, dim3(ny), dim3(256), (size_t)G * 8 * 16, 0, in, out, g); });
    timeit("y pass z-block major + pad, kx-fastest workgroups", b1, [&]() { hipLaunchKernelGGL((ypat<3, 1>), dim3(ny), dim3(256), (size_t)G * 8 * 16, 0, in, out, g); });
    timeit("y pass row-block major (G lines, stride (G/8)*a*128 B)", b1, [&]() { hipLaunchKernelGGL(ypat<4>, dim3(ny), dim3(256), (size_t)G * 8 * 16, 0, in, out, g); });
    timeit("y pass row-block major, kx-fastest workgroups", b1, [&]() { hipLaunchKernelGGL((ypat<4, 1>), dim3(ny), dim3(256), (size_t)G * 8 * 16, 0, in, out, g); });
    timeit("y pass (one G*128-B block) z-block major, kx-fastest workgroups", b1, [&]() { hipLaunchKernelGGL((ypat<2, 1>), dim3(ny), dim3(256), (size_t)G * 8 * 16, 0, in, out, g); });
    return 0;
}
