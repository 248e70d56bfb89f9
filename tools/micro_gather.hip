// Microbenchmark (no FFT, no arithmetic): what HBM rate does the ACCESS PATTERN of the fused x passes allow?
//
// An x-pass tile reads RUN complex128 (RUN*16 bytes) from each of its (field group, component, kx) planes of Ty — planes 16*(G*G+8) bytes
// apart — and writes the same pattern back.  This kernel moves exactly those bytes through the LDS and back (40 KB of LDS per workgroup
// => 4 workgroups per CU, 256 threads), once with the plane-strided addresses of the real layout (including the XCD pairing of tiles
// narrower than a 128-byte line, kd_x_pass) and once from a hypothetical tile-major layout (one contiguous block per tile):
//   the first figure is the ceiling of the fused x passes as they are laid out, the second what a device copy reaches.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro_gather.hip -o xp_tmp/micro/gather ;  run: xp_tmp/micro/gather [G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct __attribute__((aligned(16))) c16 { double re, im; };

template <int MODE, int RUN, int PAIRED>   // MODE 0: plane-strided (Ty as it is), 1: tile-major
__global__ __launch_bounds__(256) void gather(const c16* __restrict__ in, c16* __restrict__ out, int nplanes, size_t plane_stride, unsigned ntiles) {
    extern __shared__ c16 buf[];
    unsigned tile = blockIdx.x;
    if (PAIRED > 1 && blockIdx.x < (ntiles / (8 * PAIRED)) * (8 * PAIRED)) {
        const unsigned q = blockIdx.x / (8 * PAIRED), r = blockIdx.x % (8 * PAIRED);
        tile = q * (8 * PAIRED) + PAIRED * (r % 8) + r / 8;
    }
    const int tid = threadIdx.x, n = nplanes * RUN;
    for (int t = tid; t < n; t += 256) {
        const int p = t / RUN, e = t % RUN;
        const size_t off = MODE == 0 ? (size_t)p * plane_stride + (size_t)tile * RUN + e : ((size_t)tile * nplanes + p) * RUN + e;
        buf[t] = in[off];
    }
    __syncthreads();
    for (int t = tid; t < n; t += 256) {
        const int p = t / RUN, e = t % RUN;
        const size_t off = MODE == 0 ? (size_t)p * plane_stride + (size_t)tile * RUN + e : ((size_t)tile * nplanes + p) * RUN + e;
        c16 v = buf[(t * 7 + 3) % n];
        v.re += 1.0;
        out[off] = v;
    }
}

template <int RUN, int PAIRED> void run(const char* what, const c16* in, c16* out, int nplanes, size_t stride, size_t plane) {
    const unsigned ntiles = (unsigned)(plane / RUN);
    const size_t lds = (size_t)nplanes * RUN * 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) {
                if (mode == 0) hipLaunchKernelGGL((gather<0, RUN, PAIRED>), dim3(ntiles), dim3(256), lds, 0, in, out, nplanes, stride, ntiles);
                else hipLaunchKernelGGL((gather<1, RUN, PAIRED>), dim3(ntiles), dim3(256), lds, 0, in, out, nplanes, stride, ntiles);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double bytes = 2.0 * nplanes * RUN * 16 * (double)ntiles * 10;
        printf("%-34s %-14s %7.1f us per launch  %5.2f TB/s\n", what, mode ? "tile-major" : "plane-strided", best * 1e3 / 10, bytes / (best * 1e-3) / 1e12);
    }
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 384, a = G / 3;
    const size_t plane = (size_t)G * G, stride = plane + 8;
    const size_t n = (size_t)6 * a * stride;                  // up to two field groups
    c16 *in, *out;
    if (hipMalloc(&in, n * 16) != hipSuccess || hipMalloc(&out, n * 16) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(in, 0, n * 16);
    printf("G = %d: %d (c, kx) planes per field group, %.0f MB per field group\n", G, 3 * a, 3.0 * a * stride * 16 / 1e6);
    if (G <= 192) {
        run<8, 1>("forward pass (128-B runs)", in, out, 3 * a, stride, plane);
        run<4, 2>("adjoint pass (64-B runs, paired)", in, out, 6 * a, stride, plane);
    } else {
        run<4, 2>("forward pass (64-B runs, paired)", in, out, 3 * a, stride, plane);
        run<2, 4>("adjoint pass (32-B runs, paired)", in, out, 6 * a, stride, plane);
        run<8, 1>("(128-B runs, for reference)", in, out, 3 * a, stride, plane);
    }
    return 0;
}
