// L2 -> L1 stream microbenchmark in the engine's access pattern: every workgroup (8 waves, one per CU) streams the SAME
// buffer of `ws` bytes as 1 KB wave-loads (64 lanes x 16 B, fully coalesced), wave w taking fragments w, w + 8, ...,
// `depth` loads in flight per wave, no arithmetic.  What it answers: the per-CU rate the vector L1 sustains from L2 as a
// function of the working set shared by an XCD's CUs (4 MB of L2 per XCD) — the ceiling of every weights-direct kernel.
// Build: hipcc -O3 --offload-arch=gfx950 tools/l2_stream.hip -o tools/_l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512, 2) void stream_kernel(const u32x4* __restrict__ buf, size_t n_frag, int passes, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    for (int p = 0; p < passes; ++p) {
        for (size_t f = wave; f + 8 * (DEPTH - 1) < n_frag; f += 8 * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = buf[(f + 8 * d) * 64 + lane];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;  // keeps the loads alive
}

int main() {
    const size_t maxb = 64u << 20;
    u32x4* buf; unsigned* out;
    hipMalloc(&buf, maxb); hipMalloc(&out, 4);
    hipMemset(buf, 1, maxb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double mbs[] = {0.5, 1, 2, 3, 4, 5, 7, 10, 16, 32, 64};
    for (int depth : {4, 8}) {
        for (double mb : mbs) {
            const size_t ws = (size_t)(mb * 1048576), n_frag = ws / 1024;
            const int passes = (int)(256.0 * 1048576 / ws) + 1;  // ~256 MB per workgroup
            auto launch = [&] {
                if (depth == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(256), dim3(512), 0, 0, buf, n_frag, passes, out);
                else hipLaunchKernelGGL(stream_kernel<8>, dim3(256), dim3(512), 0, 0, buf, n_frag, passes, out);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double bytes = 256.0 * passes * (double)(n_frag / (8 * depth) * (8 * depth)) * 1024;
            printf("depth %d  working set %5.1f MB: %7.3f ms  %6.2f TB/s L2->L1  (%.1f GB/s per CU)\n", depth, mb, ms, bytes / ms / 1e9,
                   bytes / ms / 1e6 / 256);
        }
    }
    return 0;
}
