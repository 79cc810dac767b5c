// Sustained f16 MFMA rate of the device: every wave issues independent v_mfma_f32_32x32x16_f16 from registers only
// (no memory traffic), 2 waves per SIMD.  Prints TFLOP/s by hipEvents and the shader clock by s_memtime.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/_mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, unsigned long long* ticks, int iters) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
    f32x16 c[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 6; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256, iters = 4096;  // one 8-wave workgroup per CU
    float* out; unsigned long long* ticks;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&ticks, blocks * 8);
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    const double flops = (double)blocks * 8 * iters * 24 * 2.0 * 32 * 32 * 16;
    for (int burst : {0, 1, 5, 20, 60}) {
        for (int r = 0; r < burst; ++r) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, out, ticks, iters);
        hipEventRecord(ea);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, out, ticks, iters);
        hipEventRecord(eb);
        hipEventSynchronize(eb);
        float ms; hipEventElapsedTime(&ms, ea, eb);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("after %2d back-to-back launches: %.3f ms  %.0f TFLOP/s f16 dense  (%.0f / 3 = %.0f TF of f16x3 products); "
               "shader clock %.2f GHz, MFMA pipe %.0f %% busy by cycles\n", burst, ms, flops / (ms * 1e-3) / 1e12,
               flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 3e12, (double)h[blocks / 2] / (ms * 1e6),
               100.0 * (2.0 * iters * 24 * 32) / (double)h[blocks / 2]);
    }
    return 0;
}
