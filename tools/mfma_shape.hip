// MFMA shape A/B under the power cap (MI355X_MICROARCH.md 'DVFS give-back' item 7): the same
// FLOPs per wave and the same 96 accumulator registers per wave issued as
//   S32: 6 tiles of v_mfma_f32_32x32x16_f16      (24 per iteration, 32 cycles each)
//   S16: 24 tiles of v_mfma_f32_16x16x32_f16     (96 per iteration of half the FLOPs each, 16 cycles each)
// from registers only, on RANDOM operands (the clock the chip holds depends on the data), one or two waves per SIMD,
// timed interleaved in ONE process (rule 24).  Operand variety: each tile multiplies one of 4 distinct A fragments
// by one of 2 distinct B fragments (as the engine's 3 x 2 wave tile does), so operand-register traffic is real.
// Prints TFLOP/s by hipEvents and the in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape.hip -o tools/_mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ half8 rnd_frag(unsigned seed, bool zero) {
    half8 v;
    for (int j = 0; j < 8; ++j) {
        const unsigned h = hash32(seed * 8u + j);
        v[j] = zero ? (_Float16)0.0f : (_Float16)(((int)(h & 0xFFFF) - 32768) * (1.0f / 32768.0f));  // uniform [-1, 1)
    }
    return v;
}

template <int SHAPE, int WPS>
__global__ __launch_bounds__(256 * WPS, WPS) void mfma_loop(float* out, unsigned long long* ticks, int iters, int zero) {
    half8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = rnd_frag((blockIdx.x * 1024u + threadIdx.x) * 8u + i, zero);
    for (int i = 0; i < 2; ++i) b[i] = rnd_frag((blockIdx.x * 1024u + threadIdx.x) * 8u + 4 + i, zero);
    float s = 0.0f;
    unsigned long long t0, t1, r0, r1;
    if (SHAPE == 32) {
        f32x16 c[6];
        for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.0f;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + rep) & 3], b[i & 1], c[i], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
    } else {
        f32x4 c[24];
        for (int i = 0; i < 24; ++i) for (int e = 0; e < 4; ++e) c[i][e] = 0.0f;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)   // 2 x 24 MFMAs of 16x16x32 = the FLOPs of 4 x 6 MFMAs of 32x32x16
#pragma unroll
                for (int i = 0; i < 24; ++i)
                    c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + rep) & 3], b[i & 1], c[i], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 24; ++i) for (int e = 0; e < 4; ++e) s += c[i][e];
    }
    out[blockIdx.x * 256 * WPS + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int blocks = 256, iters = 4096;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&ticks, blocks * 16);
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    struct Var { const char* name; int shape, wps, zero; };
    const Var vars[] = {{"32x32x16 2w/SIMD random", 32, 2, 0}, {"16x16x32 2w/SIMD random", 16, 2, 0},
                        {"32x32x16 1w/SIMD random", 32, 1, 0}, {"16x16x32 1w/SIMD random", 16, 1, 0},
                        {"32x32x16 2w/SIMD zeros ", 32, 2, 1}, {"16x16x32 2w/SIMD zeros ", 16, 2, 1}};
    auto launch = [&](const Var& v) {
        if (v.shape == 32 && v.wps == 2) hipLaunchKernelGGL((mfma_loop<32, 2>), dim3(blocks), dim3(512), 0, 0, out, ticks, iters, v.zero);
        if (v.shape == 16 && v.wps == 2) hipLaunchKernelGGL((mfma_loop<16, 2>), dim3(blocks), dim3(512), 0, 0, out, ticks, iters, v.zero);
        if (v.shape == 32 && v.wps == 1) hipLaunchKernelGGL((mfma_loop<32, 1>), dim3(blocks), dim3(256), 0, 0, out, ticks, iters, v.zero);
        if (v.shape == 16 && v.wps == 1) hipLaunchKernelGGL((mfma_loop<16, 1>), dim3(blocks), dim3(256), 0, 0, out, ticks, iters, v.zero);
    };
    // warm the chip into its loaded state first (the first launches after idle run at a different clock)
    for (int r = 0; r < 40; ++r) launch(vars[0]);
    hipDeviceSynchronize();
    for (int round = 0; round < 4; ++round) {
        for (const Var& v : vars) {
            for (int r = 0; r < 6; ++r) launch(v);  // settle on this variant's clock
            hipEventRecord(ea);
            for (int r = 0; r < 4; ++r) launch(v);
            hipEventRecord(eb);
            hipEventSynchronize(eb);
            float ms; hipEventElapsedTime(&ms, ea, eb);
            ms /= 4;
            std::vector<unsigned long long> h(2 * blocks);
            hipMemcpy(h.data(), ticks, blocks * 16, hipMemcpyDeviceToHost);
            std::vector<double> clk(blocks);
            for (int i = 0; i < blocks; ++i) clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
            std::sort(clk.begin(), clk.end());
            const double flops = (double)blocks * 4 * v.wps * iters * 24 * 2.0 * 32 * 32 * 16;
            printf("round %d  %s: %.3f ms  %.0f TFLOP/s f16 dense (/3 = %.0f TF f16x3)  in-kernel clock %.2f GHz\n", round, v.name, ms,
                   flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 3e12, clk[blocks / 2]);
        }
    }
    return 0;
}
