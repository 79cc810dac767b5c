// Instruction-fetch microbenchmark: how many cycles does STRAIGHT-LINE code cost per instruction as a function of the
// size of the code a CU (pair) cycles through?  The layer kernel is 75 KB of mostly once-through code per workgroup
// (prologue, LayerNorms, epilogues) around a 16 KB main loop; if a cyclic walk over more code than the instruction
// cache holds misses on every line, those phases run at the fetch rate, not the issue rate.
//   kernel<KB>: a loop of `iters` passes over KB x 1024 bytes of 8-byte VALU instructions (v_fma_f32, VOP3).
//   DEP = 1: one dependent chain (issue-latency-bound when the code is resident); DEP = 0: four independent chains.
// One 512-thread workgroup per CU (8 waves, like the layer kernel), 256 workgroups; cycles by s_memtime in wave 0.
// Build: hipcc -O3 --offload-arch=gfx950 tools/icache_probe.hip -o tools/_icache_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define STR2(x) #x
#define STR(x) STR2(x)

template <int KB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void walk_kernel(float* out, int iters, unsigned long long* cyc) {
    float a = (float)threadIdx.x, b = 1.0001f, c = a + 1.0f, d = a + 2.0f, e = a + 3.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KB == 8) asm volatile(".rept 256\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n .endr" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if constexpr (KB == 16) asm volatile(".rept 512\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n .endr" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if constexpr (KB == 24) asm volatile(".rept 768\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n .endr" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if constexpr (KB == 32) asm volatile(".rept 1024\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n .endr" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if constexpr (KB == 48) asm volatile(".rept 1536\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n .endr" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if constexpr (KB == 56) asm volatile(".rept 1792\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n .endr" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if constexpr (KB == 64) asm volatile(".rept 2048\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n .endr" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if constexpr (KB == 72) asm volatile(".rept 2304\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n .endr" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if constexpr (KB == 96) asm volatile(".rept 3072\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n .endr" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if constexpr (KB == 112) asm volatile(".rept 3584\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n .endr" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d + e;
}

template <int KB, int WAVES>
static void run(float* out, unsigned long long* cyc, int grid) {
    const int iters = (4096 / KB) > 4 ? (4096 / KB) : 4;  // ~4 MB of instruction bytes per wave
    unsigned long long* h = (unsigned long long*)malloc(grid * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((walk_kernel<KB, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((walk_kernel<KB, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, iters, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid; ++i) s += (double)h[i];
    const double n_inst = (double)iters * KB * 128.0;
    printf("code %3d KB, %d waves/WG, %4d WGs: %8.3f ms, %9.0f cycles per workgroup, %.2f cycles per instruction per wave (%d passes)\n",
           KB, WAVES, grid, ms, s / grid, s / grid / n_inst, iters);
    free(h);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int grid : {256, 512}) {
        run<8, 8>(out, cyc, grid); run<16, 8>(out, cyc, grid); run<24, 8>(out, cyc, grid); run<32, 8>(out, cyc, grid);
        run<48, 8>(out, cyc, grid); run<56, 8>(out, cyc, grid); run<64, 8>(out, cyc, grid); run<72, 8>(out, cyc, grid);
        run<96, 8>(out, cyc, grid); run<112, 8>(out, cyc, grid);
    }
    printf("-- one wave per workgroup (no sharing of fetched lines between waves)\n");
    run<16, 1>(out, cyc, 256); run<64, 1>(out, cyc, 256); run<96, 1>(out, cyc, 256); run<112, 1>(out, cyc, 256);
    return 0;
}
