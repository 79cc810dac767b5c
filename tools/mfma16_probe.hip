// Lane maps of v_mfma_f32_16x16x32_f16 checked with exact integer data (A asymmetric, B asymmetric):
//   A: lane l holds A[row l&15][k = 8(l>>4) + j]   B: lane l holds B[k = 8(l>>4) + j][col l&15]
//   D: lane l holds D[row 4(l>>4) + reg][col l&15]
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma16_probe.hip -o tools/_mfma16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline float Aval(int r, int k) { return (float)((r * 3 + k * 5) % 7 - 3); }
__host__ __device__ inline float Bval(int k, int c) { return (float)((k * 2 + c * 11 + (k * c) % 3) % 5 - 2); }
__global__ void probe(float* out) {
    const int l = threadIdx.x, c = l & 15, g = l >> 4;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)Aval(c, 8 * g + j); b[j] = (_Float16)Bval(8 * g + j, c); }
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + c] = d[r];
}
int main() {
    float* o; hipMalloc(&o, 256 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, o);
    float h[256]; hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
        float s = 0; for (int k = 0; k < 32; ++k) s += Aval(r, k) * Bval(k, c);
        if (s != h[r * 16 + c]) ++bad;
    }
    printf("mfma_f32_16x16x32_f16 lane maps: %s (%d of 256 wrong)\n", bad ? "MISMATCH" : "OK", bad);
    return bad != 0;
}
