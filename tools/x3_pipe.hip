// Prototype: pipelined f16x3 main loop (2 LDS stages, 1 barrier/slab, depth-2 register prefetch,
// double-buffered fragments), checked against the straightforward loop.  Exploration only.
#include "gemm_x3.h"
#include <vector>
using namespace icrec;

typedef TileCfg<2, 4, 2, 1> C8;  // 128x128, 8 waves of 64x32

// ---------------------------------------------------------------- reference (current library loop)
__global__ __launch_bounds__(512) void k_ref(const _Float16* Ah, const _Float16* Al, int M, int K, const _Float16* Wh,
                                             const _Float16* Wl, int N, float* out, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / 4, wn = wave % 4;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(bid / ntn) * 128, n0 = (int64_t)(bid % ntn) * 128;
    f32x16 a0[2][1], a1[2][1];
    tile_gemm_h<C8>(a0, a1, Ah, Al, m0, M, Wh, Wl, n0, N, K, (_Float16*)sm);
    const int64_t col = n0 + wn * 32 + (lane & 31);
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = m0 + (wm * 2 + i) * 32 + acc_row(e, lane);
            if (row < M && col < N) out[row * N + col] = fmaf(a1[i][0][e], LO_UNSCALE, a0[i][0][e]);
        }
}

// ---------------------------------------------------------------- pipelined
struct Frag { half8 ah[2], al[2], bh, bl; };

__device__ __forceinline__ void load_frag(Frag& f, const _Float16* st, int wm, int wn, int r, int h, int ks) {
    const _Float16* Ahs = st; const _Float16* Als = st + 128 * HLD; const _Float16* Bhs = st + 2 * 128 * HLD; const _Float16* Bls = st + 3 * 128 * HLD;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int o = ((wm * 2 + i) * 32 + r) * HLD + ks * 16 + h * 8;
        f.ah[i] = *(const half8*)(Ahs + o); f.al[i] = *(const half8*)(Als + o);
    }
    const int o = (wn * 32 + r) * HLD + ks * 16 + h * 8;
    f.bh = *(const half8*)(Bhs + o); f.bl = *(const half8*)(Bls + o);
}
__device__ __forceinline__ void mma_frag(f32x16 (&a0)[2], f32x16 (&a1)[2], const Frag& f) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a0[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh, a0[i], 0, 0, 0);
        a1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl, a1[i], 0, 0, 0);
        a1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh, a1[i], 0, 0, 0);
    }
}
struct Pre { u32x4 ah[2], al[2], bh[2], bl[2]; };
__device__ __forceinline__ void gload(Pre& p, const _Float16* Ah, const _Float16* Al, int64_t m0, int M, const _Float16* Wh,
                                      const _Float16* Wl, int64_t n0, int N, int K, int slab, int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = t + 512 * i;
        int64_t row = m0 + (id >> 3); row = row < M ? row : M - 1;
        const int64_t off = row * K + slab * HBK + (id & 7) * 8;
        p.ah[i] = *(const u32x4*)(Ah + off); p.al[i] = *(const u32x4*)(Al + off);
        int64_t rw = n0 + (id >> 3); rw = rw < N ? rw : N - 1;
        const int64_t ofw = rw * K + slab * HBK + (id & 7) * 8;
        p.bh[i] = *(const u32x4*)(Wh + ofw); p.bl[i] = *(const u32x4*)(Wl + ofw);
    }
}
__device__ __forceinline__ void lstore(const Pre& p, _Float16* st, int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = t + 512 * i;
        const int o = (id >> 3) * HLD + (id & 7) * 8;
        *(u32x4*)(st + o) = p.ah[i]; *(u32x4*)(st + 128 * HLD + o) = p.al[i];
        *(u32x4*)(st + 2 * 128 * HLD + o) = p.bh[i]; *(u32x4*)(st + 3 * 128 * HLD + o) = p.bl[i];
    }
}

// one slab: MMA on `cur` (frags double-buffered), then park slab s+1 (regs `nxt_regs`) into `nxt` stage
__device__ __forceinline__ void slab_step(f32x16 (&a0)[2], f32x16 (&a1)[2], const _Float16* cur, _Float16* nxt, const Pre& nxt_regs,
                                          bool store, int wm, int wn, int r, int h, int t) {
    Frag f0, f1;
#define SB __builtin_amdgcn_sched_barrier(0)
    load_frag(f0, cur, wm, wn, r, h, 0);
    SB;
    load_frag(f1, cur, wm, wn, r, h, 1);
    SB;
    mma_frag(a0, a1, f0);
    SB;
    load_frag(f0, cur, wm, wn, r, h, 2);
    SB;
    mma_frag(a0, a1, f1);
    SB;
    load_frag(f1, cur, wm, wn, r, h, 3);
    SB;
    mma_frag(a0, a1, f0);
    SB;
    if (store) lstore(nxt_regs, nxt, t);
    SB;
    mma_frag(a0, a1, f1);
    SB;
}

__global__ __launch_bounds__(512, 2) void k_pipe(const _Float16* Ah, const _Float16* Al, int M, int K, const _Float16* Wh,
                                                 const _Float16* Wl, int N, float* out, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    _Float16* st0 = (_Float16*)sm;
    _Float16* st1 = st0 + 4 * 128 * HLD;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / 4, wn = wave % 4, r = lane & 31, h = lane >> 5;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(bid / ntn) * 128, n0 = (int64_t)(bid % ntn) * 128;
    f32x16 a0[2], a1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { a0[i][e] = 0; a1[i][e] = 0; }
    const int nslab = K / HBK;  // even, >= 2
    Pre X, Y;
    gload(X, Ah, Al, m0, M, Wh, Wl, n0, N, K, 0, t);
    gload(Y, Ah, Al, m0, M, Wh, Wl, n0, N, K, 1, t);
    lstore(X, st0, t);
    __syncthreads();
    // invariant at loop top (s even): st0 holds slab s, Y holds slab s+1 (in flight or landed)
    for (int s = 0; s < nslab; s += 2) {
        if (s + 2 < nslab) gload(X, Ah, Al, m0, M, Wh, Wl, n0, N, K, s + 2, t);
        slab_step(a0, a1, st0, st1, Y, true, wm, wn, r, h, t);       // MMA slab s, park slab s+1 in st1
        __syncthreads();
        if (s + 3 < nslab) gload(Y, Ah, Al, m0, M, Wh, Wl, n0, N, K, s + 3, t);
        slab_step(a0, a1, st1, st0, X, s + 2 < nslab, wm, wn, r, h, t);  // MMA slab s+1, park slab s+2 in st0
        __syncthreads();
    }
    const int64_t col = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = m0 + (wm * 2 + i) * 32 + acc_row(e, lane);
            if (row < M && col < N) out[row * N + col] = fmaf(a1[i][e], LO_UNSCALE, a0[i][e]);
        }
}

template <class KERN>
float timeit(KERN kern, size_t smem, int threads, const _Float16* Ah, const _Float16* Al, int M, int K, const _Float16* Wh, const _Float16* Wl, int N, float* out) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    int mt = (M + 127) / 128, nt = (N + 127) / 128;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(mt * nt), dim3(threads), smem, 0, Ah, Al, M, K, Wh, Wl, N, out, nt);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(mt * nt), dim3(threads), smem, 0, Ah, Al, M, K, Wh, Wl, N, out, nt);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("ERR %s\n", hipGetErrorString(e));
    return ms / 5;
}

int main() {
    const int M = 131150;
    for (int shape = 0; shape < 2; ++shape) {
        const int K = shape == 0 ? 384 : 1536, N = shape == 0 ? 1536 : 384;
        _Float16 *Ah, *Al, *Wh, *Wl; float *o1, *o2;
        hipMalloc(&Ah, (size_t)M * K * 2); hipMalloc(&Al, (size_t)M * K * 2); hipMalloc(&Wh, (size_t)N * K * 2); hipMalloc(&Wl, (size_t)N * K * 2);
        hipMalloc(&o1, (size_t)M * N * 4); hipMalloc(&o2, (size_t)M * N * 4);
        std::vector<_Float16> g((size_t)M * K);
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
        for (auto& v : g) v = (_Float16)(float)((rnd() + rnd() + rnd() + rnd() - 2.0) * 1.7);
        hipMemcpy(Ah, g.data(), g.size() * 2, hipMemcpyHostToDevice);
        for (auto& v : g) v = (_Float16)(float)(rnd() - 0.5);
        hipMemcpy(Al, g.data(), g.size() * 2, hipMemcpyHostToDevice);
        for (size_t i = 0; i < (size_t)N * K; ++i) g[i] = (_Float16)(float)((rnd() - 0.5) * 0.2);
        hipMemcpy(Wh, g.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
        for (size_t i = 0; i < (size_t)N * K; ++i) g[i] = (_Float16)(float)((rnd() - 0.5));
        hipMemcpy(Wl, g.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
        float t_ref = timeit(k_ref, SmemH<C8>::BYTES, 512, Ah, Al, M, K, Wh, Wl, N, o1);
        float t_pipe = timeit(k_pipe, 2 * SmemH<C8>::BYTES, 512, Ah, Al, M, K, Wh, Wl, N, o2);
        std::vector<float> h1((size_t)1 << 22), h2((size_t)1 << 22);
        size_t off = (size_t)M * N - h1.size();
        hipMemcpy(h1.data(), o1 + off, h1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), o2 + off, h2.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < h1.size(); ++i) if (h1[i] != h2[i]) ++bad;
        hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < h1.size(); ++i) if (h1[i] != h2[i]) ++bad;
        printf("K=%d N=%d: ref %.3f ms (%.0f TF-eq) | pipelined %.3f ms (%.0f TF-eq) | mismatching outputs: %zu\n", K, N, t_ref, 2.0 * M * K * N / t_ref / 1e9, t_pipe, 2.0 * M * K * N / t_pipe / 1e9, bad);
        hipFree(Ah); hipFree(Al); hipFree(Wh); hipFree(Wl); hipFree(o1); hipFree(o2);
    }
    return 0;
}
