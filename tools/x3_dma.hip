// Prototype: f16x3 GEMM main loop with LDS-DMA staging (global_load_lds_dwordx4), 3-stage ring, one raw
// barrier per K-tile, counted vmcnt.  8 waves (4M x 2N), tile 256 x 128, 64 x 64 per wave, BK = 32 halfs.
// LDS rows are 64 B unpadded; 16-B chunk c of row r lives at chunk position c ^ ((r>>2)&3) (applied on the
// DMA source address and on the ds_read address).  Checked bit-for-bit against the library loop.
#include "gemm_x3.h"
#include <vector>
#include <algorithm>
using namespace icrec;
__device__ unsigned long long g_stamp[4 * 65536];

typedef TileCfg<2, 4, 2, 1> C8;
__global__ __launch_bounds__(512) void k_ref(const _Float16* Ah, const _Float16* Al, int M, int K, const _Float16* Wh,
                                             const _Float16* Wl, int N, float* out, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / 4, wn = wave % 4;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(bid / ntn) * 128, n0 = (int64_t)(bid % ntn) * 128;
    f32x16 a0[2][1], a1[2][1];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    tile_gemm_h<C8>(a0, a1, Ah, Al, m0, M, Wh, Wl, n0, N, K, (_Float16*)sm);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x < 65536) { g_stamp[2 * blockIdx.x] = c1 - c0; g_stamp[2 * blockIdx.x + 1] = r1 - r0; }
    const int64_t col = n0 + wn * 32 + (lane & 31);
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = m0 + (wm * 2 + i) * 32 + acc_row(e, lane);
            if (row < M && col < N) out[row * N + col] = fmaf(a1[i][0][e], LO_UNSCALE, a0[i][0][e]);
        }
}

constexpr int DBM = 256, DBN = 128, DBK = 32;                 // halfs
constexpr int A_PLANE = DBM * DBK * 2, B_PLANE = DBN * DBK * 2;  // bytes per plane per stage
constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;               // 49152
constexpr int NSTAGE = 3;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// one wave issues its 6 pieces (1 KiB each = 16 rows x 64 B) of K-tile `kt` into `stage`
__device__ __forceinline__ void dma_tile(char* stage, const _Float16* Ah, const _Float16* Al, int64_t m0, int M,
                                         const _Float16* Wh, const _Float16* Wl, int64_t n0, int N, int K, int kt, int wave, int lane) {
    const int rsub = lane >> 2;                         // row inside the 16-row piece
    const int csrc = (lane & 3) ^ ((lane >> 4) & 3);    // source chunk for LDS chunk position lane&3
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int p = wave * 6 + i;                     // 0..47: [A_hi 16][A_lo 16][B_hi 8][B_lo 8]
        const _Float16* plane; int64_t row0; int64_t rows; int rb; int off;
        if (p < 16)      { plane = Ah; row0 = m0; rows = M; rb = p;      off = 0; }
        else if (p < 32) { plane = Al; row0 = m0; rows = M; rb = p - 16; off = A_PLANE; }
        else if (p < 40) { plane = Wh; row0 = n0; rows = N; rb = p - 32; off = 2 * A_PLANE; }
        else             { plane = Wl; row0 = n0; rows = N; rb = p - 40; off = 2 * A_PLANE + B_PLANE; }
        int64_t row = row0 + rb * 16 + rsub;
        row = row < rows ? row : rows - 1;
        const _Float16* src = plane + row * K + kt * DBK + csrc * 8;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(stage + off + rb * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ half8 lds_frag(const char* plane, int row, int c) {
    return *(const half8*)(plane + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
}

__global__ __launch_bounds__(512, 2) void k_dma(const _Float16* Ah, const _Float16* Al, int M, int K, const _Float16* Wh,
                                                const _Float16* Wl, int N, float* out, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(bid / ntn) * DBM, n0 = (int64_t)(bid % ntn) * DBN;
    f32x16 a0[2][2], a1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { a0[i][j][e] = 0; a1[i][j][e] = 0; }
    const int nt = K / DBK;  // >= 2
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    dma_tile(sm, Ah, Al, m0, M, Wh, Wl, n0, N, K, 0, wave, lane);
    dma_tile(sm + STAGE, Ah, Al, m0, M, Wh, Wl, n0, N, K, 1, wave, lane);
    for (int kt = 0; kt < nt; ++kt) {
        // tile kt landed (this wave's pieces), then everybody's; stage (kt+2)%3 is free after the barrier
        if (kt + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nt) dma_tile(sm + ((kt + 2) % NSTAGE) * STAGE, Ah, Al, m0, M, Wh, Wl, n0, N, K, kt + 2, wave, lane);
        const char* st = sm + (kt % NSTAGE) * STAGE;
        const char* pAh = st; const char* pAl = st + A_PLANE; const char* pBh = st + 2 * A_PLANE; const char* pBl = pBh + B_PLANE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wm * 2 + i) * 32 + r;
                ah[i] = lds_frag(pAh, row, 2 * ks + h); al[i] = lds_frag(pAl, row, 2 * ks + h);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (wn * 2 + j) * 32 + r;
                bh[j] = lds_frag(pBh, row, 2 * ks + h); bl[j] = lds_frag(pBl, row, 2 * ks + h);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    a0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], a0[i][j], 0, 0, 0);
                    a1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], a1[i][j], 0, 0, 0);
                    a1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], a1[i][j], 0, 0, 0);
                }
        }
    }
    {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0 && blockIdx.x < 65536) { g_stamp[2 * blockIdx.x] = c1 - c0; g_stamp[2 * blockIdx.x + 1] = r1 - r0; }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + (wn * 2 + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + (wm * 2 + i) * 32 + acc_row(e, lane);
                if (row < M && col < N) out[row * N + col] = fmaf(a1[i][j][e], LO_UNSCALE, a0[i][j][e]);
            }
    }
}

template <class KERN>
float timeit(KERN kern, size_t smem, int threads, int bm, int bn, const _Float16* Ah, const _Float16* Al, int M, int K, const _Float16* Wh, const _Float16* Wl, int N, float* out) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    int mt = (M + bm - 1) / bm, nt = (N + bn - 1) / bn;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(mt * nt), dim3(threads), smem, 0, Ah, Al, M, K, Wh, Wl, N, out, nt);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(mt * nt), dim3(threads), smem, 0, Ah, Al, M, K, Wh, Wl, N, out, nt);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("ERR %s\n", hipGetErrorString(e));
    {
        std::vector<unsigned long long> h(2 * 4096);
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamp), h.size() * 8);
        std::vector<double> clk, cyc;
        for (int i = 1000; i < 3000; ++i) if (h[2 * i + 1]) { clk.push_back((double)h[2 * i] / h[2 * i + 1] * 0.1); cyc.push_back((double)h[2 * i]); }
        std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
        if (!clk.empty()) printf("    in-kernel clock median %.2f GHz; main-loop cycles per block median %.0f\n", clk[clk.size() / 2], cyc[cyc.size() / 2]);
    }
    return ms / 5;
}

int main() {
    const int M = 131150;
    for (int shape = 0; shape < 2; ++shape) {
        const int K = shape == 0 ? 384 : 1536, N = shape == 0 ? 1536 : 384;
        _Float16 *Ah, *Al, *Wh, *Wl; float *o1, *o2;
        hipMalloc(&Ah, (size_t)M * K * 2); hipMalloc(&Al, (size_t)M * K * 2); hipMalloc(&Wh, (size_t)N * K * 2); hipMalloc(&Wl, (size_t)N * K * 2);
        hipMalloc(&o1, (size_t)M * N * 4); hipMalloc(&o2, (size_t)M * N * 4);
        hipMemset(o2, 0xFF, (size_t)M * N * 4);
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
        float t_ref = timeit(k_ref, SmemH<C8>::BYTES, 512, 128, 128, Ah, Al, M, K, Wh, Wl, N, o1);
        float t_dma = timeit(k_dma, (size_t)NSTAGE * STAGE, 512, DBM, DBN, Ah, Al, M, K, Wh, Wl, N, o2);
        std::vector<float> h1((size_t)1 << 22), h2((size_t)1 << 22);
        size_t bad = 0;
        for (size_t off : {(size_t)0, (size_t)M * N / 2, (size_t)M * N - h1.size()}) {
            hipMemcpy(h1.data(), o1 + off, h1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), o2 + off, h2.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < h1.size(); ++i) if (memcmp(&h1[i], &h2[i], 4)) ++bad;
        }
        printf("K=%d N=%d: ref %.3f ms (%.0f TF-eq) | LDS-DMA %.3f ms (%.0f TF-eq) | mismatching outputs: %zu\n", K, N, t_ref, 2.0 * M * K * N / t_ref / 1e9, t_dma, 2.0 * M * K * N / t_dma / 1e9, bad);
        hipFree(Ah); hipFree(Al); hipFree(Wh); hipFree(Wl); hipFree(o1); hipFree(o2);
    }
    return 0;
}
