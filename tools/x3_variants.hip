// Tile/slab variants of the f16x3 main loop (+ fp32 bias epilogue) — exploration only.
#include "common.h"
#include <vector>
using namespace icrec;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int WM, int WN, int TM, int TN, int HBK>
struct V {
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, THREADS = WM * WN * 64;
    static constexpr int CPR = HBK / 8;                  // 16-B chunks per row per slab
    static constexpr int HLD = HBK + 8;                  // padded LDS row (halfs)
    static constexpr int A_CH = BM * CPR / THREADS, B_CH = BN * CPR / THREADS;
    static constexpr size_t SMEM = (size_t)(2 * BM + 2 * BN) * HLD * 2;
    static_assert(BM * CPR % THREADS == 0 && BN * CPR % THREADS == 0, "");
};

template <class C, int ST, int STAG = 0>
__global__ __launch_bounds__(C::THREADS) void k(const _Float16* Ah, const _Float16* Al, int M, int K, const _Float16* Wh,
                                                const _Float16* Wl, int N, float* out, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    _Float16* Ahs = (_Float16*)sm; _Float16* Als = Ahs + C::BM * C::HLD; _Float16* Bhs = Als + C::BM * C::HLD; _Float16* Bls = Bhs + C::BN * C::HLD;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / (C::THREADS / 64 / (C::BM / (C::BM / 1)) ? 1 : 1);
    (void)wm;
    const int WNc = C::BN / (32 * (C::BN / 32 / ((C::THREADS / 64) / (C::BM / 32 / (C::BM / 32 / 1)) ? 1 : 1)));
    (void)WNc;
    // wave grid: derive from template ints
    constexpr int TMc = 2;  // fixed below via specialisation-free math
    (void)TMc;
    if (ST == 3 && blockIdx.x < 512 && ((blockIdx.x >> 3) & 1)) {  // stagger: half of the first wave of blocks starts late
        for (int i = 0; i < STAG; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / ntn, nt = bid % ntn;
    const int64_t m0 = (int64_t)mt * C::BM, n0 = (int64_t)nt * C::BN;
    // per-wave tile counts
    constexpr int WAVES = C::THREADS / 64;
    constexpr int TILES_M = C::BM / 32, TILES_N = C::BN / 32;
    constexpr int WN_ = (TILES_N >= 4 && WAVES >= 4) ? (WAVES == 8 ? 4 : 2) : 1;
    constexpr int WM_ = WAVES / WN_;
    constexpr int TM = TILES_M / WM_, TN = TILES_N / WN_;
    const int wmi = wave / WN_, wni = wave % WN_;
    f32x16 acc0[TM][TN], acc1[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0; acc1[i][j][e] = 0; }
    u32x4 pah[C::A_CH], pal[C::A_CH], pbh[C::B_CH], pbl[C::B_CH];
    auto load = [&](int slab) {
#pragma unroll
        for (int i = 0; i < C::A_CH; ++i) { int id = t + C::THREADS * i; int64_t row = m0 + id / C::CPR; row = row < M ? row : M - 1; int64_t off = row * K + slab * (C::CPR * 8) + (id % C::CPR) * 8; pah[i] = *(const u32x4*)(Ah + off); pal[i] = *(const u32x4*)(Al + off); }
#pragma unroll
        for (int i = 0; i < C::B_CH; ++i) { int id = t + C::THREADS * i; int64_t row = n0 + id / C::CPR; row = row < N ? row : N - 1; int64_t off = row * K + slab * (C::CPR * 8) + (id % C::CPR) * 8; pbh[i] = *(const u32x4*)(Wh + off); pbl[i] = *(const u32x4*)(Wl + off); }
    };
    const int nslab = K / (C::CPR * 8);
    load(0);
    const int r = lane & 31, h = lane >> 5;
    for (int s = 0; s < nslab; ++s) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < C::A_CH; ++i) { int id = t + C::THREADS * i; int o = (id / C::CPR) * C::HLD + (id % C::CPR) * 8; *(u32x4*)(Ahs + o) = pah[i]; *(u32x4*)(Als + o) = pal[i]; }
#pragma unroll
        for (int i = 0; i < C::B_CH; ++i) { int id = t + C::THREADS * i; int o = (id / C::CPR) * C::HLD + (id % C::CPR) * 8; *(u32x4*)(Bhs + o) = pbh[i]; *(u32x4*)(Bls + o) = pbl[i]; }
        __syncthreads();
        if (s + 1 < nslab) load(s + 1);
#pragma unroll
        for (int ks = 0; ks < C::CPR / 2; ++ks) {
            half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) { int o = ((wmi * TM + i) * 32 + r) * C::HLD + ks * 16 + h * 8; ah[i] = *(const half8*)(Ahs + o); al[i] = *(const half8*)(Als + o); }
#pragma unroll
            for (int j = 0; j < TN; ++j) { int o = ((wni * TN + j) * 32 + r) * C::HLD + ks * 16 + h * 8; bh[j] = *(const half8*)(Bhs + o); bl[j] = *(const half8*)(Bls + o); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc1[i][j], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int64_t col = n0 + (wni * TN + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + (wmi * TM + i) * 32 + acc_row(e, lane);
                if (row < M && col < N) { float v = fmaf(acc1[i][j][e], 1.0f / 2048.0f, acc0[i][j][e]); if (ST == 0 || ST == 3) out[row * N + col] = v; else if (ST == 1) __builtin_nontemporal_store(v, &out[row * N + col]); else if (v == 1.2345f) out[0] = v; }
            }
    }
}

template <class C, int ST, int STAG = 0>
float run(const char* name, const _Float16* Ah, const _Float16* Al, int M, int K, const _Float16* Wh, const _Float16* Wl, int N, float* out) {
    auto kern = k<C, ST, STAG>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
    int mt = (M + C::BM - 1) / C::BM, nt = (N + C::BN - 1) / C::BN;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(mt * nt), dim3(C::THREADS), C::SMEM, 0, Ah, Al, M, K, Wh, Wl, N, out, nt);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(mt * nt), dim3(C::THREADS), C::SMEM, 0, Ah, Al, M, K, Wh, Wl, N, out, nt);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipError_t e = hipGetLastError();
    printf("%-34s tile %3dx%3d thr %3d smem %6zu : %.3f ms  (%.0f TF-eq)%s\n", name, C::BM, C::BN, C::THREADS, C::SMEM, ms / 5, 2.0 * M * K * N / (ms / 5) / 1e9, e == hipSuccess ? "" : hipGetErrorString(e));
    return ms / 5;
}

int main() {
    const int M = 131150;
    for (int shape = 0; shape < 2; ++shape) {
        const int K = shape == 0 ? 384 : 1536, N = shape == 0 ? 1536 : 384;
        _Float16 *Ah, *Al, *Wh, *Wl; float* out;
        hipMalloc(&Ah, (size_t)M * K * 2); hipMalloc(&Al, (size_t)M * K * 2); hipMalloc(&Wh, (size_t)N * K * 2); hipMalloc(&Wl, (size_t)N * K * 2); hipMalloc(&out, (size_t)M * N * 4);
        std::vector<_Float16> g((size_t)M * K);
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
        for (auto& v : g) v = (_Float16)(float)((rnd() + rnd() + rnd() + rnd() - 2.0) * 1.7);
        hipMemcpy(Ah, g.data(), g.size() * 2, hipMemcpyHostToDevice); hipMemcpy(Al, g.data(), g.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(Wh, g.data(), (size_t)N * K * 2, hipMemcpyHostToDevice); hipMemcpy(Wl, g.data() + 12345, (size_t)N * K * 2, hipMemcpyHostToDevice);
        printf("== M=%d K=%d N=%d ==\n", M, K, N);
        run<V<2, 4, 2, 1, 64>, 0>("8w 128x128 BK64 normal store", Ah, Al, M, K, Wh, Wl, N, out);
        run<V<2, 4, 2, 1, 64>, 3, 1>("  + stagger 1 x s_sleep(127)", Ah, Al, M, K, Wh, Wl, N, out);
        run<V<2, 4, 2, 1, 64>, 3, 2>("  + stagger 2", Ah, Al, M, K, Wh, Wl, N, out);
        run<V<2, 4, 2, 1, 64>, 3, 3>("  + stagger 3", Ah, Al, M, K, Wh, Wl, N, out);
        run<V<2, 4, 2, 1, 64>, 3, 5>("  + stagger 5", Ah, Al, M, K, Wh, Wl, N, out);
        run<V<2, 4, 2, 1, 64>, 2>("8w 128x128 BK64 no store", Ah, Al, M, K, Wh, Wl, N, out);
        hipFree(Ah); hipFree(Al); hipFree(Wh); hipFree(Wl); hipFree(out);
    }
    return 0;
}
