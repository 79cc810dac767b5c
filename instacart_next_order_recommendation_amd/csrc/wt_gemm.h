// wt_gemm.h — the "weights-direct" f16x3 GEMM engine of the encoder (gfx950).
//
// Every linear layer of the encoder is out[T, N] = X[T, K] . W[N, K]^T with a STATIC weight matrix and M = T
// tokens.  The engine computes the transposed tile  out^T = W . X^T  so that
//   * the weight fragments are the A operand of v_mfma_f32_32x32x16_f16.  Each wave owns distinct output
//     features, so no other wave of the workgroup needs its weight fragments: they are loaded straight from
//     global memory (L2) into registers and never touch LDS.  Weights are pre-packed once at encoder creation
//     in FRAGMENT ORDER — the 1 KB a wave needs for one (32-feature tile, 16-deep k-step, plane) is contiguous,
//     lane l's 16 bytes at l * 16 — so every weight load is one fully coalesced global_load_dwordx4;
//   * the token rows are the B operand: a [BM tokens x 64 k] slab of the activation planes is staged through
//     LDS (XOR-swizzled 128-B rows, conflict-free ds_read_b128) and shared by the four waves;
//   * the accumulators come out with the TOKEN on the lane and 4 consecutive FEATURES in consecutive
//     registers: epilogues store 16 B (fp32) or 8 B (f16 planes) per lane, a GELU'd tile can be handed to the
//     next GEMM as 8-byte LDS writes, and LayerNorm statistics never need a cross-lane transpose.
//
// Arithmetic (fp32-accurate on the f16 matrix cores): every fp32 operand is carried as two f16 planes
//     hi = f16(x * S)        lo = f16(x * S - hi)                 S = 16 (activations), 1024 (weights)
// and a product is   x*w*S_a*S_w ~= hi_w*hi_x + lo_w*hi_x + hi_w*lo_x   — three MFMAs into ONE fp32
// accumulator, in that order, k-steps ascending; the result is scaled back by 2^-14 in the epilogue.  f16 x f16
// products are exact in the fp32 accumulator; the dropped lo*lo term and the split residuals are <= 3 * 2^-22
// relative per product.  The power-of-two scales keep `lo` out of the f16 subnormal range for every value that
// matters (|x| >= 2^-7 / S: absolute error below 2^-25 / S otherwise) and are exact; |x| * S is clamped to the
// f16 range (|activation| <= 4094, |weight| <= 63.9: far outside what LayerNorm-bounded BERT tensors reach).
// One accumulator set instead of two (the previous round's hi/cross split) halves the accumulator registers,
// which is what lets a wave own a 96-feature x 64-token output tile.
#pragma once
#include "common.h"
#include "gemm_x3.h"  // half8 / half4 / u32x4

namespace icrec {

#ifdef __HIPCC__

constexpr float WT_SA = 16.0f;                          // activation plane scale
constexpr float WT_SW = 1024.0f;                        // weight plane scale
constexpr float WT_UNSCALE = 1.0f / (16.0f * 1024.0f);  // 2^-14
constexpr float F16_MAX = 65504.0f;

__device__ __forceinline__ void split_scaled(float x, float scale, _Float16& hi, _Float16& lo) {
    const float s = __builtin_amdgcn_fmed3f(x * scale, -F16_MAX, F16_MAX);
    hi = (_Float16)s;
    lo = (_Float16)(s - (float)hi);
}
// Activations are split on the fly (LayerNorm / GELU / attention epilogues), so their split is the cheap form:
//   s = 16 x;  hi = s with the low 13 mantissa bits cleared (exactly an 11-bit value, i.e. an f16 for normal
//   magnitudes);  lo = s - hi (exact in fp32);  both converted with v_cvt_pkrtz_f16_f32, two elements per instruction
// = 3 VALU per element instead of 6.  hi + lo still carries s to 2^-21 relative (lo is rounded toward zero: one ulp of
// an 11-bit residual); below the f16 normal range (|x| < 3.8e-6) the absolute error is < 2^-24 / 16; beyond it
// (|x| > 4094) the conversion saturates at +-65504 instead of overflowing.  Every producer of activation planes uses
// this one function, so a tensor's planes are the same bits whichever kernel wrote them.
typedef _Float16 half2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_prescaled(float s0, float s1, half2w& hi, half2w& lo) {
    const float h0 = __uint_as_float(__float_as_uint(s0) & 0xFFFFE000u);
    const float h1 = __uint_as_float(__float_as_uint(s1) & 0xFFFFE000u);
    hi = __builtin_bit_cast(half2w, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    lo = __builtin_bit_cast(half2w, __builtin_amdgcn_cvt_pkrtz(s0 - h0, s1 - h1));
}
// four consecutive elements -> the 8-byte hi and lo pieces the epilogues store
__device__ __forceinline__ void split_act4(const f32x4& v, half4& hi, half4& lo) {
    half2w a, b, c, d;
    split_pair_prescaled(v[0] * WT_SA, v[1] * WT_SA, a, b);
    split_pair_prescaled(v[2] * WT_SA, v[3] * WT_SA, c, d);
    hi = half4{a[0], a[1], c[0], c[1]};
    lo = half4{b[0], b[1], d[0], d[1]};
}
__device__ __forceinline__ void split_act(float x, _Float16& hi, _Float16& lo) {
    half2w a, b;
    split_pair_prescaled(x * WT_SA, 0.0f, a, b);
    hi = a[0];
    lo = b[0];
}

// ---------------------------------------------------------------- packed weights
// W [N, K] fp32 row-major -> fragment order: fragment (nt, ks, plane) = 512 halfs at ((nt * K/16 + ks) * 2 + plane) * 512,
// lane l = (h << 5 | r) holds W[nt*32 + r][ks*16 + 8h .. +7] (the A-operand map of v_mfma_f32_32x32x16_f16).
constexpr int WT_FRAG = 512;  // halfs per fragment (64 lanes x 8)
__device__ __forceinline__ size_t wt_frag_off(int nt, int ks, int KS) { return ((size_t)nt * KS + ks) * (2 * WT_FRAG); }

// ---------------------------------------------------------------- activation slab ring (LDS)
// A slab = BM = 32 * TTW token rows x 64 k of both planes; row = 128 B per plane, 16-B chunk c of row `row` is
// stored at chunk c ^ ((row >> 1) & 7): the 16 lanes of a ds_read_b128 group land on 16 distinct 16-B slots
// of the 256-B bank row.
template <int TTW>
struct XRing {
    static constexpr int BM = 32 * TTW;
    static constexpr int PLANE_BYTES = BM * 128;
    static constexpr int STAGE_BYTES = 2 * PLANE_BYTES;
    static constexpr int BYTES = 2 * STAGE_BYTES;
};

template <int TTW>
__device__ __forceinline__ void x_load(u32x4 (&xr)[2 * TTW], const _Float16* __restrict__ Xh,
                                       const _Float16* __restrict__ Xl, int64_t m0, int64_t T, int K, int slab) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < TTW; ++i) {
        const int id = t + 256 * i;
        int64_t row = m0 + (id >> 3);
        row = row < T ? row : T - 1;
        const int64_t off = row * K + slab * 64 + (id & 7) * 8;
        xr[2 * i] = *reinterpret_cast<const u32x4*>(Xh + off);
        xr[2 * i + 1] = *reinterpret_cast<const u32x4*>(Xl + off);
    }
}

template <int TTW>
__device__ __forceinline__ void x_store(const u32x4 (&xr)[2 * TTW], char* stage) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < TTW; ++i) {
        const int id = t + 256 * i;
        const int row = id >> 3, c = id & 7;
        const int pos = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
        *reinterpret_cast<u32x4*>(stage + pos) = xr[2 * i];
        *reinterpret_cast<u32x4*>(stage + XRing<TTW>::PLANE_BYTES + pos) = xr[2 * i + 1];
    }
}

// B fragment of token tile tt, k-step j (0..3) of the slab: lane (r, h) -> token tt*32 + r, k = 16 j + 8 h .. +7
__device__ __forceinline__ half8 x_frag(const char* plane, int tt, int j, int r, int h) {
    const int row = tt * 32 + r;
    return *reinterpret_cast<const half8*>(plane + row * 128 + (((2 * j + h) ^ ((row >> 1) & 7)) << 4));
}

// ---------------------------------------------------------------- one k-step of MFMAs
// acc[i][tt] += W_i . X_tt^T over 16 k: (w_hi, x_hi), (w_lo, x_hi), (w_hi, x_lo) — the canonical order every
// kernel of the engine uses, so a token's result is independent of which kernel / tile shape computed it.
// SHAPE16 (tools/ffn_bench.hip only, TIMING ONLY - the results are meaningless): every 32x32x16 MFMA is issued as
// two v_mfma_f32_16x16x32_f16 on quarters of the same accumulator with the same operand registers: the FLOPs, the
// matrix-pipe cycles and the operand traffic of a 16x16x32 port of the engine without its data layout.
template <int NTW, int TTW, bool SHAPE16 = false>
__device__ __forceinline__ void wt_mma(f32x16 (&acc)[NTW][TTW], const half8 (&wh)[NTW], const half8 (&wl)[NTW],
                                       const half8 (&xh)[TTW], const half8 (&xl)[TTW]) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int tt = 0; tt < TTW; ++tt) {
            if constexpr (SHAPE16) {
                f32x4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = f32x4{acc[i][tt][4 * u], acc[i][tt][4 * u + 1], acc[i][tt][4 * u + 2], acc[i][tt][4 * u + 3]};
                q[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[tt], q[0], 0, 0, 0);
                q[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[tt], q[1], 0, 0, 0);
                q[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[tt], q[2], 0, 0, 0);
                q[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[tt], q[3], 0, 0, 0);
                q[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[tt], q[0], 0, 0, 0);
                q[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[tt], q[1], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][tt][4 * u + e] = q[u][e];
            } else {
                acc[i][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], xh[tt], acc[i][tt], 0, 0, 0);
                acc[i][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[i], xh[tt], acc[i][tt], 0, 0, 0);
                acc[i][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], xl[tt], acc[i][tt], 0, 0, 0);
            }
        }
}

// Weight fragments of k-step `ks` for NTW tiles.  `wp[i]` and `ks` are wave-uniform (scalar registers), `lo8` is the
// lane's offset in halfs (lane * 8): the loads are global_load_dwordx4 with a scalar base, no per-load VALU.
template <int NTW>
__device__ __forceinline__ void w_load(half8 (&wh)[NTW], half8 (&wl)[NTW], const _Float16* const (&wp)[NTW], int ks,
                                       unsigned lo8) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const _Float16* p = wp[i] + (size_t)ks * (2 * WT_FRAG);
        wh[i] = *reinterpret_cast<const half8*>(p + lo8);
        wl[i] = *reinterpret_cast<const half8*>(p + WT_FRAG + lo8);
    }
}

__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// In-kernel phase stamps for tools/ffn_bench.hip (diagnostic builds only: -DICREC_STAMPS; the product never defines it).
#ifdef ICREC_STAMPS
__device__ unsigned long long g_stamps[1 << 22];
#define ICREC_STAMP(slot_wave, k)                                                                        \
    do {                                                                                                 \
        if ((threadIdx.x & 63) == 0 && (int)(threadIdx.x >> 6) == (slot_wave))                           \
            g_stamps[((size_t)blockIdx.x * 2 + ((slot_wave) ? 1 : 0)) * 64 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ICREC_STAMP(slot_wave, k) do { } while (0)
#endif

// ---------------------------------------------------------------- whole-K loop of one output tile
// acc[i][tt] = sum_k W[(nt0 + i) tile][k] . X[m0 + tt tile][k], K in slabs of 64 (4 k-steps); weight fragments
// D k-steps ahead in registers (D divides 4), the next activation slab one slab ahead in registers, two LDS
// stages, one barrier per slab.  `smem`: XRing<TTW>::BYTES.
template <int NTW, int TTW, int D, bool ZERO = true>
__device__ __forceinline__ void wt_kloop(f32x16 (&acc)[NTW][TTW], const _Float16* __restrict__ Wp, int nt0, int K,
                                         const _Float16* __restrict__ Xh, const _Float16* __restrict__ Xl, int64_t m0,
                                         int64_t T, char* smem) {
    static_assert(D == 1 || D == 2 || D == 4, "prefetch depth must divide the 4 k-steps of a slab");
    ICREC_STAMP(0, 0);
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const unsigned lo8 = lane * 8;
    const int KS = K / 16, nslab = K / 64;
    const _Float16* wp[NTW];  // wave-uniform (nt0 must be)
#pragma unroll
    for (int i = 0; i < NTW; ++i) wp[i] = Wp + wt_frag_off(nt0 + i, 0, KS);
    if (ZERO) {  // !ZERO: the caller has put the residual + bias into the accumulators (wt_res_init_*)
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int tt = 0; tt < TTW; ++tt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][tt][e] = 0.0f;
    }
    half8 wh[D][NTW], wl[D][NTW];
    // activation slabs: TWO in flight in registers (slab s+1 is written to LDS at the end of slab s, slab s+2 was
    // requested a whole slab earlier) - with one, every slab boundary waited for an HBM round trip (stamps: the first
    // slabs of a block took 2x the MFMA time)
    u32x4 xa[2 * TTW], xb[2 * TTW];
    x_load<TTW>(xa, Xh, Xl, m0, T, K, 0);
#pragma unroll
    for (int d = 0; d < D; ++d) w_load<NTW>(wh[d], wl[d], wp, d, lo8);
    if (nslab > 1) x_load<TTW>(xb, Xh, Xl, m0, T, K, 1);
    x_store<TTW>(xa, smem);
    if (nslab > 2) x_load<TTW>(xa, Xh, Xl, m0, T, K, 2);
    __syncthreads();
    ICREC_STAMP(0, 1);
    auto slab = [&](int s, u32x4 (&xnext)[2 * TTW]) {  // xnext holds slab s+1 on entry, slab s+3 on exit
        const char* st = smem + (s & 1) * XRing<TTW>::STAGE_BYTES;
        half8 xh[2][TTW], xl[2][TTW];  // fragments of the next k-step are read under the current one's MFMAs
#pragma unroll
        for (int tt = 0; tt < TTW; ++tt) {
            xh[0][tt] = x_frag(st, tt, 0, r, h);
            xl[0][tt] = x_frag(st + XRing<TTW>::PLANE_BYTES, tt, 0, r, h);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3) {
#pragma unroll
                for (int tt = 0; tt < TTW; ++tt) {
                    xh[(j + 1) & 1][tt] = x_frag(st, tt, j + 1, r, h);
                    xl[(j + 1) & 1][tt] = x_frag(st + XRing<TTW>::PLANE_BYTES, tt, j + 1, r, h);
                }
            }
            wt_mma<NTW, TTW>(acc, wh[j % D], wl[j % D], xh[j & 1], xl[j & 1]);
            int nk = 4 * s + j + D;  // past the end: re-read the last fragment (never consumed) - no branch in the loop body
            nk = nk < KS ? nk : KS - 1;
            w_load<NTW>(wh[j % D], wl[j % D], wp, nk, lo8);
            __builtin_amdgcn_sched_barrier(0);  // keep every prefetch in its k-step (the scheduler otherwise sinks the loads to their uses)
        }
        if (s + 1 < nslab) {  // slab-granular (uniform) branches; the k-step body above is straight-line code
            x_store<TTW>(xnext, smem + ((s + 1) & 1) * XRing<TTW>::STAGE_BYTES);
            if (s + 3 < nslab) x_load<TTW>(xnext, Xh, Xl, m0, T, K, s + 3);
        }
        __syncthreads();
        if (s < 24) ICREC_STAMP(0, 2 + s);
    };
    for (int s = 0; s < nslab; s += 2) {  // nslab is even (K is a multiple of 128 for every layer of the encoder)
        slab(s, xb);
        slab(s + 1, xa);
    }
}

// Feature (row of out^T) held in accumulator register e by a lane of half h: 8 (e >> 2) + 4 h + (e & 3).
__device__ __forceinline__ int wt_feat(int e, int h) { return 8 * (e >> 2) + 4 * h + (e & 3); }

#endif  // __HIPCC__

}  // namespace icrec
