// wt_gemm.h — the "weights-direct" f16x3 GEMM engine of the encoder (gfx950).
//
// Every linear layer of the encoder is out[T, N] = X[T, K] . W[N, K]^T with a STATIC weight matrix and M = T
// tokens.  The engine computes the transposed tile  out^T = W . X^T  so that
//   * the weight fragments are the A operand of the MFMA.  Each wave owns distinct output features, so no other
//     wave of the workgroup needs its weight fragments: they are loaded straight from global memory (L2) into
//     registers and never touch LDS.  Weights are pre-packed once at encoder creation in FRAGMENT ORDER — the 4 KB a
//     wave needs for one (32-feature block, 32-deep k-step) are contiguous, lane l's 16 bytes of each of its four
//     fragments at l * 16 — so every weight load is one fully coalesced global_load_dwordx4;
//   * the token rows are the B operand: the activation planes are staged through LDS (XOR-swizzled rows,
//     conflict-free ds_read_b128) and shared by the waves;
//   * the accumulators come out with the TOKEN on the lane and 4 consecutive FEATURES in consecutive
//     registers: epilogues store 16 B (fp32) or 8 B (f16 planes) per lane, a GELU'd tile is handed to the next GEMM
//     as 8-byte LDS writes, and LayerNorm statistics need two shuffles and no transpose.
//
// MFMA shape: v_mfma_f32_16x16x32_f16 (round 3; rounds 1-2 used 32x32x16).  Same FLOPs per cycle, same operand and
// accumulator bytes per FLOP - but under this engine's load the chip is power-limited, and it holds a higher clock
// on the 16x16x32 form: on random operands a registers-only loop delivers 1,970 vs 1,710 TFLOP/s (1.95 vs 1.69 GHz,
// tools/mfma_shape.hip, profiles/r03_mfma_shape_microbench.txt), and the fused FFN kernel issued in this shape ran
// 1.33x faster before its data layout existed (tools/ffn_bench.hip VAR 64, profiles/r03_ffn_harness_shape_and_stream.txt).
//   lane l = (g << 4 | c), c = l & 15, g = l >> 4
//   A (16 rows x 32 k):  lane holds A[row c][k = 8g .. 8g+7]        B (32 k x 16 cols): lane holds B[k = 8g .. 8g+7][col c]
//   D (16 x 16):         lane holds D[row 4g + reg][col c], reg = 0..3      (checked with integer data: tools/mfma16_probe.hip)
// A 32-feature x 32-token BLOCK of out^T is 2 x 2 such tiles ([fi][ti], 16 accumulator registers as before): lane
// (c, g) holds features fi*16 + 4g + reg of tokens ti*16 + c.
//
// Arithmetic (fp32-accurate on the f16 matrix cores): every fp32 operand is carried as two f16 planes
//     hi = f16(x * S)        lo = f16(x * S - hi)                 S = 16 (activations), 1024 (weights)
// and a product is   x*w*S_a*S_w ~= hi_w*hi_x + lo_w*hi_x + hi_w*lo_x   — three MFMAs into ONE fp32
// accumulator, in that order, k-steps of 32 ascending; the result is scaled back by 2^-14 in the epilogue.  f16 x f16
// products are exact in the fp32 accumulator; the dropped lo*lo term and the split residuals are <= 3 * 2^-22
// relative per product.  The power-of-two scales keep `lo` out of the f16 subnormal range for every value that
// matters (|x| >= 2^-7 / S: absolute error below 2^-25 / S otherwise) and are exact; |x| * S is clamped to the
// f16 range (|activation| <= 4094, |weight| <= 63.9: far outside what LayerNorm-bounded BERT tensors reach).
// An output's value depends only on its own row of W and its own token's row of X (the matrix core's summation
// order over the 32 k of a step is a function of k alone), so a token encodes to the same bits whichever kernel,
// tile position or batch computed it — the bitwise tests between the small-batch, batch and fused forms rest on this.
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
// Activations are split on the fly (LayerNorm / GELU / attention epilogues), so their split is the cheap form, four
// instructions per PAIR of elements:
//   s = 16 x;  hi = f16(s) rounded toward zero (v_cvt_pkrtz_f16_f32: two elements per instruction);
//   lo = f16(s - hi), the subtraction by v_fma_mix_f32 (hi read as the f16 it is, times -1, plus s: exact in fp32, no
//   conversion back), again v_cvt_pkrtz.
// (Rounds 1-3 built hi by clearing the low 13 mantissa bits of s: the same value wherever s is in f16's normal range,
// six instructions per pair.  In their compute phase the attention kernels are co-limited by vector issue and the matrix
// pipe, and a third of their per-score vector work was this split: 7-8-tile bucket -13 %.)  hi + lo carries s to 2^-21 relative (lo is rounded toward zero:
// one ulp of an 11-bit residual); below the f16 normal range (|x| < 3.8e-6) the absolute error is < 2^-24 / 16; beyond
// it (|x| > 4094) hi saturates at +-65504 and lo takes what it can of the rest.  Every producer of activation planes
// uses this one function, so a tensor's planes are the same bits whichever kernel wrote them.
typedef _Float16 half2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_prescaled(float s0, float s1, half2w& hi, half2w& lo) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(s0, s1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(s0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(s1));
    hi = __builtin_bit_cast(half2w, h);
    lo = __builtin_bit_cast(half2w, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
// four consecutive elements, already multiplied by the plane scale -> the 8-byte hi and lo pieces
__device__ __forceinline__ void split4_prescaled(const f32x4& s, half4& hi, half4& lo) {
    half2w a, b, c, d;
    split_pair_prescaled(s[0], s[1], a, b);
    split_pair_prescaled(s[2], s[3], c, d);
    hi = half4{a[0], a[1], c[0], c[1]};
    lo = half4{b[0], b[1], d[0], d[1]};
}
__device__ __forceinline__ void split_act4(const f32x4& v, half4& hi, half4& lo) {
    split4_prescaled(f32x4{v[0] * WT_SA, v[1] * WT_SA, v[2] * WT_SA, v[3] * WT_SA}, hi, lo);
}
__device__ __forceinline__ void split_act(float x, _Float16& hi, _Float16& lo) {
    half2w a, b;
    split_pair_prescaled(x * WT_SA, 0.0f, a, b);
    hi = a[0];
    lo = b[0];
}

// ---------------------------------------------------------------- packed weights
// W [N, K] fp32 row-major -> fragment order.  Fragment = the 1 KB one wave feeds to one MFMA as its A operand: lane
// l = (g << 4 | c) holds W[16-feature tile row c][k = ks*32 + 8g .. +7] of one plane.  The four fragments of a
// (32-feature block nt, k-step ks) - [fi = 0, 1 feature halves][plane = hi, lo] - are contiguous: 2,048 halfs at
// ((nt * K/32 + ks) * 4 + fi * 2 + plane) * 512.
constexpr int WT_FRAG = 512;  // halfs per fragment (64 lanes x 8)
__device__ __forceinline__ size_t wt_frag_off(int nt, int ks, int KS) { return ((size_t)nt * KS + ks) * (4 * WT_FRAG); }

struct WFrag { half8 h[2], l[2]; };   // weight fragments of one 32-feature block, one k-step: [fi] hi / lo
struct XFrag { half8 h[2], l[2]; };   // activation fragments of one 32-token block, one k-step: [ti] hi / lo
struct Acc32 { f32x4 t[2][2]; };      // a 32-feature x 32-token block of out^T: [fi][ti]

__device__ __forceinline__ void acc_zero(Acc32& a) {
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) a.t[fi][ti] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// ---------------------------------------------------------------- activation slab ring (LDS)
// A slab = BM = 32 * TTW token rows x 64 k of both planes; row = 128 B per plane, 16-B chunk ch of row `row` is
// stored at chunk ch ^ ((row >> 1) & 7): the 16 lanes of every ds_read_b128 lane group of a fragment read (rows
// c, chunks 4j + g) land on 16 distinct 16-B slots of the 256-B bank row.
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

// B fragments of token block tt, k-step j (0, 1) of the slab: lane (c, g) -> tokens tt*32 + ti*16 + c, k = 32 j + 8 g .. +7
template <int TTW>
__device__ __forceinline__ void x_frag(XFrag& x, const char* stage, int tt, int j, int c, int g) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int row = tt * 32 + ti * 16 + c;
        const int pos = row * 128 + (((4 * j + g) ^ ((row >> 1) & 7)) << 4);
        x.h[ti] = *reinterpret_cast<const half8*>(stage + pos);
        x.l[ti] = *reinterpret_cast<const half8*>(stage + XRing<TTW>::PLANE_BYTES + pos);
    }
}

// ---------------------------------------------------------------- one k-step of MFMAs for one 32-token block
// acc[i] += W_i . X^T over 32 k: per 16x16 tile (w_hi, x_hi), (w_lo, x_hi), (w_hi, x_lo) — the canonical order every
// kernel of the engine uses, so a token's result is independent of which kernel / tile shape computed it.
__device__ __forceinline__ void wt_mma_block(Acc32& a, const WFrag& w, const XFrag& x) {
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            a.t[fi][ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.h[fi], x.h[ti], a.t[fi][ti], 0, 0, 0);
            a.t[fi][ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.l[fi], x.h[ti], a.t[fi][ti], 0, 0, 0);
            a.t[fi][ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.h[fi], x.l[ti], a.t[fi][ti], 0, 0, 0);
        }
}
template <int NTW, int TTW, int TT>
__device__ __forceinline__ void wt_mma(Acc32 (&acc)[NTW][TTW], const WFrag (&w)[NTW], const XFrag& x) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) wt_mma_block(acc[i][TT], w[i], x);
}

// Weight fragments of k-step `ks` for NTW blocks.  `wp[i]` and `ks` are wave-uniform (scalar registers), `lo8` is the
// lane's offset in halfs (lane * 8): the loads are global_load_dwordx4 with a scalar base, no per-load VALU.
template <int NTW>
__device__ __forceinline__ void w_load(WFrag (&w)[NTW], const _Float16* const (&wp)[NTW], int ks, unsigned lo8) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const _Float16* p = wp[i] + (size_t)ks * (4 * WT_FRAG);
        w[i].h[0] = *reinterpret_cast<const half8*>(p + lo8);
        w[i].l[0] = *reinterpret_cast<const half8*>(p + WT_FRAG + lo8);
        w[i].h[1] = *reinterpret_cast<const half8*>(p + 2 * WT_FRAG + lo8);
        w[i].l[1] = *reinterpret_cast<const half8*>(p + 3 * WT_FRAG + lo8);
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
// the 100 MHz constant-rate counter beside a shader-clock stamp: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
#define ICREC_STAMP_RT(slot_wave, k)                                                                     \
    do {                                                                                                 \
        if ((threadIdx.x & 63) == 0 && (int)(threadIdx.x >> 6) == (slot_wave))                           \
            g_stamps[((size_t)blockIdx.x * 2 + ((slot_wave) ? 1 : 0)) * 64 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define ICREC_STAMP(slot_wave, k) do { } while (0)
#define ICREC_STAMP_RT(slot_wave, k) do { } while (0)
#endif

// ---------------------------------------------------------------- whole-K loop of one output tile
// acc[i][tt] = sum_k W[(nt0 + i) block][k] . X[m0 + tt block][k], K in slabs of 64 (2 k-steps of 32); weight fragments
// D k-steps ahead in registers (D = 1, 2 or 4: the latency form of single requests keeps four - its few workgroups
// find a layer's fragments in the Infinity Cache at best, half a microsecond away; eight measured SLOWER at 4 KB per wave and
// k-step, and faster once a wave's k-step is 2 KB: wt_linear_half_kernel, encoder_x3.h), the next activation slab one slab
// ahead in registers, two LDS stages, one barrier per slab.  `smem`: XRing<TTW>::BYTES.  The fragments of the next (k-step, token block) unit are read from LDS
// under the current unit's MFMAs.
template <int NTW, int TTW, int D>
__device__ __forceinline__ void wt_kloop(Acc32 (&acc)[NTW][TTW], const _Float16* __restrict__ Wp, int nt0, int K,
                                         const _Float16* __restrict__ Xh, const _Float16* __restrict__ Xl, int64_t m0,
                                         int64_t T, char* smem) {
    static_assert(D == 1 || D == 2 || D == 4, "prefetch depth: 1, 2 or 4 k-steps");
    static_assert(TTW == 1 || TTW == 2, "1 or 2 token blocks per wave");
    ICREC_STAMP(0, 0);
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const unsigned lo8 = lane * 8;
    const int KS = K / 32, nslab = K / 64;
    const _Float16* wp[NTW];  // wave-uniform (nt0 must be)
#pragma unroll
    for (int i = 0; i < NTW; ++i) wp[i] = Wp + wt_frag_off(nt0 + i, 0, KS);
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int tt = 0; tt < TTW; ++tt) acc_zero(acc[i][tt]);
    WFrag w[D][NTW];
    // activation slabs: TWO in flight in registers (slab s+1 is written to LDS at the end of slab s, slab s+2 was
    // requested a whole slab earlier) - with one, every slab boundary waited for an HBM round trip
    u32x4 xa[2 * TTW], xb[2 * TTW];
    x_load<TTW>(xa, Xh, Xl, m0, T, K, 0);
#pragma unroll
    for (int d = 0; d < D; ++d) w_load<NTW>(w[d], wp, d, lo8);
    if (nslab > 1) x_load<TTW>(xb, Xh, Xl, m0, T, K, 1);
    x_store<TTW>(xa, smem);
    if (nslab > 2) x_load<TTW>(xa, Xh, Xl, m0, T, K, 2);
    __syncthreads();
    ICREC_STAMP(0, 1);
    auto slab = [&](int s, auto parity, u32x4 (&xnext)[2 * TTW]) {  // xnext holds slab s+1 on entry, slab s+3 on exit
        constexpr int PAR = decltype(parity)::value;  // s & 1: the ring slot of k-step 2 s + j is (2 PAR + j) % D
        const char* st = smem + (s & 1) * XRing<TTW>::STAGE_BYTES;
        XFrag x[2];  // unit u = j * TTW + tt: the next unit's fragments are read under the current one's MFMAs
        x_frag<TTW>(x[0], st, 0, 0, c, g);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int tt = 0; tt < TTW; ++tt) {
                constexpr int U = 2 * TTW;
                const int u = j * TTW + tt;
                if (u + 1 < U) {
                    x_frag<TTW>(x[(u + 1) & 1], st, (u + 1) % TTW, (u + 1) / TTW, c, g);
                    __builtin_amdgcn_sched_barrier(0);  // issue the next unit's LDS reads before this unit's MFMAs
                }
                if (tt == 0) wt_mma<NTW, TTW, 0>(acc, w[(2 * PAR + j) % D], x[u & 1]);
                else wt_mma<NTW, TTW, TTW - 1>(acc, w[(2 * PAR + j) % D], x[u & 1]);
                __builtin_amdgcn_sched_barrier(0);  // keep every prefetch in its unit (the scheduler otherwise sinks the loads to their uses)
            }
            int nk = 2 * s + j + D;  // past the end: re-read the last fragment (never consumed) - no branch in the loop body
            nk = nk < KS ? nk : KS - 1;
            w_load<NTW>(w[(2 * PAR + j) % D], wp, nk, lo8);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s + 1 < nslab) {  // slab-granular (uniform) branches; the k-step body above is straight-line code
            x_store<TTW>(xnext, smem + ((s + 1) & 1) * XRing<TTW>::STAGE_BYTES);
            if (s + 3 < nslab) x_load<TTW>(xnext, Xh, Xl, m0, T, K, s + 3);
        }
        __syncthreads();
        if (s < 24) ICREC_STAMP(0, 2 + s);
    };
    for (int s = 0; s < nslab; s += 2) {  // nslab is even (K is a multiple of 128 for every layer of the encoder)
        slab(s, std::integral_constant<int, 0>{}, xb);
        slab(s + 1, std::integral_constant<int, 1>{}, xa);
    }
}

#endif  // __HIPCC__

}  // namespace icrec
