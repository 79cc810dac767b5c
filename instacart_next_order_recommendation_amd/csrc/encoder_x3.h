// encoder_x3.h — the linear layers of the encoder in f16x3 arithmetic (gemm_mode F16X3, the default) on the
// weights-direct engine of wt_gemm.h: QKV projection, attention-output projection + residual + LayerNorm, the fused
// FFN block, their small-batch forms, the packed-weight builder.  Included by encoder.hip only.
//
// Reference arithmetic: transformers/models/bert/modeling_bert.py (tf:) — BertSelfAttention.query/key/value tf:175-177,
// BertSelfOutput tf:289-293, BertIntermediate tf:334-337 (erf-GELU), BertOutput tf:347-351.
#pragma once
#include <type_traits>

#include "wt_gemm.h"

namespace icrec {

// ---------------------------------------------------------------- erf-GELU of the engine
// times the activation plane scale (16): one polynomial, one v_exp_f32, 16 VALU.
//   gelu(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2),     erfc(t / sqrt 2) = 2^Q10(min(t, 5.75))
// (x >= 0: x - 0.5 x erfc = 0.5 x (1 + erf);  x < 0: 0.5 x erfc(|z|) = 0.5 x (1 + erf(z)).)  Only ABSOLUTE accuracy of
// erfc matters here - it multiplies |x| and is added to a term of the size of x.  Q10: Chebyshev fit of
// log2(erfc(t / sqrt 2)) on [0, 5.75] (tools/fit_erf.py --gelu); in emulated fp32 fma arithmetic over 440k points:
// max |error| 2.4e-7 (half an ulp at x = 4.3), relative error <= 1.05e-6 wherever |gelu| > 1e-3.
__device__ __forceinline__ float gelu16_wt(float x) {
    const float t = fminf(fabsf(x), 5.75f);
    float q = -1.428101063e-08f;
    q = fmaf(q, t, 4.683960178e-07f);
    q = fmaf(q, t, -6.560040814e-06f);
    q = fmaf(q, t, 4.923233760e-05f);
    q = fmaf(q, t, -1.793856253e-04f);
    q = fmaf(q, t, -2.251562182e-04f);
    q = fmaf(q, t, 7.249582803e-03f);
    q = fmaf(q, t, -5.267105742e-02f);
    q = fmaf(q, t, -4.591336602e-01f);
    q = fmaf(q, t, -1.151116827e+00f);
    q = fmaf(q, t, 2.960897358e-07f);
    const float e = __builtin_amdgcn_exp2f(q);
    return fmaf(fabsf(x) * -8.0f, e, fmaxf(x * 16.0f, 0.0f));
}
// bias + GELU + split of the four consecutive intermediates one lane holds of a tile (acc in units of 2^-14)
__device__ __forceinline__ void gelu_split4(const f32x4& acc, const f32x4& bias, half4& hi, half4& lo) {
    f32x4 s;
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = gelu16_wt(fmaf(acc[j], WT_UNSCALE, bias[j]));
    split4_prescaled(s, hi, lo);
}

// ---------------------------------------------------------------- residual stream + LayerNorm of the f16x3 engine
// In f16x3 mode the residual stream x lives in HBM ONLY as its two f16 planes (xh, xl: 16 x to 22 significant bits,
// wt_gemm.h) — the planes every GEMM reads anyway.  The two LayerNorm sites of a layer (tf:292 attention output,
// tf:350 FFN output: LN(dense(.) + bias + x)) take the residual from those planes, in accumulator units:
//     r    = fmaf(float(xh) + float(xl), 1024, bias * 2^14)      (hi + lo is exact in fp32; one rounding)
//     acc  = the k-steps of the GEMM from 0, ascending           (units of 2^-14, like every accumulator of the engine)
//     v    = (acc + r) * 2^-14                                   (one fp32 add; the scaling is exact)
// (added AFTER the K loop: in the fused kernel the residual rows are staged by one half of the waves while the other
// half runs the attention-output GEMM, and the two meet in one add)
// then the engine's LayerNorm order over the 384 features of a token.  Wave q (0..3 of the waves holding
// accumulators) owns features q*96 .. q*96+95: lane (c, g) holds, for token block tt and token half ti, the 24 values
// of token tt*32 + ti*16 + c at features q*96 + i*32 + fi*16 + 4g + reg (i < 3, fi < 2, reg < 4):
//   part(q, g) = sum over (i, fi, reg), i outermost, of v       sequential fp32 adds from 0
//   Pq   = (part(q,0) + part(q,1)) + (part(q,2) + part(q,3))    two xor-shuffles (16, 32): a + b == b + a exactly
//   sum  = ((P0 + P1) + P2) + P3
//   mean = sum / 384;   d = v - mean;   the same tree over fmaf(d, d, .) chains;   var = that / 384
//   y = fmaf(d * (1 / sqrtf(var + eps)), gamma, beta);   planes = split(16 y)
// (wt_ln_block; ln_wt_kernel is the unfused form with the same order — same bits.)
//
// Global memory is touched only by coalesced accesses: a lane-per-token access pattern costs 4x the whole K loop
// (in-kernel stamps, round 2).  Each wave transposes its own [32 tokens x 96 features] of one plane through a PRIVATE
// 6.5 KB LDS tile (rows of 192 B + 16 B pad): no workgroup barrier, 192 contiguous bytes per token row on the
// global side.  LDS instructions of one wave execute in order; lds_order() keeps the compiler from reordering
// across the hand-over and waits for the data.
constexpr int LNT_ROW = 208;                        // bytes per tile row: 96 halfs + 16 B
constexpr int LNT_TILE = 32 * LNT_ROW;              // one wave's tile
constexpr int LNT_RED = 2 * 64 * 4 * 4;             // the two 4-partial exchanges: [2][64 tokens][4 waves] floats
constexpr int LNT_PAR = 2 * 384 * 4;                // gamma, beta
constexpr int LNT_BYTES = LNT_RED + 4 * LNT_TILE + LNT_PAR;   // 31,744 B

__device__ __forceinline__ void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float res_init_val(_Float16 hi, _Float16 lo, float b) {
    return fmaf((float)hi + (float)lo, WT_SW, b * (WT_SA * WT_SW));
}

// chunk k of this lane in the flat [32 rows][12 x 16 B] view of a wave's tile: row = f / 12, c = f % 12, f = lane + 64 k
__device__ __forceinline__ void lnt_flat(int lane, int k, int& row, int& c) {
    const int f = lane + 64 * k;
    row = (f * 43691) >> 19;  // f / 12 for f < 384
    c = f - row * 12;
}

// r (above) of a wave's [96 features x 32 tokens] (token block tb of the workgroup's 64) from the residual planes in
// global memory, in two steps so that a caller can put work between the request and the use: wt_res_rows_load issues
// the 12 coalesced 16-byte loads of the rows (12 lanes cover one token's 192 bytes of a plane), wt_res_rows_acc
// transposes them through the wave's private LDS tile (`tile`: LNT_TILE bytes) into the accumulator layout:
// acc[.][TT] = r (ADD = false) or += r (ADD = true).
struct ResRows { u32x4 v[2][6]; };  // [plane][chunk]
__device__ __forceinline__ void wt_res_rows_load(ResRows& rr, int q, int tb, const _Float16* __restrict__ xh,
                                                 const _Float16* __restrict__ xl, int64_t m0, int64_t T) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int row, ch;
        lnt_flat(lane, k, row, ch);
        int64_t gr = m0 + tb * 32 + row;
        gr = gr < T ? gr : T - 1;
        const int64_t off = gr * 384 + q * 96 + ch * 8;
        rr.v[0][k] = *reinterpret_cast<const u32x4*>(xh + off);
        rr.v[1][k] = *reinterpret_cast<const u32x4*>(xl + off);
    }
}
template <bool ADD, int TTW, int TT>
__device__ __forceinline__ void wt_res_rows_acc(Acc32 (&acc)[3][TTW], const ResRows& rr, int q, const float* __restrict__ bias,
                                                char* tile) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));  // opaque (see wt_ln_block)
    const int c = lane & 15, g = lane >> 4;
    int lpos[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int row, ch;
        lnt_flat(lane, k, row, ch);
        lpos[k] = row * LNT_ROW + ch * 16;
    }
    half4 hi[3][2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int k = 0; k < 6; ++k) *reinterpret_cast<u32x4*>(tile + lpos[k]) = rr.v[pl][k];
        lds_order();
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi) {
                const int fl = i * 32 + fi * 16 + 4 * g;
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    const half4 a = *reinterpret_cast<const half4*>(tile + (ti * 16 + c) * LNT_ROW + fl * 2);
                    if (pl == 0) {
                        hi[i][fi][ti] = a;
                    } else {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + q * 96 + fl);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float r = res_init_val(hi[i][fi][ti][j], a[j], b[j]);
                            acc[i][TT].t[fi][ti][j] = ADD ? acc[i][TT].t[fi][ti][j] + r : r;
                        }
                    }
                }
            }
        lds_order();
    }
}

// acc += r for the single-block waves of the small-batch kernels: 8-byte loads straight from global memory (a
// handful of tokens: latency, not bandwidth).
__device__ __forceinline__ void wt_res_add_direct(Acc32 (&acc)[1][1], int nt0, const float* __restrict__ bias,
                                                  const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                  int N, int64_t m0, int64_t T) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
        const int feat = nt0 * 32 + fi * 16 + 4 * g;
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + feat);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            int64_t tok = m0 + ti * 16 + c;
            tok = tok < T ? tok : T - 1;
            const half4 a = *reinterpret_cast<const half4*>(xh + tok * N + feat);
            const half4 d = *reinterpret_cast<const half4*>(xl + tok * N + feat);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][0].t[fi][ti][j] = acc[0][0].t[fi][ti][j] + res_init_val(a[j], d[j], b[j]);
        }
    }
}

// this thread's 16 bytes of the [gamma | beta] table wt_ln_block keeps in LDS (threads 0 .. 191 of the callers)
__device__ __forceinline__ f32x4 wt_ln_par_load(const float* __restrict__ gam, const float* __restrict__ bet, int ptid) {
    const int t = ptid < 192 ? ptid : 0;
    return *reinterpret_cast<const f32x4*>((t < 96 ? gam : bet - 384) + 4 * t);
}

constexpr int FFN2_XPLANE = 64 * 768;  // one plane of the resident activation image of the fused kernels (below)
__device__ __forceinline__ int ffn_x_pos(int tok, int ch) { return tok * 768 + (((ch & ~15) | ((ch ^ tok) & 15)) << 4); }

// LayerNorm of ONE 32-token block held by four waves q = 0..3 (96 features each; accumulators acc[.][TT] = residual +
// bias + dense in units of 2^-14) -> the two planes of x, written to global memory (xh / xl != nullptr: rows m0 + tb*32 ..)
// and / or - xs != nullptr - into the resident activation image of the fused kernel in LDS (layout ffn_x_pos; the
// caller's next barrier orders those writes against their readers).  The fused kernel runs it on all eight waves at once: four
// hold token block 0, four token block 1, every wave with a transposition tile of its own (`slot`).
// Every thread of the workgroup calls; `sync` = the workgroup barrier (two of them).  lds: LNT_BYTES8 for eight slots.
// ptid: index of the thread among the callers (0 .. 191 must be present).
constexpr int LNT_BYTES8 = LNT_RED + 8 * LNT_TILE + LNT_PAR;   // 58,368 B
template <int NSLOT, int TTW, int TT, class Sync>
__device__ __forceinline__ void wt_ln_block(Acc32 (&acc)[3][TTW], int q, int tb, int slot, _Float16* __restrict__ xh,
                                            _Float16* __restrict__ xl, int64_t m0, int64_t T,
                                            const float* __restrict__ gam, const float* __restrict__ bet, float eps,
                                            char* lds, Sync sync, int ptid, char* xs = nullptr) {
    float* const red = reinterpret_cast<float*>(lds);
    char* const tile = lds + LNT_RED + slot * LNT_TILE;
    float* const par = reinterpret_cast<float*>(lds + LNT_RED + NSLOT * LNT_TILE);
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));  // opaque: the per-lane addresses below are computed HERE, not hoisted to the kernel's
                                    // entry and held (or spilled) across the loops in front of this call
    const int c = lane & 15, g = lane >> 4;
    ICREC_STAMP(0, 32); ICREC_STAMP(4, 32);
    // gamma / beta -> LDS (visible after the first barrier): each lane needs the 24 values of its quarter of the wave's
    // 96 features - as vector loads that is dozens of 1 KB requests through the texture path per wave for 768 distinct bytes
    if (ptid < 192) *reinterpret_cast<f32x4*>(par + 4 * ptid) = wt_ln_par_load(gam, bet, ptid);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        float part = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[i][TT].t[fi][ti][e] * WT_UNSCALE;
                    acc[i][TT].t[fi][ti][e] = v;
                    part = part + v;
                }
        part = part + __shfl_xor(part, 16, 64);
        part = part + __shfl_xor(part, 32, 64);  // Pq: the same bits in the four lanes of a token
        if (g == 0) red[(tb * 32 + ti * 16 + c) * 4 + q] = part;
    }
    ICREC_STAMP(0, 33); ICREC_STAMP(4, 33);
    sync();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(red + (tb * 32 + ti * 16 + c) * 4);
        const float mean = (((s4[0] + s4[1]) + s4[2]) + s4[3]) / 384.0f;
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = acc[i][TT].t[fi][ti][e] - mean;
                    acc[i][TT].t[fi][ti][e] = d;
                    sq = fmaf(d, d, sq);
                }
        sq = sq + __shfl_xor(sq, 16, 64);
        sq = sq + __shfl_xor(sq, 32, 64);
        if (g == 0) red[256 + (tb * 32 + ti * 16 + c) * 4 + q] = sq;
    }
    ICREC_STAMP(0, 34); ICREC_STAMP(4, 34);
    sync();
    ICREC_STAMP(0, 35); ICREC_STAMP(4, 35);
    float rstd[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(red + 256 + (tb * 32 + ti * 16 + c) * 4);
        const float var = (((s4[0] + s4[1]) + s4[2]) + s4[3]) / 384.0f;
        rstd[ti] = 1.0f / sqrtf(var + eps);
    }
    const int64_t t0 = m0 + tb * 32;
    half4 lo[3][2][2];
    // the hi plane goes straight into the tile; lo waits in registers for its turn
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
            const int fl = i * 32 + fi * 16 + 4 * g;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(par + q * 96 + fl);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(par + 384 + q * 96 + fl);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = fmaf(acc[i][TT].t[fi][ti][j] * rstd[ti], gm[j], bt[j]);
                half4 hi;
                split_act4(y, hi, lo[i][fi][ti]);
                *reinterpret_cast<half4*>(tile + (ti * 16 + c) * LNT_ROW + fl * 2) = hi;
            }
        }
    lds_order();
    u32x4 o[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int row, ch;
        lnt_flat(lane, k, row, ch);
        o[k] = *reinterpret_cast<const u32x4*>(tile + row * LNT_ROW + ch * 16);
    }
    lds_order();
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
                *reinterpret_cast<half4*>(tile + (ti * 16 + c) * LNT_ROW + (i * 32 + fi * 16 + 4 * g) * 2) = lo[i][fi][ti];
#pragma unroll
    for (int k = 0; k < 6; ++k) {  // the hi rows leave while the lo tile is written
        int row, ch;
        lnt_flat(lane, k, row, ch);
        if (xs != nullptr) *reinterpret_cast<u32x4*>(xs + ffn_x_pos(tb * 32 + row, q * 12 + ch)) = o[k];
        if (xh != nullptr && t0 + row < T) *reinterpret_cast<u32x4*>(xh + (t0 + row) * 384 + q * 96 + ch * 8) = o[k];
    }
    lds_order();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int row, ch;
        lnt_flat(lane, k, row, ch);
        o[k] = *reinterpret_cast<const u32x4*>(tile + row * LNT_ROW + ch * 16);
    }
    lds_order();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int row, ch;
        lnt_flat(lane, k, row, ch);
        if (xs != nullptr) *reinterpret_cast<u32x4*>(xs + FFN2_XPLANE + ffn_x_pos(tb * 32 + row, q * 12 + ch)) = o[k];
        if (xl != nullptr && t0 + row < T) *reinterpret_cast<u32x4*>(xl + (t0 + row) * 384 + q * 96 + ch * 8) = o[k];
    }
    ICREC_STAMP(0, 36); ICREC_STAMP(4, 36);
}

// The unfused form of the same LayerNorm (small batches; the unfused reference chain): planes(x) <- LN(a), `a` =
// dense(.) + bias + residual as the EPI 2 GEMM wrote it.  16 threads per token: thread (q, g) sums its 24 values in
// (i, fi, reg) order, the 16 partials are combined by shuffles in the fixed tree of wt_ln_block.
__global__ __launch_bounds__(256) void ln_wt_kernel(const float* __restrict__ a, int T, const float* __restrict__ gam,
                                                    const float* __restrict__ bet, float eps,
                                                    _Float16* __restrict__ xh, _Float16* __restrict__ xl) {
    const int tid = threadIdx.x, slot = tid & 15, q = slot >> 2, g = slot & 3;
    int64_t tok = (int64_t)blockIdx.x * 16 + (tid >> 4);
    const bool ok = tok < T;
    tok = ok ? tok : (int64_t)T - 1;
    f32x4 v[3][2];
    float part = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
            v[i][fi] = *reinterpret_cast<const f32x4*>(a + tok * 384 + q * 96 + i * 32 + fi * 16 + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) part = part + v[i][fi][j];
        }
    const int base = (tid & 63) & ~15;  // first lane of this token's 16 threads
    auto tree = [&](float p) {  // ((P0 + P1) + P2) + P3 with Pq = (part(q,0) + part(q,1)) + (part(q,2) + part(q,3))
        p = p + __shfl_xor(p, 1, 64);
        p = p + __shfl_xor(p, 2, 64);
        const float p0 = __shfl(p, base, 64), p1 = __shfl(p, base + 4, 64), p2 = __shfl(p, base + 8, 64),
                    p3 = __shfl(p, base + 12, 64);
        return ((p0 + p1) + p2) + p3;
    };
    const float mean = tree(part) / 384.0f;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][fi][j] - mean;
                v[i][fi][j] = d;
                sq = fmaf(d, d, sq);
            }
    const float var = tree(sq) / 384.0f;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
            const int feat = q * 96 + i * 32 + fi * 16 + 4 * g;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + feat);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(bet + feat);
            f32x4 y;
            half4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fmaf(v[i][fi][j] * rstd, gm[j], bt[j]);
            split_act4(y, hi, lo);
            if (ok) {
                *reinterpret_cast<half4*>(xh + tok * 384 + feat) = hi;
                *reinterpret_cast<half4*>(xl + tok * 384 + feat) = lo;
            }
        }
}

constexpr int LN_LD = 388;  // floats per staged output row in LDS (+16 B: the 16-B accesses of consecutive tokens hit distinct banks)

// ---------------------------------------------------------------- f16x3 linear layers, slab-ring form
// out^T = W . X^T with the weights streamed straight from L2 into registers (packed fragment order) and the
// token slab staged through LDS.  Block = 4 waves; wave q owns NTW 32-feature blocks x TTW 32-token blocks.
//   EPI 0: out fp32 [T, N] = acc * 2^-14 + bias           (QKV)
//   EPI 1: erf-GELU (tf:336), result as f16 hi/lo planes   (FFN-up of small batches)
//   EPI 2: residual + bias (planes rh / rl, row stride N) added to the accumulators after the K loop, out fp32 =
//          acc * 2^-14: the LayerNorm input of attention-out / FFN-down for small batches (ln_wt_kernel follows)
// <1, 1, 4, EPI>: the small-batch form (weights four k-steps ahead) (<= 512 tokens and the remainder of a batch): 32-token x 128-feature
// workgroups, latency-bound.  <3, 2, 1, EPI>: the 64-token x 384-feature form of the UNFUSED reference chain
// (ICREC_FUSE=0: tests compare the fused kernels against it bit for bit).
// Each lane holds 4 consecutive features of one token per register group: 16-B (fp32) / 8-B (planes) stores.
template <int NTW, int TTW, int D, int EPI>
__global__ __launch_bounds__(256, 2) void wt_linear_kernel(const _Float16* __restrict__ Xh,
                                                           const _Float16* __restrict__ Xl, int T, int K,
                                                           const _Float16* __restrict__ Wp, int N,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           _Float16* __restrict__ oh, _Float16* __restrict__ ol,
                                                           int n_blocks_n) {
    constexpr bool STAGED = (NTW == 3 && TTW == 2 && EPI != 1);  // results leave through an LDS stage, coalesced
    constexpr int SM = (STAGED && 32 * LN_LD * 4 > XRing<TTW>::BYTES) ? 32 * LN_LD * 4 : XRing<TTW>::BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SM];
    const int lane = threadIdx.x & 63, q = wave_uniform(threadIdx.x >> 6), c = lane & 15, g = lane >> 4;
    // the n_blocks_n workgroups that read the same token rows get consecutive logical ids = the same XCD = one L2
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_blocks_n, nb = bid % n_blocks_n;
    const int64_t m0 = (int64_t)mt * (32 * TTW);
    const int nt0 = (nb * 4 + q) * NTW;
    Acc32 acc[NTW][TTW];
    wt_kloop<NTW, TTW, D>(acc, Wp, nt0, K, Xh, Xl, m0, T, smem);  // ends with a barrier: the slab ring is free
    if constexpr (EPI == 2) {  // oh / ol carry the residual planes here: acc += r
        if constexpr (NTW == 3 && TTW == 2) {
            ResRows r0, r1;
            wt_res_rows_load(r0, q, 0, oh + nb * 384, ol + nb * 384, m0, T);
            wt_res_rows_load(r1, q, 1, oh + nb * 384, ol + nb * 384, m0, T);
            wt_res_rows_acc<true, 2, 0>(acc, r0, q, bias + nb * 384, smem + q * LNT_TILE);
            wt_res_rows_acc<true, 2, 1>(acc, r1, q, bias + nb * 384, smem + q * LNT_TILE);
            __syncthreads();  // the private tiles become the output stage
        } else {
            static_assert(NTW == 1 && TTW == 1, "residual: 3 x 2 or 1 x 1 wave tiles");
            wt_res_add_direct(acc, nt0, bias, oh, ol, N, m0, T);
        }
    }
    if constexpr (STAGED) {
        // [384 features x 32 tokens] per pass -> stage[token][feature] (16-B LDS writes), then 16-B chunks in flat
        // order: every wave store instruction writes 1 KB of at most two output rows
        float* const stage = reinterpret_cast<float*>(smem);
        const int n0 = nb * 384;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi) {
                    const int fl = q * 96 + i * 32 + fi * 16 + 4 * g;
                    f32x4 b = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    if (EPI == 0) b = *reinterpret_cast<const f32x4*>(bias + n0 + fl);
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti) {
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            v[j] = EPI == 2 ? acc[i][tt].t[fi][ti][j] * WT_UNSCALE : fmaf(acc[i][tt].t[fi][ti][j], WT_UNSCALE, b[j]);
                        *reinterpret_cast<f32x4*>(stage + (ti * 16 + c) * LN_LD + fl) = v;
                    }
                }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int f = threadIdx.x + 256 * k, row = f / 96, ch = f - row * 96;
                const int64_t tok = m0 + tt * 32 + row;
                if (tok < T)
                    *reinterpret_cast<f32x4*>(out + tok * N + n0 + ch * 4) = *reinterpret_cast<const f32x4*>(stage + row * LN_LD + ch * 4);
            }
            if (tt == 0) __syncthreads();
        }
        ICREC_STAMP(0, 30);
        return;
    }

#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
            const int feat = (nt0 + i) * 32 + fi * 16 + 4 * g;
            f32x4 b = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (EPI != 2) b = *reinterpret_cast<const f32x4*>(bias + feat);
#pragma unroll
            for (int tt = 0; tt < TTW; ++tt)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    const int64_t tok = m0 + tt * 32 + ti * 16 + c;
                    if (tok < T) {
                        if (EPI == 1) {
                            half4 hi, lo;
                            gelu_split4(acc[i][tt].t[fi][ti], b, hi, lo);
                            *reinterpret_cast<half4*>(oh + tok * N + feat) = hi;
                            *reinterpret_cast<half4*>(ol + tok * N + feat) = lo;
                        } else {
                            f32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                v[j] = EPI == 2 ? acc[i][tt].t[fi][ti][j] * WT_UNSCALE : fmaf(acc[i][tt].t[fi][ti][j], WT_UNSCALE, b[j]);
                            *reinterpret_cast<f32x4*>(out + tok * N + feat) = v;
                        }
                    }
                }
        }
    ICREC_STAMP(0, 30);
}

// ---------------------------------------------------------------- resident activation image (fused FFN, QKV)
// [64 tokens][384 k] halfs per plane in LDS: 768-B rows as three 256-B sub-rows, 16-B chunk ch of token row t at
// sub-row ch >> 4, slot (ch ^ t) & 15.  A fragment read of token half ti, k-step ks — lane (c, g): row ti*16 + c (+ 32 tt),
// chunk 4 ks + g — puts the 16 lanes of every ds_read_b128 lane group on 16 distinct slots.
constexpr int FFN2_X_BYTES = 2 * FFN2_XPLANE;            // hi, lo (FFN2_XPLANE / ffn_x_pos: defined above wt_ln_block)

// the block's activation planes -> LDS (NT threads, all of them call)
template <int NT>
__device__ __forceinline__ void ffn_x_stage(char* Xs, const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                            int64_t m0, int64_t T) {
    constexpr int N = 64 * 48 / NT;
    u32x4 vh[N], vl[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int id = threadIdx.x + NT * i, row = id / 48, ch = id - row * 48;
        int64_t gr = m0 + row;
        gr = gr < T ? gr : T - 1;
        vh[i] = *reinterpret_cast<const u32x4*>(xh + gr * 384 + ch * 8);
        vl[i] = *reinterpret_cast<const u32x4*>(xl + gr * 384 + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int id = threadIdx.x + NT * i, row = id / 48, ch = id - row * 48;
        const int pos = ffn_x_pos(row, ch);
        *reinterpret_cast<u32x4*>(Xs + pos) = vh[i];
        *reinterpret_cast<u32x4*>(Xs + FFN2_XPLANE + pos) = vl[i];
    }
}

// Per-lane base addresses of the fragment reads from the resident image: xb[tt][ti] = row * 768 + ((g ^ c) << 4) with
// row = tt*32 + ti*16 + c; the fragment of k-step ks sits at (xb ^ ((ks & 3) << 6)) + (ks >> 2) * 256 (the XOR only
// touches bits 6-7, which the row term leaves zero).
__device__ __forceinline__ void ffn_x_bases(int (&xb)[2][2], int c, int g) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) xb[tt][ti] = (tt * 32 + ti * 16 + c) * 768 + ((g ^ c) << 4);
}
__device__ __forceinline__ void ffn_x_frag(XFrag& x, const char* Xs, const int (&xb)[2][2], int tt, int ks) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int pos = (xb[tt][ti] ^ ((ks & 3) << 6)) + (ks >> 2) * 256;
        x.h[ti] = *reinterpret_cast<const half8*>(Xs + pos);
        x.l[ti] = *reinterpret_cast<const half8*>(Xs + FFN2_XPLANE + pos);
    }
}

// ---------------------------------------------------------------- small batches: every GEMM on twice as many CUs
// A single request's GEMMs are a handful of workgroups, each pulling its weight fragments through ONE CU's vector L1: FFN-down
// (K = 1,536: 48 k-steps, the longest dependent chain of a request) moved 768 KB per 32-token x 128-feature workgroup, twelve
// workgroups in all.  Here a wave owns ONE 16-feature tile (half a 32-feature block: two of a k-step's four fragments, six
// MFMAs) and a workgroup 32 tokens x 64 features: twice the workgroups, half the bytes and half the MFMA chain per CU
// (measured, same box: FFN-down -12 us per request, attention-out another -10 us; profiles/r04_single_request_anatomy.txt).
// Per output the chain is wt_kloop's (k-steps ascending, (w_hi,x_hi), (w_lo,x_hi), (w_hi,x_lo)): identical bits.
template <int EPI, int D = 4>  // 0: out = acc * 2^-14 + bias; 2: residual planes rh / rl + bias added after the loop, out = that * 2^-14
__global__ __launch_bounds__(256, 2) void wt_linear_half_kernel(const _Float16* __restrict__ Xh,
                                                                    const _Float16* __restrict__ Xl, int T, int K,
                                                                    const _Float16* __restrict__ Wp, int N,
                                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                                    const _Float16* __restrict__ rh,
                                                                    const _Float16* __restrict__ rl, int n_blocks_n) {
    __shared__ __attribute__((aligned(16))) char smem[XRing<1>::BYTES];
    const int lane = threadIdx.x & 63, q = wave_uniform(threadIdx.x >> 6), c = lane & 15, g = lane >> 4;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_blocks_n, nb = bid % n_blocks_n;
    const int64_t m0 = (int64_t)mt * 32;
    const int ft = nb * 4 + q, nt = ft >> 1, fi = ft & 1;  // 16-feature tile, its 32-feature block, which half
    const unsigned lo8 = lane * 8;
    const int KS = K / 32, nslab = K / 64;
    const _Float16* const wp = Wp + wt_frag_off(nt, 0, KS) + (size_t)fi * (2 * WT_FRAG);
    static_assert(D == 4 || D == 6 || D == 8, "ring depth: 4, 6 (K a multiple of 384) or 8 (K a multiple of 256) k-steps");
    half8 wh[D], wl[D];
    auto w_load1 = [&](int slot, int ks) {
        const _Float16* p = wp + (size_t)ks * (4 * WT_FRAG);
        wh[slot] = *reinterpret_cast<const half8*>(p + lo8);
        wl[slot] = *reinterpret_cast<const half8*>(p + WT_FRAG + lo8);
    };
    f32x4 acc[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
    u32x4 xa[2], xb[2];
    x_load<1>(xa, Xh, Xl, m0, T, K, 0);
#pragma unroll
    for (int d = 0; d < D; ++d) w_load1(d, d);
    if (nslab > 1) x_load<1>(xb, Xh, Xl, m0, T, K, 1);
    x_store<1>(xa, smem);
    if (nslab > 2) x_load<1>(xa, Xh, Xl, m0, T, K, 2);
    __syncthreads();
    auto slab = [&](int s, auto parity, u32x4 (&xnext)[2]) {  // as wt_kloop<1, 1, D>: ring slot of k-step 2 s + j is 2 PAR + j
        constexpr int PAR = decltype(parity)::value;
        const char* st = smem + (s & 1) * XRing<1>::STAGE_BYTES;
        XFrag x[2];
        x_frag<1>(x[0], st, 0, 0, c, g);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 0) {
                x_frag<1>(x[1], st, 0, 1, c, g);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[2 * PAR + j], x[j].h[ti], acc[ti], 0, 0, 0);
                acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[2 * PAR + j], x[j].h[ti], acc[ti], 0, 0, 0);
                acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[2 * PAR + j], x[j].l[ti], acc[ti], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            int nk = 2 * s + j + D;
            nk = nk < KS ? nk : KS - 1;
            w_load1(2 * PAR + j, nk);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s + 1 < nslab) {
            x_store<1>(xnext, smem + ((s + 1) & 1) * XRing<1>::STAGE_BYTES);
            if (s + 3 < nslab) x_load<1>(xnext, Xh, Xl, m0, T, K, s + 3);
        }
        __syncthreads();
    };
    // the ring slot of k-step 2 s + j is 2 (s % (D / 2)) + j, the register buffer of the next slab alternates with s & 1: a trip of
    // the unrolled loop covers lcm(D / 2, 2) slabs (the host checks that nslab is a multiple of it)
    if constexpr (D == 6) {
        for (int s = 0; s < nslab; s += 6) {
            slab(s, std::integral_constant<int, 0>{}, xb);
            slab(s + 1, std::integral_constant<int, 1>{}, xa);
            slab(s + 2, std::integral_constant<int, 2>{}, xb);
            slab(s + 3, std::integral_constant<int, 0>{}, xa);
            slab(s + 4, std::integral_constant<int, 1>{}, xb);
            slab(s + 5, std::integral_constant<int, 2>{}, xa);
        }
    } else {
        for (int s = 0; s < nslab; s += D / 2) {
            slab(s, std::integral_constant<int, 0>{}, xb);
            slab(s + 1, std::integral_constant<int, 1>{}, xa);
            if constexpr (D == 8) {
                slab(s + 2, std::integral_constant<int, 2>{}, xb);
                slab(s + 3, std::integral_constant<int, 3>{}, xa);
            }
        }
    }
    const int feat = nt * 32 + fi * 16 + 4 * g;
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + feat);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int64_t tok = m0 + ti * 16 + c;
        if (tok < T) {
            f32x4 v;
            if constexpr (EPI == 2) {
                const half4 a = *reinterpret_cast<const half4*>(rh + tok * N + feat);
                const half4 d = *reinterpret_cast<const half4*>(rl + tok * N + feat);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (acc[ti][j] + res_init_val(a[j], d[j], b[j])) * WT_UNSCALE;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaf(acc[ti][j], WT_UNSCALE, b[j]);
            }
            *reinterpret_cast<f32x4*>(out + tok * N + feat) = v;
        }
    }
}

// ---------------------------------------------------------------- small batches: LayerNorm folded into the consuming GEMM
// A replayed hipGraph pays ~5 us per node whatever the node does (profiles/r04_single_request_anatomy.txt), and a single
// request's LayerNorm nodes did 5 us of nothing else.  The two K = 384 GEMMs that CONSUME a LayerNorm's output (FFN-up
// behind the attention-output LayerNorm, the next layer's QKV projection behind the FFN LayerNorm) read whole token rows
// anyway, so every workgroup normalises its 32 rows itself: t1 rows (fp32: dense + bias + residual, as the EPI 2 GEMM
// wrote them) -> ln_wt_kernel's arithmetic, thread for thread (16 threads per token, the same partial sums, the same
// shuffle tree: identical bits) -> the planes go into a resident LDS image ([32][384] x 2, layout ffn_x_pos) and - from
// the workgroups of feature block 0 only - to xh / xl, the residual of the next EPI 2 GEMM.  The weight ring's first four
// k-steps are requested BEFORE the LayerNorm, so the L2 latency of the first fragments runs under it.  K loop: twelve
// straight-line k-steps off the image, one 32 x 32 block per wave, per output the chain of wt_kloop (same bits as
// wt_linear_kernel<1, 1, 4, EPI> on ln_wt_kernel's planes).  EPI 0: fp32 out + bias; EPI 1: erf-GELU planes.
constexpr int LNIN_XPLANE = 32 * 768;
template <int EPI, bool HALF = false>  // HALF: a wave owns one 16-feature tile (64 features per workgroup) instead of a 32-feature block
__global__ __launch_bounds__(256, 2) void wt_linear_lnin_kernel(const float* __restrict__ a, int T,
                                                                const float* __restrict__ gam,
                                                                const float* __restrict__ bet, float eps,
                                                                _Float16* __restrict__ xh, _Float16* __restrict__ xl,
                                                                const _Float16* __restrict__ Wp, int N,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                _Float16* __restrict__ oh, _Float16* __restrict__ ol,
                                                                int n_blocks_n) {
    constexpr int KS1 = 12;
    __shared__ __attribute__((aligned(16))) char Xs[2 * LNIN_XPLANE];
    const int tid = threadIdx.x, lane = tid & 63, wq = wave_uniform(tid >> 6), c = lane & 15, g = lane >> 4;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_blocks_n, nb = bid % n_blocks_n;
    const int64_t m0 = (int64_t)mt * 32;
    const int ft = nb * 4 + wq;                       // HALF: 16-feature tile index
    const int nt = HALF ? ft >> 1 : ft, hf = ft & 1;  // the 32-feature block; HALF: which half of it
    const unsigned lo8 = lane * 8;
    constexpr int RD = 8;  // weight ring depth in k-steps: eight of the twelve are requested in front of the LayerNorm
    WFrag w[RD][1];  // HALF: only [hf] of h / l is loaded and used
    const _Float16* const wp1[1] = {Wp + wt_frag_off(nt, 0, KS1)};
    auto w_ring_load = [&](int slot, int ks) {
        if constexpr (HALF) {
            const _Float16* p = wp1[0] + (size_t)ks * (4 * WT_FRAG) + (size_t)hf * (2 * WT_FRAG);
            w[slot][0].h[0] = *reinterpret_cast<const half8*>(p + lo8);
            w[slot][0].l[0] = *reinterpret_cast<const half8*>(p + WT_FRAG + lo8);
        } else {
            w_load<1>(w[slot], wp1, ks, lo8);
        }
    };
#pragma unroll
    for (int d = 0; d < RD; ++d) w_ring_load(d, d);
    f32x4 bv[2];
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) bv[fi] = *reinterpret_cast<const f32x4*>(bias + nt * 32 + (HALF ? hf : fi) * 16 + 4 * g);
    // ---- LayerNorm of the block's 32 rows: two passes of 16 tokens, ln_wt_kernel's thread mapping and order
    {
        const int slot = tid & 15, q = slot >> 2, gg = slot & 3;
        const int base = (tid & 63) & ~15;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 16 + (tid >> 4);
            int64_t tok = m0 + row;
            const bool ok = tok < T;
            tok = ok ? tok : (int64_t)T - 1;
            f32x4 v[3][2];
            float part = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi) {
                    v[i][fi] = *reinterpret_cast<const f32x4*>(a + tok * 384 + q * 96 + i * 32 + fi * 16 + 4 * gg);
#pragma unroll
                    for (int j = 0; j < 4; ++j) part = part + v[i][fi][j];
                }
            auto tree = [&](float p) {
                p = p + __shfl_xor(p, 1, 64);
                p = p + __shfl_xor(p, 2, 64);
                const float p0 = __shfl(p, base, 64), p1 = __shfl(p, base + 4, 64), p2 = __shfl(p, base + 8, 64),
                            p3 = __shfl(p, base + 12, 64);
                return ((p0 + p1) + p2) + p3;
            };
            const float mean = tree(part) / 384.0f;
            float sq = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = v[i][fi][j] - mean;
                        v[i][fi][j] = d;
                        sq = fmaf(d, d, sq);
                    }
            const float var = tree(sq) / 384.0f;
            const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi) {
                    const int feat = q * 96 + i * 32 + fi * 16 + 4 * gg;
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + feat);
                    const f32x4 bt = *reinterpret_cast<const f32x4*>(bet + feat);
                    f32x4 y;
                    half4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = fmaf(v[i][fi][j] * rstd, gm[j], bt[j]);
                    split_act4(y, hi, lo);
                    const int pos = ffn_x_pos(row, feat >> 3) + 8 * ((feat >> 2) & 1);
                    *reinterpret_cast<half4*>(Xs + pos) = hi;
                    *reinterpret_cast<half4*>(Xs + LNIN_XPLANE + pos) = lo;
                    if (nb == 0 && ok) {
                        *reinterpret_cast<half4*>(xh + tok * 384 + feat) = hi;
                        *reinterpret_cast<half4*>(xl + tok * 384 + feat) = lo;
                    }
                }
        }
    }
    __syncthreads();  // image resident
    // ---- K = 384 off the image
    int xb[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) xb[ti] = (ti * 16 + c) * 768 + ((g ^ c) << 4);
    auto x_frag1 = [&](XFrag& x, int ks) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int pos = (xb[ti] ^ ((ks & 3) << 6)) + (ks >> 2) * 256;
            x.h[ti] = *reinterpret_cast<const half8*>(Xs + pos);
            x.l[ti] = *reinterpret_cast<const half8*>(Xs + LNIN_XPLANE + pos);
        }
    };
    Acc32 S;
    acc_zero(S);
    XFrag x[2];
    x_frag1(x[0], 0);
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        if (ks + 1 < KS1) {
            x_frag1(x[(ks + 1) & 1], ks + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (HALF) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                S.t[0][ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks % RD][0].h[0], x[ks & 1].h[ti], S.t[0][ti], 0, 0, 0);
                S.t[0][ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks % RD][0].l[0], x[ks & 1].h[ti], S.t[0][ti], 0, 0, 0);
                S.t[0][ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks % RD][0].h[0], x[ks & 1].l[ti], S.t[0][ti], 0, 0, 0);
            }
        } else {
            wt_mma_block(S, w[ks % RD][0], x[ks & 1]);
        }
        if (ks + RD < KS1) w_ring_load(ks % RD, ks + RD);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int fi = 0; fi < (HALF ? 1 : 2); ++fi) {
        const int feat = nt * 32 + (HALF ? hf : fi) * 16 + 4 * g;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int64_t tok = m0 + ti * 16 + c;
            if (tok < T) {
                if (EPI == 1) {
                    half4 hi, lo;
                    gelu_split4(S.t[fi][ti], bv[fi], hi, lo);
                    *reinterpret_cast<half4*>(oh + tok * N + feat) = hi;
                    *reinterpret_cast<half4*>(ol + tok * N + feat) = lo;
                } else {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaf(S.t[fi][ti][j], WT_UNSCALE, bv[fi][j]);
                    *reinterpret_cast<f32x4*>(out + tok * N + feat) = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- QKV projection, activation-resident form (large batches)
// out[T, N] = X . W^T + bias for a block of 64 tokens and ALL N = 1,152 features on the eight waves of a workgroup,
// built like the producer half of the fused kernel: the block's 64 x 384 activation planes are resident in LDS (96 KB),
// every wave walks whole 32-feature blocks - wave w: blocks w, w + 8, ... (a SIMD hosts waves s and s + 4 = 9 of the
// 36 blocks) - with K = 384 as 12 straight-line k-steps, the weight ring 4 k-steps deep and running across block
// boundaries, no workgroup barrier.  Results leave through a private per-wave LDS tile ([32 tokens][32 features] fp32,
// 144-B rows) as 16-B chunks: 128 B per token row per store.
// Per output the MFMA chain is wt_kloop's: the same bits as wt_linear_kernel<3, 2, 1, 0>.
// Two callers: qkv_resident_kernel (layer 0, after the embeddings) and the fused post-attention kernel, which ends
// with the NEXT layer's projection of the block it has just normalised.
constexpr int QKVR_STG_LD = 36;                                  // floats per staged row (+16 B)
constexpr int QKVR_STG_BYTES = 32 * QKVR_STG_LD * 4;             // 4,608 B per wave
constexpr int QKVR_LDS = FFN2_X_BYTES + 8 * QKVR_STG_BYTES;      // 135,168 B

__device__ __forceinline__ void qkv_block_walk(const char* Xs, float* stg, const _Float16* __restrict__ Wp,
                                               const float* __restrict__ bias, float* __restrict__ out, int N, int64_t m0,
                                               int64_t T, int wave, int lane, int c, int g, unsigned lo8) {
    constexpr int KS1 = 12;
    const int NT = N / 32;
    int xb[2][2];
    ffn_x_bases(xb, c, g);
    WFrag w[4][1];
    {
        const _Float16* const wp0[1] = {Wp + wt_frag_off(wave, 0, KS1)};
#pragma unroll
        for (int d = 0; d < 4; ++d) w_load<1>(w[d], wp0, d, lo8);
    }
    for (int nt = wave; nt < NT; nt += 8) {
        const int nn = nt + 8 < NT ? nt + 8 : nt;  // past the last block: re-read this one's fragments (never consumed)
        const _Float16* const wp1[1] = {Wp + wt_frag_off(nt, 0, KS1)};
        const _Float16* const wpn[1] = {Wp + wt_frag_off(nn, 0, KS1)};
        f32x4 bv[2];  // loaded BEFORE the k-loop (vmcnt counts in order: behind the ring it would wait for the whole ring)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) bv[fi] = *reinterpret_cast<const f32x4*>(bias + nt * 32 + fi * 16 + 4 * g);
        Acc32 S[1][2];
        acc_zero(S[0][0]);
        acc_zero(S[0][1]);
        XFrag x[2];
        ffn_x_frag(x[0], Xs, xb, 0, 0);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int u = 2 * ks + tt;
                if (u + 1 < 2 * KS1) {
                    ffn_x_frag(x[(u + 1) & 1], Xs, xb, (u + 1) & 1, (u + 1) >> 1);
                    __builtin_amdgcn_sched_barrier(0);  // the next unit's LDS reads are issued before this unit's 12 MFMAs
                }
                if (tt == 0) wt_mma<1, 2, 0>(S, w[ks & 3], x[u & 1]);
                else wt_mma<1, 2, 1>(S, w[ks & 3], x[u & 1]);
            }
            if (ks + 4 < KS1) w_load<1>(w[ks & 3], wp1, ks + 4, lo8);
            else w_load<1>(w[ks & 3], wpn, ks + 4 - KS1, lo8);
            __builtin_amdgcn_sched_barrier(0);  // pin the prefetch to its k-step
        }
        // ---- this block out: [32 tokens][32 features] per pass through the wave's private LDS tile
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaf(S[0][tt].t[fi][ti][j], WT_UNSCALE, bv[fi][j]);
                    *reinterpret_cast<f32x4*>(stg + (ti * 16 + c) * QKVR_STG_LD + fi * 16 + 4 * g) = v;
                }
            lds_order();
            f32x4 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int f = lane + 64 * k;
                o[k] = *reinterpret_cast<const f32x4*>(stg + (f >> 3) * QKVR_STG_LD + (f & 7) * 4);
            }
            lds_order();  // the tile is free for the next pass before the stores are issued
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int f = lane + 64 * k;
                const int64_t tok = m0 + tt * 32 + (f >> 3);
                if (tok < T) *reinterpret_cast<f32x4*>(out + tok * N + nt * 32 + (f & 7) * 4) = o[k];
            }
        }
    }
}

// ---------------------------------------------------------------- fused FFN (large batches)
// x <- LN(W2 . gelu(W1 . x + b1) + b2 + x)   (tf:334-351: BertIntermediate, BertOutput) for a block of 64 tokens,
// without the [T, 1536] intermediate ever leaving the CU.  The 1,536 intermediate features are walked in 12 chunks
// of 128; per chunk
//   P1  S^T[128 x 64 tok] = W1[chunk] . X^T          K = 384; wave q: intermediates q*32..+31 (2 token blocks)
//   G   H = split(gelu(S * 2^-14 + b1))              registers -> 8-byte LDS writes (4 consecutive k of a token)
//   P2  Y^T[384 x 64 tok] += W2[:, chunk] . H^T      K = 128; wave q: features q*96..+95 (3 x 2 blocks, 96 regs)
// then the residual + LayerNorm epilogue.  Per output the MFMA chain is exactly wt_kloop's (k-steps ascending,
// the same three products per step), so the result equals FFN-up -> FFN-down -> add_ln through wt_linear_kernel
// bit for bit.
//
// Producer / consumer form: one 8-wave workgroup per CU owning ALL 160 KB of LDS:
//   * the block's 64 x 384 activation planes stay resident in LDS (96 KB) for all 12 chunks — no restream, no
//     slab barriers;
//   * waves 0-3 (producers) run P1 + GELU for chunk c+1 and write H into one half of a double buffer (2 x 32 KB)
//     while waves 4-7 (consumers) run P2 of chunk c from the other half: ONE workgroup barrier per chunk;
//   * each SIMD hosts one producer and one consumer.  They issue the same number of MFMAs per chunk (288 each);
//     the GELU of chunk c-1 (VALU) is software-pipelined between the MFMAs of P1(c), one 16x16 tile per k-step;
//   * specialisation frees registers for deep weight prefetch rings (W1: 4 k-steps, W2: 2 k-steps ahead).
constexpr int FFN_IC = 128;
constexpr int FFN2_HPLANE = 64 * 256;                    // [64 tokens][128 k] halfs
constexpr int FFN2_HBUF = 2 * FFN2_HPLANE;               // hi, lo
constexpr int FFN2_LDS = FFN2_X_BYTES + 2 * FFN2_HBUF;   // 163,840 B = the whole LDS of a CU
static_assert(FFN2_LDS == 160 * 1024, "fused FFN LDS budget");

__device__ __forceinline__ void bar_lds() {  // LDS hand-off barrier that leaves global loads in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

//
// AO = true (the product): the kernel is the whole post-attention half of a layer.  Its prologue is the attention
// output projection + residual + LayerNorm (tf:289-293) of the same 64 tokens:
//   the block's CONTEXT planes (ch / cl) go to the resident image; the producer waves run K = 384 over Wo while the
//   consumer waves - they own the [96 features x 64 tokens] accumulator layout and its LayerNorm epilogue - stage
//   residual + bias; the consumers add the two and LayerNorm straight into the resident image: x1 = LN(Wo . ctx + bo
//   + x) never travels to HBM (-2 x 768 B per token per layer each way, one launch fewer), the FFN proceeds on it as
//   before.  Per output the same chain and the same LayerNorm order as wt_linear_kernel<.., 2> + ln_wt_kernel:
//   identical bits.
// VAR (tools/ffn_bench.hip only; the product uses 0), timing ablations: 1 = no GELU, 4 = no in-loop weight loads,
// 8 = no in-loop LDS fragment reads, 32 = no sched_group_barrier interleave, 64 = ~16 k idle cycles at the start
template <int VAR, bool AO = false>
__global__ __launch_bounds__(512, 2) void ffn_fused2_kernel(_Float16* __restrict__ xh, _Float16* __restrict__ xl,
                                                            int T, int I,
                                                            const _Float16* __restrict__ W1p,
                                                            const float* __restrict__ b1,
                                                            const _Float16* __restrict__ W2p,
                                                            const float* __restrict__ b2,
                                                            const float* __restrict__ gam,
                                                            const float* __restrict__ bet, float eps,
                                                            const _Float16* __restrict__ ch = nullptr,
                                                            const _Float16* __restrict__ cl = nullptr,
                                                            const _Float16* __restrict__ Wop = nullptr,
                                                            const float* __restrict__ bo = nullptr,
                                                            const float* __restrict__ gam1 = nullptr,
                                                            const float* __restrict__ bet1 = nullptr,
                                                            const _Float16* __restrict__ Wqp = nullptr,
                                                            const float* __restrict__ bq = nullptr,
                                                            float* __restrict__ qkv_out = nullptr, int Nq = 0) {
    constexpr int KS1 = 12;  // k-steps of 32 over K = 384
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    char* const Xs = smem2;
    char* const Hs = smem2 + FFN2_X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), q = wave & 3;
    int c = lane & 15, g = lane >> 4;
    const bool producer = wave < 4;
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int NC = I / FFN_IC, KS2 = I / 32;

    ICREC_STAMP(0, 0);
    ICREC_STAMP(4, 0);
    ICREC_STAMP_RT(0, 62);
    if constexpr ((VAR & 64) != 0) {  // harness: ~16 k idle cycles per workgroup (does time follow cycles or power?)
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
    }
    unsigned lo8 = lane * 8;
    if constexpr (AO) {
        // x1 = LayerNorm(Wo . ctx + bo + x) -> the resident image.
        // All eight waves share the work; wave (q, half) owns features q*96 .. +95 of token block `half`:
        //  * it stages its residual rows into the accumulator layout (global -> its private LDS tile -> r);
        //  * the GEMM runs in the producers' form (one 32-feature block x 64 tokens at a time, weight ring 4 k-steps deep,
        //    the loop of qkv_resident_kernel): producer p takes blocks p, p + 4 and p + 8 while the consumer on its SIMD
        //    stages its residual rows (a 2 : 1 split with the consumers' rows staged first left the producers waiting);
        //  * the products travel through the context image, dead once every wave has left its K loops: 12 blocks x 2
        //    token blocks x 4 tiles x 1 KB = the 96 KB of the image, each tile in the lane order both sides hold it in
        //    (conflict-free 16-B accesses); every wave picks up the 12 tiles of its (q, half), adds r and takes part in
        //    the LayerNorm of its token block, which lands in the image.
        // Scratch (LayerNorm exchange, eight transposition tiles): the H buffers, idle until the FFN starts.
        // (the two roles run the same steps from separate instantiations: a value defined on one path only - the early
        // residual rows, a ring - would otherwise be carried, and spilled, along the other path too)
        auto prologue = [&](auto role_tag) {
            constexpr bool PROD = decltype(role_tag)::value;
            constexpr int tb = PROD ? 0 : 1;  // the token block this wave stages, adds and normalises
            ResRows rr;  // the wave's residual rows.  Consumers (one GEMM block, registers to spare): requested first of
                         // all, they arrive under the staging of the context planes and are transposed before the K
                         // loop.  Producers (two blocks: beside two accumulator sets, the ring and the fragments the
                         // rows would spill): requested behind the K loops - they finish first and wait anyway.
            if (!PROD) wt_res_rows_load(rr, q, tb, xh, xl, m0, T);
            ffn_x_stage<512>(Xs, ch, cl, m0, T);  // the context planes
            ICREC_STAMP(0, 44); ICREC_STAMP(4, 44);
            int xb[2][2];
            ffn_x_bases(xb, c, g);
            WFrag w[4][1];
            {
                const _Float16* const wp0[1] = {Wop + wt_frag_off(q, 0, KS1)};
#pragma unroll
                for (int d = 0; d < 4; ++d) w_load<1>(w[d], wp0, d, lo8);
            }
            __syncthreads();  // context planes resident
            ICREC_STAMP(0, 45); ICREC_STAMP(4, 45);
            // K = 384 of one block; the ring continues into block `nn` (or re-reads this one's last fragments, never consumed)
            auto ao_block = [&](Acc32 (&S)[2], int nt, int nn) {
                const _Float16* const wp1[1] = {Wop + wt_frag_off(nt, 0, KS1)};
                const _Float16* const wpn[1] = {Wop + wt_frag_off(nn, 0, KS1)};
                acc_zero(S[0]);
                acc_zero(S[1]);
                XFrag x[2];
                ffn_x_frag(x[0], Xs, xb, 0, 0);
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int u = 2 * ks + tt;
                        if (u + 1 < 2 * KS1) {
                            ffn_x_frag(x[(u + 1) & 1], Xs, xb, (u + 1) & 1, (u + 1) >> 1);
                            __builtin_amdgcn_sched_barrier(0);  // the next unit's LDS reads before this unit's 12 MFMAs
                        }
                        wt_mma_block(S[tt], w[ks & 3][0], x[u & 1]);
                    }
                    if (ks + 4 < KS1) w_load<1>(w[ks & 3], wp1, ks + 4, lo8);
                    else w_load<1>(w[ks & 3], wpn, ks + 4 - KS1, lo8);
                    __builtin_amdgcn_sched_barrier(0);  // pin the prefetch to its k-step
                }
            };
            auto put = [&](const Acc32 (&S)[2], int nt) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                        for (int ti = 0; ti < 2; ++ti)
                            *reinterpret_cast<f32x4*>(Xs + (((nt * 2 + tt) * 2 + fi) * 2 + ti) * 1024 + lane * 16) = S[tt].t[fi][ti];
            };
            Acc32 Y[3][1];
            auto finish_r = [&]() {  // r is FINISHED here: otherwise the compiler sinks the final fma of every element below
                                     // the barriers and carries its two inputs instead
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                        for (int ti = 0; ti < 2; ++ti) asm volatile("" : "+v"(Y[i][0].t[fi][ti]));
            };
            if constexpr (PROD) {
                Acc32 S0[2], S1[2], S2[2];
                ao_block(S0, q, q + 4);
                ao_block(S1, q + 4, q + 8);
                ao_block(S2, q + 8, q + 8);
                ICREC_STAMP(0, 28);
                wt_res_rows_load(rr, q, tb, xh, xl, m0, T);
                __syncthreads();  // every wave has left the context image: it takes the products
                put(S0, q);
                put(S1, q + 4);
                put(S2, q + 8);
                wt_res_rows_acc<false, 1, 0>(Y, rr, q, bo, Hs + LNT_RED + wave * LNT_TILE);  // per-wave private tiles
                finish_r();
            } else {
                wt_res_rows_acc<false, 1, 0>(Y, rr, q, bo, Hs + LNT_RED + wave * LNT_TILE);
                ICREC_STAMP(4, 46);
                finish_r();
                ICREC_STAMP(4, 28);
                __syncthreads();  // every wave has left the context image
            }
            __syncthreads();  // products visible
            ICREC_STAMP(0, 29);
            ICREC_STAMP(4, 29);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti) {
                        const f32x4 sv = *reinterpret_cast<const f32x4*>(Xs + ((((q * 3 + i) * 2 + tb) * 2 + fi) * 2 + ti) * 1024 + lane * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) Y[i][0].t[fi][ti][j] = sv[j] + Y[i][0].t[fi][ti][j];  // acc + r
                    }
            ICREC_STAMP(4, 31);
            // its first barrier also orders every wave's reads of the products before the first write of x1
            wt_ln_block<8, 1, 0>(Y, q, tb, wave, nullptr, nullptr, m0, T, gam1, bet1, eps, Hs, [] { __syncthreads(); }, tid, Xs);
        };
        if (producer) prologue(std::true_type{});
        else prologue(std::false_type{});
        ICREC_STAMP(0, 31);
    } else {
        ffn_x_stage<512>(Xs, xh, xl, m0, T);
    }
    if constexpr (AO) {  // the FFN part's per-lane addresses - and with them its first weight-ring loads, 96 registers in
                         // the consumers - are derived from here on, not hoisted above the prologue (where they spill)
        asm volatile("" : "+v"(c), "+v"(g), "+v"(lo8));
    }
    if (producer) {
        int xb[2][2];
        ffn_x_bases(xb, c, g);
        WFrag w[4][1];
        {
            const _Float16* const wp0[1] = {W1p + wt_frag_off(q, 0, KS1)};
#pragma unroll
            for (int d = 0; d < 4; ++d) w_load<1>(w[d], wp0, d, lo8);
        }
        __syncthreads();  // X resident
        ICREC_STAMP(0, 1);
        // Software pipeline: iteration ch runs P1(ch) with the GELU of chunk ch-1 spread over its k-steps (one 16x16
        // tile = 4 consecutive intermediates of one token per lane on 8 of the 12 k-steps), so the producer's VALU
        // work sits between its own MFMAs and the consumers' instead of behind them.  H[ch-1] is handed over at the
        // end of iteration ch.
        Acc32 S[1][2], Sp[1][2];  // this chunk's accumulators, the previous chunk's (being GELU'd)
        f32x4 bias[2], biasp[2];
        acc_zero(Sp[0][0]);
        acc_zero(Sp[0][1]);
        biasp[0] = biasp[1] = bias[0] = bias[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // one pipeline iteration: P1(ch) (MMA = true) with G(ch - 1) spread over its 12 k-steps; the drain iteration
        // (MMA = false) runs only the G slices
        auto iteration = [&](int ch, auto mma_tag) {
            constexpr bool MMA = decltype(mma_tag)::value;
            const _Float16* const wp1[1] = {W1p + wt_frag_off((MMA ? ch : 0) * 4 + q, 0, KS1)};
            const _Float16* const wpn[1] = {W1p + wt_frag_off((MMA && ch + 1 < NC ? ch + 1 : 0) * 4 + q, 0, KS1)};
            if (MMA) {  // this chunk's biases, loaded BEFORE the k-loop: a load issued behind the weight ring would make
                        // its consumer wait for the whole ring (vmcnt counts in order)
                const float* bp = b1 + ch * FFN_IC + q * 32 + 4 * g;
                bias[0] = *reinterpret_cast<const f32x4*>(bp);
                bias[1] = *reinterpret_cast<const f32x4*>(bp + 16);
            }
            acc_zero(S[0][0]);
            acc_zero(S[0][1]);
            char* const Hb = Hs + ((ch + 1) & 1) * FFN2_HBUF;  // H[(ch - 1) & 1]
            XFrag x[2];  // unit u = 2 ks + tt: the next unit's fragments are read under the current one's MFMAs
            half4 ghi, glo;  // the GELU group being assembled
            if (MMA) ffn_x_frag(x[0], Xs, xb, 0, 0);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int u = 2 * ks + tt;
                    if (MMA) {
                        if (u + 1 < 2 * KS1 && !(VAR & 8)) ffn_x_frag(x[(u + 1) & 1], Xs, xb, (u + 1) & 1, (u + 1) >> 1);
                        if (tt == 0) wt_mma<1, 2, 0>(S, w[ks & 3], x[(VAR & 8) ? 0 : (u & 1)]);
                        else wt_mma<1, 2, 1>(S, w[ks & 3], x[(VAR & 8) ? 0 : (u & 1)]);
                        if (tt == 1 && !(VAR & 4)) {  // straight-line refill: this chunk's k-step ks+4, or the next chunk's ks+4-12
                            if (ks + 4 < KS1) w_load<1>(w[ks & 3], wp1, ks + 4, lo8);
                            else w_load<1>(w[ks & 3], wpn, ks + 4 - KS1, lo8);
                        }
                    }
                    if (u % 3 != 2) {  // G of the PREVIOUS chunk, two elements per unit (16 of the 24 units carry a slice): bias +
                                       // erf-GELU + split; a finished group of 4 consecutive k goes out as one 8-byte LDS
                                       // write per plane
                        const int s = u - u / 3, tile = s >> 1, j = 2 * (s & 1);
                        const int gfi = tile >> 2, gtt = (tile >> 1) & 1, gti = tile & 1;
                        const float p0 = fmaf(Sp[0][gtt].t[gfi][gti][j], WT_UNSCALE, biasp[gfi][j]);
                        const float p1 = fmaf(Sp[0][gtt].t[gfi][gti][j + 1], WT_UNSCALE, biasp[gfi][j + 1]);
                        half2w a, d;
                        split_pair_prescaled((VAR & 1) ? p0 : gelu16_wt(p0), (VAR & 1) ? p1 : gelu16_wt(p1), a, d);
                        ghi[j] = a[0];
                        ghi[j + 1] = a[1];
                        glo[j] = d[0];
                        glo[j + 1] = d[1];
                        if (s & 1) {
                            const int tok = gtt * 32 + gti * 16 + c;
                            const int pos = tok * 256 + (((4 * q + 2 * gfi + (g >> 1)) ^ c) << 4) + 8 * (g & 1);
                            *reinterpret_cast<half4*>(Hb + pos) = ghi;  // iteration 0 writes GELU(0) into a buffer nobody reads yet
                            *reinterpret_cast<half4*>(Hb + FFN2_HPLANE + pos) = glo;
                        }
                    }
                    if (MMA && !(VAR & 32)) {  // the next unit's LDS reads first; then one MFMA and a few of the slice's VALU
                                               // instructions, twelve times (the scheduler otherwise sinks the reads to their uses)
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                        for (int m = 0; m < 12; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (u % 3 != 2) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);  // pin the prefetch (and the GELU slice) to its unit
                }
            }
            Sp[0][0] = S[0][0];
            Sp[0][1] = S[0][1];
            biasp[0] = bias[0];
            biasp[1] = bias[1];
        };
        for (int ch = 0; ch < NC; ++ch) {
            iteration(ch, std::true_type{});
            ICREC_STAMP(0, 2 + 2 * ch);
            if (ch > 0) bar_lds();  // B(ch): H[ch - 1] is complete; the consumers have left H[ch & 1]
            ICREC_STAMP(0, 3 + 2 * ch);
        }
        iteration(NC, std::false_type{});
        bar_lds();  // B(NC): H[NC - 1]
        ICREC_STAMP(0, 26);
        ICREC_STAMP(0, 27);
        // ---- the LayerNorm of token block 1: its accumulators come over from the consumers through the (dead) image
        __syncthreads();  // every reader of the LDS is done: image and H buffers are free
        __syncthreads();  // token block 1 parked
        Acc32 Yp[3][1];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
                    Yp[i][0].t[fi][ti] = *reinterpret_cast<const f32x4*>(Xs + ((q * 12) + (i * 2 + fi) * 2 + ti) * 1024 + lane * 16);
        // scratch: the H buffers; x2 goes to global memory and - when the next layer's QKV projection follows - into the
        // image (written behind the LayerNorm's barriers: every parked tile has been picked up by then)
        wt_ln_block<8, 1, 0>(Yp, q, 1, wave, xh, xl, m0, T, gam, bet, eps, Hs, [] { __syncthreads(); }, tid, Wqp != nullptr ? Xs : nullptr);
    } else {
        Acc32 Y[3][2];
        const _Float16* w2p[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) w2p[i] = W2p + wt_frag_off(q * 3 + i, 0, KS2);
        int hb[2][2];  // tok*256 + ((g ^ c) << 4); the fragment of k-step k2 of the chunk sits at hb ^ (k2 << 6)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) hb[tt][ti] = (tt * 32 + ti * 16 + c) * 256 + ((g ^ c) << 4);
        WFrag w[2][3];
#pragma unroll
        for (int d = 0; d < 2; ++d) w_load<3>(w[d], w2p, d, lo8);
        __syncthreads();  // X resident (matches the producers' first barrier)
        ICREC_STAMP(4, 1);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) acc_zero(Y[i][tt]);
        bar_lds();        // B1: H[0] is ready
        for (int ch = 0; ch < NC; ++ch) {
            ICREC_STAMP(4, 2 + 2 * ch);
            const char* const Hb = Hs + (ch & 1) * FFN2_HBUF;
            XFrag x[2];  // unit u = 2 k2 + tt
            auto h_frag = [&](XFrag& xf, int tt, int k2) {
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    const int pos = hb[tt][ti] ^ (k2 << 6);
                    xf.h[ti] = *reinterpret_cast<const half8*>(Hb + pos);
                    xf.l[ti] = *reinterpret_cast<const half8*>(Hb + FFN2_HPLANE + pos);
                }
            };
            h_frag(x[0], 0, 0);
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int u = 2 * k2 + tt;
                    if (u + 1 < 8 && !(VAR & 8)) {
                        h_frag(x[(u + 1) & 1], (u + 1) & 1, (u + 1) >> 1);
                        __builtin_amdgcn_sched_barrier(0);  // issue the next unit's LDS reads before this unit's 36 MFMAs
                    }
                    if (tt == 0) wt_mma<3, 2, 0>(Y, w[k2 & 1], x[(VAR & 8) ? 0 : (u & 1)]);
                    else wt_mma<3, 2, 1>(Y, w[k2 & 1], x[(VAR & 8) ? 0 : (u & 1)]);
                }
                if (!(VAR & 4)) {
                    int nk = ch * 4 + k2 + 2;
                    nk = nk < KS2 ? nk : KS2 - 1;  // past the end: re-read the last fragment (never consumed)
                    w_load<3>(w[k2 & 1], w2p, nk, lo8);
                }
                __builtin_amdgcn_sched_barrier(0);  // pin the prefetch to its k-step
            }
            ICREC_STAMP(4, 3 + 2 * ch);
            if (ch + 1 < NC) bar_lds();  // B(ch+2): done with H[ch & 1]; H[(ch + 1) & 1] is ready
        }
        ICREC_STAMP(4, 27);
        // ---- + residual + bias: the block's own planes, still resident (the same r as wt_res_global / wt_res_add_direct)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi) {
                const int feat = q * 96 + i * 32 + fi * 16 + 4 * g;  // 4 consecutive features: half a 16-B chunk
                const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + feat);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti) {
                        const int tok = tt * 32 + ti * 16 + c;
                        const int pos = ffn_x_pos(tok, feat >> 3) + 8 * (g & 1);
                        const half4 a = *reinterpret_cast<const half4*>(Xs + pos);
                        const half4 d = *reinterpret_cast<const half4*>(Xs + FFN2_XPLANE + pos);
#pragma unroll
                        for (int j = 0; j < 4; ++j) Y[i][tt].t[fi][ti][j] = Y[i][tt].t[fi][ti][j] + res_init_val(a[j], d[j], b[j]);
                    }
            }
        // ---- LayerNorm: token block 0 here, token block 1 on the producer wave of the same q (idle otherwise): its
        // accumulators are parked in the dead image (12 tiles of 1 KB per wave, lane order)
        __syncthreads();  // every reader of the LDS is done: image and H buffers are free
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
                    *reinterpret_cast<f32x4*>(Xs + ((q * 12) + (i * 2 + fi) * 2 + ti) * 1024 + lane * 16) = Y[i][1].t[fi][ti];
        __syncthreads();  // token block 1 parked
        wt_ln_block<8, 2, 0>(Y, q, 0, wave, xh, xl, m0, T, gam, bet, eps, Hs, [] { __syncthreads(); }, tid, Wqp != nullptr ? Xs : nullptr);
    }
    ICREC_STAMP(0, 30);
    ICREC_STAMP(4, 30);
    // ---- the NEXT layer's QKV projection of this block (tf:175-177), on the planes just normalised, still on chip: no
    // separate launch, no re-read of x, and the 4.6 KB of fp32 Q / K / V rows per token leave spread over the whole
    // kernel's time instead of in one burst.  Staging tiles: the H buffers (the LayerNorm scratch is done with).
    if (Wqp != nullptr) {  // uniform over the grid
        __syncthreads();  // x2 resident in the image; LayerNorm scratch free
        asm volatile("" : "+v"(c), "+v"(g), "+v"(lo8));  // (addresses and first ring loads derived here, not hoisted)
        qkv_block_walk(Xs, reinterpret_cast<float*>(Hs + wave * QKVR_STG_BYTES), Wqp, bq, qkv_out, Nq, m0, T, wave, lane, c, g, lo8);
    }
    ICREC_STAMP(0, 43);
    ICREC_STAMP(4, 43);
    ICREC_STAMP_RT(0, 63);
}

__global__ __launch_bounds__(512, 2) void qkv_resident_kernel(const _Float16* __restrict__ xh,
                                                              const _Float16* __restrict__ xl, int T,
                                                              const _Float16* __restrict__ Wp,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ out, int N) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    char* const Xs = smem2;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), c = lane & 15, g = lane >> 4;
    float* const stg = reinterpret_cast<float*>(smem2 + FFN2_X_BYTES + wave * QKVR_STG_BYTES);
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    ffn_x_stage<512>(Xs, xh, xl, m0, T);
    __syncthreads();  // X resident
    qkv_block_walk(Xs, stg, Wp, bias, out, N, m0, T, wave, lane, c, g, lane * 8);
}

// W (fp32 [N, K]) -> packed f16 hi/lo fragments (wt_gemm.h), once at encoder creation.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int N, int K,
                                                           _Float16* __restrict__ out) {
    const int KS = K / 32;
    const size_t n = (size_t)(N / 16) * KS * 64;  // one thread per (16-feature tile, k-step, lane)
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (size_t)gridDim.x * 256) {
        const size_t fr = id >> 6;
        const int lane = (int)(id & 63), c = lane & 15, g = lane >> 4;
        const int nt16 = (int)(fr / KS), ks = (int)(fr % KS);
        const float* src = w + (size_t)(nt16 * 16 + c) * K + ks * 32 + 8 * g;
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 a, b;
            split_scaled(src[j], WT_SW, a, b);
            hi[j] = a;
            lo[j] = b;
        }
        _Float16* dst = out + wt_frag_off(nt16 >> 1, ks, KS) + (size_t)(nt16 & 1) * (2 * WT_FRAG) + lane * 8;
        *reinterpret_cast<half8*>(dst) = hi;
        *reinterpret_cast<half8*>(dst + WT_FRAG) = lo;
    }
}

}  // namespace icrec
