// encoder.hip — the SentenceTransformer forward on gfx950: BertModel (6 post-LN layers),
// masked mean pooling, L2 normalisation; token-packed (varlen), fp32 throughout, every GEMM
// and both attention products on v_mfma_f32_32x32x2_f32.
//
// Replaces the device work of SentenceTransformer.encode as called at
//   /root/reference/src/inference/serve_recommendations.py:195-200 (catalog index build)
//   /root/reference/src/inference/serve_recommendations.py:213, :246 (per-request query)
// Arithmetic follows transformers/models/bert/modeling_bert.py (tf:) as cited per kernel,
// and oracle/icrec_oracle.c reduction orders where a kernel says "oracle order".
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"
#include "gemm_x3.h"
#include "wt_gemm.h"

namespace icrec {

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ int find_seq(const int32_t* __restrict__ cu, int n_seqs, int t) {
    int lo = 0, hi = n_seqs;  // largest s with cu[s] <= t
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (cu[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// LayerNorm of one 384-wide row held 6 values per lane (element i = lane + 64*j); oracle order.
// SPLIT additionally writes the row as f16 hi/lo planes for the f16x3 GEMMs (gemm_x3.h).
template <int H, bool SPLIT>
__device__ __forceinline__ void ln_row(float (&v)[H / 64], const float* __restrict__ g, const float* __restrict__ b,
                                       float eps, float* __restrict__ out, _Float16* __restrict__ oh,
                                       _Float16* __restrict__ ol, int lane) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < H / 64; ++j) s = s + v[j];
    const float mean = wave_sum_f32(s) / (float)H;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < H / 64; ++j) {
        float d = v[j] - mean;
        q = fmaf(d, d, q);
    }
    const float var = wave_sum_f32(q) / (float)H;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int j = 0; j < H / 64; ++j) {
        const int i = lane + 64 * j;
        const float y = fmaf((v[j] - mean) * rstd, g[i], b[i]);
        if (SPLIT) {  // f16x3 mode: the residual stream exists only as its two planes
            _Float16 hi, lo;
            split_act(y, hi, lo);
            oh[i] = hi;
            ol[i] = lo;
        } else {
            out[i] = y;
        }
    }
}

// ---------------------------------------------------------------- K1: embeddings + LN (tf:98-107)
template <int H, bool SPLIT>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ ids,
                                                       const int32_t* __restrict__ cu, int n_seqs, int T,
                                                       const float* __restrict__ word, const float* __restrict__ pos,
                                                       const float* __restrict__ type, const float* __restrict__ g,
                                                       const float* __restrict__ b, float eps, int vocab, int max_pos,
                                                       float* __restrict__ x, _Float16* __restrict__ xh,
                                                       _Float16* __restrict__ xl) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int s = find_seq(cu, n_seqs, t);
    int id = ids[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int p = t - cu[s];
    p = p >= max_pos ? max_pos - 1 : p;
    float v[H / 64];
#pragma unroll
    for (int j = 0; j < H / 64; ++j) {
        const int i = lane + 64 * j;
        v[j] = (word[(size_t)id * H + i] + type[i]) + pos[(size_t)p * H + i];
    }
    ln_row<H, SPLIT>(v, g, b, eps, x + (size_t)t * H, xh + (size_t)t * H, xl + (size_t)t * H, lane);
}

// ---------------------------------------------------------------- residual + LN (tf:292, tf:350)
// x <- LN(a + x); `a` already holds dense(.) + bias.
template <int H, bool SPLIT>
__global__ __launch_bounds__(256) void add_ln_kernel(const float* __restrict__ a, float* __restrict__ x, int T,
                                                     const float* __restrict__ g, const float* __restrict__ b,
                                                     float eps, _Float16* __restrict__ xh,
                                                     _Float16* __restrict__ xl) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    float v[H / 64];
#pragma unroll
    for (int j = 0; j < H / 64; ++j) {
        const int i = lane + 64 * j;
        v[j] = a[(size_t)t * H + i] + x[(size_t)t * H + i];
    }
    ln_row<H, SPLIT>(v, g, b, eps, x + (size_t)t * H, xh + (size_t)t * H, xl + (size_t)t * H, lane);
}

// ---------------------------------------------------------------- GEMM: out = A . W^T + bias [, GELU]
// torch.nn.Linear (tf:175-177 QKV, tf:290 attention output, tf:335 intermediate, tf:348 output).
// GELU is the exact erf form (tf:336, ACT2FN["gelu"]).
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf for the f16x3 GELU epilogue: branch-free, one v_exp_f32.  |x| <= 0.921875: x * P6(x^2); else
// sign(x) * (1 - 2^Q8(min(|x|, 4))).  Coefficients: Chebyshev fits (tools/fit_erf.py); checked over 620k points
// with fma emulation: max |error| 1.1e-7, max relative error 2.8 ulp — the same class as the library erff, at
// about half its instruction count (the library version carries a full expf range reduction and two branches).
__device__ __forceinline__ float erf_fast(float x) {
    const float t = fminf(fabsf(x), 4.0f), s = x * x;
    float a = 8.392696624e-05f;
    a = fmaf(a, s, -8.148506071e-04f);
    a = fmaf(a, s, 5.201591808e-03f);
    a = fmaf(a, s, -2.685964751e-02f);
    a = fmaf(a, s, 1.128370053e-01f);
    a = fmaf(a, s, -3.761263411e-01f);
    a = fmaf(a, s, 1.128379167e+00f);
    float b = 2.327214131e-06f;
    b = fmaf(b, t, -6.574532254e-05f);
    b = fmaf(b, t, 8.549435650e-04f);
    b = fmaf(b, t, -6.837534407e-03f);
    b = fmaf(b, t, 3.803287522e-02f);
    b = fmaf(b, t, -1.586590115e-01f);
    b = fmaf(b, t, -9.116990640e-01f);
    b = fmaf(b, t, -1.630481967e+00f);
    b = fmaf(b, t, 4.364755836e-04f);
    const float far = copysignf(1.0f - __builtin_amdgcn_exp2f(b), x);
    return t <= 0.921875f ? x * a : far;
}
__device__ __forceinline__ float gelu_erf_fast(float x) { return x * 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

// erf-GELU of the f16x3 engine, times the activation plane scale (16): one polynomial, one v_exp_f32, 16 VALU.
//   gelu(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2),     erfc(t / sqrt 2) = 2^Q10(min(t, 5.75))
// (x >= 0: x - 0.5 x erfc = 0.5 x (1 + erf);  x < 0: 0.5 x erfc(|z|) = 0.5 x (1 + erf(z)).)  Only ABSOLUTE accuracy of
// erfc matters here - it multiplies |x| and is added to a term of the size of x - so the separate small-|x| branch of
// erf_fast (kept for relative accuracy of erf itself) is not needed.  Q10: Chebyshev fit of log2(erfc(t / sqrt 2)) on
// [0, 5.75] (tools/fit_erf.py --gelu); in emulated fp32 fma arithmetic over 440k points: max |error| 2.4e-7 (half an
// ulp at x = 4.3), relative error <= 1.05e-6 wherever |gelu| > 1e-3.
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

template <class Cfg, bool GELU>
__global__ __launch_bounds__(Cfg::THREADS, 2) void linear_kernel(const float* __restrict__ A, int M, int K,
                                                                 const float* __restrict__ W, int N,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ out, int n_tiles_n) {
    __shared__ __attribute__((aligned(16))) float smem[Cfg::LDS_FLOATS];
    float* As = smem;
    float* Bs = smem + Cfg::BM * LDK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_tiles_n, nt = bid % n_tiles_n;  // tiles sharing an A row panel are neighbours
    const int64_t m0 = (int64_t)mt * Cfg::BM, n0 = (int64_t)nt * Cfg::BN;
    TileRegs<Cfg> pre;
    f32x16 acc[Cfg::TM][Cfg::TN];
    tile_gemm<Cfg>(acc, A, m0, M, W, n0, N, K, As, Bs, pre, false);
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
        const int64_t col = n0 + (wn * Cfg::TN + j) * 32 + (lane & 31);
        const float bv = col < N ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + (wm * Cfg::TM + i) * 32 + acc_row(e, lane);
                if (row < M && col < N) {
                    float v = acc[i][j][e] + bv;
                    if (GELU) v = gelu_erf(v);
                    out[row * N + col] = v;
                }
            }
    }
}

// ---------------------------------------------------------------- residual stream + LayerNorm of the f16x3 engine
// In f16x3 mode the residual stream x lives in HBM ONLY as its two f16 planes (xh, xl: 16 x to 22 significant bits,
// wt_gemm.h) — the planes every GEMM reads anyway.  The two LayerNorm sites of a layer (tf:292 attention output,
// tf:350 FFN output: LN(dense(.) + bias + x)) start their accumulators from the residual instead of adding it at
// the end:
//     acc0 = fmaf(float(xh) + float(xl), 1024, bias * 2^14)      (hi + lo is exact in fp32; one rounding)
//     acc  = acc0 + the k-steps of the GEMM, ascending           (units of 2^-14, like every accumulator of the engine)
//     v    = acc * 2^-14                                         (exact)
// then the engine's LayerNorm order over the 384 features of a token.  Wave q (0..3 of the waves holding
// accumulators) owns features q*96 .. q*96+95: lane (r, h) holds, for token tile tt, the 48 values of token
// tt*32 + r at features q*96 + i*32 + 8g + 4h + j (i < 3, g < 4, j < 4):
//   part(q, h) = sum over (i, g, j), i outermost, of v            sequential fp32 adds from 0
//   sum        = ((P0 + P1) + P2) + P3,   Pq = part(q, 0) + part(q, 1)
//   mean = sum / 384;   d = v - mean;   the same tree over fmaf(d, d, .) chains;   var = that / 384
//   y = fmaf(d * (1 / sqrtf(var + eps)), gamma, beta);   planes = split(16 y)
// (ln_wt_kernel is the unfused form with the same order — same bits.)  Compared with an fp32 copy of x beside the
// planes this drops 8 of the 12 bytes per element every LayerNorm site moved, and the residual needs no global
// read where the planes are already in LDS (fused FFN).
//
// Global memory is touched only by coalesced accesses: a lane-per-token access pattern costs 4x the whole K loop
// (in-kernel stamps: 73k cycles per block).  Each wave transposes its own [32 tokens x 96 features] of one plane
// through a PRIVATE 6.5 KB LDS tile (rows of 192 B + 16 B pad): no workgroup barrier, 192 contiguous bytes per
// token row on the global side.  LDS instructions of one wave execute in order; lds_order() keeps the compiler
// from reordering across the hand-over and waits for the data.
constexpr int LNT_ROW = 208;                        // bytes per tile row: 96 halfs + 16 B
constexpr int LNT_TILE = 32 * LNT_ROW;              // one wave's tile
constexpr int LNT_RED = 2 * 64 * 4 * 4;             // the two 4-partial exchanges: [2][64 tokens][4 waves] floats
constexpr int LNT_PAR = 2 * 384 * 4;                // gamma, beta
constexpr int LNT_BYTES = LNT_RED + 4 * LNT_TILE + LNT_PAR;   // 31,744 B

__device__ __forceinline__ void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float res_init_val(_Float16 hi, _Float16 lo, float b) {
    return fmaf((float)hi + (float)lo, WT_SW, b * (WT_SA * WT_SW));
}

// acc0 of a [96-feature x 64-token] wave tile from the residual planes in global memory.  `tile`: this wave's
// LNT_TILE bytes of LDS.
__device__ __forceinline__ void wt_res_init_global(f32x16 (&acc)[3][2], int q, const float* __restrict__ bias,
                                                   const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                   int64_t m0, int64_t T, char* tile) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    u32x4 v[2][2][6];
    int lpos[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {  // 16-B chunk f of the wave's [32 rows][12 chunks]: 12 lanes cover one row's 192 B
        const int f = lane + 64 * k, row = f / 12, c = f - row * 12;
        lpos[k] = row * LNT_ROW + c * 16;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            int64_t g = m0 + tt * 32 + row;
            g = g < T ? g : T - 1;
            const int64_t off = g * 384 + q * 96 + c * 8;
            v[tt][0][k] = *reinterpret_cast<const u32x4*>(xh + off);
            v[tt][1][k] = *reinterpret_cast<const u32x4*>(xl + off);
        }
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int k = 0; k < 6; ++k) *reinterpret_cast<u32x4*>(tile + lpos[k]) = v[tt][pl][k];
            lds_order();
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int fl = i * 32 + 8 * g + 4 * h;
                    const half4 a = *reinterpret_cast<const half4*>(tile + r * LNT_ROW + fl * 2);
                    if (pl == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][tt][4 * g + j] = (float)a[j];
                    } else {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + q * 96 + fl);
#pragma unroll
                        for (int j = 0; j < 4; ++j)  // float(hi) + float(lo) is exact
                            acc[i][tt][4 * g + j] = fmaf(acc[i][tt][4 * g + j] + (float)a[j], WT_SW, b[j] * (WT_SA * WT_SW));
                    }
                }
            lds_order();
        }
}

// The same for the single-tile waves of the small-batch kernels: 8-byte loads straight from global memory (a
// handful of tokens: latency, not bandwidth).
__device__ __forceinline__ void wt_res_init_direct(f32x16 (&acc)[1][1], int nt0, const float* __restrict__ bias,
                                                   const _Float16* __restrict__ xh, const _Float16* __restrict__ xl,
                                                   int N, int64_t m0, int64_t T) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    int64_t tok = m0 + r;
    tok = tok < T ? tok : T - 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int feat = nt0 * 32 + 8 * g + 4 * h;
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + feat);
        const half4 a = *reinterpret_cast<const half4*>(xh + tok * N + feat);
        const half4 d = *reinterpret_cast<const half4*>(xl + tok * N + feat);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][0][4 * g + j] = res_init_val(a[j], d[j], b[j]);
    }
}

// this thread's 16 bytes of the [gamma | beta] table wt_ln_out keeps in LDS (threads 0 .. 191 of the callers)
__device__ __forceinline__ f32x4 wt_ln_par_load(const float* __restrict__ gam, const float* __restrict__ bet, int ptid) {
    const int t = ptid < 192 ? ptid : 0;
    return *reinterpret_cast<const f32x4*>((t < 96 ? gam : bet - 384) + 4 * t);
}

// LayerNorm of a block's accumulators (acc = residual + bias + dense, in units of 2^-14) -> the two planes of x.
// NT threads call; `active` = this wave holds accumulators (wave-uniform), `sync` = the workgroup barrier.
// lds: LNT_BYTES.  ptid: index of the thread among the callers (0 .. 191 must be present).
template <class Sync>
__device__ __forceinline__ void wt_ln_out(f32x16 (&acc)[3][2], bool active, int q, _Float16* __restrict__ xh,
                                          _Float16* __restrict__ xl, int64_t m0, int64_t T,
                                          const float* __restrict__ gam, const float* __restrict__ bet, float eps,
                                          char* lds, Sync sync, int ptid, int salt = 0,
                                          const f32x4* par_pre = nullptr) {
    // salt: 0, opaque to the compiler when the call sits in a loop - otherwise the per-lane addresses below are
    // hoisted out of the loop and held (or spilled) across it
    float* const red = reinterpret_cast<float*>(lds);
    char* const tile = lds + LNT_RED + q * LNT_TILE;
    float* const par = reinterpret_cast<float*>(lds + LNT_RED + 4 * LNT_TILE);
    const int lane = (threadIdx.x & 63) + salt, r = lane & 31, h = lane >> 5;
    ICREC_STAMP(0, 32); ICREC_STAMP(4, 32);
    // gamma / beta -> LDS (visible after the first barrier): each lane needs the 48 values of its half of the wave's 96
    // features, the same in every lane of the half - as vector loads that is 96 x 1 KB through the texture path per wave
    // for 768 distinct bytes (measured: the normalise + write-out phase was bound by them)
    // par_pre: the caller loaded this thread's 16 bytes earlier (wt_ln_par_load), off the critical path
    if (ptid < 192) *reinterpret_cast<f32x4*>(par + 4 * ptid) = par_pre ? *par_pre : wt_ln_par_load(gam, bet, ptid);
    if (active) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float part = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = acc[i][p][e] * WT_UNSCALE;
                    acc[i][p][e] = v;
                    part = part + v;
                }
            part = part + __shfl_xor(part, 32, 64);  // Pq (a + b == b + a exactly: both halves hold the same bits)
            if (h == 0) red[(p * 32 + r) * 4 + q] = part;
        }
    }
    ICREC_STAMP(0, 33); ICREC_STAMP(4, 33);
    sync();
    if (active) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(red + (p * 32 + r) * 4);
            const float mean = (((s4[0] + s4[1]) + s4[2]) + s4[3]) / 384.0f;
            float sq = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float d = acc[i][p][e] - mean;
                    acc[i][p][e] = d;
                    sq = fmaf(d, d, sq);
                }
            sq = sq + __shfl_xor(sq, 32, 64);
            if (h == 0) red[256 + (p * 32 + r) * 4 + q] = sq;
        }
    }
    ICREC_STAMP(0, 34); ICREC_STAMP(4, 34);
    sync();
    ICREC_STAMP(0, 35); ICREC_STAMP(4, 35);
    if (active) {
        // chunk k of this lane in the flat [32 rows][12 x 16 B] view of a tile: row = f / 12, c = f % 12, f = lane + 64 k
        auto flat = [&](int k, int& row, int& c) {
            const int f = lane + 64 * k;
            row = (f * 43691) >> 19;  // f / 12 for f < 384
            c = f - row * 12;
        };
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(red + 256 + (p * 32 + r) * 4);
            const float var = (((s4[0] + s4[1]) + s4[2]) + s4[3]) / 384.0f;
            const float rstd = 1.0f / sqrtf(var + eps);
            const int64_t t0 = m0 + p * 32;
            half4 lo[3][4];
            // the hi plane goes straight into the tile; lo waits in registers for its turn
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int feat = q * 96 + i * 32 + 8 * g + 4 * h;
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(par + feat);
                    const f32x4 bt = *reinterpret_cast<const f32x4*>(par + 384 + feat);
                    f32x4 y;
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = fmaf(acc[i][p][4 * g + j] * rstd, gm[j], bt[j]);
                    half4 hi;
                    split_act4(y, hi, lo[i][g]);
                    *reinterpret_cast<half4*>(tile + r * LNT_ROW + (i * 32 + 8 * g + 4 * h) * 2) = hi;
                }
            ICREC_STAMP(0, 37 + 3 * p); ICREC_STAMP(4, 37 + 3 * p);
            lds_order();
            u32x4 o[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                int row, c;
                flat(k, row, c);
                o[k] = *reinterpret_cast<const u32x4*>(tile + row * LNT_ROW + c * 16);
            }
            lds_order();
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<half4*>(tile + r * LNT_ROW + (i * 32 + 8 * g + 4 * h) * 2) = lo[i][g];
#pragma unroll
            for (int k = 0; k < 6; ++k) {  // the hi rows leave while the lo tile is written
                int row, c;
                flat(k, row, c);
                if (t0 + row < T) *reinterpret_cast<u32x4*>(xh + (t0 + row) * 384 + q * 96 + c * 8) = o[k];
            }
            lds_order();
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                int row, c;
                flat(k, row, c);
                o[k] = *reinterpret_cast<const u32x4*>(tile + row * LNT_ROW + c * 16);
            }
            lds_order();
            ICREC_STAMP(0, 38 + 3 * p); ICREC_STAMP(4, 38 + 3 * p);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                int row, c;
                flat(k, row, c);
                if (t0 + row < T) *reinterpret_cast<u32x4*>(xl + (t0 + row) * 384 + q * 96 + c * 8) = o[k];
            }
            ICREC_STAMP(0, 39 + 3 * p); ICREC_STAMP(4, 39 + 3 * p);
        }
    }
    ICREC_STAMP(0, 36); ICREC_STAMP(4, 36);
}

// The unfused form of the same LayerNorm (small batches; ICREC_FUSE=0): planes(x) <- LN(a), `a` = dense(.) + bias +
// residual as the EPI 2 GEMM wrote it.  8 threads per token: thread (q, h) sums its 48 values in (i, g, j) order,
// the 8 partials are combined by shuffles in the fixed tree of wt_ln_out.
__global__ __launch_bounds__(256) void ln_wt_kernel(const float* __restrict__ a, int T, const float* __restrict__ gam,
                                                    const float* __restrict__ bet, float eps,
                                                    _Float16* __restrict__ xh, _Float16* __restrict__ xl) {
    const int tid = threadIdx.x, slot = tid & 7, q = slot >> 1, h = slot & 1;
    int64_t tok = (int64_t)blockIdx.x * 32 + (tid >> 3);
    const bool ok = tok < T;
    tok = ok ? tok : (int64_t)T - 1;
    f32x4 v[3][4];
    float part = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v[i][g] = *reinterpret_cast<const f32x4*>(a + tok * 384 + q * 96 + i * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) part = part + v[i][g][j];
        }
    const int base = (tid & 63) & ~7;  // first lane of this token's 8 threads
    auto tree = [&](float p) {  // ((P0 + P1) + P2) + P3 with Pq = part(q,0) + part(q,1); every lane gets the same bits
        p = p + __shfl_xor(p, 1, 64);
        const float p0 = __shfl(p, base, 64), p1 = __shfl(p, base + 2, 64), p2 = __shfl(p, base + 4, 64),
                    p3 = __shfl(p, base + 6, 64);
        return ((p0 + p1) + p2) + p3;
    };
    const float mean = tree(part) / 384.0f;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][g][j] - mean;
                v[i][g][j] = d;
                sq = fmaf(d, d, sq);
            }
    const float var = tree(sq) / 384.0f;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int feat = q * 96 + i * 32 + 8 * g + 4 * h;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + feat);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(bet + feat);
            f32x4 y;
            half4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fmaf(v[i][g][j] * rstd, gm[j], bt[j]);
            split_act4(y, hi, lo);
            if (ok) {
                *reinterpret_cast<half4*>(xh + tok * 384 + feat) = hi;
                *reinterpret_cast<half4*>(xl + tok * 384 + feat) = lo;
            }
        }
}

constexpr int LN_LD = 388;  // floats per staged output row in LDS (+16 B: the 16-B accesses of consecutive tokens hit distinct banks)

// ---------------------------------------------------------------- f16x3 linear layers (wt_gemm.h)
// out^T = W . X^T with the weights streamed straight from L2 into registers (packed fragment order) and the
// token slab staged through LDS.  Block = 4 waves; wave q owns NTW 32-feature tiles x TTW 32-token tiles.
//   EPI 0: out fp32 [T, N] = acc * 2^-14 + bias           (QKV)
//   EPI 1: erf-GELU (tf:336), result as f16 hi/lo planes   (FFN-up of small batches)
//   EPI 2: accumulators start from residual + bias (planes rh / rl, row stride N), out fp32 = acc * 2^-14: the
//          LayerNorm input of attention-out / FFN-down for small batches (ln_wt_kernel follows)
// Each lane holds 4 consecutive features of one token per register group: 16-B (fp32) / 8-B (planes) stores.
template <int NTW, int TTW, int D, int EPI>
__global__ __launch_bounds__(256, 2) void wt_linear_kernel(const _Float16* __restrict__ Xh,
                                                           const _Float16* __restrict__ Xl, int T, int K,
                                                           const _Float16* __restrict__ Wp, int N,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           _Float16* __restrict__ oh, _Float16* __restrict__ ol,
                                                           int n_blocks_n) {
    constexpr bool STAGED = (NTW == 3 && TTW == 2 && EPI != 1);  // batch form: results leave through an LDS stage, coalesced
    constexpr int SM = (STAGED && 32 * LN_LD * 4 > XRing<TTW>::BYTES) ? 32 * LN_LD * 4 : XRing<TTW>::BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SM];
    const int lane = threadIdx.x & 63, q = wave_uniform(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
    // the n_blocks_n workgroups that read the same token rows get consecutive logical ids = the same XCD = one L2
    // (PMC: without the remap the QKV launch fetched its activations three times, 930 MB instead of ~330 MB)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_blocks_n, nb = bid % n_blocks_n;
    const int64_t m0 = (int64_t)mt * (32 * TTW);
    const int nt0 = (nb * 4 + q) * NTW;
    f32x16 acc[NTW][TTW];
    if constexpr (EPI == 2) {  // oh / ol carry the residual planes here
        if constexpr (NTW == 3 && TTW == 2) {
            wt_res_init_global(acc, q, bias + nb * 384, oh + nb * 384, ol + nb * 384, m0, T, smem + q * LNT_TILE);
            __syncthreads();
        } else {
            static_assert(NTW == 1 && TTW == 1, "residual init: 3 x 2 or 1 x 1 wave tiles");
            wt_res_init_direct(acc, nt0, bias, oh, ol, N, m0, T);
        }
    }
    wt_kloop<NTW, TTW, D, EPI != 2>(acc, Wp, nt0, K, Xh, Xl, m0, T, smem);
    if constexpr (STAGED) {
        // [384 features x 32 tokens] per pass -> stage[token][feature] (16-B LDS writes), then 16-B chunks in flat
        // order: every wave store instruction writes 1 KB of at most two output rows.  (Storing 16 B per lane with
        // the lanes 4,608 B apart took as long as the whole K loop: 32k of 69k cycles per block by in-kernel stamps.)
        float* const stage = reinterpret_cast<float*>(smem);
        const int n0 = nb * 384;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int fl = q * 96 + i * 32 + 8 * g + 4 * h;
                    f32x4 b = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    if (EPI == 0) b = *reinterpret_cast<const f32x4*>(bias + n0 + fl);
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = EPI == 2 ? acc[i][tt][4 * g + j] * WT_UNSCALE : fmaf(acc[i][tt][4 * g + j], WT_UNSCALE, b[j]);
                    *reinterpret_cast<f32x4*>(stage + r * LN_LD + fl) = v;
                }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int f = threadIdx.x + 256 * k, row = f / 96, c = f - row * 96;
                const int64_t tok = m0 + tt * 32 + row;
                if (tok < T)
                    *reinterpret_cast<f32x4*>(out + tok * N + n0 + c * 4) = *reinterpret_cast<const f32x4*>(stage + row * LN_LD + c * 4);
            }
            if (tt == 0) __syncthreads();
        }
        ICREC_STAMP(0, 30);
        return;
    }

#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int feat = (nt0 + i) * 32 + 8 * g + 4 * h;
            f32x4 b = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (EPI != 2) b = *reinterpret_cast<const f32x4*>(bias + feat);
#pragma unroll
            for (int tt = 0; tt < TTW; ++tt) {
                const int64_t tok = m0 + tt * 32 + r;
                if (tok < T) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = EPI == 2 ? acc[i][tt][4 * g + j] * WT_UNSCALE : fmaf(acc[i][tt][4 * g + j], WT_UNSCALE, b[j]);
                    if (EPI == 1) {
                        half4 hi, lo;
                        half2w a, b, c, d;
                        split_pair_prescaled(gelu16_wt(v[0]), gelu16_wt(v[1]), a, b);
                        split_pair_prescaled(gelu16_wt(v[2]), gelu16_wt(v[3]), c, d);
                        hi = half4{a[0], a[1], c[0], c[1]};
                        lo = half4{b[0], b[1], d[0], d[1]};
                        *reinterpret_cast<half4*>(oh + tok * N + feat) = hi;
                        *reinterpret_cast<half4*>(ol + tok * N + feat) = lo;
                    } else {
                        *reinterpret_cast<f32x4*>(out + tok * N + feat) = v;
                    }
                }
            }
        }
    ICREC_STAMP(0, 30);
}

// Attention-output projection + residual + LayerNorm in one kernel (large batches): block = 64 tokens x all 384
// features, K = 384.
// VAR (tools/ffn_bench.hip only; the product uses 0): 1 = accumulators start from zero (no residual rows in),
// 2 = no LayerNorm / write-out (one dummy store per thread keeps the K loop alive)
template <int D, int VAR = 0>
__global__ __launch_bounds__(256, 2) void wt_linear_ln_kernel(const _Float16* __restrict__ Ah,
                                                              const _Float16* __restrict__ Al, int T, int K,
                                                              const _Float16* __restrict__ Wp,
                                                              const float* __restrict__ bias,
                                                              _Float16* __restrict__ xh, _Float16* __restrict__ xl,
                                                              const float* __restrict__ gam,
                                                              const float* __restrict__ bet, float eps) {
    static_assert(XRing<2>::BYTES >= LNT_BYTES, "the slab ring doubles as the LayerNorm scratch");
    __shared__ __attribute__((aligned(16))) char smem[XRing<2>::BYTES];
    const int q = wave_uniform(threadIdx.x >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    f32x16 acc[3][2];
    if (!(VAR & 1)) {
        wt_res_init_global(acc, q, bias, xh, xl, m0, T, smem + LNT_RED + q * LNT_TILE);
        __syncthreads();  // the private tiles become the slab ring
    }
    wt_kloop<3, 2, D, (VAR & 1) != 0>(acc, Wp, q * 3, K, Ah, Al, m0, T, smem);  // ends with a barrier: the slab ring is free
    if (VAR & 2) {
        float sdum = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 16; ++e) sdum += acc[i][tt][e];
        if (sdum == 123.456f) xh[m0 * 384 + threadIdx.x] = (_Float16)sdum;
        return;
    }
    wt_ln_out(acc, true, q, xh, xl, m0, T, gam, bet, eps, smem, [] { __syncthreads(); }, threadIdx.x);
    ICREC_STAMP(0, 30);
}

// ---------------------------------------------------------------- fused FFN (large batches)
// x <- LN(W2 . gelu(W1 . x + b1) + b2 + x)   (tf:334-351: BertIntermediate, BertOutput) for a block of 64 tokens,
// without the [T, 1536] intermediate ever leaving the CU.  The 1,536 intermediate features are walked in 12 chunks
// of 128; per chunk
//   P1  S^T[128 x 64 tok] = W1[chunk] . X^T          K = 384; wave q: intermediates q*32..+31 (2 token tiles)
//   G   H = split(gelu(S * 2^-14 + b1))              registers -> 8-byte LDS writes (4 consecutive k of a token)
//   P2  Y^T[384 x 64 tok] += W2[:, chunk] . H^T      K = 128; wave q: features q*96..+95 (3 x 2 tiles, 96 regs)
// then the residual + LayerNorm epilogue.  Per output the MFMA chain is exactly wt_kloop's (k-steps ascending,
// the same three products per step), so the result equals FFN-up -> FFN-down -> add_ln through wt_linear_kernel
// bit for bit.
constexpr int FFN_IC = 128;

// Producer / consumer form: one 8-wave workgroup per CU owning ALL 160 KB of LDS:
//   * the block's 64 x 384 activation planes stay resident in LDS (96 KB) for all 12 chunks — no restream, no
//     slab barriers;
//   * waves 0-3 (producers) run P1 + GELU for chunk c+1 and write H into one half of a double buffer (2 x 32 KB)
//     while waves 4-7 (consumers) run P2 of chunk c from the other half: ONE workgroup barrier per chunk;
//   * each SIMD hosts one producer and one consumer.  They issue the same number of MFMAs per chunk (144 each);
//     the producer runs at raised priority, so it finishes P1 early and its GELU (VALU) overlaps the consumer's
//     MFMAs;
//   * specialisation frees registers for deep weight prefetch rings (W1: 8 k-steps, W2: 4 k-steps ahead).
constexpr int FFN2_XPLANE = 64 * 768;                    // [64 tokens][384 k] halfs: 768-B rows, 3 sub-rows of 256 B
constexpr int FFN2_X_BYTES = 2 * FFN2_XPLANE;            // hi, lo
constexpr int FFN2_HPLANE = 64 * 256;                    // [64 tokens][128 k] halfs
constexpr int FFN2_HBUF = 2 * FFN2_HPLANE;               // hi, lo
constexpr int FFN2_LDS = FFN2_X_BYTES + 2 * FFN2_HBUF;   // 163,840 B = the whole LDS of a CU
static_assert(FFN2_LDS == 160 * 1024, "fused FFN LDS budget");

__device__ __forceinline__ void bar_lds() {  // LDS hand-off barrier that leaves global loads in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int VAR>
__global__ __launch_bounds__(512, 2) void ffn_fused2_kernel(_Float16* __restrict__ xh, _Float16* __restrict__ xl,
                                                            int T, int I,
                                                            const _Float16* __restrict__ W1p,
                                                            const float* __restrict__ b1,
                                                            const _Float16* __restrict__ W2p,
                                                            const float* __restrict__ b2,
                                                            const float* __restrict__ gam,
                                                            const float* __restrict__ bet, float eps) {
    constexpr int KS1 = 24;
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    char* const Xs = smem2;
    char* const Hs = smem2 + FFN2_X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), q = wave & 3, r = lane & 31, h = lane >> 5;
    const bool producer = wave < 4;
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int NC = I / FFN_IC, KS2 = I / 16;

    ICREC_STAMP(0, 0);
    ICREC_STAMP(4, 0);
    // ---- the block's activation planes -> LDS, once (16-B chunk c of token row t at sub-row c >> 4, slot (c ^ t) & 15)
    {
        u32x4 vh[6], vl[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 512 * i, row = id / 48, c = id - row * 48;
            int64_t g = m0 + row;
            g = g < T ? g : (int64_t)T - 1;
            vh[i] = *reinterpret_cast<const u32x4*>(xh + g * 384 + c * 8);
            vl[i] = *reinterpret_cast<const u32x4*>(xl + g * 384 + c * 8);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 512 * i, row = id / 48, c = id - row * 48;
            const int pos = row * 768 + (((c & ~15) | ((c ^ row) & 15)) << 4);
            *reinterpret_cast<u32x4*>(Xs + pos) = vh[i];
            *reinterpret_cast<u32x4*>(Xs + FFN2_XPLANE + pos) = vl[i];
        }
    }

    const unsigned lo8 = lane * 8;
    f32x16 Y[3][2];  // consumers only (dead in the producer branch)
    if (producer) {
        // LDS byte address of this lane's fragment of token tile tt at 16-B chunk ch = 2 ks + h:
        //   tok*768 + (ch >> 4)*256 + (((ch & 15) ^ (tok & 15)) << 4),  (ch & 15) ^ t = 2 (ks & 7) ^ (h ^ t)
        int xb[8][2];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int tok = tt * 32 + r;
                xb[m][tt] = tok * 768 + (((2 * m) ^ h ^ (tok & 15)) << 4);
            }
        half8 wh[8][1], wl[8][1];
        {
            const _Float16* const wp0[1] = {W1p + wt_frag_off(q, 0, KS1)};
#pragma unroll
            for (int d = 0; d < 8; ++d) w_load<1>(wh[d], wl[d], wp0, d, lo8);
        }
        __syncthreads();  // X resident
        ICREC_STAMP(0, 1);
        // Software pipeline: iteration c runs P1(c) with the GELU of chunk c-1 spread over its k-steps (one group
        // of 4 intermediates x 1 token tile every third k-step), so the producer's VALU work sits between its own
        // MFMAs and the consumers' instead of behind them.  H[c-1] is handed over at the end of iteration c.
        f32x16 S[1][2], Sp[1][2];  // this chunk's accumulators, the previous chunk's (being GELU'd)
        f32x4 bias[4], biasp[4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e = 0; e < 16; ++e) Sp[0][tt][e] = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) biasp[g] = bias[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // one pipeline iteration: P1(c) (MMA = true) with G(c - 1) spread over its 24 k-steps; the drain iteration
        // (MMA = false) runs only the G slices
        auto iteration = [&](int c, auto mma_tag) {
            constexpr bool MMA = decltype(mma_tag)::value;
            const _Float16* const wp1[1] = {W1p + wt_frag_off((MMA ? c : 0) * 4 + q, 0, KS1)};
            const _Float16* const wpn[1] = {W1p + wt_frag_off((MMA && c + 1 < NC ? c + 1 : 0) * 4 + q, 0, KS1)};
            if (MMA) {  // this chunk's biases, loaded BEFORE the k-loop: a load issued behind the weight ring would make
                        // its consumer wait for the whole ring (vmcnt counts in order)
                const float* bp = b1 + c * FFN_IC + q * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const f32x4*>(bp + 8 * g);
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 16; ++e) S[0][tt][e] = 0.0f;
            char* const Hb = Hs + ((c + 1) & 1) * FFN2_HBUF;  // H[(c - 1) & 1]
            half8 fh[2][2], fl[2][2];  // the next k-step's fragments are read under the current one's MFMAs
            half4 ghi, glo;            // the GELU group being assembled
            if (MMA) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    fh[0][tt] = *reinterpret_cast<const half8*>(Xs + xb[0][tt]);
                    fl[0][tt] = *reinterpret_cast<const half8*>(Xs + FFN2_XPLANE + xb[0][tt]);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                if (MMA) {
                    if (ks + 1 < KS1 && !(VAR & 8)) {
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            const int pos = xb[(ks + 1) & 7][tt] + ((ks + 1) >> 3) * 256;
                            fh[(ks + 1) & 1][tt] = *reinterpret_cast<const half8*>(Xs + pos);
                            fl[(ks + 1) & 1][tt] = *reinterpret_cast<const half8*>(Xs + FFN2_XPLANE + pos);
                        }
                    }
                    wt_mma<1, 2, (VAR & 64) != 0>(S, wh[ks & 7], wl[ks & 7], fh[(VAR & 8) ? 0 : (ks & 1)], fl[(VAR & 8) ? 0 : (ks & 1)]);
                    if (!(VAR & 4) && !((VAR & 128) && (ks & 1))) {  // straight-line refill: this chunk's k-step ks+8, or the next chunk's ks+8-24
                                                                    // (VAR 128, timing only: every second refill skipped = half the weight stream)
                        if (ks + 8 < KS1) w_load<1>(wh[ks & 7], wl[ks & 7], wp1, ks + 8, lo8);
                        else w_load<1>(wh[ks & 7], wl[ks & 7], wpn, ks + 8 - KS1, lo8);
                    }
                }
                if (ks % 3 != 2) {  // G of the PREVIOUS chunk, two elements per k-step (16 of the 24 k-steps carry a slice):
                                    // bias + erf-GELU + split; a finished group of 4 consecutive k goes out as one
                                    // 8-byte LDS write per plane
                    constexpr int dummy_ = 0;
                    (void)dummy_;
                    const int u = ks - ks / 3;
                    {
                        const int n = 2 * u, gi = n >> 2, j = n & 3, g = gi >> 1, tt = gi & 1;  // elements j, j + 1 of group gi
                        const float p0 = fmaf(Sp[0][tt][4 * g + j], WT_UNSCALE, biasp[g][j]);
                        const float p1 = fmaf(Sp[0][tt][4 * g + j + 1], WT_UNSCALE, biasp[g][j + 1]);
                        half2w a, d;
                        split_pair_prescaled((VAR & 1) ? p0 : gelu16_wt(p0), (VAR & 1) ? p1 : gelu16_wt(p1), a, d);
                        ghi[j] = a[0];
                        ghi[j + 1] = a[1];
                        glo[j] = d[0];
                        glo[j + 1] = d[1];
                    }
                    if (u & 1) {
                        const int gi = u >> 1, g = gi >> 1, tt = gi & 1;
                        const int tok = tt * 32 + r;
                        const int pos = tok * 256 + (((4 * q + g) ^ (tok & 15)) << 4) + 8 * h;
                        *reinterpret_cast<half4*>(Hb + pos) = ghi;  // iteration 0 writes GELU(0) into a buffer nobody reads yet
                        *reinterpret_cast<half4*>(Hb + FFN2_HPLANE + pos) = glo;
                    }
                    if (MMA && !(VAR & 32)) {  // the next step's LDS reads first; then one MFMA and a dozen of the slice's VALU
                                               // instructions, six times
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                        for (int m = 0; m < 6; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // pin the prefetch (and the GELU slice) to its k-step
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) Sp[0][tt] = S[0][tt];
#pragma unroll
            for (int g = 0; g < 4; ++g) biasp[g] = bias[g];
        };
        for (int c = 0; c < NC; ++c) {
            iteration(c, std::true_type{});
            ICREC_STAMP(0, 2 + 2 * c);
            if (c > 0) bar_lds();  // B(c): H[c - 1] is complete; the consumers have left H[c & 1]
            ICREC_STAMP(0, 3 + 2 * c);
        }
        iteration(NC, std::false_type{});
        bar_lds();  // B(NC): H[NC - 1]
        ICREC_STAMP(0, 26);
    } else {
        const _Float16* w2p[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) w2p[i] = W2p + wt_frag_off(q * 3 + i, 0, KS2);
        int hb[2];  // tok*256 + ((h ^ (tok & 15)) << 4); chunk 2 k2 + h lands at hb ^ (k2 << 5)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) hb[tt] = (tt * 32 + r) * 256 + ((h ^ (r & 15)) << 4);
        half8 wh[4][3], wl[4][3];
#pragma unroll
        for (int d = 0; d < 4; ++d) w_load<3>(wh[d], wl[d], w2p, d, lo8);
        __syncthreads();  // X resident (matches the producers' first barrier)
        ICREC_STAMP(4, 1);
        // Y starts from the residual + bias: the block's own planes, already in LDS (wt_res_init_*: same value)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = q * 12 + i * 4 + g;  // 16-B chunk of the token row holding features q*96 + i*32 + 8g .. +7
                const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + q * 96 + i * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int tok = tt * 32 + r;
                    const int pos = tok * 768 + (((c & ~15) | ((c ^ tok) & 15)) << 4) + 8 * h;
                    const half4 a = *reinterpret_cast<const half4*>(Xs + pos);
                    const half4 d = *reinterpret_cast<const half4*>(Xs + FFN2_XPLANE + pos);
#pragma unroll
                    for (int j = 0; j < 4; ++j) Y[i][tt][4 * g + j] = res_init_val(a[j], d[j], b[j]);
                }
            }
        bar_lds();        // B1: H[0] is ready
        for (int c = 0; c < NC; ++c) {
            ICREC_STAMP(4, 2 + 2 * c);
            const char* const Hb = Hs + (c & 1) * FFN2_HBUF;
            half8 fh[2][2], fl[2][2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                fh[0][tt] = *reinterpret_cast<const half8*>(Hb + hb[tt]);
                fl[0][tt] = *reinterpret_cast<const half8*>(Hb + FFN2_HPLANE + hb[tt]);
            }
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                if (k2 + 1 < 8 && !(VAR & 8)) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int pos = hb[tt] ^ ((k2 + 1) << 5);
                        fh[(k2 + 1) & 1][tt] = *reinterpret_cast<const half8*>(Hb + pos);
                        fl[(k2 + 1) & 1][tt] = *reinterpret_cast<const half8*>(Hb + FFN2_HPLANE + pos);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // issue the next step's LDS reads before this step's 18 MFMAs
                }
                wt_mma<3, 2, (VAR & 64) != 0>(Y, wh[k2 & 3], wl[k2 & 3], fh[(VAR & 8) ? 0 : (k2 & 1)], fl[(VAR & 8) ? 0 : (k2 & 1)]);
                if (!(VAR & 4) && !((VAR & 128) && (k2 & 1))) {
                    int nk = c * 8 + k2 + 4;
                    nk = nk < KS2 ? nk : KS2 - 1;  // past the end: re-read the last fragment (never consumed)
                    w_load<3>(wh[k2 & 3], wl[k2 & 3], w2p, nk, lo8);
                }
                __builtin_amdgcn_sched_barrier(0);  // pin the prefetch to its k-step
            }
            ICREC_STAMP(4, 3 + 2 * c);
            if (c + 1 < NC) bar_lds();  // B(c+2): done with H[c & 1]; H[(c + 1) & 1] is ready
        }
    }
    ICREC_STAMP(0, 27);
    ICREC_STAMP(4, 27);
    // ---- LayerNorm on the consumers' accumulators (the producers only join the two barriers)
    __syncthreads();  // every reader of the LDS is done: it becomes the LayerNorm scratch
    wt_ln_out(Y, !producer, q, xh, xl, m0, T, gam, bet, eps, smem2, [] { __syncthreads(); }, threadIdx.x);
    ICREC_STAMP(0, 30);
    ICREC_STAMP(4, 30);
}

// Persistent form of the same computation: one workgroup per CU walks blocks j = blockIdx.x, + gridDim.x, ... and the
// block boundary is software-pipelined away.  Time is cut into epochs (one workgroup barrier each); in epoch (j, k)
//   producers   k = 0: P1(0)   k = 1..11: P1(k) + G(k-1)   k = 12 ("drain"): G(11), and block j+1's planes -> LDS
//   consumers   k = 0: P2(11) of block j-1   k = 1: LayerNorm + write-out of block j-1, then Y0 of block j
//               k = 2..12: P2(k-2)
// so the LayerNorm epilogue of a block runs in the slot where the consumers would wait for the first H chunk of the
// next one, under the producers' MFMAs; its scratch is the H buffer that is idle in that epoch (H[1]: P2(11) has
// consumed it, G(1) is written an epoch later), its two internal barriers are matched by two extra barriers inside
// the producers' iteration 1.  The next block's activation planes travel global -> registers -> LDS inside the
// drain epoch (the producers' MFMA-free epoch: weight ring, S and the fragment registers are idle), into the X
// region that P1(11) has just left.  Arithmetic and order per output are those of ffn_fused2_kernel.
__global__ __launch_bounds__(512, 2) void ffn_fused3_kernel(_Float16* __restrict__ xh, _Float16* __restrict__ xl,
                                                            int T, int I, const _Float16* __restrict__ W1p,
                                                            const float* __restrict__ b1,
                                                            const _Float16* __restrict__ W2p,
                                                            const float* __restrict__ b2,
                                                            const float* __restrict__ gam,
                                                            const float* __restrict__ bet, float eps) {
    constexpr int KS1 = 24;
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    char* const Xs = smem3;
    char* const Hs = smem3 + FFN2_X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), q = wave & 3, r = lane & 31, h = lane >> 5;
    const bool producer = wave < 4;
    const int G = gridDim.x, nblk = (T + 63) / 64;
    const int J = (nblk - (int)blockIdx.x + G - 1) / G;  // blocks of this workgroup (host: gridDim.x <= nblk)
    const int NC = I / FFN_IC, KS2 = I / 16;
    auto block_m0 = [&](int j) { return (int64_t)(j * G + (int)blockIdx.x) * 64; };

    // ---- stagger: the workgroups of a launch would otherwise walk their blocks in lockstep, and every block boundary
    // (64 x 768 B of planes in, the same out) would hit HBM from all CUs in the same few microseconds.  Eight phases,
    // ~1.5k cycles apart (about one epoch in total: the weight chunks the CUs of an XCD share stay the same ones).
    for (int ph = ((int)blockIdx.x >> 3) & 7; ph > 0; --ph) __builtin_amdgcn_s_sleep(23);
    // ---- the first block's activation planes -> LDS (16-B chunk c of token row t at sub-row c >> 4, slot (c ^ t) & 15)
    {
        const int64_t m0 = block_m0(0);
        u32x4 vh[6], vl[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 512 * i, row = id / 48, c = id - row * 48;
            int64_t g = m0 + row;
            g = g < T ? g : (int64_t)T - 1;
            vh[i] = *reinterpret_cast<const u32x4*>(xh + g * 384 + c * 8);
            vl[i] = *reinterpret_cast<const u32x4*>(xl + g * 384 + c * 8);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 512 * i, row = id / 48, c = id - row * 48;
            const int pos = row * 768 + (((c & ~15) | ((c ^ row) & 15)) << 4);
            *reinterpret_cast<u32x4*>(Xs + pos) = vh[i];
            *reinterpret_cast<u32x4*>(Xs + FFN2_XPLANE + pos) = vl[i];
        }
    }

    const unsigned lo8 = lane * 8;
    if (producer) {
        int xb[8][2];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int tok = tt * 32 + r;
                xb[m][tt] = tok * 768 + (((2 * m) ^ h ^ (tok & 15)) << 4);
            }
        half8 wh[8][1], wl[8][1];
        {
            const _Float16* const wp0[1] = {W1p + wt_frag_off(q, 0, KS1)};
#pragma unroll
            for (int d = 0; d < 8; ++d) w_load<1>(wh[d], wl[d], wp0, d, lo8);
        }
        __syncthreads();  // X(0) resident
        f32x16 S[1][2], Sp[1][2];
        f32x4 bias[4], biasp[4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e = 0; e < 16; ++e) Sp[0][tt][e] = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) biasp[g] = bias[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // one epoch of the producers: P1(c) if MMA, the GELU of the previous chunk spread over the k-steps if GEL;
        // sync2: join the two internal barriers of the consumers' LayerNorm; the drain (!MMA) also moves the next
        // block's planes (rows m0n ..) into LDS when there is one
        // W1j / b1j: the weight and bias pointers, re-defined (opaquely) once per block - the same addresses are read
        // for every block, and as loop-invariant loads the compiler would hoist a whole iteration's weight stream
        // out of the block loop and spill it
        const _Float16* W1j = W1p;
        const float* b1j = b1;
        auto iteration = [&](int c, auto mma_tag, auto gel_tag, bool sync2, int64_t m0n, bool has_next) {
            constexpr bool MMA = decltype(mma_tag)::value, GEL = decltype(gel_tag)::value;
            const _Float16* const wp1[1] = {W1j + wt_frag_off((MMA ? c : 0) * 4 + q, 0, KS1)};
            const _Float16* const wpn[1] = {W1j + wt_frag_off((MMA && c + 1 < NC ? c + 1 : 0) * 4 + q, 0, KS1)};
            if (MMA) {  // biases before the k-loop: a load issued behind the weight ring would wait for the whole ring
                const float* bp = b1j + c * FFN_IC + q * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const f32x4*>(bp + 8 * g);
            }
            // drain: the next block's planes in four quarters of 6 chunks per producer thread (hi rows 0-31, hi rows 32-63,
            // lo likewise), each requested 6 k-steps before it is written to LDS
            u32x4 xv[6];
            int tids = tid;  // opaque copy: keeps the drain's address arithmetic inside the drain (not hoisted across the block loop)
            if (!MMA) asm volatile("" : "+v"(tids));
            auto x_next_load = [&](int quarter) {
                const _Float16* plane = quarter < 2 ? xh : xl;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int id = tids + 256 * (i + 6 * (quarter & 1)), row = id / 48, cc = id - row * 48;
                    int64_t g = m0n + row;
                    g = g < T ? g : (int64_t)T - 1;
                    xv[i] = *reinterpret_cast<const u32x4*>(plane + g * 384 + cc * 8);
                }
            };
            auto x_next_store = [&](int quarter) {
                char* dst = Xs + (quarter < 2 ? 0 : FFN2_XPLANE);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int id = tids + 256 * (i + 6 * (quarter & 1)), row = id / 48, cc = id - row * 48;
                    *reinterpret_cast<u32x4*>(dst + row * 768 + (((cc & ~15) | ((cc ^ row) & 15)) << 4)) = xv[i];
                }
            };
            if (!MMA && has_next) x_next_load(0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 16; ++e) S[0][tt][e] = 0.0f;
            char* const Hb = Hs + ((c + 1) & 1) * FFN2_HBUF;  // H[(c - 1) & 1]
            half8 fh[2][2], fl[2][2];
            half4 ghi, glo;
            if (MMA) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    fh[0][tt] = *reinterpret_cast<const half8*>(Xs + xb[0][tt]);
                    fl[0][tt] = *reinterpret_cast<const half8*>(Xs + FFN2_XPLANE + xb[0][tt]);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                if (MMA && GEL && (ks == 8 || ks == 16)) {
                    if (sync2) bar_lds();
                }
                if (MMA) {
                    if (ks + 1 < KS1) {
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            const int pos = xb[(ks + 1) & 7][tt] + ((ks + 1) >> 3) * 256;
                            fh[(ks + 1) & 1][tt] = *reinterpret_cast<const half8*>(Xs + pos);
                            fl[(ks + 1) & 1][tt] = *reinterpret_cast<const half8*>(Xs + FFN2_XPLANE + pos);
                        }
                    }
                    wt_mma<1, 2>(S, wh[ks & 7], wl[ks & 7], fh[ks & 1], fl[ks & 1]);
                    if (ks + 8 < KS1) w_load<1>(wh[ks & 7], wl[ks & 7], wp1, ks + 8, lo8);
                    else w_load<1>(wh[ks & 7], wl[ks & 7], wpn, ks + 8 - KS1, lo8);
                }
                if (GEL && ks % 3 != 2) {
                    const int u = ks - ks / 3;
                    {
                        const int n = 2 * u, gi = n >> 2, j = n & 3, g = gi >> 1, tt = gi & 1;
                        const float p0 = fmaf(Sp[0][tt][4 * g + j], WT_UNSCALE, biasp[g][j]);
                        const float p1 = fmaf(Sp[0][tt][4 * g + j + 1], WT_UNSCALE, biasp[g][j + 1]);
                        half2w a, d;
                        split_pair_prescaled(gelu16_wt(p0), gelu16_wt(p1), a, d);
                        ghi[j] = a[0];
                        ghi[j + 1] = a[1];
                        glo[j] = d[0];
                        glo[j + 1] = d[1];
                    }
                    if (u & 1) {
                        const int gi = u >> 1, g = gi >> 1, tt = gi & 1;
                        const int tok = tt * 32 + r;
                        const int pos = tok * 256 + (((4 * q + g) ^ (tok & 15)) << 4) + 8 * h;
                        *reinterpret_cast<half4*>(Hb + pos) = ghi;
                        *reinterpret_cast<half4*>(Hb + FFN2_HPLANE + pos) = glo;
                    }
                    if (MMA) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                        for (int m = 0; m < 6; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                        }
                    }
                }
                if (!MMA && (ks == 5 || ks == 11 || ks == 17) && has_next) {
                    x_next_store(ks / 6);
                    x_next_load(ks / 6 + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!MMA && has_next) x_next_store(3);
            if (MMA) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) Sp[0][tt] = S[0][tt];
#pragma unroll
                for (int g = 0; g < 4; ++g) biasp[g] = bias[g];
            }
        };
        for (int j = 0; j < J; ++j) {
            {   // an opaque zero OFFSET (not an opaque pointer: that would turn the loads into flat_load, which count
                // against lgkmcnt as well and serialise with the LDS fragment reads)
                int zero = 0;
                asm volatile("" : "+s"(zero));
                W1j = W1p + zero;
                b1j = b1 + zero;
            }
            if (j == 1) ICREC_STAMP(0, 0);
            iteration(0, std::true_type{}, std::false_type{}, false, 0, false);
            if (j == 1) ICREC_STAMP(0, 1);
            bar_lds();
            if (j == 1) ICREC_STAMP(0, 2);
            for (int c = 1; c < NC; ++c) {
                iteration(c, std::true_type{}, std::true_type{}, c == 1 && j > 0, 0, false);
                if (j == 1) ICREC_STAMP(0, 1 + 2 * c);
                bar_lds();
                if (j == 1) ICREC_STAMP(0, 2 + 2 * c);
            }
            iteration(NC, std::false_type{}, std::true_type{}, false, block_m0(j + 1), j + 1 < J);
            if (j == 1) ICREC_STAMP(0, 25);
            bar_lds();
            if (j == 1) ICREC_STAMP(0, 26);
        }
        bar_lds();  // epoch (J, 0): the consumers' last P2
        bar_lds();  // the two barriers inside the last LayerNorm
        bar_lds();
    } else {
        f32x16 Y[3][2];
        const _Float16* w2p[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) w2p[i] = W2p + wt_frag_off(q * 3 + i, 0, KS2);
        int hb[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) hb[tt] = (tt * 32 + r) * 256 + ((h ^ (r & 15)) << 4);
        half8 wh[4][3], wl[4][3];
#pragma unroll
        for (int d = 0; d < 4; ++d) w_load<3>(wh[d], wl[d], w2p, d, lo8);
        __syncthreads();  // X(0) resident (matches the producers' first barrier)
        // LAST (chunk NC-1): no prefetch past the end - the ring is dead across the LayerNorm (which needs the
        // registers) and is refilled for the next block right after it
        auto P2 = [&](int c, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            const char* const Hb = Hs + (c & 1) * FFN2_HBUF;
            half8 fh[2][2], fl[2][2];
            int hbc[2] = {hb[0], hb[1]};  // opaque per chunk: the 14 derived fragment addresses are recomputed here (one
                                          // v_xor each) instead of living in registers across the whole block loop
            asm volatile("" : "+v"(hbc[0]), "+v"(hbc[1]));
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                fh[0][tt] = *reinterpret_cast<const half8*>(Hb + hbc[tt]);
                fl[0][tt] = *reinterpret_cast<const half8*>(Hb + FFN2_HPLANE + hbc[tt]);
            }
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                if (k2 + 1 < 8) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int pos = hbc[tt] ^ ((k2 + 1) << 5);
                        fh[(k2 + 1) & 1][tt] = *reinterpret_cast<const half8*>(Hb + pos);
                        fl[(k2 + 1) & 1][tt] = *reinterpret_cast<const half8*>(Hb + FFN2_HPLANE + pos);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                wt_mma<3, 2>(Y, wh[k2 & 3], wl[k2 & 3], fh[k2 & 1], fl[k2 & 1]);
                if (!LAST || k2 < 4) w_load<3>(wh[k2 & 3], wl[k2 & 3], w2p, c * 8 + k2 + 4, lo8);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        const float *b2j = b2, *gamj = gam, *betj = bet;
        for (int j = 0; j <= J; ++j) {
            {   // per-block opaque re-definition of the weight / parameter pointers (see the producers)
                int zero = 0;
                asm volatile("" : "+s"(zero));
                b2j = b2 + zero;
                gamj = gam + zero;
                betj = bet + zero;
#pragma unroll
                for (int i = 0; i < 3; ++i) w2p[i] = W2p + zero + wt_frag_off(q * 3 + i, 0, KS2);
            }
            if (j == 1) ICREC_STAMP(4, 0);
            // requested now, used after the LayerNorm write-out: a load issued behind those stores would wait for them
            const f32x4 parv = wt_ln_par_load(gamj, betj, tid - 256);
            const f32x4 biasv = *reinterpret_cast<const f32x4*>(b2j + q * 96 + 4 * (lane < 24 ? lane : 0));
            if (j > 0) P2(NC - 1, std::true_type{});
            if (j == 1) ICREC_STAMP(4, 1);
            bar_lds();  // end of epoch (j, 0)
            if (j == 1) ICREC_STAMP(4, 2);
            int salt = 0;  // see wt_ln_out
            asm volatile("" : "+v"(salt));
            if (j > 0)
                wt_ln_out(Y, true, q, xh, xl, block_m0(j - 1), T, gamj, betj, eps, Hs + FFN2_HBUF, [] { bar_lds(); }, tid - 256, salt, &parv);
            if (j == J) break;
            if (j > 0) {
#pragma unroll
                for (int d = 0; d < 4; ++d) w_load<3>(wh[d], wl[d], w2p, d, lo8);
            }
            // Y starts from the residual + bias: the block's own planes, already in LDS (wt_res_init_*: same value); the
            // wave's 96 bias values through its own (now idle) LayerNorm tile
            float* const bq = reinterpret_cast<float*>(Hs + FFN2_HBUF + LNT_RED + q * LNT_TILE);
            if (lane < 24) *reinterpret_cast<f32x4*>(bq + 4 * lane) = biasv;
            lds_order();
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = q * 12 + i * 4 + g;
                    const f32x4 b = *reinterpret_cast<const f32x4*>(bq + i * 32 + 8 * g + 4 * h + salt);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int tok = tt * 32 + r + salt;
                        const int pos = tok * 768 + (((c & ~15) | ((c ^ tok) & 15)) << 4) + 8 * h;
                        const half4 a = *reinterpret_cast<const half4*>(Xs + pos);
                        const half4 d = *reinterpret_cast<const half4*>(Xs + FFN2_XPLANE + pos);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) Y[i][tt][4 * g + jj] = res_init_val(a[jj], d[jj], b[jj]);
                    }
                }
            if (j == 1) ICREC_STAMP(4, 3);
            bar_lds();  // end of epoch (j, 1)
            if (j == 1) ICREC_STAMP(4, 4);
            for (int c = 0; c + 1 < NC; ++c) {
                P2(c, std::false_type{});
                if (j == 1) ICREC_STAMP(4, 5 + 2 * c);
                bar_lds();
                if (j == 1) ICREC_STAMP(4, 6 + 2 * c);
            }
        }
    }
}

// ---------------------------------------------------------------- QKV projection, activation-resident form (large batches)
// out[T, N] = X . W^T + bias for a block of 64 tokens and ALL N = 1,152 features in one 8-wave workgroup, built like
// the producer half of ffn_fused2_kernel: the block's 64 x 384 activation planes are loaded into LDS ONCE (96 KB; the
// slab-ring form restreams them per 384-feature workgroup and pays a prologue, five slab barriers and an epilogue
// barrier pair per 64 x 384 outputs), every wave walks whole feature tiles - wave w: tiles w, w + 8, ... (a SIMD
// hosts waves s and s + 4 = 9 of the 36 tiles) - with K = 384 as 24 straight-line k-steps, the weight ring 8 k-steps
// deep and running across tile boundaries, no workgroup barrier after the first.  Results leave through a private
// per-wave LDS tile ([32 tokens][32 features] fp32, 144-B rows) as 16-B chunks: 128 B per token row per store.
// Per output the MFMA chain is wt_kloop's: the same bits as wt_linear_kernel<3, 2, 2, 0>.
constexpr int QKVR_STG_LD = 36;                                  // floats per staged row (+16 B)
constexpr int QKVR_STG_BYTES = 32 * QKVR_STG_LD * 4;             // 4,608 B per wave
constexpr int QKVR_LDS = FFN2_X_BYTES + 8 * QKVR_STG_BYTES;      // 135,168 B

__global__ __launch_bounds__(512, 2) void qkv_resident_kernel(const _Float16* __restrict__ xh,
                                                              const _Float16* __restrict__ xl, int T,
                                                              const _Float16* __restrict__ Wp,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ out, int N) {
    constexpr int KS1 = 24;
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    char* const Xs = smem2;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), r = lane & 31, h = lane >> 5;
    float* const stg = reinterpret_cast<float*>(smem2 + FFN2_X_BYTES + wave * QKVR_STG_BYTES);
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int NT = N / 32;
    // ---- the block's activation planes -> LDS, once (layout of ffn_fused2_kernel)
    {
        u32x4 vh[6], vl[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 512 * i, row = id / 48, c = id - row * 48;
            int64_t g = m0 + row;
            g = g < T ? g : (int64_t)T - 1;
            vh[i] = *reinterpret_cast<const u32x4*>(xh + g * 384 + c * 8);
            vl[i] = *reinterpret_cast<const u32x4*>(xl + g * 384 + c * 8);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 512 * i, row = id / 48, c = id - row * 48;
            const int pos = row * 768 + (((c & ~15) | ((c ^ row) & 15)) << 4);
            *reinterpret_cast<u32x4*>(Xs + pos) = vh[i];
            *reinterpret_cast<u32x4*>(Xs + FFN2_XPLANE + pos) = vl[i];
        }
    }
    const unsigned lo8 = lane * 8;
    int xb[8][2];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int tok = tt * 32 + r;
            xb[m][tt] = tok * 768 + (((2 * m) ^ h ^ (tok & 15)) << 4);
        }
    half8 wh[8][1], wl[8][1];
    {
        const _Float16* const wp0[1] = {Wp + wt_frag_off(wave, 0, KS1)};
#pragma unroll
        for (int d = 0; d < 8; ++d) w_load<1>(wh[d], wl[d], wp0, d, lo8);
    }
    __syncthreads();  // X resident
    // (Spreading a tile's write-out over the next tile's k-steps - one store per k-step instead of a burst of eight -
    // was measured at the same speed with 256 instead of 190 VGPRs; running without any store is 72 us per launch
    // faster: the 604 MB of fp32 QKV rows per launch are what the rest of the time buys.)
    for (int nt = wave; nt < NT; nt += 8) {
        const int nn = nt + 8 < NT ? nt + 8 : nt;  // past the last tile: re-read this one's fragments (never consumed)
        const _Float16* const wp1[1] = {Wp + wt_frag_off(nt, 0, KS1)};
        const _Float16* const wpn[1] = {Wp + wt_frag_off(nn, 0, KS1)};
        f32x4 bv[4];  // loaded BEFORE the k-loop (vmcnt counts in order: behind the ring it would wait for the whole ring)
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(bias + nt * 32 + 8 * g + 4 * h);
        f32x16 S[1][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e = 0; e < 16; ++e) S[0][tt][e] = 0.0f;
        half8 fh[2][2], fl[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            fh[0][tt] = *reinterpret_cast<const half8*>(Xs + xb[0][tt]);
            fl[0][tt] = *reinterpret_cast<const half8*>(Xs + FFN2_XPLANE + xb[0][tt]);
        }
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            if (ks + 1 < KS1) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int pos = xb[(ks + 1) & 7][tt] + ((ks + 1) >> 3) * 256;
                    fh[(ks + 1) & 1][tt] = *reinterpret_cast<const half8*>(Xs + pos);
                    fl[(ks + 1) & 1][tt] = *reinterpret_cast<const half8*>(Xs + FFN2_XPLANE + pos);
                }
            }
            wt_mma<1, 2>(S, wh[ks & 7], wl[ks & 7], fh[ks & 1], fl[ks & 1]);
            if (ks + 8 < KS1) w_load<1>(wh[ks & 7], wl[ks & 7], wp1, ks + 8, lo8);
            else w_load<1>(wh[ks & 7], wl[ks & 7], wpn, ks + 8 - KS1, lo8);
            __builtin_amdgcn_sched_barrier(0);  // pin the prefetch to its k-step
        }
        // ---- this tile out: [32 tokens][32 features] per pass through the wave's private LDS tile
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaf(S[0][tt][4 * g + j], WT_UNSCALE, bv[g][j]);
                *reinterpret_cast<f32x4*>(stg + r * QKVR_STG_LD + 8 * g + 4 * h) = v;
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

// W (fp32 [N, K]) -> packed f16 hi/lo fragments (wt_gemm.h), once at encoder creation.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int N, int K,
                                                           _Float16* __restrict__ out) {
    const int KS = K / 16;
    const size_t n = (size_t)(N / 32) * KS * 64;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (size_t)gridDim.x * 256) {
        const size_t fr = id >> 6;
        const int lane = (int)(id & 63), r = lane & 31, h = lane >> 5;
        const int nt = (int)(fr / KS), ks = (int)(fr % KS);
        const float* src = w + (size_t)(nt * 32 + r) * K + ks * 16 + 8 * h;
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 a, b;
            split_scaled(src[j], WT_SW, a, b);
            hi[j] = a;
            lo[j] = b;
        }
        *reinterpret_cast<half8*>(out + fr * (2 * WT_FRAG) + lane * 8) = hi;
        *reinterpret_cast<half8*>(out + fr * (2 * WT_FRAG) + WT_FRAG + lane * 8) = lo;
    }
}

// ---------------------------------------------------------------- attention (tf:111-136, 164-203)
// One workgroup = one (sequence, head) and up to four 32-row query blocks (one per wave).
// S^T = K.Q^T is computed with keys on the accumulator rows, so each lane ends up with the
// scores of ONE query (column = lane & 31) against 16 keys per 32-key tile.  The softmax is
// then lane-local plus one exchange with lane^32, and the exponentiated accumulator registers
// are fed back unchanged as the A operand of P.V (A[i=query][k=key]): no transpose, no LDS
// round trip.  Key order inside the P.V chain is therefore, per 32-key tile,
//   e = 0..15: key (e&3)+8(e>>2) then key (e&3)+8(e>>2)+4
// and the softmax denominator is the sum of the two half-wave partial sums; the oracle
// (icrec_oracle.c, attention block) accumulates in exactly this order.
constexpr int DH = 32;
constexpr int LDQ = 36;  // K LDS row stride (even/odd split layout, like the GEMM tiles)

// One launch per length bucket: NKT = max 32-key tiles (1, 2, 4, 8), WAVES = query blocks per
// workgroup.  A workgroup whose sequence belongs to another bucket exits at once, so short
// sequences run with the LDS footprint / occupancy of their own bucket even in a mixed batch.
// K and V of the (sequence, head) live in LDS; each wave's 32 query rows come straight from
// global memory into the B-operand registers.
template <int NKT, int WAVES, bool SPLIT>
__global__ __launch_bounds__(WAVES * 64) void attention_kernel(const float* __restrict__ qkv,
                                                               const int32_t* __restrict__ cu, int heads, int H,
                                                               float scale_log2e, float* __restrict__ ctx,
                                                               _Float16* __restrict__ ch, _Float16* __restrict__ cl) {
    __shared__ __attribute__((aligned(16))) float Ks[NKT * 32 * LDQ];
    __shared__ __attribute__((aligned(16))) float Vs[NKT * 32 * DH];
    __shared__ float Ls[WAVES * 32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int t0 = cu[s], L = cu[s + 1] - t0;
    const int nkt = (L + 31) >> 5;
    if (nkt > NKT || (NKT > 1 && nkt <= NKT / 2)) return;  // another bucket's sequence
    const int qb0 = blockIdx.y * WAVES;
    if (qb0 >= nkt) return;
    const int ld = 3 * H;

    // stage K (even/odd split) and V; rows past L are clamped (masked below / never stored)
    for (int id = tid; id < nkt * 32 * 8; id += WAVES * 64) {
        const int row = id >> 3, c = id & 7;
        const int rr = row < L ? row : L - 1;
        const float* src = qkv + (size_t)(t0 + rr) * ld + hd * DH + c * 4;
        const float4 kv = *reinterpret_cast<const float4*>(src + H);
        const float4 vv = *reinterpret_cast<const float4*>(src + 2 * H);
        float* kp = Ks + row * LDQ + (c >> 1) * 8 + (c & 1) * 2;
        *reinterpret_cast<float2*>(kp) = make_float2(kv.x, kv.z);
        *reinterpret_cast<float2*>(kp + 4) = make_float2(kv.y, kv.w);
        *reinterpret_cast<float4*>(Vs + row * DH + c * 4) = vv;
    }
    const int r = lane & 31, h = lane >> 5;
    const int qb = qb0 + wave;
    // this lane's query row -> B fragments (lane half h supplies the even / odd head dims)
    float4 qf[4];
    {
        int qr = qb * 32 + r;
        qr = qr < L ? qr : L - 1;
        const float4* qp = reinterpret_cast<const float4*>(qkv + (size_t)(t0 + qr) * ld + hd * DH);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const float4 a = qp[2 * kq], b = qp[2 * kq + 1];
            qf[kq] = h == 0 ? make_float4(a.x, a.z, b.x, b.z) : make_float4(a.y, a.w, b.y, b.w);
        }
    }
    __syncthreads();
    if (qb >= nkt) return;  // idle wave (no barrier below)

    // ---- S^T tiles: sc[kt][e] = sum_d K[kt*32 + krow(e)][d] * Q[qb*32 + r][d]
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[kt][e] = 0.0f;
        if (kt < nkt) {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const float4 kf = *reinterpret_cast<const float4*>(Ks + (kt * 32 + r) * LDQ + kq * 8 + h * 4);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[kq].x, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[kq].y, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[kq].z, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[kq].w, sc[kt], 0, 0, 0);
            }
        }
    }
    // ---- scores in log2 units (scale * log2(e) folded), key tail masked, row max
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
            const bool last = kt == nkt - 1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = sc[kt][e] * scale_log2e;
                if (last && kt * 32 + acc_row(e, lane) >= L) v = -INFINITY;
                sc[kt][e] = v;
                mx = fmaxf(mx, v);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // ---- p = 2^(v - max); denominator = this half-wave's keys ascending, then the two halves added
    float lsum = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(sc[kt][e] - mx);
                sc[kt][e] = p;
                lsum = lsum + p;
            }
        }
    }
    {
        const float other = __shfl_xor(lsum, 32, 64);
        lsum = h == 0 ? lsum + other : other + lsum;  // l0 + l1 in both halves
    }
    if (h == 0) Ls[wave * 32 + r] = lsum;
    // ---- O = P.V with P taken straight from the accumulator registers
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[kt][e], Vs[key * DH + r], o, 0, 0, 0);
            }
        }
    }
    // ---- normalise rows by their denominator and store (row = query, column = head dim);
    // Ls was written by this wave's own lanes (wave-local LDS ordering makes it visible).
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int qrow = acc_row(e, lane);
        const int tq = qb * 32 + qrow;
        if (tq < L) {
            const float v = o[e] / Ls[wave * 32 + qrow];
            const size_t at = (size_t)(t0 + tq) * H + hd * DH + r;
            if (SPLIT) {
                _Float16 hi, lo;
                split_act(v, hi, lo);
                ch[at] = hi;
                cl[at] = lo;
            } else {
                ctx[at] = v;
            }
        }
    }
}

// ---------------------------------------------------------------- dispatch order of the attention workgroups
// order[0 .. n_seqs) = the sequences sorted by length, longest first (counting sort over the 256 possible lengths;
// the order inside one length is whatever the atomics give - the workgroups are independent, results do not depend
// on it).  The attention launches map workgroup b to sequence order[b / heads]: each bucket's workgroups are then
// dispatched longest-first with the other buckets' (empty) workgroups behind them instead of in between, so a launch
// does not end on a few 8-tile sequences that started last.
__global__ __launch_bounds__(1024) void seq_order_kernel(const int32_t* __restrict__ cu, int n_seqs,
                                                         int32_t* __restrict__ order) {
    __shared__ int hist[257];
    const int tid = threadIdx.x;
    for (int i = tid; i < 257; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int s = tid; s < n_seqs; s += 1024) {
        int L = cu[s + 1] - cu[s];
        L = L < 1 ? 1 : (L > 256 ? 256 : L);
        atomicAdd(&hist[256 - L], 1);
    }
    __syncthreads();
    if (tid == 0) {  // exclusive prefix sum: hist[b] becomes the first slot of bin b
        int run = 0;
        for (int b = 0; b < 257; ++b) { const int c = hist[b]; hist[b] = run; run += c; }
    }
    __syncthreads();
    for (int s = tid; s < n_seqs; s += 1024) {
        int L = cu[s + 1] - cu[s];
        L = L < 1 ? 1 : (L > 256 ? 256 : L);
        order[atomicAdd(&hist[256 - L], 1)] = s;
    }
}

// ---------------------------------------------------------------- attention, f16x3 arithmetic
// Same structure as attention_kernel (one block per (sequence, head), S^T on the accumulator rows, P fed
// back from the accumulators), with both products on the f16 MFMA by the 3-term split of gemm_x3.h:
//   S^T = K_hi.Q_hi + 2^-11 (K_hi.Q_lo + K_lo.Q_hi)        O = P_hi.V_hi + 2^-11 (P_hi.V_lo + P_lo.V_hi)
// K is staged as hi/lo f16 planes [key][32] (64-B rows, 16-B chunks XOR-swizzled by (key>>2)&3), V as
// TRANSPOSED hi/lo planes [dim][key] so that a lane's eight k-slots (keys) of one head dim are two 8-B
// reads; Q (per wave) and P (per tile, straight from the accumulators) are split in registers.
// For a 32-key tile and k-step s, slot j of lane-half h is key 4h + (j&3) + 8(2s + (j>>2)) — the keys
// accumulator register e = 8s + j holds — on both operands.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split8(const float (&x)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 a = (_Float16)x[j];
        hi[j] = a;
        lo[j] = (_Float16)((x[j] - (float)a) * LO_SCALE);
    }
}

template <int NKT, int WAVES, bool SPLIT>
__global__ __launch_bounds__(WAVES * 64, (NKT >= 8 ? 4 : 1)) void attention_x3_kernel(const float* __restrict__ qkv,
                                                                  const int32_t* __restrict__ cu, int heads, int H,
                                                                  float scale_log2e, float* __restrict__ ctx,
                                                                  _Float16* __restrict__ ch, _Float16* __restrict__ cl,
                                                                  const int32_t* __restrict__ order) {
    // Single-accumulator form of the split (wt_gemm.h): every operand is carried as hi/lo f16 planes of 16 x (Q, K, V)
    // or 1024 p (the probabilities), the three products of a k-step accumulate into ONE fp32 tile, and the power-of-two
    // scales are folded into constants: S' = 256 S, O' = 16384 sum_k p_k V_k, l' = 1024 sum_k p_k, O = O' / (16 l').
    // All splits are the 3-instruction form (mask / subtract / v_cvt_pkrtz pairs).
    //
    // Long bucket (NKT = 8, RECOMP): the score tiles are computed TWICE - once for the row maximum, once more for the
    // exponentials, each tile consumed by the PV product as soon as it exists - instead of all eight being held in
    // 128 registers between the two passes.  Same MFMA chains, same order of every sum: the same bits, 48 more MFMAs
    // per wave on an idle matrix pipe, and the kernel drops under 128 VGPRs; with the output tile parked on the K
    // planes (behind one more barrier) it also drops to 67 KB of LDS - TWO workgroups per CU, so one's staging and
    // barrier phases run under the other's arithmetic.
    constexpr bool RECOMP = NKT >= 8;
    constexpr int VT = NKT * 32 + 4;  // V^T row stride in halfs (+8 B: the 32 dims land on distinct banks)
    __shared__ __attribute__((aligned(16))) _Float16 Kbuf[2 * NKT * 32 * 32];
    _Float16* const Kh = Kbuf;
    _Float16* const Kl = Kbuf + NKT * 32 * 32;
    __shared__ __attribute__((aligned(16))) _Float16 Vh[32 * VT];
    __shared__ __attribute__((aligned(16))) _Float16 Vl[32 * VT];
    __shared__ float Ls[WAVES * 32];
    // Plane output (SPLIT): every wave parks its 32 x 32 output tile (hi | lo) on the K planes once all waves have left
    // them (second barrier) and writes it out 16 B per lane - 4 store instructions instead of 32 two-byte ones.
    constexpr bool OB = SPLIT;
    static_assert(!OB || WAVES * 2 * 32 * 32 <= 2 * NKT * 32 * 32, "the output tiles reuse the K planes");
    _Float16* const Ob = Kbuf;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sidx = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int s = order != nullptr ? order[sidx] : sidx;  // seq_order_kernel: longest first
    const int t0 = cu[s], L = cu[s + 1] - t0;
    const int nkt = (L + 31) >> 5;
    if (nkt > NKT || (NKT > 1 && nkt <= NKT / 2)) return;  // another bucket's sequence
    const int qb0 = blockIdx.y * WAVES;
    if (qb0 >= nkt) return;
    const int ld = 3 * H;

    const int r = lane & 31, h = lane >> 5;
    const int qb = qb0 + wave;
    ICREC_STAMP(0, 0);
    half8 qh[2], ql[2];  // B operand of S^T: this lane's query row, dims 16s + 8h .. +7
    {
        int qr = qb * 32 + r;
        qr = qr < L ? qr : L - 1;
        const float* qp = qkv + (size_t)(t0 + qr) * ld + hd * DH;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 8 * h);
            const f32x4 b = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 8 * h + 4);
            half4 ah, al, bh, bl;
            split_act4(a, ah, al);
            split_act4(b, bh, bl);
            qh[ks] = half8{ah[0], ah[1], ah[2], ah[3], bh[0], bh[1], bh[2], bh[3]};
            ql[ks] = half8{al[0], al[1], al[2], al[3], bl[0], bl[1], bl[2], bl[3]};
        }
    }
    // K/V staging: all of this thread's loads are issued before the first one is consumed
    constexpr int STG = NKT * 32 * 8 / (WAVES * 64);  // = 4 for every bucket
    f32x4 kreg[STG], vreg[STG];
#pragma unroll
    for (int it = 0; it < STG; ++it) {
        const int id = tid + it * WAVES * 64;
        const int key = id >> 3, c = id & 7;  // c: 4-dim group
        const int rr = key < L ? key : L - 1;
        const float* src = qkv + (size_t)(t0 + rr) * ld + hd * DH + c * 4;
        kreg[it] = *reinterpret_cast<const f32x4*>(src + H);
        vreg[it] = *reinterpret_cast<const f32x4*>(src + 2 * H);
    }
#pragma unroll
    for (int it = 0; it < STG; ++it) {
        const int id = tid + it * WAVES * 64;
        if (id < nkt * 32 * 8) {
            const int key = id >> 3, c = id & 7;
            half4 khi, klo, vhi, vlo;
            split_act4(kreg[it], khi, klo);
            split_act4(vreg[it], vhi, vlo);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Vh[(c * 4 + i) * VT + key] = vhi[i];
                Vl[(c * 4 + i) * VT + key] = vlo[i];
            }
            const int off = key * 32 + ((((c >> 1) ^ ((key >> 2) & 3)) << 3) | ((c & 1) << 2));
            *reinterpret_cast<half4*>(Kh + off) = khi;
            *reinterpret_cast<half4*>(Kl + off) = klo;
        }
    }
    ICREC_STAMP(0, 1);
    __syncthreads();
    ICREC_STAMP(0, 2);
    const bool active = qb < nkt;
    if (!RECOMP && !OB && !active) return;  // idle wave (no barrier below)

    // raw scores S' = 256 S of key tile kt for this wave's 32 queries, keys beyond the sequence at -inf (only the one
    // tile that has any pays for the selects: uniform branch)
    auto score_tile = [&](int kt) {
        f32x16 t;
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = 0.0f;
        const int key = kt * 32 + r;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = key * 32 + (((2 * ks + h) ^ ((key >> 2) & 3)) << 3);
            const half8 kh = *reinterpret_cast<const half8*>(Kh + off);
            const half8 kl = *reinterpret_cast<const half8*>(Kl + off);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], t, 0, 0, 0);
        }
        if (kt == nkt - 1 && kt * 32 + 32 > L) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (kt * 32 + acc_row(e, lane) >= L) t[e] = -INFINITY;
        }
        return t;
    };
    // P (a tile of p' = 1024 p in the accumulator layout) times V, into o
    auto pv_tile = [&](int kt, const f32x16& pt, f32x16& o) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                half2w a, b;
                split_pair_prescaled(pt[8 * ks + j], pt[8 * ks + j + 1], a, b);
                ph[j] = a[0]; ph[j + 1] = a[1];
                pl[j] = b[0]; pl[j + 1] = b[1];
            }
            // this lane's head dim r, keys base .. base+3 and base+8 .. base+11
            const int base = kt * 32 + 4 * h + 16 * ks;
            const half4 v0h = *reinterpret_cast<const half4*>(Vh + r * VT + base);
            const half4 v1h = *reinterpret_cast<const half4*>(Vh + r * VT + base + 8);
            const half4 v0l = *reinterpret_cast<const half4*>(Vl + r * VT + base);
            const half4 v1l = *reinterpret_cast<const half4*>(Vl + r * VT + base + 8);
            const half8 vh = {v0h[0], v0h[1], v0h[2], v0h[3], v1h[0], v1h[1], v1h[2], v1h[3]};
            const half8 vl = {v0l[0], v0l[1], v0l[2], v0l[3], v1l[0], v1l[1], v1l[2], v1l[3]};
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, o, 0, 0, 0);
        }
    };
    typedef float float2w __attribute__((ext_vector_type(2)));
    const float cs = scale_log2e * (1.0f / 256.0f);  // scores in log2 units from S' = 256 S
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.0f;
    if (RECOMP) {
        if (active) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt < nkt) {
                    const f32x16 t = score_tile(kt);
#pragma unroll
                    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, t[e]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float shift = fmaf(-mx, cs, 10.0f);
            float2w ls2 = float2w{0.0f, 0.0f};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt < nkt) {
                    f32x16 t = score_tile(kt);
#pragma unroll
                    for (int e = 0; e < 16; ++e) t[e] = __builtin_amdgcn_exp2f(fmaf(t[e], cs, shift));
#pragma unroll
                    for (int e = 0; e < 16; e += 2) ls2 = ls2 + float2w{t[e], t[e + 1]};
                    pv_tile(kt, t, o);
                }
            }
            float lrow = ls2[0] + ls2[1];
            const float other = __shfl_xor(lrow, 32, 64);
            lrow = h == 0 ? lrow + other : other + lrow;
            if (h == 0) Ls[wave * 32 + r] = lrow;
        }
        ICREC_STAMP(0, 3);
        __syncthreads();  // every wave has left the K planes: they become the output tiles
        if (!active) return;
    } else {
    if (active) {
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
        if (kt < nkt) sc[kt] = score_tile(kt);
    // Softmax on the raw scores (S' = 256 S): the keys beyond the sequence are masked in the one tile that has any
    // (uniform branch), the maximum is taken before scaling, and scale, shift and the 2^10 factor of p' = 1024 p go
    // into one fma in front of the exponential: p' = 2^(S' cs - max' cs + 10).
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[kt][e]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float shift = fmaf(-mx, cs, 10.0f);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[kt][e] = __builtin_amdgcn_exp2f(fmaf(sc[kt][e], cs, shift));
        }
    }
    // row sums l' = sum_k p'_k: two interleaved chains per lane (packed adds), the halves of a row joined by a shuffle
    float2w ls2 = float2w{0.0f, 0.0f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) ls2 = ls2 + float2w{sc[kt][e], sc[kt][e + 1]};
        }
    }
    float lrow = ls2[0] + ls2[1];
    {
        const float other = __shfl_xor(lrow, 32, 64);
        lrow = h == 0 ? lrow + other : other + lrow;
    }
    if (h == 0) Ls[wave * 32 + r] = lrow;
    ICREC_STAMP(0, 3);

#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
        if (kt < nkt) pv_tile(kt, sc[kt], o);
    }
    if (OB) {
        __syncthreads();  // every wave has left the K planes
        if (!active) return;
    }
    }
    ICREC_STAMP(0, 4);
    if (OB) {
        // park the wave's 32 x 32 output tile (hi and lo planes) in LDS row-major, then write it out 16 B
        // per lane: 4 store instructions instead of 32 two-byte ones
        _Float16* ob = Ob + wave * (2 * 32 * 32);
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const int q0 = acc_row(e, lane), q1 = acc_row(e + 1, lane);
            const float v0 = o[e] * (0.0625f * __builtin_amdgcn_rcpf(Ls[wave * 32 + q0]));
            const float v1 = o[e + 1] * (0.0625f * __builtin_amdgcn_rcpf(Ls[wave * 32 + q1]));
            half2w hi, lo;
            split_pair_prescaled(v0 * WT_SA, v1 * WT_SA, hi, lo);
            ob[q0 * 32 + r] = hi[0];
            ob[q1 * 32 + r] = hi[1];
            ob[32 * 32 + q0 * 32 + r] = lo[0];
            ob[32 * 32 + q1 * 32 + r] = lo[1];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int id = lane + 64 * t, qrow = id >> 2, c8 = (id & 3) * 8;
            const int tq = qb * 32 + qrow;
            if (tq < L) {
                const size_t at = (size_t)(t0 + tq) * H + hd * DH + c8;
                *reinterpret_cast<u32x4*>(ch + at) = *reinterpret_cast<const u32x4*>(ob + qrow * 32 + c8);
                *reinterpret_cast<u32x4*>(cl + at) = *reinterpret_cast<const u32x4*>(ob + 32 * 32 + qrow * 32 + c8);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int qrow = acc_row(e, lane);
            const int tq = qb * 32 + qrow;
            if (tq < L) {
                const float v = o[e] * (0.0625f * __builtin_amdgcn_rcpf(Ls[wave * 32 + qrow]));
                const size_t at = (size_t)(t0 + tq) * H + hd * DH + r;
                if (SPLIT) {
                    _Float16 hi, lo;
                    split_act(v, hi, lo);
                    ch[at] = hi;
                    cl[at] = lo;
                } else {
                    ctx[at] = v;
                }
            }
        }
    }
    ICREC_STAMP(0, 5);
#ifdef ICREC_STAMPS
    if (threadIdx.x == 0) {  // where this workgroup ran: HW_ID (cu / sh / se in bits 8..15) and XCC_ID, for the per-CU timeline of the harness
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        g_stamps[((size_t)blockIdx.x * 2) * 64 + 8] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

// ---------------------------------------------------------------- mean pooling + L2 normalise
// sentence_transformers Pooling(mean): sum_t h_t / clamp(count, 1e-9); then n_norm times
// x / max(|x|_2, 1e-12) (Normalize module, normalize_embeddings=True).  One workgroup of
// H threads per sequence; norm in oracle order by wave 0.
template <int H, bool PLANES>
__global__ __launch_bounds__(H) void pool_norm_kernel(const float* __restrict__ x, const _Float16* __restrict__ xh,
                                                      const _Float16* __restrict__ xl, const int32_t* __restrict__ cu,
                                                      int n_norm, float* __restrict__ out) {
    // PLANES (f16x3 mode): the hidden state is its two planes, h_t = (float(hi) + float(lo)) / 16 exactly
    auto at = [&](size_t idx) { return PLANES ? ((float)xh[idx] + (float)xl[idx]) * (1.0f / WT_SA) : x[idx]; };
    __shared__ float v[H];
    __shared__ float den_s;
    const int s = blockIdx.x, i = threadIdx.x;
    const int t0 = cu[s], t1 = cu[s + 1];
    float acc = 0.0f;
    int t = t0;
    for (; t + 8 <= t1; t += 8) {  // 8 independent loads in flight, summed in ascending token order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = at((size_t)(t + u) * H + i);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = acc + v[u];
    }
    for (; t < t1; ++t) acc = acc + at((size_t)t * H + i);
    float cnt = (float)(t1 - t0);
    cnt = cnt < 1e-9f ? 1e-9f : cnt;
    float val = acc / cnt;
    for (int rep = 0; rep < n_norm; ++rep) {
        v[i] = val;
        __syncthreads();
        if (i < 64) {
            float a = 0.0f;
#pragma unroll
            for (int j = 0; j < H / 64; ++j) a = fmaf(v[i + 64 * j], v[i + 64 * j], a);
            const float nrm = sqrtf(wave_sum_f32(a));
            if (i == 0) den_s = nrm > 1e-12f ? nrm : 1e-12f;
        }
        __syncthreads();
        val = val / den_s;
        __syncthreads();
    }
    out[(size_t)s * H + i] = val;
}

// ---------------------------------------------------------------- host side
constexpr int HID = 384;

struct LayerW {
    float *Wqkv, *bqkv, *Wo, *bo, *g1, *b1n, *W1, *b1, *W2, *b2, *g2, *b2n;
    // the four weight matrices as packed f16 hi/lo fragments (wt_gemm.h; gemm_mode F16X3 only)
    _Float16 *Wqkv_p, *Wo_p, *W1_p, *W2_p;
};
struct Encoder {
    icrec_bert_cfg cfg;
    int device = 0;
    int n_cu = 256;
    float* blob = nullptr;      // the uploaded weight blob
    float* extra = nullptr;     // repacked Wqkv / bqkv
    _Float16* planes = nullptr; // packed weight fragments (F16X3)
    float *word, *pos, *type, *eg, *eb;
    LayerW layers[64];
    // Side stream + events of one caller stream: the short remainder of a large batch (batch_split) and the shorter
    // attention buckets run beside the batch kernels of the same layer instead of behind them.  One set per caller
    // stream (created on first use, kept for the encoder's life), so that concurrent icrec_encode calls on different
    // streams - DeviceEncoder runs the two halves of a batch that way - do not queue behind each other's side work.
    struct Side {
        hipStream_t caller = nullptr, side = nullptr;
        hipEvent_t ev_main = nullptr, ev_qkv_tail = nullptr, ev_att = nullptr, ev_tail = nullptr, ev_q = nullptr, ev_sa = nullptr;
    };
    std::mutex side_mu;
    std::vector<Side*> sides;
};

static int side_for(Encoder* e, hipStream_t caller, Encoder::Side** out) {
    std::lock_guard<std::mutex> lock(e->side_mu);
    *out = nullptr;
    for (Encoder::Side* sd : e->sides)
        if (sd->caller == caller) { *out = sd; return ICREC_OK; }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (caller && hipStreamIsCapturing(caller, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return ICREC_OK;  // no stream / event creation inside a capture: the call stays on the caller's stream
    Encoder::Side* sd = new Encoder::Side();
    sd->caller = caller;
    ICREC_HIP(hipStreamCreateWithFlags(&sd->side, hipStreamNonBlocking));
    for (hipEvent_t* ev : {&sd->ev_main, &sd->ev_qkv_tail, &sd->ev_att, &sd->ev_tail, &sd->ev_q, &sd->ev_sa})
        ICREC_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    e->sides.push_back(sd);
    *out = sd;
    return ICREC_OK;
}

static size_t weight_count(const icrec_bert_cfg* c) {
    const size_t H = c->hidden, I = c->intermediate;
    const size_t emb = (size_t)c->vocab_size * H + (size_t)c->max_position * H + (size_t)c->type_vocab * H + 2 * H;
    const size_t per = 4 * (H * H + H) + 2 * H + (I * H + I) + (H * I + H) + 2 * H;
    return emb + per * c->layers;
}

struct EncWs {
    size_t x, xs, qkv, ctx, t1, h, total;
};
static EncWs enc_ws(const icrec_bert_cfg& c, int64_t T) {
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    EncWs w;
    size_t o = 0;
    w.x = o;   o += al((size_t)T * c.hidden * 4);
    w.xs = o;  o += al((size_t)T * c.hidden * 4);        // x as f16 hi/lo planes (F16X3)
    w.qkv = o; o += al((size_t)T * 3 * c.hidden * 4);
    w.ctx = o; o += al((size_t)T * c.hidden * 4);        // fp32 ctx, or its two f16 planes
    w.t1 = o;  o += al((size_t)T * c.hidden * 4);
    w.h = o;   o += al((size_t)T * c.intermediate * 4);  // fp32 h, or its two f16 planes
    w.total = o;
    return w;
}

typedef TileCfg<2, 2, 2, 2> GemmBig;  // 128 x 128 output tile, 4 waves
constexpr int X3_SMALL_M = 512;  // <= this many tokens: the latency form (32-token x 128-feature blocks, many workgroups)

template <bool GELU>
static void launch_linear(const float* A, int M, int K, const float* W, int N, const float* bias, float* out,
                          hipStream_t st) {
    const int mt = (M + GemmBig::BM - 1) / GemmBig::BM, nt = (N + GemmBig::BN - 1) / GemmBig::BN;
    hipLaunchKernelGGL((linear_kernel<GemmBig, GELU>), dim3(mt * nt), dim3(GemmBig::THREADS), 0, st, A, M, K, W, N,
                       bias, out, nt);
}

// f16x3 linear layer through the weights-direct engine.  Single requests / micro-batches are latency-bound (a
// handful of workgroups, each walking its K loop): they use 32-token x 128-feature blocks — one 32x32 tile per
// wave, as many workgroups as the shape allows; batches use 64-token x 384-feature blocks (3 x 2 tiles per
// wave).  Per-output arithmetic is the same chain in both, so a request encodes to the same bits either way.
template <int EPI>
static void launch_wt_linear(const _Float16* Xh, const _Float16* Xl, int T, int K, const _Float16* Wp, int N,
                             const float* bias, float* out, _Float16* oh, _Float16* ol, hipStream_t st) {
    if (T <= X3_SMALL_M) {
        const int nbn = N / 128;
        hipLaunchKernelGGL((wt_linear_kernel<1, 1, 4, EPI>), dim3(((T + 31) / 32) * nbn), dim3(256), 0, st, Xh, Xl, T, K,
                           Wp, N, bias, out, oh, ol, nbn);
    } else {
        const int nbn = N / 384;
        hipLaunchKernelGGL((wt_linear_kernel<3, 2, 2, EPI>), dim3(((T + 63) / 64) * nbn), dim3(256), 0, st, Xh, Xl, T, K,
                           Wp, N, bias, out, oh, ol, nbn);
    }
}

// Launch every length bucket that can occur for max_seqlen (a bucket whose workgroups all exit
// costs a few microseconds; single-sequence calls launch exactly one bucket).
template <bool SPLIT, bool X3>
static void launch_attention(const float* qkv, const int32_t* cu, int n_seqs, int heads, int H, int max_seqlen,
                             float* ctx, _Float16* ch, _Float16* cl, hipStream_t st, int buckets = 15,
                             const int32_t* order = nullptr) {
    // buckets: bit b set = launch the bucket of 2^b key tiles (callers split the buckets over two streams)
    const float sl2e = (1.0f / sqrtf((float)DH)) * 1.44269504088896340736f;
    const int nkt_max = (max_seqlen + 31) / 32;
    const bool single = n_seqs == 1;
    const dim3 grid1(n_seqs * heads, 1);
#define ICREC_ATT(NKT, W)                                                                                        \
    do {                                                                                                         \
        if (X3) hipLaunchKernelGGL((attention_x3_kernel<NKT, W, SPLIT>), grid1, dim3(W * 64), 0, st, qkv, cu, heads, H, sl2e, ctx, ch, cl, order); \
        else hipLaunchKernelGGL((attention_kernel<NKT, W, SPLIT>), grid1, dim3(W * 64), 0, st, qkv, cu, heads, H, sl2e, ctx, ch, cl);      \
    } while (0)
    if ((buckets & 1) && (single ? nkt_max == 1 : true)) ICREC_ATT(1, 1);
    if ((buckets & 2) && (single ? nkt_max == 2 : nkt_max >= 2)) ICREC_ATT(2, 2);
    if ((buckets & 4) && (single ? (nkt_max == 3 || nkt_max == 4) : nkt_max >= 3)) ICREC_ATT(4, 4);
    if ((buckets & 8) && nkt_max >= 5) ICREC_ATT(8, 8);
#undef ICREC_ATT
}

}  // namespace icrec

using namespace icrec;

extern "C" {

size_t icrec_encoder_weight_count(const icrec_bert_cfg* cfg) { return cfg ? weight_count(cfg) : 0; }

int icrec_encoder_create(const float* weights_host, size_t n_floats, const icrec_bert_cfg* cfg, int device,
                         icrec_encoder** out) {
    ICREC_REQUIRE(weights_host && cfg && out, "icrec_encoder_create: NULL argument");
    ICREC_REQUIRE(cfg->hidden == HID, "icrec_encoder_create: this build supports hidden=384 only (got %d)", cfg->hidden);
    ICREC_REQUIRE(cfg->heads * DH == cfg->hidden, "icrec_encoder_create: head_dim must be 32 (heads=%d)", cfg->heads);
    ICREC_REQUIRE(cfg->intermediate >= 384 && cfg->intermediate % 384 == 0, "icrec_encoder_create: intermediate size must be a multiple of 384 (got %d)", cfg->intermediate);
    ICREC_REQUIRE(cfg->layers >= 1 && cfg->layers <= 64, "icrec_encoder_create: layers must be in [1,64]");
    ICREC_REQUIRE(cfg->vocab_size >= 1 && cfg->max_position >= 1 && cfg->type_vocab >= 1, "icrec_encoder_create: bad vocab/position sizes");
    ICREC_REQUIRE(cfg->n_normalize >= 0 && cfg->n_normalize <= 4, "icrec_encoder_create: n_normalize must be in [0,4]");
    ICREC_REQUIRE(cfg->gemm_mode == ICREC_GEMM_F32 || cfg->gemm_mode == ICREC_GEMM_F16X3, "icrec_encoder_create: unknown gemm_mode %d", cfg->gemm_mode);
    ICREC_REQUIRE(n_floats == weight_count(cfg), "icrec_encoder_create: weight blob has %zu floats, expected %zu", n_floats, weight_count(cfg));
    ICREC_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ICREC_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("icrec_encoder_create: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
        return ICREC_ENODEV;
    }
    Encoder* e = new Encoder();
    e->cfg = *cfg;
    e->device = device;
    e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t H = cfg->hidden, I = cfg->intermediate;
    const size_t mat_per_layer = 3 * H * H + H * H + I * H + H * I;
    const bool x3 = cfg->gemm_mode == ICREC_GEMM_F16X3;
    if (hipMalloc(&e->blob, n_floats * 4) != hipSuccess ||
        hipMalloc(&e->extra, (size_t)cfg->layers * (3 * H * H + 3 * H) * 4) != hipSuccess ||
        (x3 && hipMalloc(&e->planes, (size_t)cfg->layers * mat_per_layer * 2 * sizeof(_Float16)) != hipSuccess)) {
        set_error("icrec_encoder_create: hipMalloc failed");
        if (e->blob) (void)hipFree(e->blob);
        if (e->extra) (void)hipFree(e->extra);
        delete e;
        return ICREC_ENOMEM;
    }
    ICREC_HIP(hipMemcpy(e->blob, weights_host, n_floats * 4, hipMemcpyHostToDevice));
    float* p = e->blob;
    e->word = p; p += (size_t)cfg->vocab_size * H;
    e->pos = p;  p += (size_t)cfg->max_position * H;
    e->type = p; p += (size_t)cfg->type_vocab * H;
    e->eg = p;   p += H;
    e->eb = p;   p += H;
    float* x = e->extra;
    _Float16* pl = e->planes;
    auto pack = [&](const float* w, int N, int K, _Float16*& out) {  // [N, K] fp32 -> packed hi/lo fragments
        out = pl;
        pl += (size_t)2 * N * K;
        hipLaunchKernelGGL(pack_weights_kernel, dim3(1024), dim3(256), 0, 0, w, N, K, out);
    };
    for (int l = 0; l < cfg->layers; ++l) {
        LayerW& L = e->layers[l];
        L.Wqkv = x; x += 3 * H * H;
        L.bqkv = x; x += 3 * H;
        for (int part = 0; part < 3; ++part) {  // Wq,bq | Wk,bk | Wv,bv are interleaved in the blob
            ICREC_HIP(hipMemcpy(L.Wqkv + part * H * H, p, H * H * 4, hipMemcpyDeviceToDevice)); p += H * H;
            ICREC_HIP(hipMemcpy(L.bqkv + part * H, p, H * 4, hipMemcpyDeviceToDevice)); p += H;
        }
        L.Wo = p; p += H * H; L.bo = p; p += H;
        L.g1 = p; p += H; L.b1n = p; p += H;
        L.W1 = p; p += I * H; L.b1 = p; p += I;
        L.W2 = p; p += H * I; L.b2 = p; p += H;
        L.g2 = p; p += H; L.b2n = p; p += H;
        if (x3) {
            pack(L.Wqkv, (int)(3 * H), (int)H, L.Wqkv_p);
            pack(L.Wo, (int)H, (int)H, L.Wo_p);
            pack(L.W1, (int)I, (int)H, L.W1_p);
            pack(L.W2, (int)H, (int)I, L.W2_p);
        }
    }
    ICREC_HIP(hipGetLastError());
    ICREC_HIP(hipDeviceSynchronize());
    *out = reinterpret_cast<icrec_encoder*>(e);
    return ICREC_OK;
}

int icrec_encoder_destroy(icrec_encoder* h) {
    Encoder* e = reinterpret_cast<Encoder*>(h);
    if (!e) return ICREC_OK;
    (void)hipSetDevice(e->device);
    (void)hipFree(e->blob);
    (void)hipFree(e->extra);
    if (e->planes) (void)hipFree(e->planes);
    for (Encoder::Side* sd : e->sides) {
        if (sd->side) {
            (void)hipStreamSynchronize(sd->side);
            (void)hipStreamDestroy(sd->side);
        }
        for (hipEvent_t ev : {sd->ev_main, sd->ev_qkv_tail, sd->ev_att, sd->ev_tail, sd->ev_q, sd->ev_sa})
            if (ev) (void)hipEventDestroy(ev);
        delete sd;
    }
    delete e;
    return ICREC_OK;
}

size_t icrec_encode_workspace_bytes(const icrec_encoder* h, int64_t total_tokens, int32_t n_seqs) {
    const Encoder* e = reinterpret_cast<const Encoder*>(h);
    if (!e || total_tokens < 1 || n_seqs < 1) return 0;
    return enc_ws(e->cfg, total_tokens).total;
}

// How icrec_encode splits a batch of T tokens (f16x3 mode): [0, main) through the batch kernels in whole rounds of
// one 64-token workgroup per CU, [main, T) — a short remainder, or everything for small batches — through the
// small-batch kernels.
static void batch_split(const Encoder* e, int T, int* t_main, int* t_tail) {
    const int round_tokens = 64 * e->n_cu;
    *t_main = T;
    *t_tail = 0;
    if (T > round_tokens && T % round_tokens != 0 && T % round_tokens <= X3_SMALL_M) {
        *t_tail = T % round_tokens;
        *t_main = T - *t_tail;
    }
}

int icrec_encode_batch_split(const icrec_encoder* h, int64_t total_tokens, int64_t* main_tokens, int64_t* tail_tokens) {
    const Encoder* e = reinterpret_cast<const Encoder*>(h);
    ICREC_REQUIRE(e && main_tokens && tail_tokens && total_tokens >= 1 && total_tokens < (1ll << 31), "icrec_encode_batch_split: bad argument");
    int m, t;
    batch_split(e, (int)total_tokens, &m, &t);
    *main_tokens = m;
    *tail_tokens = t;
    return ICREC_OK;
}

int icrec_encode(icrec_encoder* h, const int32_t* ids_dev, const int32_t* cu_dev, int32_t n_seqs, int64_t T64,
                 int32_t max_seqlen, float* out_dev, void* ws, size_t ws_bytes, void* stream) {
    Encoder* e = reinterpret_cast<Encoder*>(h);
    ICREC_REQUIRE(e && ids_dev && cu_dev && out_dev, "icrec_encode: NULL argument");
    ICREC_REQUIRE(n_seqs >= 1 && T64 >= n_seqs && T64 < (1ll << 31), "icrec_encode: bad n_seqs/total_tokens (%d, %lld)", n_seqs, (long long)T64);
    ICREC_REQUIRE(max_seqlen >= 1 && max_seqlen <= 256 && max_seqlen <= e->cfg.max_position, "icrec_encode: max_seqlen must be in [1, 256] (got %d)", max_seqlen);
    const int T = (int)T64;
    const EncWs w = enc_ws(e->cfg, T);
    if (!ws || ws_bytes < w.total) {
        set_error("icrec_encode: workspace too small (%zu < %zu)", ws_bytes, w.total);
        return ICREC_ENOMEM;
    }
    ICREC_HIP(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    ScopedTimer whole(T_ENCODE, st);
    char* base = reinterpret_cast<char*>(ws);
    float* x = reinterpret_cast<float*>(base + w.x);
    float* qkv = reinterpret_cast<float*>(base + w.qkv);
    float* ctx = reinterpret_cast<float*>(base + w.ctx);
    float* t1 = reinterpret_cast<float*>(base + w.t1);
    float* hb = reinterpret_cast<float*>(base + w.h);
    const icrec_bert_cfg& c = e->cfg;
    const int H = c.hidden, I = c.intermediate;
    const int rows_grid = (T + 3) / 4;
    const bool x3 = c.gemm_mode == ICREC_GEMM_F16X3;
    // whole rounds of the fused FFN kernel (one 64-token workgroup per CU) + a short remainder, see the layer loop
    int T_main, T_tail;
    batch_split(e, T, &T_main, &T_tail);
    const char* fuse_env = getenv("ICREC_FUSE");  // ICREC_FUSE=0: A/B switch to the unfused kernels (read per call: tests flip it)
    const bool fuse = !(fuse_env && fuse_env[0] == '0');
    const char* side_env = getenv("ICREC_SIDE_STREAM");  // ICREC_SIDE_STREAM=0: the remainder's kernels stay on the caller's stream (A/B)
    const bool side_stream = !(side_env && side_env[0] == '0');
    const bool persist = fuse_env && fuse_env[0] == '3';
    const char* qr_env = getenv("ICREC_QKV_RESIDENT");  // ICREC_QKV_RESIDENT=0: A/B switch to the slab-ring QKV kernel (same bits)
    const bool qkv_res = !(qr_env && qr_env[0] == '0') && H == 384;
    const bool split_att = x3 && side_stream && n_seqs >= 64 && max_seqlen > 128;  // batches with a long bucket
    Encoder::Side* sd = nullptr;
    if (x3 && side_stream && (T_tail || split_att))
        if (int rc_ = side_for(e, st, &sd)) return rc_;
    const bool use_side = sd != nullptr;  // ICREC_FUSE=3: the persistent, block-pipelined fused FFN kernel (same bits, same speed: DESIGN.md 4.2)
    // f16 hi/lo planes (F16X3): x, ctx and h; ctx/h planes alias the fp32 regions they replace
    _Float16* xh = reinterpret_cast<_Float16*>(base + w.xs);
    _Float16* xl = xh + (size_t)T * H;
    _Float16* ch = reinterpret_cast<_Float16*>(ctx);
    _Float16* cl = ch + (size_t)T * H;
    _Float16* hh = reinterpret_cast<_Float16*>(hb);
    _Float16* hl = hh + (size_t)T * I;

    // f16x3 mode: the fp32 x region of the workspace is unused (the residual stream is its two planes): it carries the
    // attention dispatch order of batches (ICREC_ATT_ORDER=0: workgroup b serves sequence b / heads, A/B)
    const int32_t* order = nullptr;
    {
        const char* ao = getenv("ICREC_ATT_ORDER");
        if (x3 && n_seqs >= 64 && !(ao && ao[0] == '0')) {
            int32_t* ord = reinterpret_cast<int32_t*>(x);
            hipLaunchKernelGGL(seq_order_kernel, dim3(1), dim3(1024), 0, st, cu_dev, n_seqs, ord);
            order = ord;
        }
    }
    if (x3)
        hipLaunchKernelGGL((embed_ln_kernel<HID, true>), dim3(rows_grid), dim3(256), 0, st, ids_dev, cu_dev, n_seqs, T,
                           e->word, e->pos, e->type, e->eg, e->eb, c.ln_eps, c.vocab_size, c.max_position, x, xh, xl);
    else
        hipLaunchKernelGGL((embed_ln_kernel<HID, false>), dim3(rows_grid), dim3(256), 0, st, ids_dev, cu_dev, n_seqs, T,
                           e->word, e->pos, e->type, e->eg, e->eb, c.ln_eps, c.vocab_size, c.max_position, x, xh, xl);
    for (int l = 0; l < c.layers; ++l) {
        const LayerW& L = e->layers[l];
        if (x3) {
            // Token ranges: [0, T_main) goes through the batch kernels in whole rounds of one 64-token workgroup per
            // CU, a short remainder [T_main, T) through the small-batch kernels (same arithmetic, same bits) instead
            // of costing every batch kernel an extra, almost empty round.
            auto qkv_stage = [&](int r0, int Tn, hipStream_t st) -> int {
                if (Tn > X3_SMALL_M && qkv_res) {  // activation-resident form: one 64-token workgroup per CU
                    auto kern = qkv_resident_kernel;
                    if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), QKVR_LDS)) return rc_;
                    hipLaunchKernelGGL(kern, dim3((Tn + 63) / 64), dim3(512), QKVR_LDS, st, xh + (size_t)r0 * H,
                                       xl + (size_t)r0 * H, Tn, L.Wqkv_p, L.bqkv, qkv + (size_t)r0 * 3 * H, 3 * H);
                    return ICREC_OK;
                }
                launch_wt_linear<0>(xh + (size_t)r0 * H, xl + (size_t)r0 * H, Tn, H, L.Wqkv_p, 3 * H, L.bqkv,
                                    qkv + (size_t)r0 * 3 * H, nullptr, nullptr, st);
                return ICREC_OK;
            };
            auto post_stage = [&](int r0, int Tn, hipStream_t st) -> int {
                float* const t1r = t1 + (size_t)r0 * H;
                _Float16 *const xhr = xh + (size_t)r0 * H, *const xlr = xl + (size_t)r0 * H;
                const _Float16 *const chr = ch + (size_t)r0 * H, *const clr = cl + (size_t)r0 * H;
                if (Tn > X3_SMALL_M && fuse) {
                    // attention-out + residual + LN, then the whole FFN block + residual + LN: two kernels per half layer
                    hipLaunchKernelGGL((wt_linear_ln_kernel<2>), dim3((Tn + 63) / 64), dim3(256), 0, st, chr, clr, Tn, H,
                                       L.Wo_p, L.bo, xhr, xlr, L.g1, L.b1n, c.ln_eps);
                    ScopedTimer tm(T_FFN_UP, st);
                    const int nblk = (Tn + 63) / 64;
                    if (persist) {
                        auto kern = ffn_fused3_kernel;
                        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), FFN2_LDS)) return rc_;
                        hipLaunchKernelGGL(kern, dim3(nblk < e->n_cu ? nblk : e->n_cu), dim3(512), FFN2_LDS, st, xhr, xlr, Tn,
                                           I, L.W1_p, L.b1, L.W2_p, L.b2, L.g2, L.b2n, c.ln_eps);
                    } else {
                        auto kern = ffn_fused2_kernel<0>;
                        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), FFN2_LDS)) return rc_;
                        hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), FFN2_LDS, st, xhr, xlr, Tn, I, L.W1_p, L.b1, L.W2_p,
                                           L.b2, L.g2, L.b2n, c.ln_eps);
                    }
                } else {
                    _Float16 *const hhr = hh + (size_t)r0 * I, *const hlr = hl + (size_t)r0 * I;
                    launch_wt_linear<2>(chr, clr, Tn, H, L.Wo_p, H, L.bo, t1r, xhr, xlr, st);  // residual: x planes
                    hipLaunchKernelGGL(ln_wt_kernel, dim3((Tn + 31) / 32), dim3(256), 0, st, t1r, Tn, L.g1, L.b1n,
                                       c.ln_eps, xhr, xlr);
                    {
                        ScopedTimer tm(Tn > X3_SMALL_M ? T_FFN_UP : T_NSLOTS - 1, st);
                        launch_wt_linear<1>(xhr, xlr, Tn, H, L.W1_p, I, L.b1, nullptr, hhr, hlr, st);
                    }
                    launch_wt_linear<2>(hhr, hlr, Tn, I, L.W2_p, H, L.b2, t1r, xhr, xlr, st);
                    hipLaunchKernelGGL(ln_wt_kernel, dim3((Tn + 31) / 32), dim3(256), 0, st, t1r, Tn, L.g2, L.b2n,
                                       c.ln_eps, xhr, xlr);
                }
                return ICREC_OK;
            };
            // The remainder's kernels run on the side stream: its QKV beside the batch QKV, its attention-out / FFN chain
            // beside the batch's.  Attention covers all rows, so it joins both (ev_qkv_tail in, ev_att out); the side
            // stream's in-order execution keeps its own layers apart.
            hipStream_t ts = st;
            if (T_tail && use_side) {
                ts = sd->side;
                if (l == 0) {  // the side stream starts behind the embeddings
                    ICREC_HIP(hipEventRecord(sd->ev_main, st));
                    ICREC_HIP(hipStreamWaitEvent(ts, sd->ev_main, 0));
                }
                if (int rc_ = qkv_stage(T_main, T_tail, ts)) return rc_;
                ICREC_HIP(hipEventRecord(sd->ev_qkv_tail, ts));
            } else if (T_tail) {
                if (int rc_ = qkv_stage(T_main, T_tail, st)) return rc_;
            }
            if (int rc_ = qkv_stage(0, T_main, st)) return rc_;
            if (T_tail && use_side) ICREC_HIP(hipStreamWaitEvent(st, sd->ev_qkv_tail, 0));
            if (split_att && use_side) {
                // the long bucket keeps one 8-wave workgroup per CU busy (LDS) with issue slots to spare: the shorter
                // buckets' workgroups run beside it from the side stream instead of after it
                ICREC_HIP(hipEventRecord(sd->ev_q, st));
                ICREC_HIP(hipStreamWaitEvent(sd->side, sd->ev_q, 0));
                launch_attention<true, true>(qkv, cu_dev, n_seqs, c.heads, H, max_seqlen, ctx, ch, cl, sd->side, 7, order);
                ICREC_HIP(hipEventRecord(sd->ev_sa, sd->side));
                launch_attention<true, true>(qkv, cu_dev, n_seqs, c.heads, H, max_seqlen, ctx, ch, cl, st, 8, order);
                ICREC_HIP(hipStreamWaitEvent(st, sd->ev_sa, 0));
            } else {
                launch_attention<true, true>(qkv, cu_dev, n_seqs, c.heads, H, max_seqlen, ctx, ch, cl, st, 15, order);
            }
            if (T_tail) {
                if (use_side) {
                    ICREC_HIP(hipEventRecord(sd->ev_att, st));
                    ICREC_HIP(hipStreamWaitEvent(ts, sd->ev_att, 0));
                }
                if (int rc_ = post_stage(T_main, T_tail, ts)) return rc_;
            }
            if (int rc_ = post_stage(0, T_main, st)) return rc_;
            if (T_tail && use_side && l + 1 == c.layers) {  // pooling reads every row: the side stream joins here
                ICREC_HIP(hipEventRecord(sd->ev_tail, ts));
                ICREC_HIP(hipStreamWaitEvent(st, sd->ev_tail, 0));
            }
        } else {
            launch_linear<false>(x, T, H, L.Wqkv, 3 * H, L.bqkv, qkv, st);
            launch_attention<false, false>(qkv, cu_dev, n_seqs, c.heads, H, max_seqlen, ctx, ch, cl, st);
            launch_linear<false>(ctx, T, H, L.Wo, H, L.bo, t1, st);
            hipLaunchKernelGGL((add_ln_kernel<HID, false>), dim3(rows_grid), dim3(256), 0, st, t1, x, T, L.g1, L.b1n,
                               c.ln_eps, xh, xl);
            {
                ScopedTimer tm(T_FFN_UP, st);
                launch_linear<true>(x, T, H, L.W1, I, L.b1, hb, st);
            }
            launch_linear<false>(hb, T, I, L.W2, H, L.b2, t1, st);
            hipLaunchKernelGGL((add_ln_kernel<HID, false>), dim3(rows_grid), dim3(256), 0, st, t1, x, T, L.g2, L.b2n,
                               c.ln_eps, xh, xl);
        }
    }
    if (x3) hipLaunchKernelGGL((pool_norm_kernel<HID, true>), dim3(n_seqs), dim3(HID), 0, st, x, xh, xl, cu_dev, c.n_normalize, out_dev);
    else hipLaunchKernelGGL((pool_norm_kernel<HID, false>), dim3(n_seqs), dim3(HID), 0, st, x, xh, xl, cu_dev, c.n_normalize, out_dev);
    ICREC_HIP(hipGetLastError());
    return ICREC_OK;
}

}  // extern "C"
