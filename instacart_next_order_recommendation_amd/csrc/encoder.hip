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
#include "encoder_x3.h"

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
    for (int s0 = tid; s0 < n_seqs; s0 += 1024) {
        const int s = n_seqs - 1 - s0;
        int L = cu[s + 1] - cu[s];
        L = L < 1 ? 1 : (L > 256 ? 256 : L);
        atomicAdd(&hist[256 - ((L + 31) & ~31)], 1);
    }
    __syncthreads();
    if (tid == 0) {  // exclusive prefix sum: hist[b] becomes the first slot of bin b
        int run = 0;
        for (int b = 0; b < 257; ++b) { const int c = hist[b]; hist[b] = run; run += c; }
    }
    __syncthreads();
    for (int s0 = tid; s0 < n_seqs; s0 += 1024) {
        const int s = n_seqs - 1 - s0;
        int L = cu[s + 1] - cu[s];
        L = L < 1 ? 1 : (L > 256 ? 256 : L);
        order[atomicAdd(&hist[256 - ((L + 31) & ~31)], 1)] = s;
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

// ds_read_b64_tr_b16 (gfx950): all 64 lanes must be active; `p` is this lane's Mechanism address (8-byte aligned)
__device__ __forceinline__ half4 lds_read_tr(const _Float16* p) {
    typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    return __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                                         (__attribute__((address_space(3))) fp16x4*)(reinterpret_cast<uintptr_t>(p))));
}

// nlo: sequences of nlo < nkt <= NKT key tiles belong to this launch (the others' workgroups exit at once)
template <int NKT, int WAVES, bool SPLIT>
__global__ __launch_bounds__(WAVES * 64, (NKT >= 6 ? 4 : NKT == 4 ? 3 : 1)) void attention_x3_kernel(const float* __restrict__ qkv,
                                                                  const int32_t* __restrict__ cu, int heads, int H,
                                                                  float scale_log2e, float* __restrict__ ctx,
                                                                  _Float16* __restrict__ ch, _Float16* __restrict__ cl,
                                                                  const int32_t* __restrict__ order, int nlo) {
    // Single-accumulator form of the split (wt_gemm.h): every operand is carried as hi/lo f16 planes of 16 x (Q, K, V)
    // or 1024 p (the probabilities), the three products of a k-step accumulate into ONE fp32 tile, and the power-of-two
    // scales are folded into constants: S' = 256 S, O' = 16384 sum_k p_k V_k, l' = 1024 sum_k p_k, O = O' / (16 l').
    // All splits are the 3-instruction form (mask / subtract / v_cvt_pkrtz pairs).
    //
    // Long buckets (NKT >= 6, RECOMP): the score tiles are computed TWICE - once for the row maximum (on the hi x hi
    // products alone, see score_tile), once more in full for the exponentials, each tile consumed by the PV product as
    // soon as it exists - instead of all of them being held in 128 registers between the two passes: 16 more MFMAs per
    // wave, and the kernel drops under 128 VGPRs; with the output tile parked on the K
    // planes (behind one more barrier) it also drops to 67 KB of LDS - TWO workgroups per CU, so one's staging and
    // barrier phases run under the other's arithmetic.
    constexpr bool RECOMP = NKT >= 6;
    __shared__ __attribute__((aligned(16))) _Float16 Kbuf[2 * NKT * 32 * 32];
    _Float16* const Kh = Kbuf;
    _Float16* const Kl = Kbuf + NKT * 32 * 32;
    // V as it arrives: row-major hi / lo planes [key][32 dims] (64-B rows, one 8-byte store per thread and plane where the
    // transposed image took four 2-byte ones); the P.V product reads its B operand - 4 consecutive keys of one head
    // dimension per lane - with the transposing LDS read (ds_read_b64_tr_b16: per 16-lane group a block of 4 keys x 16
    // dims, lane 4q + p supplying the address of key q, dims 4p..4p+3, lane i receiving dim i of the 4 keys; a 32-lane
    // half covers 4 rows of 64 B = every bank once).
    __shared__ __attribute__((aligned(16))) _Float16 Vh[NKT * 32 * 32];
    __shared__ __attribute__((aligned(16))) _Float16 Vl[NKT * 32 * 32];
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
    if (nkt > NKT || nkt <= nlo) return;  // another bucket's sequence
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
    ICREC_STAMP(0, 6);  // Q rows arrived and split
    // K/V staging: all of this thread's loads are issued before the first one is consumed
    constexpr int STG = NKT * 32 * 8 / (WAVES * 64);  // = 4 for every bucket (WAVES == NKT)
    static_assert(NKT * 32 * 8 % (WAVES * 64) == 0, "staging: whole rounds");
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
#ifdef ICREC_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ICREC_STAMP(0, 7);  // K / V rows arrived
#endif
#pragma unroll
    for (int it = 0; it < STG; ++it) {
        const int id = tid + it * WAVES * 64;
        if (id < nkt * 32 * 8) {
            const int key = id >> 3, c = id & 7;
            half4 khi, klo, vhi, vlo;
            split_act4(kreg[it], khi, klo);
            split_act4(vreg[it], vhi, vlo);
            *reinterpret_cast<half4*>(Vh + key * 32 + c * 4) = vhi;
            *reinterpret_cast<half4*>(Vl + key * 32 + c * 4) = vlo;
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
    // hi_only (the row-maxima pass of the long buckets): the hi x hi products alone - 2 MFMAs per tile instead of 6.  The
    // softmax does not care which shift it is given as long as the exponentials stay in range; this approximate maximum
    // can sit below the true one by at most 2^-10 |q| |k| in score units, the planes of p' = 2^10 p have 2^6 of head room,
    // and the exponent is clamped for whatever lies beyond (|q| |k| > 2.4e4: two hundred times a BERT head's).
    auto score_tile = [&](int kt, bool hi_only = false) {
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
            if (hi_only) continue;
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
#if defined(ICREC_ATT_ABL) && (ICREC_ATT_ABL & 1)  // timing ablation (tools/ffn_bench.hip): no lo plane of P
                a = __builtin_bit_cast(half2w, __builtin_amdgcn_cvt_pkrtz(pt[8 * ks + j], pt[8 * ks + j + 1]));
                b = a;
#else
                split_pair_prescaled(pt[8 * ks + j], pt[8 * ks + j + 1], a, b);
#endif
                ph[j] = a[0]; ph[j + 1] = a[1];
                pl[j] = b[0]; pl[j + 1] = b[1];
            }
            // this lane's head dim r, keys base .. base+3 and base+8 .. base+11: two transposed 4-key x 16-dim blocks per plane
            const int base = kt * 32 + 4 * h + 16 * ks;
            const int tr_at = (base + ((lane & 15) >> 2)) * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            const half4 v0h = lds_read_tr(Vh + tr_at);
            const half4 v1h = lds_read_tr(Vh + tr_at + 8 * 32);
            const half4 v0l = lds_read_tr(Vl + tr_at);
            const half4 v1l = lds_read_tr(Vl + tr_at + 8 * 32);
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
                    const f32x16 t = score_tile(kt, true);
#pragma unroll
                    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, t[e]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            ICREC_STAMP(0, 9);  // first score pass (row maxima) done (slot 8 holds the HW id)
            const float shift = fmaf(-mx, cs, 10.0f);
            float2w ls2 = float2w{0.0f, 0.0f};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt < nkt) {
                    f32x16 t = score_tile(kt);
#pragma unroll
#if defined(ICREC_ATT_ABL) && (ICREC_ATT_ABL & 2)  // timing ablation: no exponential
                    for (int e = 0; e < 16; ++e) t[e] = fmaf(t[e], cs, shift);
#else
                    for (int e = 0; e < 16; ++e) t[e] = __builtin_amdgcn_exp2f(fminf(fmaf(t[e], cs, shift), 15.9f));
#endif
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
    // A/B switches, read from the environment ONCE, at icrec_encoder_create (never on the hot path; a test that wants
    // the other form creates a second encoder under the other setting):
    //   ICREC_FUSE=0         the UNFUSED reference chain for batches: slab-ring QKV, attention-out GEMM + LayerNorm,
    //                        FFN-up, FFN-down + LayerNorm as separate launches in natural sequence order (same bits)
    //   ICREC_SIDE_STREAM=0  every kernel on the caller's stream (no side stream for the batch remainder / short buckets)
    bool fuse = true, side_stream = true;
    //   ICREC_SMALL_M=n      token count up to which a call takes the latency-form kernels (32-token x 64-feature
    //                        workgroups, every GEMM a launch of its own) instead of the layer kernel (one 64-token workgroup
    //                        per CU).  Round 4: 3,584 - measured crossover ~4,000 tokens (tools/small_m_sweep.py: 897 tokens
    //                        0.39 ms against 0.79, 2,900 tokens 0.74 against 0.86, 4,485 tokens 0.86 against 0.82); rounds 1-3: 512
    int small_m = 3584;
    //   ICREC_TAIL_M=n       a batch's remainder (tokens beyond whole rounds of one 64-token workgroup per CU) of up to n tokens
    //                        goes through the latency-form kernels on the side stream instead of a partial round (which
    //                        costs every layer kernel ~0.1 ms: +0.57 ms per step).  Round 4: 2,560 - measured on 1,030- /
    //                        1,040- / 1,050-context batches (remainders 650 / 1,839 / 3,117 tokens): 9.83 -> 9.35 ms, 9.86 -> 9.60,
    //                        9.89 -> 9.91 (tools/tail_sweep.sh, profiles/r04_small_m_sweep.txt); rounds 2-3: 512
    int tail_m = 2560;
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
constexpr int X3_SMALL_M = 512;  // a batch's remainder of up to this many tokens goes through the latency-form kernels (batch_split)

template <bool GELU>
static void launch_linear(const float* A, int M, int K, const float* W, int N, const float* bias, float* out,
                          hipStream_t st) {
    const int mt = (M + GemmBig::BM - 1) / GemmBig::BM, nt = (N + GemmBig::BN - 1) / GemmBig::BN;
    hipLaunchKernelGGL((linear_kernel<GemmBig, GELU>), dim3(mt * nt), dim3(GemmBig::THREADS), 0, st, A, M, K, W, N,
                       bias, out, nt);
}

// f16x3 linear layer through the weights-direct engine.  Single requests / micro-batches (up to small_m tokens) are
// latency-bound - a handful of workgroups, each walking its K loop at the rate ONE CU's vector L1 pulls fragments from
// L2: they use 32-token x 64-feature workgroups, one 16-feature tile per wave (wt_linear_half_kernel; EPI 1 keeps the
// 32 x 128 form: its small-batch consumer is wt_linear_lnin_kernel), as many workgroups as the shape allows; batches use
// 64-token x 384-feature blocks (3 x 2 tiles per wave).  Per-output arithmetic is the same chain in all of them, so a
// request encodes to the same bits either way.
template <int EPI>
static void launch_wt_linear(const _Float16* Xh, const _Float16* Xl, int T, int K, const _Float16* Wp, int N,
                             const float* bias, float* out, _Float16* oh, _Float16* ol, hipStream_t st, int small_m) {
    if (T <= small_m) {
        if constexpr (EPI == 0 || EPI == 2) {  // 16-feature tiles per wave, twice the workgroups (wt_linear_half_kernel)
            if (N % 64 == 0 && K % 128 == 0) {
                const int nbn = N / 64;
                if (K % 256 == 0 && K >= 1024)  // FFN-down: weights eight k-steps ahead
                    hipLaunchKernelGGL((wt_linear_half_kernel<EPI, 8>), dim3(((T + 31) / 32) * nbn), dim3(256), 0, st, Xh, Xl, T, K, Wp,
                                       N, bias, out, (const _Float16*)oh, (const _Float16*)ol, nbn);
                else if (K % 384 == 0)  // K = 384: six of the twelve k-steps ahead
                    hipLaunchKernelGGL((wt_linear_half_kernel<EPI, 6>), dim3(((T + 31) / 32) * nbn), dim3(256), 0, st, Xh, Xl, T, K, Wp,
                                       N, bias, out, (const _Float16*)oh, (const _Float16*)ol, nbn);
                else
                hipLaunchKernelGGL((wt_linear_half_kernel<EPI>), dim3(((T + 31) / 32) * nbn), dim3(256), 0, st, Xh, Xl, T, K, Wp, N,
                                   bias, out, (const _Float16*)oh, (const _Float16*)ol, nbn);
                return;
            }
        }
        const int nbn = N / 128;
        hipLaunchKernelGGL((wt_linear_kernel<1, 1, 4, EPI>), dim3(((T + 31) / 32) * nbn), dim3(256), 0, st, Xh, Xl, T, K,
                           Wp, N, bias, out, oh, ol, nbn);
    } else {
        const int nbn = N / 384;
        hipLaunchKernelGGL((wt_linear_kernel<3, 2, 1, EPI>), dim3(((T + 63) / 64) * nbn), dim3(256), 0, st, Xh, Xl, T, K,
                           Wp, N, bias, out, oh, ol, nbn);
    }
}

// Launch every length bucket that can occur for max_seqlen (a bucket whose workgroups all exit
// costs a few microseconds; single-sequence calls launch exactly one bucket).
template <bool SPLIT, bool X3>
static void launch_attention(const float* qkv, const int32_t* cu, int n_seqs, int heads, int H, int max_seqlen,
                             float* ctx, _Float16* ch, _Float16* cl, hipStream_t st, int buckets = 31,
                             const int32_t* order = nullptr) {
    // buckets: bits 0..2 = the 1-, 2-, 3-4-tile buckets, bit 3 = 5-6 tiles (5-8 outside f16x3 batches), bit 4 = 7-8 tiles
    // (callers split the buckets over two streams)
    const float sl2e = (1.0f / sqrtf((float)DH)) * 1.44269504088896340736f;
    const int nkt_max = (max_seqlen + 31) / 32;
    const bool single = n_seqs == 1;
    const dim3 grid1(n_seqs * heads, 1);
#define ICREC_ATT(NKT, W, NLO)                                                                                   \
    do {                                                                                                         \
        if (X3) hipLaunchKernelGGL((attention_x3_kernel<NKT, W, SPLIT>), grid1, dim3(W * 64), 0, st, qkv, cu, heads, H, sl2e, ctx, ch, cl, order, NLO); \
        else hipLaunchKernelGGL((attention_kernel<NKT, W, SPLIT>), grid1, dim3(W * 64), 0, st, qkv, cu, heads, H, sl2e, ctx, ch, cl);      \
    } while (0)
    if ((buckets & 1) && (single ? nkt_max == 1 : true)) ICREC_ATT(1, 1, 0);
    if ((buckets & 2) && (single ? nkt_max == 2 : nkt_max >= 2)) ICREC_ATT(2, 2, 1);
    if ((buckets & 4) && (single ? (nkt_max == 3 || nkt_max == 4) : nkt_max >= 3)) ICREC_ATT(4, 4, 2);
    if (X3 && !single) {  // f16x3 batches: the long sequences in two buckets (5-6 and 7-8 key tiles)
        if ((buckets & 8) && nkt_max >= 5) ICREC_ATT(6, 6, 4);
        if ((buckets & 16) && nkt_max >= 7) ICREC_ATT(8, 8, 6);
    } else if ((buckets & 8) && nkt_max >= 5) {
        ICREC_ATT(8, 8, 4);
    }
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
    {
        const char* fuse_env = getenv("ICREC_FUSE");
        const char* side_env = getenv("ICREC_SIDE_STREAM");
        e->fuse = !(fuse_env && fuse_env[0] == '0');
        if (const char* sm = getenv("ICREC_SMALL_M")) { const int v = atoi(sm); if (v >= 0) e->small_m = v; }
        if (const char* tm = getenv("ICREC_TAIL_M")) { const int v = atoi(tm); if (v >= 0) e->tail_m = v; }
        e->side_stream = !(side_env && side_env[0] == '0');
    }
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
    if (T > round_tokens && T % round_tokens != 0 && T % round_tokens <= e->tail_m) {
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
    const bool fuse = e->fuse, side_stream = e->side_stream;
    const int small_m = e->small_m;
    const bool qkv_res = fuse && H == 384;
    const bool lnin = fuse && H == 384 && I % 64 == 0;  // small ranges: LayerNorms folded into the consuming GEMMs
    const bool split_att = x3 && side_stream && n_seqs >= 64 && max_seqlen > 128;  // batches with a long bucket
    Encoder::Side* sd = nullptr;
    if (x3 && side_stream && (T_tail || split_att))
        if (int rc_ = side_for(e, st, &sd)) return rc_;
    const bool use_side = sd != nullptr;
    // f16 hi/lo planes (F16X3): x, ctx and h; ctx/h planes alias the fp32 regions they replace
    _Float16* xh = reinterpret_cast<_Float16*>(base + w.xs);
    _Float16* xl = xh + (size_t)T * H;
    _Float16* ch = reinterpret_cast<_Float16*>(ctx);
    _Float16* cl = ch + (size_t)T * H;
    _Float16* hh = reinterpret_cast<_Float16*>(hb);
    _Float16* hl = hh + (size_t)T * I;

    // f16x3 mode: the fp32 x region of the workspace is unused (the residual stream is its two planes): it carries the
    // attention dispatch order of batches (the unfused reference chain keeps the natural order: workgroup b serves
    // sequence b / heads)
    const int32_t* order = nullptr;
    if (x3 && n_seqs >= 64 && fuse) {
        int32_t* ord = reinterpret_cast<int32_t*>(x);
        hipLaunchKernelGGL(seq_order_kernel, dim3(1), dim3(1024), 0, st, cu_dev, n_seqs, ord);
        order = ord;
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
                if (Tn > small_m && qkv_res) {  // activation-resident form: one 64-token workgroup per CU
                    auto kern = qkv_resident_kernel;
                    if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), QKVR_LDS)) return rc_;
                    hipLaunchKernelGGL(kern, dim3((Tn + 63) / 64), dim3(512), QKVR_LDS, st, xh + (size_t)r0 * H,
                                       xl + (size_t)r0 * H, Tn, L.Wqkv_p, L.bqkv, qkv + (size_t)r0 * 3 * H, 3 * H);
                    return ICREC_OK;
                }
                if (lnin && l > 0 && Tn <= small_m) {
                    // small ranges: the previous layer's FFN LayerNorm is this kernel's prologue (t1 rows -> planes in LDS
                    // and, from the workgroups of feature block 0, to xh / xl): one graph node fewer per layer
                    const LayerW& Lp = e->layers[l - 1];
                    const int nbn = 3 * H / 64;
                    hipLaunchKernelGGL((wt_linear_lnin_kernel<0, true>), dim3(((Tn + 31) / 32) * nbn), dim3(256), 0, st,
                                       (const float*)(t1 + (size_t)r0 * H), Tn, Lp.g2, Lp.b2n, c.ln_eps, xh + (size_t)r0 * H,
                                       xl + (size_t)r0 * H, L.Wqkv_p, 3 * H, L.bqkv, qkv + (size_t)r0 * 3 * H,
                                       (_Float16*)nullptr, (_Float16*)nullptr, nbn);
                    return ICREC_OK;
                }
                launch_wt_linear<0>(xh + (size_t)r0 * H, xl + (size_t)r0 * H, Tn, H, L.Wqkv_p, 3 * H, L.bqkv,
                                    qkv + (size_t)r0 * 3 * H, nullptr, nullptr, st, small_m);
                return ICREC_OK;
            };
            auto post_stage = [&](int r0, int Tn, hipStream_t st) -> int {
                float* const t1r = t1 + (size_t)r0 * H;
                _Float16 *const xhr = xh + (size_t)r0 * H, *const xlr = xl + (size_t)r0 * H;
                const _Float16 *const chr = ch + (size_t)r0 * H, *const clr = cl + (size_t)r0 * H;
                if (Tn > small_m && fuse) {
                    // attention-out + residual + LN and the whole FFN block + residual + LN: ONE kernel per half layer
                    // (x1 stays on chip between the two LayerNorm sites)
                    ScopedTimer tm(T_FFN_UP, st);
                    const int nblk = (Tn + 63) / 64;
                    auto kern = ffn_fused2_kernel<0, true>;
                    if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), FFN2_LDS)) return rc_;
                    // ... and, but for the last layer, the NEXT layer's QKV projection of the rows it has just normalised
                    const bool next_qkv = l + 1 < c.layers;
                    const LayerW& Ln = e->layers[next_qkv ? l + 1 : l];
                    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), FFN2_LDS, st, xhr, xlr, Tn, I, L.W1_p, L.b1, L.W2_p,
                                       L.b2, L.g2, L.b2n, c.ln_eps, chr, clr, L.Wo_p, L.bo, L.g1, L.b1n,
                                       next_qkv ? (const _Float16*)Ln.Wqkv_p : (const _Float16*)nullptr, (const float*)Ln.bqkv,
                                       qkv + (size_t)r0 * 3 * H, 3 * H);
                } else {
                    _Float16 *const hhr = hh + (size_t)r0 * I, *const hlr = hl + (size_t)r0 * I;
                    launch_wt_linear<2>(chr, clr, Tn, H, L.Wo_p, H, L.bo, t1r, xhr, xlr, st, small_m);  // residual: x planes
                    if (lnin && Tn <= small_m) {  // LayerNorm + FFN-up in one node (wt_linear_lnin_kernel)
                        const int nbn = I / 64;
                        hipLaunchKernelGGL((wt_linear_lnin_kernel<1, true>), dim3(((Tn + 31) / 32) * nbn), dim3(256), 0, st,
                                           (const float*)t1r, Tn, L.g1, L.b1n, c.ln_eps, xhr, xlr, L.W1_p, I, L.b1,
                                           (float*)nullptr, hhr, hlr, nbn);
                    } else {
                    hipLaunchKernelGGL(ln_wt_kernel, dim3((Tn + 15) / 16), dim3(256), 0, st, t1r, Tn, L.g1, L.b1n,
                                       c.ln_eps, xhr, xlr);
                    {
                        ScopedTimer tm(Tn > small_m ? T_FFN_UP : T_NSLOTS - 1, st);
                        launch_wt_linear<1>(xhr, xlr, Tn, H, L.W1_p, I, L.b1, nullptr, hhr, hlr, st, small_m);
                    }
                    }
                    launch_wt_linear<2>(hhr, hlr, Tn, I, L.W2_p, H, L.b2, t1r, xhr, xlr, st, small_m);
                    // the FFN LayerNorm: the prologue of the next layer's QKV projection (qkv_stage) - but for the last layer
                    if (!(lnin && Tn <= small_m && l + 1 < c.layers))
                    hipLaunchKernelGGL(ln_wt_kernel, dim3((Tn + 15) / 16), dim3(256), 0, st, t1r, Tn, L.g2, L.b2n,
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
            // batches: layer 0 projects Q / K / V in a launch of its own; every later layer's projection is the epilogue of
            // the previous layer's fused kernel
            const bool qkv_in_fused = fuse && T_main > small_m && H == 384;
            if (l == 0 || !qkv_in_fused)
                if (int rc_ = qkv_stage(0, T_main, st)) return rc_;
            if (T_tail && use_side) ICREC_HIP(hipStreamWaitEvent(st, sd->ev_qkv_tail, 0));
            if (split_att && use_side) {
                // the long bucket keeps one 8-wave workgroup per CU busy (LDS) with issue slots to spare: the shorter
                // buckets' workgroups run beside it from the side stream instead of after it
                ICREC_HIP(hipEventRecord(sd->ev_q, st));
                ICREC_HIP(hipStreamWaitEvent(sd->side, sd->ev_q, 0));
                launch_attention<true, true>(qkv, cu_dev, n_seqs, c.heads, H, max_seqlen, ctx, ch, cl, sd->side, 7 | 16, order);
                ICREC_HIP(hipEventRecord(sd->ev_sa, sd->side));
                launch_attention<true, true>(qkv, cu_dev, n_seqs, c.heads, H, max_seqlen, ctx, ch, cl, st, 8, order);
                ICREC_HIP(hipStreamWaitEvent(st, sd->ev_sa, 0));
            } else {
                launch_attention<true, true>(qkv, cu_dev, n_seqs, c.heads, H, max_seqlen, ctx, ch, cl, st, 31, order);
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
