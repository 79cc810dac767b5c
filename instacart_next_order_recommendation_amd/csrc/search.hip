// search.hip — cosine scoring fused with per-query top-k selection (no score matrix is
// ever written), the k-way merge of sorted partial lists, and row L2-normalisation.
//
// Replaces, on the device:
//   sentence_transformers.util.cos_sim(query_emb, product_embeddings)  serve_recommendations.py:214/:250
//   scores.argsort(descending=True)                                      :215/:251
//   the exclusion / top-k Python loop                                    :216-225/:254-262
// of /root/reference/src/inference/serve_recommendations.py.
#include "common.h"
#include "gemm_x3.h"
#include "wt_gemm.h"

namespace icrec {

// ---------------------------------------------------------------- row normalisation
// out = x / max(|x|_2, eps) — torch.nn.functional.normalize(p=2, dim=1) as cos_sim applies it.
// One wavefront per row; reduction order = 64 strided fmaf partials + xor butterfly, identical
// to oracle/icrec_oracle.c:wave_sum(mode 2).
// float -> bfloat16 bits, round to nearest even (inputs are finite: normalised rows)
__device__ __forceinline__ uint16_t bf16_rne(float v) {
    const unsigned u = __float_as_uint(v);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// tr_cols > 0 (fp32 only): write the result transposed, out[i * tr_cols + row] (n_out_rows == tr_cols) — the
// k-major query layout of stream_search_kernel.
template <bool OUT16>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, void* __restrict__ outv,
                                                             int64_t n_rows, int64_t n_out_rows, int dim, float eps,
                                                             int tr_cols = 0) {
    float* out = static_cast<float*>(outv);
    uint16_t* out16 = static_cast<uint16_t*>(outv);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_out_rows) return;
    if (row >= n_rows) {  // zero padding rows (query tiles are padded to the tile width)
        for (int i = lane; i < dim; i += 64) {
            if (OUT16) out16[row * dim + i] = 0;
            else if (tr_cols > 0) out[(int64_t)i * tr_cols + row] = 0.0f;
            else out[row * dim + i] = 0.0f;
        }
        return;
    }
    const float* xr = x + row * dim;
    float acc = 0.0f;
    for (int i = lane; i < dim; i += 64) {
        float v = xr[i];
        acc = fmaf(v, v, acc);
    }
    float nrm = sqrtf(wave_sum_f32(acc));
    float den = nrm > eps ? nrm : eps;
    for (int i = lane; i < dim; i += 64) {
        const float v = xr[i] / den;
        if (OUT16) out16[row * dim + i] = bf16_rne(v);
        else if (tr_cols > 0) out[(int64_t)i * tr_cols + row] = v;
        else out[row * dim + i] = v;
    }
}

// bf16 rows widened back to fp32 (icrec_index_export)
__global__ __launch_bounds__(256) void widen_bf16_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __uint_as_float((unsigned)in[i] << 16);
}

// ---------------------------------------------------------------- score + select
// candidate queue slots per query between two list merges (more when few queries share the LDS)
// Candidate queue entries per query: 64 for 32-query tiles, 16 for wider ones.  The resident filter pass takes 32 when
// the lists are short (k <= 32) and a block walks few rounds (res_qcap): the first rounds of a block offer 16-48
// candidates per query and would otherwise take two or three offer / merge iterations each - at 49,688 rows x 1,024
// queries those rounds are most of the kernel (0.274 -> 0.229 ms, same box); blocks that walk thousands of rounds
// keep 16 (10 M rows: 1 % faster with it).
template <class Cfg> struct QCap { static constexpr int V = Cfg::BN <= 32 ? 64 : 16; };
constexpr int RES_QCAP_MAX_ROUNDS = 64;
__host__ __device__ constexpr int res_qcap(int k, int tiles_per_chunk) {
    return k <= 32 && tiles_per_chunk <= RES_QCAP_MAX_ROUNDS ? 32 : 16;
}

// Resident filter pass (PMODE 3): the query tile's two activation planes, [64 queries][384] halfs each, live in LDS
// for the whole block in the layout of the encoder's fused kernels (768-B rows as three XOR-swizzled 256-B sub-rows).
constexpr int RES_XPLANE = 64 * 768;
constexpr int RES_X_BYTES = 2 * RES_XPLANE;
constexpr int RES_KS = 24;  // k-steps of 16: the resident pass is built for dim = 384

template <class Cfg, int MODE = 0>  // MODE: 0 fp32 tiles, 2 f16 hi/lo planes staged per tile, 3 resident query planes
struct SearchSmem {
    // operand staging: fp32 tiles (common.h) or the f16 hi/lo planes of the filter pass (gemm_x3.h)
    static constexpr size_t GEMM = MODE == 3 ? (size_t)RES_X_BYTES
                                   : MODE == 2 ? (size_t)(2 * Cfg::BM + 2 * Cfg::BN) * HLD * 2 : (size_t)Cfg::LDS_FLOATS * 4;
    // dynamic LDS carve (all offsets multiples of 16 B)
    static __host__ __device__ size_t bytes(int k) {
        return GEMM + (size_t)Cfg::BN * (8 /*thr*/ + 4 /*cnt*/) + 16 /*flags*/ +
               (size_t)Cfg::BN * k * 8 + (size_t)Cfg::BN * (MODE == 3 && k <= 32 ? 32 : QCap<Cfg>::V) * 8;
    }
};

// Is local row `row` in the sorted exclusion segment [lo, hi)?
__device__ __forceinline__ bool excluded(const int32_t* __restrict__ ex, int lo, int hi, int row) {
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        int v = ex[mid];
        if (v == row) return true;
        if (v < row) lo = mid + 1; else hi = mid;
    }
    return false;
}

// Merge the queue of query q into its sorted list; one wavefront, all 64 lanes call this.
// Every element's new position is its rank in the union (keys are unique): a list entry keeps its
// index plus the number of better candidates; candidate i gets (#better candidates) + (#better list
// entries), the latter from a ballot over the lanes holding the list — no search, no extra LDS trips.
__device__ __forceinline__ void merge_queue(u64* list, const u64* queue, int n, int k, int lane) {
    const u64 c = lane < n ? queue[lane] : 0ull;
    const u64 e0 = lane < k ? list[lane] : 0ull;
    const u64 e1 = lane + 64 < k ? list[lane + 64] : 0ull;
    int rc = 0, r0 = 0, r1 = 0, lc = 0;
    for (int i = 0; i < n; ++i) {
        const u64 ci = queue[i];  // LDS broadcast
        rc += ci > c;
        r0 += ci > e0;
        r1 += ci > e1;
        const int better = __popcll(__ballot(e0 > ci)) + (k > 64 ? __popcll(__ballot(e1 > ci)) : 0);
        lc = lane == i ? better : lc;
    }
    const int pc = rc + lc, p0 = lane + r0, p1 = lane + 64 + r1;
    // all reads above are complete (their values are consumed) before any lane writes
    if (lane < n && pc < k) list[pc] = c;
    if (lane < k && p0 < k) list[p0] = e0;
    if (lane + 64 < k && p1 < k) list[p1] = e1;
}

// Two queries per call for k <= 32 and queues of <= 32 slots: lanes 0-31 merge query qa, lanes 32-63 query qb
// (nb == 0: no partner).  Same ranking rule as merge_queue; the ballot's two halves serve the two queries.  Also
// publishes the new thresholds and clears the queue counters.
__device__ __forceinline__ void merge_queue2(u64* list, const u64* queue, int qcap, int qa, int na, int qb, int nb, int k,
                                             int lane, u64* thr, int* cnt) {
    // Two queries per call (k, n <= 32): lanes 0-31 merge query qa, lanes 32-63 query qb.  Lane l holds list entry l and
    // candidate l of its query; every element's new position is its rank in the union (keys are unique):
    //   list entry:  l + #{candidates better than it}
    //   candidate:   #{candidates better than it} + #{list entries better than it}
    // The candidate counts come from ONE pass over the query's queue in LDS (the 32 lanes of a half read the same
    // address: a broadcast), the list count from a binary search (the list is sorted, best first, empty slots = 0 last).
    // (Round 2 broadcast the candidates with v_readlane and counted with ballots: ~5 k cycles per call against ~1 k.)
    const int half = lane >> 5, l = lane & 31;
    const int q = half ? qb : qa, n = half ? nb : na;
    u64* lst = list + (size_t)q * k;
    const u64* qp = queue + (size_t)q * qcap;
    const u64 c = l < n ? qp[l] : 0ull;
    const u64 e0 = l < k ? lst[l] : 0ull;
    int rc = 0, r0 = 0;
    const int nmax = na > nb ? na : nb;
    for (int i0 = 0; i0 < nmax; i0 += 8) {  // eight reads in flight (one dependent read per candidate is all latency)
        u64 ci[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ci[u] = qp[i0 + u];  // inside the queue (qcap is 16 or 32, nmax <= qcap), possibly stale
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u64 v = i0 + u < n ? ci[u] : 0ull;  // 0 past this query's own count: compares false everywhere
            rc += v > c;
            r0 += v > e0;
        }
    }
    int lc = 0;  // number of list entries better than c = the first index whose entry is not
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) {
        const int probe = lc + step;
        if (probe <= k && lst[probe - 1] > c) lc = probe;
    }
    if (n > 0) {
        const int pc = rc + lc, p0 = l + r0;
        // all reads above are complete (their values are consumed) before any lane writes
        if (l < n && pc < k) lst[pc] = c;
        if (l < k && p0 < k) lst[p0] = e0;
        if (l == 0) { thr[q] = lst[k - 1]; cnt[q] = 0; }
    }
}

// Grid: n_chunks * n_qtiles blocks (XCD-remapped).  Block (chunk, qtile) scores catalog row
// tiles [chunk*tiles_per_chunk, ...) against query tile qtile and keeps, per query, the k best
// (score, row) seen, then writes them (sorted, as keys) to partial[chunk][query][0..k).
// ---- fragment helpers of the RESIDENT filter pass (PMODE 3).  This pass keeps the v_mfma_f32_32x32x16_f16 form of
// the weights-direct engine (its selection code is written for the 32x32 accumulator map; the pass is ~3 % of a
// recommend step): catalog rows packed as [32-row tile][16-deep k-step][plane] fragments of 1 KB, lane (h << 5 | r)
// holding row r, k = 16 ks + 8 h .. +7.  (The encoder's linear layers use the 16x16x32 form, wt_gemm.h.)
__device__ __forceinline__ size_t r32_frag_off(int nt, int ks, int KS) { return ((size_t)nt * KS + ks) * (2 * WT_FRAG); }
__device__ __forceinline__ void r32_w_load(half8& wh, half8& wl, const _Float16* wp, int ks, unsigned lo8) {
    const _Float16* p = wp + (size_t)ks * (2 * WT_FRAG);
    wh = *reinterpret_cast<const half8*>(p + lo8);
    wl = *reinterpret_cast<const half8*>(p + WT_FRAG + lo8);
}
__device__ __forceinline__ void r32_mma(f32x16 (&acc)[1][2], const half8& wh, const half8& wl, const half8 (&xh)[2],
                                        const half8 (&xl)[2]) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        acc[0][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[tt], acc[0][tt], 0, 0, 0);
        acc[0][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[tt], acc[0][tt], 0, 0, 0);
        acc[0][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[tt], acc[0][tt], 0, 0, 0);
    }
}

// EMIT = true additionally stores every score to scores_out[q*N + row] (parity checks only).
// PMODE 0: fp32 rows.  1: rows stored as bfloat16 (ICREC_ROWS_BF16), widened on their way into LDS.
// 2: the FILTER pass of ICREC_ROWS_F32_FILTER — rows and queries as f16 hi/lo planes (P/P2, Qn/Q2), scores from
//    three f16 MFMAs per product (gemm_x3.h): within ~1e-7 of the exact chain at 5x its MFMA rate, NOT bit-exact;
//    its lists only nominate candidates for verify_kernel.
// 3: the RESIDENT form of the filter pass (dim = 384, catalogs up to RES_MAX_ROWS): P = the rows as packed weight
//    fragments (wt_gemm.h: the 1 KB one wave feeds to one MFMA is contiguous), Qn/Q2 = the queries' activation planes.
//    The block's 64 queries are loaded into LDS ONCE and stay there for all its row tiles; every wave owns one
//    32-row tile of each 256-row round and streams its fragments L2 -> registers through an 8-deep ring that runs
//    across rounds (the next round's first fragments land under the selection): no operand staging barriers at all -
//    PMODE 2 re-stages both operands through LDS for every 128-row tile (two barriers per 64-deep slab).
// run_flag != NULL: the whole grid exits unless *run_flag != 0 (the exact pass behind a filter pass).
#ifndef ICREC_STAMP_ROUND0
#define ICREC_STAMP_ROUND0 0  // tools/search_stamps.hip: first round of a block whose phases are stamped
#endif
template <class Cfg, bool EMIT, int PMODE>
__global__ __launch_bounds__(Cfg::THREADS, 2) void search_kernel(
    const void* __restrict__ P, const void* __restrict__ P2, int64_t N, int K, const void* __restrict__ Qn,
    const void* __restrict__ Q2, int Qpad, int Q, int k, const int32_t* __restrict__ excl_idx,
    const int32_t* __restrict__ excl_off, uint32_t row_base, int n_row_tiles, int tiles_per_chunk, int n_qtiles,
    u64* __restrict__ partial, float* __restrict__ scores_out, const int* __restrict__ run_flag) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (run_flag != nullptr && *run_flag == 0) return;  // uniform over the grid
    constexpr bool P16 = PMODE == 1;
    float* As = reinterpret_cast<float*>(smem_raw);
    float* Bs = As + Cfg::BM * LDK;
    u64* thr = reinterpret_cast<u64*>(smem_raw + SearchSmem<Cfg, PMODE >= 2 ? PMODE : 0>::GEMM);
    int* cnt = reinterpret_cast<int*>(thr + Cfg::BN);
    int* flags = cnt + Cfg::BN;  // [0],[1]: alternating "some candidate did not fit" flags
    u64* list = reinterpret_cast<u64*>(flags + 4);
    u64* queue = list + (size_t)Cfg::BN * k;

    const int QCAP = PMODE == 3 ? res_qcap(k, tiles_per_chunk) : QCap<Cfg>::V;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    // consecutive logical ids share a catalog chunk (and hence an XCD's L2)
    const int chunk = bid / n_qtiles, qtile = bid % n_qtiles;
    const int q0 = qtile * Cfg::BN;

    ICREC_STAMP(0, 60);
    for (int i = tid; i < Cfg::BN; i += Cfg::THREADS) { thr[i] = (q0 + i < Q) ? 0ull : ~0ull; cnt[i] = 0; }
    for (int i = tid; i < Cfg::BN * k; i += Cfg::THREADS) list[i] = 0ull;
    if (tid < 4) flags[tid] = 0;
    __syncthreads();

    // this lane's queries: column (lane & 31) of each of its TN column tiles
    int myq[Cfg::TN];
    u64 mythr[Cfg::TN];
    int ex_lo[Cfg::TN], ex_hi[Cfg::TN];
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
        myq[j] = (wn * Cfg::TN + j) * 32 + (lane & 31);
        const int gq = q0 + myq[j];
        mythr[j] = gq < Q ? 0ull : ~0ull;  // padding columns never produce candidates
        ex_lo[j] = ex_hi[j] = 0;
        if (excl_off != nullptr && gq < Q) { ex_lo[j] = excl_off[gq]; ex_hi[j] = excl_off[gq + 1]; }
    }

    const int t_begin = chunk * tiles_per_chunk;
    const int t_end = min(n_row_tiles, t_begin + tiles_per_chunk);
    int round = 0;
    TileRegs<Cfg, P16> pre;
    f32x16 acc[Cfg::TM][Cfg::TN];

    // ---- PMODE 3: query planes -> LDS (once), weight ring of the first round
    half8 rwh[8], rwl[8];
    int xb0[2] = {0, 0};
    const unsigned lo8 = lane * 8;
    if constexpr (PMODE == 3) {
        static_assert(PMODE != 3 || (Cfg::TM == 1 && Cfg::TN == 2 && Cfg::WAVES_N == 1 && Cfg::WAVES_M == 8),
                      "resident pass: 8 waves x (1 row tile x 2 query tiles)");
        const _Float16* qh = static_cast<const _Float16*>(Qn);
        const _Float16* ql = static_cast<const _Float16*>(Q2);
        char* const Xs = smem_raw;
        u32x4 vh[6], vl[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 512 * i, row = id / 48, c = id - row * 48;
            const int64_t g = (int64_t)(q0 + row) * 384 + c * 8;  // Qpad is a multiple of 64: every row exists
            vh[i] = *reinterpret_cast<const u32x4*>(qh + g);
            vl[i] = *reinterpret_cast<const u32x4*>(ql + g);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 512 * i, row = id / 48, c = id - row * 48;
            const int pos = row * 768 + (((c & ~15) | ((c ^ row) & 15)) << 4);
            *reinterpret_cast<u32x4*>(Xs + pos) = vh[i];
            *reinterpret_cast<u32x4*>(Xs + RES_XPLANE + pos) = vl[i];
        }
        // this lane's fragment of query tile tt at k-step ks: xb0[tt] ^ ((ks & 7) << 5), + 256 (ks >> 3)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int tok = tt * 32 + (lane & 31);
            xb0[tt] = tok * 768 + (((lane >> 5) ^ (tok & 15)) << 4);
        }
        if (t_begin < t_end) {
            const _Float16* const wp0 = static_cast<const _Float16*>(P) + r32_frag_off((t_begin * 8 + wave) * Cfg::TM, 0, RES_KS);
#pragma unroll
            for (int d = 0; d < 8; ++d) r32_w_load(rwh[d], rwl[d], wp0, d, lo8);
        }
        __syncthreads();  // queries resident
    }

    ICREC_STAMP(0, 61);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int64_t row0 = (int64_t)tile * Cfg::BM;
        if (tile - t_begin - ICREC_STAMP_ROUND0 >= 0 && tile - t_begin - ICREC_STAMP_ROUND0 < 24) ICREC_STAMP(0, 2 * (tile - t_begin - ICREC_STAMP_ROUND0));
        if constexpr (PMODE == 3) {
            // the wave's TM 32-row tiles of this round, one after the other (fragment tiles (tile * 8 + wave) * TM + i); the
            // selection below then runs once per round over all of them: its barriers and polls are per round, not per tile
            const char* const Xs = smem_raw;
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) {
                const int rt = (tile * 8 + wave) * Cfg::TM + i;
                // the ring continues into the wave's next tile; past the block's last round: re-read (never consumed)
                const int rn = i + 1 < Cfg::TM ? rt + 1 : (tile + 1 < t_end ? ((tile + 1) * 8 + wave) * Cfg::TM : rt);
                const _Float16* const wp1 = static_cast<const _Float16*>(P) + r32_frag_off(rt, 0, RES_KS);
                const _Float16* const wpn = static_cast<const _Float16*>(P) + r32_frag_off(rn, 0, RES_KS);
                f32x16 S[1][2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) S[0][tt][e] = 0.0f;
                half8 fh[2][2], fl[2][2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    fh[0][tt] = *reinterpret_cast<const half8*>(Xs + xb0[tt]);
                    fl[0][tt] = *reinterpret_cast<const half8*>(Xs + RES_XPLANE + xb0[tt]);
                }
#pragma unroll
                for (int ks = 0; ks < RES_KS; ++ks) {
                    if (ks + 1 < RES_KS) {
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            const int pos = (xb0[tt] ^ (((ks + 1) & 7) << 5)) + ((ks + 1) >> 3) * 256;
                            fh[(ks + 1) & 1][tt] = *reinterpret_cast<const half8*>(Xs + pos);
                            fl[(ks + 1) & 1][tt] = *reinterpret_cast<const half8*>(Xs + RES_XPLANE + pos);
                        }
                    }
                    r32_mma(S, rwh[ks & 7], rwl[ks & 7], fh[ks & 1], fl[ks & 1]);
                    if (ks + 8 < RES_KS) r32_w_load(rwh[ks & 7], rwl[ks & 7], wp1, ks + 8, lo8);
                    else r32_w_load(rwh[ks & 7], rwl[ks & 7], wpn, ks + 8 - RES_KS, lo8);
                    __builtin_amdgcn_sched_barrier(0);  // pin the prefetch to its k-step
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = S[0][j][e] * WT_UNSCALE;
            }
        } else if (PMODE == 2) {
            f32x16 a0[Cfg::TM][Cfg::TN], a1[Cfg::TM][Cfg::TN];
            tile_gemm_h<Cfg>(a0, a1, static_cast<const _Float16*>(P), static_cast<const _Float16*>(P2), row0, N,
                             static_cast<const _Float16*>(Qn), static_cast<const _Float16*>(Q2), q0, Qpad, K,
                             reinterpret_cast<_Float16*>(smem_raw));
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(a1[i][j][e], LO_UNSCALE, a0[i][j][e]);
        } else {
            tile_gemm<Cfg, P16>(acc, P, row0, N, static_cast<const float*>(Qn), q0, Qpad, K, As, Bs, pre, false);
        }

        if (EMIT) {
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int64_t row = row0 + (wm * Cfg::TM + i) * 32 + acc_row(e, lane);
                        const int gq = q0 + myq[j];
                        if (row < N && gq < Q) scores_out[(int64_t)gq * N + row] = acc[i][j][e] + 0.0f;
                    }
        }

        if (tile - t_begin - ICREC_STAMP_ROUND0 >= 0 && tile - t_begin - ICREC_STAMP_ROUND0 < 24) ICREC_STAMP(0, 2 * (tile - t_begin - ICREC_STAMP_ROUND0) + 1);
        // ---- selection
        // A score is OFFERED (pushed to its query's LDS queue) when it beats the query's threshold.
        //  * warm query (list full): threshold = current k-th best key.  Queues are merged into the
        //    sorted lists only when one is at least half full (or at the block's last tile), so a
        //    stale — lower — threshold only means a few extra offers, never a missed hit.
        //  * cold query (list not full yet, threshold key 0): instead of offering all of the tile's
        //    scores, each lane first offers only its own m largest (m = ceil(k / lanes per query) + 1,
        //    so the lanes together offer >= k), the queues are merged at once, and a second pass
        //    offers whatever else still beats the now-real threshold (usually nothing).
        constexpr int LPQ = 2 * Cfg::WAVES_M;  // lanes holding scores of one query
        const int m_local = (k + LPQ - 1) / LPQ + 1;
        const bool last_tile = tile == t_end - 1;
        unsigned long long offered = 0ull;
        bool lane_cold = false;
        for (int pass = 0; pass < 2; ++pass) {
            // pend bit (j*TM+i)*16+e: score e of accumulator tile (i,j) has to be offered
            unsigned long long pend = 0ull;
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) {
                float thr_s;
                if (mythr[j] == ~0ull) {
                    thr_s = INFINITY;  // padding column
                } else if (mythr[j] == 0ull && pass == 0) {
                    lane_cold = true;
                    float t = INFINITY;  // t <- m-th largest distinct valid score of this lane
                    for (int it = 0; it < m_local; ++it) {
                        float best = -INFINITY;
#pragma unroll
                        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const bool valid = row0 + (wm * Cfg::TM + i) * 32 + acc_row(e, lane) < N;
                                const float v = acc[i][j][e] + 0.0f;
                                best = (valid && v < t && v > best) ? v : best;
                            }
                        t = best;
                    }
                    thr_s = t;
                } else {
                    thr_s = mythr[j] ? key_score(mythr[j]) : -INFINITY;
                }
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (acc[i][j][e] + 0.0f >= thr_s) pend |= 1ull << ((j * Cfg::TM + i) * 16 + e);
            }
            pend &= ~offered;
            offered |= pend;
            if (pass == 0 && lane_cold) flags[2] = 1;
            bool more, wg_cold;
            [[maybe_unused]] int it_stamp = 0;
            do {
                if (tile - t_begin - ICREC_STAMP_ROUND0 == 11 && it_stamp < 6) ICREC_STAMP(0, 30 + 3 * it_stamp);
                bool lane_pending = false;
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i) {
                        const int sh = (j * Cfg::TM + i) * 16;
                        unsigned m16 = (unsigned)(pend >> sh) & 0xFFFFu;
                        while (m16) {
                            const int e = __builtin_ctz(m16);
                            m16 &= m16 - 1;
                            float sc = acc[i][j][0];
#pragma unroll
                            for (int t = 1; t < 16; ++t) sc = e == t ? acc[i][j][t] : sc;
                            sc = sc + 0.0f;  // -0 -> +0
                            const int64_t row = row0 + (wm * Cfg::TM + i) * 32 + acc_row(e, lane);
                            bool take = row < N;
                            u64 key = 0ull;
                            if (take) {
                                key = make_key(sc, row_base + (uint32_t)row);
                                take = key > mythr[j];
                            }
                            if (take && ex_hi[j] > ex_lo[j]) take = !excluded(excl_idx, ex_lo[j], ex_hi[j], (int)row);
                            bool settled = true;
                            if (take) {
                                const int slot = atomicAdd(&cnt[myq[j]], 1);
                                if (slot < QCAP) queue[myq[j] * QCAP + slot] = key;
                                else { settled = false; lane_pending = true; }
                            }
                            if (settled) pend &= ~(1ull << (sh + e));
                        }
                    }
                if (lane_pending) flags[round & 1] = 1;
                if (tile - t_begin - ICREC_STAMP_ROUND0 == 11 && it_stamp < 6) ICREC_STAMP(0, 31 + 3 * it_stamp);
                __syncthreads();
                if (tile - t_begin - ICREC_STAMP_ROUND0 == 11 && it_stamp < 6) ICREC_STAMP(0, 32 + 3 * it_stamp);
                ++it_stamp;
                more = flags[round & 1] != 0;
                wg_cold = flags[2] != 0;
                if (tid == 0) flags[(round + 1) & 1] = 0;
                const bool force = wg_cold || last_tile;
                // each wave merges the queues of its share of the queries (query wave + NW * t is polled by lane t:
                // one LDS read for all of them instead of one dependent read per query)
                constexpr int NW = Cfg::THREADS / 64, QPW = Cfg::BN / NW;
                static_assert(QPW <= 64, "one polling lane per query");
                int myc = lane < QPW ? cnt[wave + NW * lane] : 0;
                myc = myc < QCAP ? myc : QCAP;
                // merge a queue once it holds 8 candidates (a quarter of the 32-entry queues: the room above absorbs a burst
                // without a second offer / merge iteration, the early merge keeps the thresholds fresh)
                const int trig = QCAP >= 32 ? 8 : QCAP / 2;
                unsigned long long need = __ballot(myc > 0 && (force || myc >= trig));
                if (QCAP <= 32 && k <= 32) {
                    while (need) {  // two queries per merge call
                        const int ta = __builtin_ctzll(need);
                        need &= need - 1;
                        const int tb = need ? __builtin_ctzll(need) : -1;
                        if (tb >= 0) need &= need - 1;
                        const int na = __builtin_amdgcn_readlane(myc, ta);
                        const int nb = tb >= 0 ? __builtin_amdgcn_readlane(myc, tb) : 0;
                        merge_queue2(list, queue, QCAP, wave + NW * ta, na, wave + NW * (tb >= 0 ? tb : ta), nb, k, lane, thr,
                                     cnt);
                    }
                } else {
                    while (need) {
                        const int t = __builtin_ctzll(need);
                        need &= need - 1;
                        const int q = wave + NW * t;
                        merge_queue(list + (size_t)q * k, queue + q * QCAP, __builtin_amdgcn_readlane(myc, t), k, lane);
                        if (lane == 0) { thr[q] = list[(size_t)q * k + k - 1]; cnt[q] = 0; }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) mythr[j] = thr[myq[j]];
                ++round;
            } while (more);
            if (!wg_cold) break;  // warm block: one pass (uniform: read between the barriers)
        }
        __syncthreads();
        if (tid == 0) flags[2] = 0;
    }

    ICREC_STAMP(0, 62);
    // sorted partial lists out
    for (int i = tid; i < Cfg::BN * k; i += Cfg::THREADS) {
        const int q = i / k, e = i % k;
        partial[((size_t)chunk * Qpad + q0 + q) * k + e] = list[(size_t)q * k + e];
    }
    ICREC_STAMP(0, 63);
}

// ---------------------------------------------------------------- small-batch streaming search
// For Q <= 8 queries the MFMA kernel above spends 4-32x its useful work on padding columns (its
// narrowest tile is 32 queries wide) and becomes MFMA-bound long before HBM.  This kernel is the
// HBM-bound form (SURVEY §8d "K8a"): one thread per catalog row.  A block streams tiles of 256 rows
// through LDS in 128-byte slabs (coalesced 16-B global loads, next slab in flight under the current
// one's FMAs); each thread walks ITS row of the slab out of LDS and runs one fp32 fmaf chain per
// query, k ascending from 0 — the very chain the f32 MFMA computes, so scores stay bit-identical to
// the oracle.  Query values are block-uniform (scalar loads).  Selection: the block's first tile ranks
// its 256 keys per query by counting (no merges); later tiles offer only keys above the running k-th
// best into a 256-slot LDS queue that merge_queue folds into the sorted list.
constexpr int ST_ROWS = 256, ST_LDB = 144;  // LDS row stride 144 B: conflict-free 16-B reads at one row per lane

template <int NQ>
struct StreamSmem {
    static __host__ __device__ size_t bytes(int k) {
        return (size_t)ST_ROWS * ST_LDB + 64 /*thr*/ + 64 /*cnt*/ + (size_t)((NQ * k + 1) & ~1) * 8 + (size_t)NQ * ST_ROWS * 8;
    }
};

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
// acc[j] = fmaf(q[j], p, acc[j]) for the NQ queries of one k; pairs of queries share one packed FMA
// (v_pk_fma_f32: two independent IEEE fmas, so each chain is unchanged).
template <int NQ>
__device__ __forceinline__ void stream_fma(float (&acc)[NQ], const float* __restrict__ q, float p) {
    if (NQ == 1) {
        acc[0] = fmaf(q[0], p, acc[0]);
    } else {
#pragma unroll
        for (int j = 0; j < NQ; j += 2) {
            const v2f a = {acc[j], acc[j + 1]}, qq = {q[j], q[j + 1]}, pp = {p, p};
            const v2f r = __builtin_elementwise_fma(qq, pp, a);
            acc[j] = r.x;
            acc[j + 1] = r.y;
        }
    }
}

// 128-B slab `slab` of the 256 rows of tile `tile` -> 8 x 16 B per thread (rows clamped to the last valid row)
__device__ __forceinline__ void stream_load_slab(v4f (&pre)[8], const char* __restrict__ Pb, int64_t N, int row_bytes,
                                                 int tile, int slab, int tid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = tid + ST_ROWS * i;
        int64_t row = (int64_t)tile * ST_ROWS + (id >> 3);
        row = row < N ? row : N - 1;
        pre[i] = *reinterpret_cast<const v4f*>(Pb + row * row_bytes + slab * 128 + (id & 7) * 16);
    }
}

template <int NQ, bool P16>
__global__ __launch_bounds__(ST_ROWS, 2) void stream_search_kernel(
    const void* __restrict__ P, int64_t N, int K, const float* __restrict__ Qn /* [K][NQ], zero-padded columns */, int Q,
    int k, const int32_t* __restrict__ excl_idx, const int32_t* __restrict__ excl_off, uint32_t row_base, int n_row_tiles,
    int tiles_per_chunk, u64* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* rows_s = smem_raw;
    u64* thr = reinterpret_cast<u64*>(smem_raw + ST_ROWS * ST_LDB);
    int* cnt = reinterpret_cast<int*>(thr + 8);
    u64* list = reinterpret_cast<u64*>(cnt + 16);
    u64* queue = list + (size_t)((NQ * k + 1) & ~1);  // 16-B aligned: the cold path reads it two keys at a time

    constexpr int EPW = P16 ? 2 : 1;             // elements per 32-bit word
    constexpr int SLAB_ELEMS = 32 * EPW;         // 128 B of one row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = xcd_remap(blockIdx.x, gridDim.x);
    const int row_bytes = K * (P16 ? 2 : 4);
    const int nslab = K / SLAB_ELEMS;
    const char* Pb = static_cast<const char*>(P);

    if (tid < NQ) { thr[tid] = tid < Q ? 0ull : ~0ull; cnt[tid] = 0; }
    for (int i = tid; i < NQ * k; i += ST_ROWS) list[i] = 0ull;

    u64 mythr[NQ];
    int ex_lo[NQ], ex_hi[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        mythr[j] = j < Q ? 0ull : ~0ull;
        ex_lo[j] = ex_hi[j] = 0;
        if (excl_off != nullptr && j < Q) { ex_lo[j] = excl_off[j]; ex_hi[j] = excl_off[j + 1]; }
    }

    const int t_begin = chunk * tiles_per_chunk;
    const int t_end = min(n_row_tiles, t_begin + tiles_per_chunk);
    v4f pre[8];  // native vectors: HIP's float4 struct copies become memcpys that pin the array in scratch
    stream_load_slab(pre, Pb, N, row_bytes, min(t_begin, n_row_tiles - 1), 0, tid);

    for (int tile = t_begin; tile < t_end; ++tile) {
        float acc[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) acc[j] = 0.0f;
        for (int s = 0; s < nslab; ++s) {
            __syncthreads();  // the previous slab has been consumed (also covers the list/thr initialisation)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int id = tid + ST_ROWS * i;
                *reinterpret_cast<v4f*>(rows_s + (id >> 3) * ST_LDB + (id & 7) * 16) = pre[i];
            }
            __syncthreads();
            {   // next slab (of this tile or the first of the next one) goes in flight under this slab's FMAs;
                // unconditional — the block's very last iteration re-loads its own slab — so `pre` stays in registers
                const bool wrap = s + 1 == nslab;
                const int nt = wrap ? min(tile + 1, t_end - 1) : tile, ns = wrap ? 0 : s + 1;
                stream_load_slab(pre, Pb, N, row_bytes, nt, ns, tid);
            }
            const char* my = rows_s + tid * ST_LDB;
            const float* qs = Qn + (size_t)s * SLAB_ELEMS * NQ;  // k-major: the NQ values of one k are adjacent
            // NQ (x2 for bf16 rows) x 8 scalar query values per 16-B piece: unroll only as far as SGPRs allow
#pragma unroll(NQ >= 8 ? 1 : NQ >= 4 ? 2 : 8)
            for (int c = 0; c < 8; ++c) {
                const v4f w = *reinterpret_cast<const v4f*>(my + c * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (P16) {
                        const unsigned u = __float_as_uint(w[e]);
                        const int kk = c * 8 + e * 2;
                        stream_fma<NQ>(acc, qs + kk * NQ, __uint_as_float(u << 16));
                        stream_fma<NQ>(acc, qs + (kk + 1) * NQ, __uint_as_float(u & 0xFFFF0000u));
                    } else {
                        stream_fma<NQ>(acc, qs + (c * 4 + e) * NQ, w[e]);
                    }
                }
            }
        }

        // ---- selection for this tile's 256 scores per query
        const int64_t row = (int64_t)tile * ST_ROWS + tid;
        u64 key[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const u64 kx = make_key(acc[j] + 0.0f, row_base + (uint32_t)row);
            bool take = row < N && mythr[j] != ~0ull && kx > mythr[j];
            if (take && ex_hi[j] > ex_lo[j]) take = !excluded(excl_idx, ex_lo[j], ex_hi[j], (int)row);
            key[j] = take ? kx : 0ull;
        }
        if (tile == t_begin) {
            // cold: rank every key among the tile's 256 by counting; rank r < k goes straight to list[r]
#pragma unroll
            for (int j = 0; j < NQ; ++j) queue[j * ST_ROWS + tid] = key[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (j < Q) {  // uniform
                    int r = 0;
                    const u64* qj = queue + j * ST_ROWS;
                    for (int i = 0; i < ST_ROWS; i += 2) {
                        const ulonglong2 two = *reinterpret_cast<const ulonglong2*>(qj + i);  // LDS broadcast
                        r += (two.x > key[j]) + (two.y > key[j]);
                    }
                    if (key[j] != 0ull && r < k) list[j * k + r] = key[j];
                }
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int j = 0; j < NQ; ++j)
                if (key[j] != 0ull) queue[j * ST_ROWS + atomicAdd(&cnt[j], 1)] = key[j];  // <= 256 offers per tile
            __syncthreads();
            for (int j = wave; j < NQ; j += ST_ROWS / 64) {
                const int c = cnt[j];
                for (int off = 0; off < c; off += 64)
                    merge_queue(list + (size_t)j * k, queue + j * ST_ROWS + off, min(64, c - off), k, lane);
                if (lane == 0) cnt[j] = 0;
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j)
            if (mythr[j] != ~0ull) mythr[j] = list[(size_t)j * k + k - 1];
    }

    __syncthreads();
    for (int i = tid; i < NQ * k; i += ST_ROWS) partial[(size_t)chunk * NQ * k + i] = list[i];
}

// ---------------------------------------------------------------- filter + verify (ICREC_ROWS_F32_FILTER)
// fp32 values -> f16 hi/lo planes (queries per call, catalog rows once at create); clears the fallback flag if given.
template <bool SRC16>  // SRC16: the source is bfloat16 bits (ICREC_ROWS_BF16_FILTER rows), widened exactly first
__global__ __launch_bounds__(256) void split_queries_kernel(const void* __restrict__ src, size_t n, _Float16* __restrict__ hi,
                                                            _Float16* __restrict__ lo, int* __restrict__ flag) {
    if (flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *flag = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = SRC16 ? __uint_as_float((unsigned)static_cast<const uint16_t*>(src)[i] << 16)
                              : static_cast<const float*>(src)[i];
        _Float16 a, b;
        split_f16(v, a, b);
        hi[i] = a;
        lo[i] = b;
    }
}

// Resident filter pass: the (normalised) catalog rows as packed weight fragments - fragment (row tile rt, k-step ks,
// plane) = 512 halfs at ((rt * KS + ks) * 2 + plane) * 512, lane l = (h << 5 | r) holds row rt*32 + r, k = 16 ks + 8 h
// .. +7, hi = f16(1024 x), lo = f16(1024 x - hi) (wt_gemm.h) - once at index creation; rows past n_rows are zero.
template <bool SRC16>
__global__ __launch_bounds__(256) void pack_rows_kernel(const void* __restrict__ src, int64_t n_rows, int K, int64_t n_frag,
                                                        _Float16* __restrict__ out) {
    const int KS = K / 16;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n_frag * 64; id += (int64_t)gridDim.x * 256) {
        const int64_t fr = id >> 6;
        const int lane = (int)(id & 63), r = lane & 31, h = lane >> 5;
        const int64_t rt = fr / KS;
        const int ks = (int)(fr % KS);
        const int64_t row = rt * 32 + r;
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.0f;
            if (row < n_rows) {
                const size_t at = (size_t)row * K + ks * 16 + 8 * h + j;
                v = SRC16 ? __uint_as_float((unsigned)static_cast<const uint16_t*>(src)[at] << 16)
                          : static_cast<const float*>(src)[at];
            }
            _Float16 a, b;
            split_scaled(v, WT_SW, a, b);
            hi[j] = a;
            lo[j] = b;
        }
        *reinterpret_cast<half8*>(out + fr * (2 * WT_FRAG) + lane * 8) = hi;
        *reinterpret_cast<half8*>(out + fr * (2 * WT_FRAG) + WT_FRAG + lane * 8) = lo;
    }
}

// queries -> the engine's activation planes (hi/lo of 16 x, wt_gemm.h: split_act4), row-major [Qpad][K]; clears the flag
__global__ __launch_bounds__(256) void split_queries_act_kernel(const float* __restrict__ src, size_t n4, _Float16* __restrict__ hi,
                                                                _Float16* __restrict__ lo, int* __restrict__ flag) {
    if (flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *flag = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
        half4 a, b;
        split_act4(v, a, b);
        *reinterpret_cast<half4*>(hi + 4 * i) = a;
        *reinterpret_cast<half4*>(lo + 4 * i) = b;
    }
}

// Exact re-scoring of the filter pass's candidates.  cand[q][0..kp): the kp best (approximate score, row) keys of
// query q, sorted; one wavefront per query, two candidates per lane.  Every candidate gets the exact k-ascending
// fp32 fmaf chain (the oracle's arithmetic) from the fp32 rows, candidates are ranked by (exact score desc, row asc)
// and the best k written out.  The result is THE exact top-k iff no row outside the list can reach the k-th exact
// score: outside rows have approx <= the list's last approx score, and |approx - exact| <= eps, so
//     last_approx + eps < exact_kth    (or the list is not full: it then holds every admissible row)
// proves it.  Otherwise *flag is set and the exact search that follows (it exits at once when the flag is clear)
// recomputes the batch.
template <bool P16>  // P16: rows are bfloat16 bits, widened exactly (the arithmetic of the bf16 exact search)
__global__ __launch_bounds__(256) void verify_kernel(const void* __restrict__ Pv, int K, const float* __restrict__ qn,
                                                     const u64* __restrict__ cand, int Q, int kp, int k, uint32_t row_base,
                                                     float eps, int64_t* __restrict__ out_idx, float* __restrict__ out_score,
                                                     u64* __restrict__ out_keys, int* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const float* qv = qn + (size_t)q * K;
    u64 ek[2];
    int n_valid = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int c = lane + 64 * s;
        const u64 ck = c < kp ? cand[(size_t)q * kp + c] : 0ull;
        ek[s] = 0ull;
        if (ck != 0ull) {
            const uint32_t grow = key_row(ck);
            float acc = 0.0f;
            if (P16) {
                const uint16_t* pr = static_cast<const uint16_t*>(Pv) + (size_t)(grow - row_base) * K;
                for (int j = 0; j < K; j += 4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(qv + j);
                    const uint2 w = *reinterpret_cast<const uint2*>(pr + j);
                    acc = fmaf(a[0], __uint_as_float(w.x << 16), acc);
                    acc = fmaf(a[1], __uint_as_float(w.x & 0xFFFF0000u), acc);
                    acc = fmaf(a[2], __uint_as_float(w.y << 16), acc);
                    acc = fmaf(a[3], __uint_as_float(w.y & 0xFFFF0000u), acc);
                }
            } else {
                const float* pr = static_cast<const float*>(Pv) + (size_t)(grow - row_base) * K;
                for (int j = 0; j < K; j += 4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(qv + j);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(pr + j);
                    acc = fmaf(a[0], b[0], acc);
                    acc = fmaf(a[1], b[1], acc);
                    acc = fmaf(a[2], b[2], acc);
                    acc = fmaf(a[3], b[3], acc);
                }
            }
            ek[s] = make_key(acc + 0.0f, grow);
        }
        n_valid += __popcll(__ballot(ck != 0ull));
    }
    // rank of each exact key among the candidates (keys are unique: the row is part of the key)
    int rk[2] = {0, 0};
    for (int s = 0; s < 2; ++s)
        for (int l = 0; l < 64; ++l) {
            const u64 o = shfl_u64(ek[s], l);
            rk[0] += o > ek[0];
            rk[1] += o > ek[1];
        }
    // exact k-th best score (rank k-1), if there are that many candidates
    float kth = -INFINITY;
    if (n_valid >= k) {
        float mine = -INFINITY;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (ek[s] != 0ull && rk[s] == k - 1) mine = key_score(ek[s]);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mine = fmaxf(mine, __shfl_xor(mine, m, 64));
        kth = mine;
    }
    if (n_valid == kp && lane == 0) {  // full list: rows outside it exist (or may)
        const float last_approx = key_score(cand[(size_t)q * kp + kp - 1]);
        if (!(last_approx + eps < kth)) atomicOr(flag, 1);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (ek[s] != 0ull && rk[s] < k) {
            const size_t o = (size_t)q * k + rk[s];
            if (out_keys) out_keys[o] = ek[s];
            if (out_idx) { out_idx[o] = (int64_t)key_row(ek[s]); out_score[o] = key_score(ek[s]); }
        }
    for (int e = n_valid + lane; e < k; e += 64) {  // pads when the catalog (minus exclusions) is smaller than k
        const size_t o = (size_t)q * k + e;
        if (out_keys) out_keys[o] = 0ull;
        if (out_idx) { out_idx[o] = -1; out_score[o] = 0.0f; }
    }
}

// ---------------------------------------------------------------- k-way merge of sorted lists
// keys: [n_lists][q_stride][k] sorted descending per (list, query); one wavefront per query
// runs a tournament: every lane holds the heads of up to MERGE_LPL lists.
constexpr int MERGE_MAX_LISTS = 1024;
template <int MERGE_LPL>  // lists per lane: 4 (<= 256 lists) or 16 (<= 1024)
__global__ __launch_bounds__(256) void merge_kernel(const u64* __restrict__ keys, int n_lists, int q_stride, int Q,
                                                    int k, int64_t* __restrict__ out_idx,
                                                    float* __restrict__ out_score, u64* __restrict__ out_keys,
                                                    const int* __restrict__ run_flag = nullptr) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    if (run_flag != nullptr && *run_flag == 0) return;
    u64 head[MERGE_LPL];
    int pos[MERGE_LPL];
#pragma unroll
    for (int s = 0; s < MERGE_LPL; ++s) {
        const int c = lane + 64 * s;
        pos[s] = 0;
        head[s] = c < n_lists ? keys[((size_t)c * q_stride + q) * k] : 0ull;
    }
    for (int e = 0; e < k; ++e) {
        u64 best = 0ull;
#pragma unroll
        for (int s = 0; s < MERGE_LPL; ++s) best = head[s] > best ? head[s] : best;
        const u64 w = wave_max_u64(best);
        if (w != 0ull) {
#pragma unroll
            for (int s = 0; s < MERGE_LPL; ++s) {
                if (head[s] == w) {  // keys are unique: exactly one (lane, s) advances
                    const int c = lane + 64 * s;
                    ++pos[s];
                    head[s] = pos[s] < k ? keys[((size_t)c * q_stride + q) * k + pos[s]] : 0ull;
                }
            }
        }
        if (lane == 0) {
            if (out_keys) out_keys[(size_t)q * k + e] = w;
            if (out_idx) {
                out_idx[(size_t)q * k + e] = w ? (int64_t)key_row(w) : -1;
                out_score[(size_t)q * k + e] = w ? key_score(w) : 0.0f;
            }
        }
    }
}

// Few queries, few lists (a single request: ~200 partial lists of k): ONE global round trip and three short LDS passes
// instead of k dependent rounds (merge_kernel's every round waits for a global load behind a 12-shuffle wave maximum:
// 12 us of a 0.35 ms request at 195 lists x 20).  A 256-thread workgroup per query:
//   1. all n_lists * k keys (<= MERGE_BLOCK_KEYS) -> LDS, coalesced;
//   2. h = the k-th largest list HEAD (rank by counting over the <= 256 heads): at least k keys are >= h, so the k
//      best keys are all >= h, and only lists whose head is >= h (at most k of them) hold any;
//   3. those lists hand their keys >= h to a candidate array (they are sorted: stop at the first smaller one);
//   4. every candidate's rank by counting; rank r < k goes to output r.  Keys are unique (score bits | row) and 0 is
//      the empty pad, so ranks are exact: the same (score desc, row asc) order and outputs as merge_kernel.
constexpr int MERGE_BLOCK_KEYS = 16 * 256;
__global__ __launch_bounds__(256) void merge_block_kernel(const u64* __restrict__ keys, int n_lists, int q_stride, int Q,
                                                          int k, int64_t* __restrict__ out_idx,
                                                          float* __restrict__ out_score, u64* __restrict__ out_keys) {
    __shared__ __attribute__((aligned(16))) u64 all[MERGE_BLOCK_KEYS];
    __shared__ uint16_t cand[MERGE_BLOCK_KEYS];  // candidates as indices into `all`
    __shared__ u64 h_s;
    __shared__ int ncand;
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const int n_keys = n_lists * k;
    if (tid == 0) { h_s = 0ull; ncand = 0; }
#pragma unroll
    for (int s = 0; s < MERGE_BLOCK_KEYS / 256; ++s) {
        const int f = tid + 256 * s;
        if (f < n_keys) {
            const int c = f / k, e = f - c * k;
            all[f] = keys[((size_t)c * q_stride + q) * k + e];
        }
    }
    __syncthreads();
    const u64 head = tid < n_lists ? all[tid * k] : 0ull;
    if (head != 0ull) {
        int r = 0;
        for (int i = 0; i < n_lists; ++i) r += all[i * k] > head ? 1 : 0;  // LDS broadcast reads
        if (r == k - 1) h_s = head;  // unique keys: at most one thread
    }
    __syncthreads();
    const u64 h = h_s;  // 0 when fewer than k lists are non-empty: every key is a candidate
    if (head != 0ull && head >= h) {
        for (int e = 0; e < k; ++e) {
            const u64 v = all[tid * k + e];
            if (v == 0ull || v < h) break;
            cand[atomicAdd(&ncand, 1)] = (uint16_t)(tid * k + e);
        }
    }
    __syncthreads();
    const int C = ncand;
    for (int ci = tid; ci < C; ci += 256) {
        const u64 v = all[cand[ci]];
        int r = 0;
        for (int j = 0; j < C; ++j) r += all[cand[j]] > v ? 1 : 0;
        if (r < k) {
            const size_t o = (size_t)q * k + r;
            if (out_keys) out_keys[o] = v;
            if (out_idx) { out_idx[o] = (int64_t)key_row(v); out_score[o] = key_score(v); }
        }
    }
    for (int e = (C < k ? C : k) + tid; e < k; e += 256) {  // pads: fewer than k rows left after the exclusions
        const size_t o = (size_t)q * k + e;
        if (out_keys) out_keys[o] = 0ull;
        if (out_idx) { out_idx[o] = -1; out_score[o] = 0.0f; }
    }
}

// ---------------------------------------------------------------- host side
struct Index {
    void* rows = nullptr;  // normalised [n_rows, dim], fp32 or bf16 bits
    _Float16* plane_hi = nullptr;  // ICREC_ROWS_F32_FILTER: f16 hi/lo planes of `rows` for the filter pass
    _Float16* plane_lo = nullptr;
    _Float16* frag = nullptr;      // resident filter pass (dim 384, <= RES_MAX_ROWS rows): the rows as packed fragments
    int64_t frag_row_tiles = 0;    // 32-row tiles in `frag` (whole rounds of CfgRes::BM rows)
    int storage = ICREC_ROWS_F32;
    int64_t n_rows = 0;
    int dim = 0;
    int64_t row_offset = 0;
    int device = 0;
    int n_cu = 256;
    int stream_max_q = 8;          // ICREC_STREAM_MAX_Q at creation
};

static inline bool rows_are_bf16(const Index* ix) { return ix->storage == ICREC_ROWS_BF16 || ix->storage == ICREC_ROWS_BF16_FILTER; }

typedef TileCfg<2, 2, 2, 2> CfgBig;    // 128 rows x 128 queries
typedef TileCfg<4, 1, 2, 2> CfgMid;    // 256 rows x  64 queries
typedef TileCfg<4, 1, 2, 1> CfgSmall;  // 256 rows x  32 queries
typedef TileCfg<2, 2, 2, 1> CfgFilter;  // 128 rows x  64 queries, f16 planes (filter pass)
typedef TileCfg<8, 1, 1, 2> CfgRes;     // 256 rows x  64 queries per round, 8 waves (resident filter pass)
// Filter storages of dimension 384 keep the rows as the packed fragments of the resident filter pass INSTEAD of the
// row-major planes (same bytes).  Every 64-query block streams its chunk of the shard through L2; the blocks of a
// chunk sit on one XCD (xcd_remap) and walk it in step, so HBM still sees each fragment about once.  Measured against
// the staged pass (same box, top-20): 49,688 rows x 1,024 queries 0.26 vs 0.39 ms; 2 M rows 4.5 vs 6.5 ms; 10 M rows
// 19.6 vs 30.1 ms, x 4,096 queries 74 vs 124 ms (425 TF-equivalent = 0.51 of the 3-pass f16 roof).
constexpr int64_t RES_MAX_ROWS = (int64_t)1 << 40;  // no limit (ICREC_FILTER_RESIDENT=<n> sets one, =0 forces the staged form)

// Filter + verify (ICREC_ROWS_F32_FILTER): batches of at least FILTER_MIN_Q queries are ranked by the f16x3 filter
// pass with FILTER_SLACK extra list entries, then verified exactly.  FILTER_EPS bounds |filter score - exact score|
// for unit vectors: the f16 split drops <= 3 * 2^-22 per product (Cauchy-Schwarz: <= 7.2e-7 per score) and either
// fp32 accumulation is off by at most 384 * 2^-24 = 2.3e-5 from the real dot product; 1e-4 covers the sum twice.
// Below 256 queries the pass's fixed costs (three more launches, longer lists) eat its advantage: measured at
// 49,688 rows Q=64 0.26 ms vs 0.15 ms exact, Q=256 equal, Q=1024 0.61 vs 0.82 ms; at 2M rows Q=256 2.5 vs 4.2 ms,
// Q=1024 7.5 vs 14.5 ms.
constexpr int FILTER_MIN_Q = 256, FILTER_SLACK = 12;
constexpr float FILTER_EPS = 1.0e-4f;
static inline int filter_list_len(int k) { int kp = k + FILTER_SLACK; kp = (kp + 7) & ~7; return kp; }

struct Plan {
    int variant;  // 0 big, 1 mid, 2 small
    int BM, BN, Qpad, n_qtiles, n_row_tiles, tiles_per_chunk, n_chunks;
    size_t smem;
    size_t ws_q, ws_partial, ws_total;
};

// Largest batch the streaming kernel takes (ICREC_STREAM_MAX_Q=0 disables it; tuning/diagnostic knob), read ONCE,
// when the index is created: a handle never changes its kernels between calls.
static int stream_max_q_from_env() {
    const char* e = getenv("ICREC_STREAM_MAX_Q");
    const int v = e ? atoi(e) : 8;
    return v > 8 ? 8 : v < 0 ? 0 : v;
}

static Plan make_plan(const Index* ix, int Q, int k, bool allow_stream) {
    Plan p;
    // The streaming kernel pays a per-block cold start (rank 256 keys per query by counting) that only amortises
    // over several tiles: take it for Q <= 2 always, for Q <= 8 once every block has >= 2 tiles (measured at
    // 49,688 rows, f32: Q=8 67 us streaming vs 44 us MFMA; at 2M rows 0.64 ms vs 0.76 ms).
    const int64_t st_tiles = (ix->n_rows + ST_ROWS - 1) / ST_ROWS;
    if (allow_stream && Q <= ix->stream_max_q && (Q <= 2 || st_tiles >= 2 * 3 * (int64_t)ix->n_cu)) {
        // variant 3: stream_search_kernel<NQ>, one block per chunk of 256-row tiles, no query tiling
        const int nq = Q <= 1 ? 1 : Q <= 2 ? 2 : Q <= 4 ? 4 : 8;
        p.variant = 3; p.BM = ST_ROWS; p.BN = nq;
        p.smem = nq == 1 ? StreamSmem<1>::bytes(k) : nq == 2 ? StreamSmem<2>::bytes(k) : nq == 4 ? StreamSmem<4>::bytes(k)
                                                                                                 : StreamSmem<8>::bytes(k);
        p.n_qtiles = 1;
        p.Qpad = nq;
        p.n_row_tiles = (int)((ix->n_rows + p.BM - 1) / p.BM);
        int want_chunks = 3 * ix->n_cu;  // ~3 blocks per CU are resident (LDS)
        if (want_chunks > MERGE_MAX_LISTS) want_chunks = MERGE_MAX_LISTS;
        if (want_chunks > p.n_row_tiles) want_chunks = p.n_row_tiles;
        p.tiles_per_chunk = (p.n_row_tiles + want_chunks - 1) / want_chunks;
        p.n_chunks = (p.n_row_tiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk;
    } else {
        if (Q > 64 && k <= 32) { p.variant = 0; p.BM = CfgBig::BM; p.BN = CfgBig::BN; p.smem = SearchSmem<CfgBig>::bytes(k); }
        else if (Q > 32 && k <= 64) { p.variant = 1; p.BM = CfgMid::BM; p.BN = CfgMid::BN; p.smem = SearchSmem<CfgMid>::bytes(k); }
        else { p.variant = 2; p.BM = CfgSmall::BM; p.BN = CfgSmall::BN; p.smem = SearchSmem<CfgSmall>::bytes(k); }
        p.n_qtiles = (Q + p.BN - 1) / p.BN;
        p.Qpad = p.n_qtiles * p.BN;
        p.n_row_tiles = (int)((ix->n_rows + p.BM - 1) / p.BM);
        // one full wave of resident blocks (2 per CU fit by LDS/VGPR), at most 256 chunks
        int want_chunks = (2 * ix->n_cu) / p.n_qtiles;
        if (want_chunks < 1) want_chunks = 1;
        if (want_chunks > 256) want_chunks = 256;
        if (want_chunks > p.n_row_tiles) want_chunks = p.n_row_tiles;
        p.tiles_per_chunk = (p.n_row_tiles + want_chunks - 1) / want_chunks;
        p.n_chunks = (p.n_row_tiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk;
    }
    p.ws_q = ((size_t)p.Qpad * ix->dim * 4 + 255) & ~(size_t)255;
    p.ws_partial = ((size_t)p.n_chunks * p.Qpad * k * 8 + 255) & ~(size_t)255;
    p.ws_total = p.ws_q + p.ws_partial;
    return p;
}

template <int NQ, bool P16>
static int launch_stream(const Index* ix, const Plan& p, const float* qn, int Q, int k, const int32_t* ei,
                         const int32_t* eo, u64* partial, hipStream_t st) {
    ScopedTimer tm(T_SEARCH_KERNEL, st);
    hipLaunchKernelGGL((stream_search_kernel<NQ, P16>), dim3(p.n_chunks), dim3(ST_ROWS), p.smem, st, (const void*)ix->rows,
                       ix->n_rows, ix->dim, qn, Q, k, ei, eo, (uint32_t)ix->row_offset, p.n_row_tiles, p.tiles_per_chunk,
                       partial);
    ICREC_HIP(hipGetLastError());
    return ICREC_OK;
}

template <class Cfg, bool EMIT, bool P16>
static int launch_search(const Index* ix, const Plan& p, const float* qn, int Q, int k, const int32_t* ei,
                         const int32_t* eo, u64* partial, float* scores_out, hipStream_t st,
                         const int* run_flag = nullptr) {
    auto kern = search_kernel<Cfg, EMIT, P16 ? 1 : 0>;
    if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc_;
    const int grid = p.n_chunks * p.n_qtiles;
    {
        ScopedTimer tm(run_flag == nullptr ? T_SEARCH_KERNEL : T_SEARCH_FALLBACK, st);  // the guarded pass has its own slot
        hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), p.smem, st, (const void*)ix->rows, (const void*)nullptr,
                           ix->n_rows, ix->dim, (const void*)qn, (const void*)nullptr, p.Qpad, Q, k, ei, eo,
                           (uint32_t)ix->row_offset, p.n_row_tiles, p.tiles_per_chunk, p.n_qtiles, partial, scores_out,
                           run_flag);
    }
    ICREC_HIP(hipGetLastError());
    return ICREC_OK;
}

// Workspace of the filter + verify path: [qn fp32 | q hi | q lo | flag | candidate keys | partial lists (filter pass,
// then reused by the guarded exact pass)].
struct FilterPlan {
    bool use, resident;
    int kp, Qpad, n_qtiles, n_row_tiles, tiles_per_chunk, n_chunks;
    size_t smem, off_qh, off_ql, off_flag, off_cand, off_partial, ws_total;
};

static FilterPlan make_filter_plan(const Index* ix, int Q, int k, const Plan& exact) {
    FilterPlan f;
    f.kp = filter_list_len(k);
    f.resident = ix->frag != nullptr;
    f.use = (ix->plane_hi != nullptr || f.resident) && Q >= FILTER_MIN_Q && f.kp <= ICREC_MAX_K;
    // resident form: the query planes leave 64 KB of LDS for the lists (k <= 92); longer lists take the exact search,
    // which is the faster one there anyway (measured at 49,688 rows, Q = 1,024, k = 100: staged filter 2.5 ms, exact 1.5 ms)
    if (f.resident && SearchSmem<CfgRes, 3>::bytes(f.kp) > 160 * 1024) f.use = false;
    if (!f.use) { f.ws_total = 0; return f; }
    const int BMf = f.resident ? CfgRes::BM : CfgFilter::BM;
    f.n_qtiles = (Q + CfgFilter::BN - 1) / CfgFilter::BN;  // 64 queries per tile in both forms
    f.Qpad = f.n_qtiles * CfgFilter::BN;
    f.n_row_tiles = (int)((ix->n_rows + BMf - 1) / BMf);
    // staged form: two 4-wave blocks per CU; resident form: one 8-wave block per CU (its query planes take 96 KB)
    int want_chunks = ((f.resident ? 1 : 2) * ix->n_cu) / f.n_qtiles;
    if (want_chunks < 1) want_chunks = 1;
    if (want_chunks > 256) want_chunks = 256;
    if (want_chunks > f.n_row_tiles) want_chunks = f.n_row_tiles;
    f.tiles_per_chunk = (f.n_row_tiles + want_chunks - 1) / want_chunks;
    f.n_chunks = (f.n_row_tiles + f.tiles_per_chunk - 1) / f.tiles_per_chunk;
    f.smem = f.resident ? SearchSmem<CfgRes, 3>::bytes(f.kp) : SearchSmem<CfgFilter, 2>::bytes(f.kp);
    const int qpad_max = f.Qpad > exact.Qpad ? f.Qpad : exact.Qpad;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    f.off_qh = up((size_t)qpad_max * ix->dim * 4);
    f.off_ql = f.off_qh + up((size_t)f.Qpad * ix->dim * 2);
    f.off_flag = f.off_ql + up((size_t)f.Qpad * ix->dim * 2);
    f.off_cand = f.off_flag + 256;
    f.off_partial = f.off_cand + up((size_t)Q * f.kp * 8);
    const size_t part_filter = (size_t)f.n_chunks * f.Qpad * f.kp * 8;
    f.ws_total = f.off_partial + up(part_filter > exact.ws_partial ? part_filter : exact.ws_partial);
    return f;
}

static int run_search(Index* ix, const float* q, int Q, int k, const int32_t* ei, const int32_t* eo, int64_t* out_idx,
                      float* out_score, u64* out_keys, float* scores_out, void* ws, size_t ws_bytes, hipStream_t st);

// Filter (f16x3 MFMA, approximate) -> merge -> verify (exact chains on the candidates) -> exact search that runs
// only if some query could not be proven.  Same outputs, bit for bit, as the exact search.
static int run_search_filtered(Index* ix, const FilterPlan& f, const Plan& ex, const float* q, int Q, int k,
                               const int32_t* ei, const int32_t* eo, int64_t* out_idx, float* out_score, u64* out_keys,
                               void* ws, hipStream_t st) {
    char* base = reinterpret_cast<char*>(ws);
    float* qn = reinterpret_cast<float*>(base);
    _Float16* qh = reinterpret_cast<_Float16*>(base + f.off_qh);
    _Float16* ql = reinterpret_cast<_Float16*>(base + f.off_ql);
    int* flag = reinterpret_cast<int*>(base + f.off_flag);
    u64* cand = reinterpret_cast<u64*>(base + f.off_cand);
    u64* partial = reinterpret_cast<u64*>(base + f.off_partial);
    const int qpad_max = f.Qpad > ex.Qpad ? f.Qpad : ex.Qpad;
    hipLaunchKernelGGL(normalize_rows_kernel<false>, dim3((qpad_max + 3) / 4), dim3(256), 0, st, q, (void*)qn, (int64_t)Q,
                       (int64_t)qpad_max, ix->dim, 1e-12f, 0);
    const size_t nq = (size_t)f.Qpad * ix->dim;
    if (f.resident) {
        hipLaunchKernelGGL(split_queries_act_kernel, dim3((unsigned)((nq / 4 + 255) / 256 < 1024 ? (nq / 4 + 255) / 256 : 1024)),
                           dim3(256), 0, st, (const float*)qn, nq / 4, qh, ql, flag);
        auto kern = search_kernel<CfgRes, false, 3>;
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc_;
        ScopedTimer tm(T_SEARCH_KERNEL, st);
        hipLaunchKernelGGL(kern, dim3(f.n_chunks * f.n_qtiles), dim3(CfgRes::THREADS), f.smem, st,
                           (const void*)ix->frag, (const void*)nullptr, ix->n_rows, ix->dim, (const void*)qh,
                           (const void*)ql, f.Qpad, Q, f.kp, ei, eo, (uint32_t)ix->row_offset, f.n_row_tiles,
                           f.tiles_per_chunk, f.n_qtiles, partial, (float*)nullptr, (const int*)nullptr);
    } else {
        hipLaunchKernelGGL(split_queries_kernel<false>, dim3((unsigned)((nq + 255) / 256 < 1024 ? (nq + 255) / 256 : 1024)),
                           dim3(256), 0, st, (const void*)qn, nq, qh, ql, flag);
        auto kern = search_kernel<CfgFilter, false, 2>;
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc_;
        ScopedTimer tm(T_SEARCH_KERNEL, st);
        hipLaunchKernelGGL(kern, dim3(f.n_chunks * f.n_qtiles), dim3(CfgFilter::THREADS), f.smem, st,
                           (const void*)ix->plane_hi, (const void*)ix->plane_lo, ix->n_rows, ix->dim, (const void*)qh,
                           (const void*)ql, f.Qpad, Q, f.kp, ei, eo, (uint32_t)ix->row_offset, f.n_row_tiles,
                           f.tiles_per_chunk, f.n_qtiles, partial, (float*)nullptr, (const int*)nullptr);
    }
    ICREC_HIP(hipGetLastError());
    hipLaunchKernelGGL(merge_kernel<4>, dim3((Q + 3) / 4), dim3(256), 0, st, partial, f.n_chunks, f.Qpad, Q, f.kp,
                       (int64_t*)nullptr, (float*)nullptr, cand, (const int*)nullptr);
    const bool h = rows_are_bf16(ix);
    if (h)
        hipLaunchKernelGGL(verify_kernel<true>, dim3((Q + 3) / 4), dim3(256), 0, st, (const void*)ix->rows, ix->dim, qn, cand, Q,
                           f.kp, k, (uint32_t)ix->row_offset, FILTER_EPS, out_idx, out_score, out_keys, flag);
    else
        hipLaunchKernelGGL(verify_kernel<false>, dim3((Q + 3) / 4), dim3(256), 0, st, (const void*)ix->rows, ix->dim, qn, cand, Q,
                           f.kp, k, (uint32_t)ix->row_offset, FILTER_EPS, out_idx, out_score, out_keys, flag);
    ICREC_HIP(hipGetLastError());
    // exact pass: every workgroup returns at once unless verify raised the flag
    int rc;
    if (h)
        rc = ex.variant == 0   ? launch_search<CfgBig, false, true>(ix, ex, qn, Q, k, ei, eo, partial, nullptr, st, flag)
             : ex.variant == 1 ? launch_search<CfgMid, false, true>(ix, ex, qn, Q, k, ei, eo, partial, nullptr, st, flag)
                               : launch_search<CfgSmall, false, true>(ix, ex, qn, Q, k, ei, eo, partial, nullptr, st, flag);
    else
        rc = ex.variant == 0   ? launch_search<CfgBig, false, false>(ix, ex, qn, Q, k, ei, eo, partial, nullptr, st, flag)
             : ex.variant == 1 ? launch_search<CfgMid, false, false>(ix, ex, qn, Q, k, ei, eo, partial, nullptr, st, flag)
                               : launch_search<CfgSmall, false, false>(ix, ex, qn, Q, k, ei, eo, partial, nullptr, st, flag);
    if (rc != ICREC_OK) return rc;
    hipLaunchKernelGGL(merge_kernel<4>, dim3((Q + 3) / 4), dim3(256), 0, st, partial, ex.n_chunks, ex.Qpad, Q, k, out_idx,
                       out_score, out_keys, (const int*)flag);
    ICREC_HIP(hipGetLastError());
    return ICREC_OK;
}

static int run_search(Index* ix, const float* q, int Q, int k, const int32_t* ei, const int32_t* eo, int64_t* out_idx,
                      float* out_score, u64* out_keys, float* scores_out, void* ws, size_t ws_bytes, hipStream_t st) {
    ICREC_REQUIRE(ix && q, "icrec_search: NULL index or queries");
    ICREC_REQUIRE(Q >= 1, "icrec_search: n_queries must be >= 1 (got %d)", Q);
    ICREC_REQUIRE(k >= 1 && k <= ICREC_MAX_K, "icrec_search: k must be in [1, %d] (got %d)", ICREC_MAX_K, k);
    ICREC_REQUIRE((ei == nullptr) == (eo == nullptr), "icrec_search: excl_idx and excl_off must both be set or both NULL");
    if (scores_out == nullptr && (ix->plane_hi != nullptr || ix->frag != nullptr)) {
        const Plan ex = make_plan(ix, Q, k, false);
        const FilterPlan f = make_filter_plan(ix, Q, k, ex);
        if (f.use) {
            if (ws_bytes < f.ws_total || ws == nullptr) {
                set_error("icrec_search: workspace too small (%zu < %zu)", ws_bytes, f.ws_total);
                return ICREC_ENOMEM;
            }
            ICREC_HIP(hipSetDevice(ix->device));
            ScopedTimer whole(T_SEARCH, st);
            return run_search_filtered(ix, f, ex, q, Q, k, ei, eo, out_idx, out_score, out_keys, ws, st);
        }
    }
    const Plan p = make_plan(ix, Q, k, scores_out == nullptr);
    if (ws_bytes < p.ws_total || ws == nullptr) {
        set_error("icrec_search: workspace too small (%zu < %zu)", ws_bytes, p.ws_total);
        return ICREC_ENOMEM;
    }
    ICREC_HIP(hipSetDevice(ix->device));
    ScopedTimer whole(T_SEARCH, st);
    float* qn = reinterpret_cast<float*>(ws);
    u64* partial = reinterpret_cast<u64*>(reinterpret_cast<char*>(ws) + p.ws_q);
    hipLaunchKernelGGL(normalize_rows_kernel<false>, dim3((p.Qpad + 3) / 4), dim3(256), 0, st, q, (void*)qn, (int64_t)Q,
                       (int64_t)p.Qpad, ix->dim, 1e-12f, p.variant == 3 ? p.Qpad : 0);
    int rc;
#define ICREC_SEARCH_DISPATCH(EMIT, P16)                                                                            \
    (p.variant == 0   ? launch_search<CfgBig, EMIT, P16>(ix, p, qn, Q, k, ei, eo, partial, scores_out, st)          \
     : p.variant == 1 ? launch_search<CfgMid, EMIT, P16>(ix, p, qn, Q, k, ei, eo, partial, scores_out, st)          \
                      : launch_search<CfgSmall, EMIT, P16>(ix, p, qn, Q, k, ei, eo, partial, scores_out, st))
    if (p.variant == 3) {
        const bool h = rows_are_bf16(ix);
#define ICREC_STREAM_DISPATCH(NQ) \
    (h ? launch_stream<NQ, true>(ix, p, qn, Q, k, ei, eo, partial, st) : launch_stream<NQ, false>(ix, p, qn, Q, k, ei, eo, partial, st))
        rc = p.BN == 1 ? ICREC_STREAM_DISPATCH(1) : p.BN == 2 ? ICREC_STREAM_DISPATCH(2) : p.BN == 4 ? ICREC_STREAM_DISPATCH(4)
                                                                                                   : ICREC_STREAM_DISPATCH(8);
#undef ICREC_STREAM_DISPATCH
    } else if (rows_are_bf16(ix)) rc = scores_out ? ICREC_SEARCH_DISPATCH(true, true) : ICREC_SEARCH_DISPATCH(false, true);
    else rc = scores_out ? ICREC_SEARCH_DISPATCH(true, false) : ICREC_SEARCH_DISPATCH(false, false);
#undef ICREC_SEARCH_DISPATCH
    if (rc != ICREC_OK) return rc;
    if (out_idx || out_keys) {
        if (Q <= 4 && p.n_chunks <= 256 && (int64_t)p.n_chunks * k <= MERGE_BLOCK_KEYS)
            hipLaunchKernelGGL(merge_block_kernel, dim3(Q), dim3(256), 0, st, partial, p.n_chunks, p.Qpad, Q, k, out_idx,
                               out_score, out_keys);
        else if (p.n_chunks <= 256)
            hipLaunchKernelGGL(merge_kernel<4>, dim3((Q + 3) / 4), dim3(256), 0, st, partial, p.n_chunks, p.Qpad, Q, k,
                               out_idx, out_score, out_keys);
        else
            hipLaunchKernelGGL(merge_kernel<16>, dim3((Q + 3) / 4), dim3(256), 0, st, partial, p.n_chunks, p.Qpad, Q, k,
                               out_idx, out_score, out_keys);
        ICREC_HIP(hipGetLastError());
    }
    return ICREC_OK;
}

// ---------------------------------------------------------------- full ranking (offline evaluation consumers)
// The reference's ContentBasedBaseline.rank_all / compare_untrained_vs_trained (src/baselines/content_based.py:58-63,
// scripts/compare_untrained_vs_trained.py:74-85) argsort every score row completely.  One workgroup per query sorts the
// packed keys (orderable(score) << 32 | ~row: the search kernels' total order, score descending then row ascending)
// with a bitonic network: P = next power of two >= n_rows keys per query in global scratch (pads = key 0, which sorts
// last), stages with partner distance < 4,096 run on an 8,192-key segment in LDS, the rest in global memory.
constexpr int RANK_SEG = 8192;  // keys per LDS segment (64 KB)
__global__ __launch_bounds__(1024) void rank_keys_kernel(const float* __restrict__ scores, int64_t n_rows, int64_t P,
                                                         u64* __restrict__ keys) {
    const int64_t qi = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 1024)
        keys[qi * P + i] = i < n_rows ? make_key(scores[qi * n_rows + i], (uint32_t)i) : 0ull;
}

// descending bitonic compare-exchange on a[i], a[i ^ j] inside the size-k subsequence containing i
__device__ __forceinline__ void bitonic_cx(u64& lo_slot, u64& hi_slot, bool desc) {
    const u64 a = lo_slot, b = hi_slot;
    const bool swap = desc ? (a < b) : (a > b);
    lo_slot = swap ? b : a;
    hi_slot = swap ? a : b;
}

__global__ __launch_bounds__(1024) void rank_sort_kernel(u64* __restrict__ keys, int64_t P) {
    __shared__ u64 seg[RANK_SEG];
    u64* const a = keys + (int64_t)blockIdx.x * P;
    const int t = threadIdx.x;
    const int64_t seg_len = P < RANK_SEG ? P : RANK_SEG, nseg = P / seg_len;
    // the stages j = j_hi, j_hi/2, ..., 1 of size-k merges, for one segment held in LDS (partners stay inside it)
    auto local_stages = [&](int64_t base, int64_t k_lo, int64_t k_hi, int64_t j_cap) {
        for (int64_t i = t; i < seg_len; i += 1024) seg[i] = a[base + i];
        __syncthreads();
        for (int64_t k = k_lo; k <= k_hi; k <<= 1)
            for (int64_t j = (k >> 1) < j_cap ? (k >> 1) : j_cap; j >= 1; j >>= 1) {
                for (int64_t p = t; p < seg_len / 2; p += 1024) {
                    const int64_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                    u64 x = seg[i], y = seg[i | j];
                    bitonic_cx(x, y, ((base + i) & k) == 0);
                    seg[i] = x;
                    seg[i | j] = y;
                }
                __syncthreads();
            }
        for (int64_t i = t; i < seg_len; i += 1024) a[base + i] = seg[i];
        __syncthreads();
    };
    for (int64_t sidx = 0; sidx < nseg; ++sidx) local_stages(sidx * seg_len, 2, seg_len, seg_len / 2);  // k <= seg_len
    for (int64_t k = seg_len * 2; k <= P; k <<= 1) {
        for (int64_t j = k >> 1; j >= seg_len; j >>= 1) {  // partners in different segments: global memory
            for (int64_t p = t; p < P / 2; p += 1024) {
                const int64_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                u64 x = a[i], y = a[i | j];
                bitonic_cx(x, y, (i & k) == 0);
                a[i] = x;
                a[i | j] = y;
            }
            __syncthreads();  // one workgroup owns the row; the barrier orders its global writes for its own reads
        }
        for (int64_t sidx = 0; sidx < nseg; ++sidx) local_stages(sidx * seg_len, k, k, seg_len / 2);
    }
}

__global__ __launch_bounds__(256) void rank_emit_kernel(const u64* __restrict__ keys, int64_t n_rows, int64_t P,
                                                        int64_t row_offset, int64_t* __restrict__ out) {
    const int64_t qi = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * 256)
        out[qi * n_rows + i] = row_offset + (int64_t)key_row(keys[qi * P + i]);
}

static int64_t rank_pow2(int64_t n) {
    int64_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

}  // namespace icrec

using namespace icrec;

extern "C" {

int icrec_index_create_ex(const float* rows_dev, int64_t n_rows, int32_t dim, int64_t row_offset, int device,
                          int32_t storage, icrec_index** out) {
    ICREC_REQUIRE(rows_dev && out, "icrec_index_create: NULL argument");
    ICREC_REQUIRE(n_rows >= 1, "icrec_index_create: n_rows must be >= 1");
    ICREC_REQUIRE(dim >= BK && dim % BK == 0 && dim <= 4096, "icrec_index_create: dim must be a multiple of %d (got %d)", BK, dim);
    ICREC_REQUIRE(row_offset >= 0 && row_offset + n_rows < 0xFFFFFFFFll, "icrec_index_create: row_offset + n_rows must be < 2^32-1");
    ICREC_REQUIRE(storage >= ICREC_ROWS_F32 && storage <= ICREC_ROWS_BF16_FILTER,
                  "icrec_index_create: storage must be one of ICREC_ROWS_F32 (0), _BF16 (1), _F32_FILTER (2), _BF16_FILTER (3), got %d", storage);
    const bool with_planes = storage == ICREC_ROWS_F32_FILTER || storage == ICREC_ROWS_BF16_FILTER;
    const bool rows16 = storage == ICREC_ROWS_BF16 || storage == ICREC_ROWS_BF16_FILTER;
    ICREC_REQUIRE(!with_planes || dim % HBK == 0, "icrec_index_create: the filter planes need dim %% %d == 0 (got %d)", HBK, dim);
    ICREC_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ICREC_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("icrec_index_create: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
        return ICREC_ENODEV;
    }
    Index* ix = new Index();
    ix->n_rows = n_rows; ix->dim = dim; ix->row_offset = row_offset; ix->device = device; ix->storage = storage;
    ix->n_cu = prop.multiProcessorCount;
    ix->stream_max_q = stream_max_q_from_env();
    const size_t bytes = (size_t)n_rows * dim * (rows16 ? 2 : 4);
    hipError_t e = hipMalloc(&ix->rows, bytes);
    if (e != hipSuccess) {
        delete ix;
        set_error("icrec_index_create: hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return ICREC_ENOMEM;
    }
    const dim3 grid((unsigned)((n_rows + 3) / 4));
    if (rows16)
        hipLaunchKernelGGL(normalize_rows_kernel<true>, grid, dim3(256), 0, 0, rows_dev, ix->rows, n_rows, n_rows, dim, 1e-12f);
    else
        hipLaunchKernelGGL(normalize_rows_kernel<false>, grid, dim3(256), 0, 0, rows_dev, ix->rows, n_rows, n_rows, dim, 1e-12f);
    ICREC_HIP(hipGetLastError());
    const char* res_env = getenv("ICREC_FILTER_RESIDENT");  // "0": staged form; a number > 1: row limit of the resident form (A/B)
    const int64_t res_max = res_env && atoll(res_env) > 1 ? atoll(res_env) : RES_MAX_ROWS;
    if (with_planes && dim == 16 * RES_KS && n_rows <= res_max && !(res_env && res_env[0] == '0' && res_env[1] == 0)) {
        // resident filter pass: packed fragments instead of the row-major planes
        ix->frag_row_tiles = ((n_rows + CfgRes::BM - 1) / CfgRes::BM) * (CfgRes::BM / 32);
        const int64_t n_frag = ix->frag_row_tiles * RES_KS;
        if (hipMalloc(&ix->frag, (size_t)n_frag * 2 * WT_FRAG * sizeof(_Float16)) != hipSuccess) {
            hipFree(ix->rows);
            delete ix;
            set_error("icrec_index_create: hipMalloc of the filter fragments (%zu bytes) failed", (size_t)n_frag * 2 * WT_FRAG * 2);
            return ICREC_ENOMEM;
        }
        if (rows16)  // fragments of the ROUNDED rows: the filter then approximates exactly what the exact pass computes
            hipLaunchKernelGGL(pack_rows_kernel<true>, dim3(4096), dim3(256), 0, 0, (const void*)ix->rows, n_rows, dim, n_frag, ix->frag);
        else
            hipLaunchKernelGGL(pack_rows_kernel<false>, dim3(4096), dim3(256), 0, 0, (const void*)ix->rows, n_rows, dim, n_frag, ix->frag);
        ICREC_HIP(hipGetLastError());
    } else if (with_planes) {
        const size_t n = (size_t)n_rows * dim;
        hipError_t e1 = hipMalloc(&ix->plane_hi, n * 2), e2 = hipMalloc(&ix->plane_lo, n * 2);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            hipFree(ix->plane_hi); hipFree(ix->plane_lo); hipFree(ix->rows);
            delete ix;
            set_error("icrec_index_create: hipMalloc of the filter planes (2 x %zu bytes) failed", n * 2);
            return ICREC_ENOMEM;
        }
        if (rows16)  // planes of the ROUNDED rows: the filter then approximates exactly what the exact pass computes
            hipLaunchKernelGGL(split_queries_kernel<true>, dim3(4096), dim3(256), 0, 0, (const void*)ix->rows, n, ix->plane_hi,
                               ix->plane_lo, (int*)nullptr);
        else
            hipLaunchKernelGGL(split_queries_kernel<false>, dim3(4096), dim3(256), 0, 0, (const void*)ix->rows, n, ix->plane_hi,
                               ix->plane_lo, (int*)nullptr);
        ICREC_HIP(hipGetLastError());
    }
    ICREC_HIP(hipStreamSynchronize(0));
    *out = reinterpret_cast<icrec_index*>(ix);
    return ICREC_OK;
}

int icrec_index_create(const float* rows_dev, int64_t n_rows, int32_t dim, int64_t row_offset, int device,
                       icrec_index** out) {
    return icrec_index_create_ex(rows_dev, n_rows, dim, row_offset, device, ICREC_ROWS_F32, out);
}

int32_t icrec_index_storage(const icrec_index* h) { return h ? reinterpret_cast<const Index*>(h)->storage : -1; }
int32_t icrec_index_dim(const icrec_index* h) { return h ? reinterpret_cast<const Index*>(h)->dim : 0; }
int32_t icrec_index_device(const icrec_index* h) { return h ? reinterpret_cast<const Index*>(h)->device : -1; }

int icrec_index_destroy(icrec_index* h) {
    Index* ix = reinterpret_cast<Index*>(h);
    if (!ix) return ICREC_OK;
    hipSetDevice(ix->device);
    hipFree(ix->rows);
    hipFree(ix->plane_hi);
    hipFree(ix->plane_lo);
    hipFree(ix->frag);
    delete ix;
    return ICREC_OK;
}

int64_t icrec_index_rows(const icrec_index* h) { return h ? reinterpret_cast<const Index*>(h)->n_rows : 0; }
int64_t icrec_index_row_offset(const icrec_index* h) { return h ? reinterpret_cast<const Index*>(h)->row_offset : 0; }

int icrec_index_export(const icrec_index* h, float* rows_dev, void* stream) {
    const Index* ix = reinterpret_cast<const Index*>(h);
    ICREC_REQUIRE(ix && rows_dev, "icrec_index_export: NULL argument");
    const int64_t n = ix->n_rows * ix->dim;
    if (rows_are_bf16(ix)) {
        ICREC_HIP(hipSetDevice(ix->device));
        hipLaunchKernelGGL(widen_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const uint16_t*>(ix->rows), rows_dev, n);
        ICREC_HIP(hipGetLastError());
    } else {
        ICREC_HIP(hipMemcpyAsync(rows_dev, ix->rows, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return ICREC_OK;
}

size_t icrec_search_workspace_bytes(const icrec_index* h, int32_t n_queries, int32_t k) {
    const Index* ix = reinterpret_cast<const Index*>(h);
    if (!ix || n_queries < 1 || k < 1 || k > ICREC_MAX_K) return 0;
    const Plan ex = make_plan(ix, n_queries, k, false);
    const size_t a = ex.ws_total, b = make_plan(ix, n_queries, k, true).ws_total;
    const size_t c = make_filter_plan(ix, n_queries, k, ex).ws_total;
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

int icrec_search(icrec_index* h, const float* q_dev, int32_t n_queries, int32_t k, const int32_t* excl_idx_dev,
                 const int32_t* excl_off_dev, int64_t* out_idx_dev, float* out_score_dev, void* ws, size_t ws_bytes,
                 void* stream) {
    ICREC_REQUIRE(out_idx_dev && out_score_dev, "icrec_search: NULL output");
    return run_search(reinterpret_cast<Index*>(h), q_dev, n_queries, k, excl_idx_dev, excl_off_dev, out_idx_dev,
                      out_score_dev, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

int icrec_search_partial(icrec_index* h, const float* q_dev, int32_t n_queries, int32_t k, const int32_t* excl_idx_dev,
                         const int32_t* excl_off_dev, uint64_t* out_keys_dev, void* ws, size_t ws_bytes, void* stream) {
    ICREC_REQUIRE(out_keys_dev, "icrec_search_partial: NULL output");
    return run_search(reinterpret_cast<Index*>(h), q_dev, n_queries, k, excl_idx_dev, excl_off_dev, nullptr, nullptr,
                      reinterpret_cast<u64*>(out_keys_dev), nullptr, ws, ws_bytes, (hipStream_t)stream);
}

int icrec_scores(icrec_index* h, const float* q_dev, int32_t n_queries, float* out_dev, void* ws, size_t ws_bytes,
                 void* stream) {
    ICREC_REQUIRE(out_dev, "icrec_scores: NULL output");
    return run_search(reinterpret_cast<Index*>(h), q_dev, n_queries, 1, nullptr, nullptr, nullptr, nullptr, nullptr,
                      out_dev, ws, ws_bytes, (hipStream_t)stream);
}

size_t icrec_rank_all_workspace_bytes(const icrec_index* h, int32_t n_queries) {
    const Index* ix = reinterpret_cast<const Index*>(h);
    if (!ix || n_queries < 1) return 0;
    const size_t sw = icrec_search_workspace_bytes(h, n_queries, 1);
    if (sw == 0) return 0;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return al((size_t)n_queries * ix->n_rows * 4) + al((size_t)n_queries * rank_pow2(ix->n_rows) * 8) + al(sw);
}

int icrec_rank_all(icrec_index* h, const float* q_dev, int32_t n_queries, int64_t* out_rows_dev, void* ws,
                   size_t ws_bytes, void* stream) {
    Index* ix = reinterpret_cast<Index*>(h);
    ICREC_REQUIRE(ix && q_dev && out_rows_dev && n_queries >= 1, "icrec_rank_all: bad argument");
    const size_t need = icrec_rank_all_workspace_bytes(h, n_queries);
    if (!ws || ws_bytes < need || need == 0) {
        set_error("icrec_rank_all: workspace too small (%zu < %zu)", ws_bytes, need);
        return ICREC_ENOMEM;
    }
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int64_t P = rank_pow2(ix->n_rows);
    char* base = reinterpret_cast<char*>(ws);
    float* scores = reinterpret_cast<float*>(base);
    u64* keys = reinterpret_cast<u64*>(base + al((size_t)n_queries * ix->n_rows * 4));
    char* sws = reinterpret_cast<char*>(keys) + al((size_t)n_queries * P * 8);
    if (int rc = icrec_scores(h, q_dev, n_queries, scores, sws, ws_bytes - (size_t)(sws - base), stream)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const unsigned gx = (unsigned)((P + 1023) / 1024 < 64 ? (P + 1023) / 1024 : 64);
    hipLaunchKernelGGL(rank_keys_kernel, dim3(gx, n_queries), dim3(1024), 0, st, scores, ix->n_rows, P, keys);
    hipLaunchKernelGGL(rank_sort_kernel, dim3(n_queries), dim3(1024), 0, st, keys, P);
    hipLaunchKernelGGL(rank_emit_kernel, dim3(gx * 4, n_queries), dim3(256), 0, st, keys, ix->n_rows, P, ix->row_offset,
                       out_rows_dev);
    ICREC_HIP(hipGetLastError());
    return ICREC_OK;
}

int icrec_merge_topk(const uint64_t* keys_dev, int32_t n_lists, int32_t n_queries, int32_t k, int64_t* out_idx_dev,
                     float* out_score_dev, int device, void* stream) {
    ICREC_REQUIRE(keys_dev && out_idx_dev && out_score_dev, "icrec_merge_topk: NULL argument");
    ICREC_REQUIRE(n_lists >= 1 && n_lists <= MERGE_MAX_LISTS, "icrec_merge_topk: n_lists must be in [1, %d]", MERGE_MAX_LISTS);
    ICREC_REQUIRE(n_queries >= 1 && k >= 1 && k <= ICREC_MAX_K, "icrec_merge_topk: bad n_queries/k");
    ICREC_HIP(hipSetDevice(device));
    if (n_lists <= 256)
        hipLaunchKernelGGL(merge_kernel<4>, dim3((n_queries + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const u64*>(keys_dev), n_lists, n_queries, n_queries, k, out_idx_dev,
                           out_score_dev, (u64*)nullptr);
    else
        hipLaunchKernelGGL(merge_kernel<16>, dim3((n_queries + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const u64*>(keys_dev), n_lists, n_queries, n_queries, k, out_idx_dev,
                           out_score_dev, (u64*)nullptr);
    ICREC_HIP(hipGetLastError());
    return ICREC_OK;
}

int icrec_normalize_rows(const float* x_dev, float* out_dev, int64_t n_rows, int32_t dim, float eps, int device,
                         void* stream) {
    ICREC_REQUIRE(x_dev && out_dev && n_rows >= 1 && dim >= 1, "icrec_normalize_rows: bad argument");
    ICREC_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(normalize_rows_kernel<false>, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       x_dev, (void*)out_dev, n_rows, n_rows, dim, eps);
    ICREC_HIP(hipGetLastError());
    return ICREC_OK;
}

}  // extern "C"
