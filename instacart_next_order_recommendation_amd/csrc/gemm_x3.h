// gemm_x3.h — fp32-accurate GEMM on the f16 matrix cores by operand splitting.
//
// gfx950 has no TF32/xf32 MFMA; its exact f32 MFMA runs at 1/16 of the f16 rate.  Every fp32
// operand x is therefore carried as two f16 planes
//     hi = f16(x)                     lo = f16((x - hi) * 2^11)
// (x - hi is exact in fp32; |x - hi - lo*2^-11| <= 2^-22 |x|), and a product is evaluated as
//     a*w  ~=  a_hi*w_hi + 2^-11 * (a_hi*w_lo + a_lo*w_hi)
// with v_mfma_f32_32x32x16_f16: f16 x f16 products are exact in the fp32 accumulator, the
// dropped a_lo*w_lo term and the split residuals are <= 3 * 2^-22 relative per product, i.e.
// the same order as fp32 accumulation noise.  Three f16 MFMAs replace sixteen f32-MFMA-cycles:
// 5.3x the fp32-MFMA roofline for the same algorithmic FLOPs.  The lo planes are scaled by
// 2^11 so that no value depends on f16 subnormals; the cross terms accumulate separately and
// are scaled back once in the epilogue.
//
// Tiles: 128 B (64 halfs) of K per row per slab — byte-for-byte the geometry of the fp32 engine
// (common.h), so staging pattern, LDS row stride (144 B) and bank behaviour are the same.
#pragma once
#include "common.h"

namespace icrec {

#ifdef __HIPCC__

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HBK = 64;   // halfs of K per slab
constexpr int HLD = 72;   // LDS row stride in halfs (144 B)
constexpr float LO_SCALE = 2048.0f;          // 2^11
constexpr float LO_UNSCALE = 1.0f / 2048.0f;

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * LO_SCALE);
}

template <class Cfg>
struct TileRegsH {
    u32x4 ah[Cfg::A_CHUNKS], al[Cfg::A_CHUNKS], bh[Cfg::B_CHUNKS], bl[Cfg::B_CHUNKS];
};

template <class Cfg>
struct SmemH {
    static constexpr int A_HALFS = Cfg::BM * HLD, B_HALFS = Cfg::BN * HLD;
    static constexpr size_t BYTES = (size_t)(2 * A_HALFS + 2 * B_HALFS) * 2;
};

// planes are [rows, K] f16 row-major
template <class Cfg>
__device__ __forceinline__ void tile_load_h(TileRegsH<Cfg>& r, const _Float16* __restrict__ Ah,
                                            const _Float16* __restrict__ Al, int64_t a_row0, int64_t a_rows,
                                            const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl,
                                            int64_t b_row0, int64_t b_rows, int K, int slab) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < Cfg::A_CHUNKS; ++i) {
        const int id = t + Cfg::THREADS * i;
        int64_t row = a_row0 + (id >> 3);
        row = row < a_rows ? row : a_rows - 1;
        const int64_t off = row * K + slab * HBK + (id & 7) * 8;
        r.ah[i] = *reinterpret_cast<const u32x4*>(Ah + off);
        r.al[i] = *reinterpret_cast<const u32x4*>(Al + off);
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_CHUNKS; ++i) {
        const int id = t + Cfg::THREADS * i;
        int64_t row = b_row0 + (id >> 3);
        row = row < b_rows ? row : b_rows - 1;
        const int64_t off = row * K + slab * HBK + (id & 7) * 8;
        r.bh[i] = *reinterpret_cast<const u32x4*>(Bh + off);
        r.bl[i] = *reinterpret_cast<const u32x4*>(Bl + off);
    }
}

template <class Cfg>
__device__ __forceinline__ void tile_store_lds_h(const TileRegsH<Cfg>& r, _Float16* Ahs, _Float16* Als,
                                                 _Float16* Bhs, _Float16* Bls) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < Cfg::A_CHUNKS; ++i) {
        const int id = t + Cfg::THREADS * i;
        const int o = (id >> 3) * HLD + (id & 7) * 8;
        *reinterpret_cast<u32x4*>(Ahs + o) = r.ah[i];
        *reinterpret_cast<u32x4*>(Als + o) = r.al[i];
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_CHUNKS; ++i) {
        const int id = t + Cfg::THREADS * i;
        const int o = (id >> 3) * HLD + (id & 7) * 8;
        *reinterpret_cast<u32x4*>(Bhs + o) = r.bh[i];
        *reinterpret_cast<u32x4*>(Bls + o) = r.bl[i];
    }
}

// acc0 += A_hi.W_hi ; acc1 += A_hi.W_lo + A_lo.W_hi   over one 64-deep slab
template <class Cfg>
__device__ __forceinline__ void tile_mma_h(f32x16 (&acc0)[Cfg::TM][Cfg::TN], f32x16 (&acc1)[Cfg::TM][Cfg::TN],
                                           const _Float16* Ahs, const _Float16* Als, const _Float16* Bhs,
                                           const _Float16* Bls, int wm, int wn, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < HBK / 16; ++ks) {
        half8 ah[Cfg::TM], al[Cfg::TM], bh[Cfg::TN], bl[Cfg::TN];
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
            const int o = ((wm * Cfg::TM + i) * 32 + r) * HLD + ks * 16 + h * 8;
            ah[i] = *reinterpret_cast<const half8*>(Ahs + o);
            al[i] = *reinterpret_cast<const half8*>(Als + o);
        }
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
            const int o = ((wn * Cfg::TN + j) * 32 + r) * HLD + ks * 16 + h * 8;
            bh[j] = *reinterpret_cast<const half8*>(Bhs + o);
            bl[j] = *reinterpret_cast<const half8*>(Bls + o);
        }
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) {
                acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc1[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc1[i][j], 0, 0, 0);
            }
    }
}

// Whole-K loop for one output tile (single LDS stage, register prefetch of the next slab).
template <class Cfg>
__device__ __forceinline__ void tile_gemm_h(f32x16 (&acc0)[Cfg::TM][Cfg::TN], f32x16 (&acc1)[Cfg::TM][Cfg::TN],
                                            const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                            int64_t a_row0, int64_t a_rows, const _Float16* __restrict__ Bh,
                                            const _Float16* __restrict__ Bl, int64_t b_row0, int64_t b_rows, int K,
                                            _Float16* smem) {
    _Float16* Ahs = smem;
    _Float16* Als = Ahs + SmemH<Cfg>::A_HALFS;
    _Float16* Bhs = Als + SmemH<Cfg>::A_HALFS;
    _Float16* Bls = Bhs + SmemH<Cfg>::B_HALFS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0.0f; acc1[i][j][e] = 0.0f; }
    const int nslab = K / HBK;
    TileRegsH<Cfg> pre;
    tile_load_h<Cfg>(pre, Ah, Al, a_row0, a_rows, Bh, Bl, b_row0, b_rows, K, 0);
    for (int s = 0; s < nslab; ++s) {
        __syncthreads();
        tile_store_lds_h<Cfg>(pre, Ahs, Als, Bhs, Bls);
        __syncthreads();
        if (s + 1 < nslab) tile_load_h<Cfg>(pre, Ah, Al, a_row0, a_rows, Bh, Bl, b_row0, b_rows, K, s + 1);
        tile_mma_h<Cfg>(acc0, acc1, Ahs, Als, Bhs, Bls, wm, wn, lane);
    }
}

// Latency-bound form for small M (a single request: a handful of workgroups, each k-step waiting a full
// memory round trip): D slabs are kept in flight in registers instead of one, so a K = 384 tile costs
// ~2 round trips instead of 6.  Same LDS stage, same MFMA order => bit-identical results to tile_gemm_h.
template <class Cfg, int D>
__device__ __forceinline__ void tile_gemm_h_deep(f32x16 (&acc0)[Cfg::TM][Cfg::TN], f32x16 (&acc1)[Cfg::TM][Cfg::TN],
                                                 const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                 int64_t a_row0, int64_t a_rows, const _Float16* __restrict__ Bh,
                                                 const _Float16* __restrict__ Bl, int64_t b_row0, int64_t b_rows,
                                                 int K, _Float16* smem) {
    _Float16* Ahs = smem;
    _Float16* Als = Ahs + SmemH<Cfg>::A_HALFS;
    _Float16* Bhs = Als + SmemH<Cfg>::A_HALFS;
    _Float16* Bls = Bhs + SmemH<Cfg>::B_HALFS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0.0f; acc1[i][j][e] = 0.0f; }
    const int nslab = K / HBK;
    TileRegsH<Cfg> pre[D];
#pragma unroll
    for (int d = 0; d < D; ++d)  // slabs past the end re-read the last one (never consumed)
        tile_load_h<Cfg>(pre[d], Ah, Al, a_row0, a_rows, Bh, Bl, b_row0, b_rows, K, d < nslab ? d : nslab - 1);
    for (int s0 = 0; s0 < nslab; s0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int s = s0 + d;
            if (s < nslab) {  // block-uniform
                __syncthreads();
                tile_store_lds_h<Cfg>(pre[d], Ahs, Als, Bhs, Bls);
                __syncthreads();
                const int nxt = s + D;
                tile_load_h<Cfg>(pre[d], Ah, Al, a_row0, a_rows, Bh, Bl, b_row0, b_rows, K, nxt < nslab ? nxt : nslab - 1);
                tile_mma_h<Cfg>(acc0, acc1, Ahs, Als, Bhs, Bls, wm, wn, lane);
            }
        }
    }
}

#endif  // __HIPCC__

}  // namespace icrec
