// common.h — host-side error plumbing and device helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/icrec.h"

namespace icrec {

void set_error(const char* fmt, ...);

#define ICREC_HIP(call)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ::icrec::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                               __LINE__);                                                    \
            return ICREC_EHIP;                                                               \
        }                                                                                    \
    } while (0)

#define ICREC_REQUIRE(cond, ...)               \
    do {                                       \
        if (!(cond)) {                         \
            ::icrec::set_error(__VA_ARGS__);   \
            return ICREC_EINVAL;               \
        }                                      \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device function attribute: set it once per (kernel,
// device) under a lock (a second device in the same process, or two threads racing on the first launch, would
// otherwise launch with the 64 KB default and fail).  Uses the calling thread's current device.
int ensure_dynamic_lds(const void* kernel, int bytes);

// hipEvent-based per-kernel timing on the launch stream (bench.py's roofline leg).
struct TimingSlot {
    double total_ms = 0.0;
    int64_t n = 0;
};
enum { T_SEARCH_KERNEL = 0, T_FFN_UP = 1, T_ENCODE = 2, T_SEARCH = 3, T_SEARCH_FALLBACK = 4, T_NSLOTS = 8 };
bool timing_on();
// Records [start, stop] around a launch when timing is enabled; resolved lazily at query time.
struct ScopedTimer {
    ScopedTimer(int slot, hipStream_t s);
    ~ScopedTimer();
    int slot;
    hipStream_t stream;
    hipEvent_t a = nullptr, b = nullptr;
};

typedef unsigned long long u64;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef __HIPCC__

// ------------------------------------------------------------------ ranking keys
// key = (orderable(score) << 32) | (0xFFFFFFFF - row): a larger key is a better hit under
// (score descending, row ascending).  key 0 is the "empty" pad (no real key is 0).
__device__ __forceinline__ u64 make_key(float s, uint32_t row) {
    uint32_t u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((u64)u << 32) | (u64)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ float key_score(u64 key) {
    uint32_t u = (uint32_t)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint32_t key_row(u64 key) { return 0xFFFFFFFFu - (uint32_t)key; }

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src_lane) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src_lane, 64);
    hi = __shfl(hi, src_lane, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        u64 o = shfl_xor_u64(v, m);
        v = o > v ? o : v;
    }
    return v;
}
// The fixed reduction order shared with oracle/icrec_oracle.c:wave_sum — butterfly xor 32..1.
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}

// ------------------------------------------------------------------ fp32 MFMA tile engine
// C[BM x BN] = A[BM rows, K] . B[BN rows, K]^T with v_mfma_f32_32x32x2_f32.  Each output is the
// exact k-ascending chain acc = fmaf(a[k], b[k], acc) from acc = 0 (gfx950's f32 MFMA is a
// k-ordered fmaf chain with one rounding per product), so results are bit-identical to the
// oracle's loops.
//
// Both operands are row-major with K contiguous ("NT" GEMM).  K is walked in slabs of BK = 32
// floats = one 128-B line per row, staged global -> registers -> LDS.  In LDS each group of 8
// consecutive k is stored as [k0 k2 k4 k6 | k1 k3 k5 k7] so that one ds_read_b128 gives a lane
// the four values it feeds to four consecutive MFMA steps (lanes 0-31 supply the even k of a
// step, lanes 32-63 the odd k).  Row stride 36 floats keeps ds_read_b128 conflict-free.
constexpr int BK = 32;
constexpr int LDK = 36;

template <int WAVES_M_, int WAVES_N_, int TM_, int TN_>
struct TileCfg {
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, TM = TM_, TN = TN_;
    static constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    static constexpr int THREADS = WAVES_M * WAVES_N * 64;
    static constexpr int A_CHUNKS = BM * 8 / THREADS;  // float4 chunks per thread per slab
    static constexpr int B_CHUNKS = BN * 8 / THREADS;
    static constexpr int LDS_FLOATS = (BM + BN) * LDK;
    static_assert(BM * 8 % THREADS == 0 && BN * 8 % THREADS == 0, "tile/threads mismatch");
};

// A16: operand A (the catalog rows in the search kernel) is stored as bfloat16; a thread's chunk is
// then 16 B = 8 consecutive k (half as many chunks), widened to fp32 when it is written to LDS.
template <class Cfg, bool A16 = false>
struct TileRegs {
    static constexpr int A_N = A16 ? (Cfg::A_CHUNKS + 1) / 2 : Cfg::A_CHUNKS;
    float4 a[A_N];
    float4 b[Cfg::B_CHUNKS];
};

// Issue the global loads of slab `slab` (rows clamped to the last valid row).
template <class Cfg, bool A16 = false>
__device__ __forceinline__ void tile_load(TileRegs<Cfg, A16>& r, const void* __restrict__ Av, int64_t a_row0,
                                          int64_t a_rows, const float* __restrict__ B, int64_t b_row0,
                                          int64_t b_rows, int K, int slab) {
    const int t = threadIdx.x;
    if (A16) {
        static_assert(!A16 || Cfg::BM * 4 % Cfg::THREADS == 0, "tile/threads mismatch (bf16 rows)");
        const uint16_t* A = static_cast<const uint16_t*>(Av);
#pragma unroll
        for (int i = 0; i < TileRegs<Cfg, A16>::A_N; ++i) {
            int id = t + Cfg::THREADS * i;
            int64_t row = a_row0 + (id >> 2);
            row = row < a_rows ? row : a_rows - 1;
            r.a[i] = *reinterpret_cast<const float4*>(A + row * K + slab * BK + (id & 3) * 8);  // raw bits
        }
    } else {
        const float* A = static_cast<const float*>(Av);
#pragma unroll
        for (int i = 0; i < TileRegs<Cfg, A16>::A_N; ++i) {
            int id = t + Cfg::THREADS * i;
            int64_t row = a_row0 + (id >> 3);
            row = row < a_rows ? row : a_rows - 1;
            r.a[i] = *reinterpret_cast<const float4*>(A + row * K + slab * BK + (id & 7) * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_CHUNKS; ++i) {
        int id = t + Cfg::THREADS * i;
        int64_t row = b_row0 + (id >> 3);
        row = row < b_rows ? row : b_rows - 1;
        r.b[i] = *reinterpret_cast<const float4*>(B + row * K + slab * BK + (id & 7) * 4);
    }
}

// Write the staged slab into LDS in the even/odd-split order.
template <class Cfg, bool A16 = false>
__device__ __forceinline__ void tile_store_lds(const TileRegs<Cfg, A16>& r, float* As, float* Bs) {
    const int t = threadIdx.x;
    if (A16) {
#pragma unroll
        for (int i = 0; i < TileRegs<Cfg, A16>::A_N; ++i) {
            int id = t + Cfg::THREADS * i;
            int row = id >> 2, c = id & 3;
            // word w holds k = 2w (low half) and k = 2w+1 (high half); bf16 -> fp32 is a 16-bit shift
            const unsigned w0 = __float_as_uint(r.a[i].x), w1 = __float_as_uint(r.a[i].y),
                           w2 = __float_as_uint(r.a[i].z), w3 = __float_as_uint(r.a[i].w);
            float* p = As + row * LDK + c * 8;
            *reinterpret_cast<float4*>(p) = make_float4(__uint_as_float(w0 << 16), __uint_as_float(w1 << 16),
                                                        __uint_as_float(w2 << 16), __uint_as_float(w3 << 16));
            *reinterpret_cast<float4*>(p + 4) =
                make_float4(__uint_as_float(w0 & 0xFFFF0000u), __uint_as_float(w1 & 0xFFFF0000u),
                            __uint_as_float(w2 & 0xFFFF0000u), __uint_as_float(w3 & 0xFFFF0000u));
        }
    } else {
#pragma unroll
        for (int i = 0; i < TileRegs<Cfg, A16>::A_N; ++i) {
            int id = t + Cfg::THREADS * i;
            int row = id >> 3, c = id & 7;
            float* p = As + row * LDK + (c >> 1) * 8 + (c & 1) * 2;
            *reinterpret_cast<float2*>(p) = make_float2(r.a[i].x, r.a[i].z);      // even k
            *reinterpret_cast<float2*>(p + 4) = make_float2(r.a[i].y, r.a[i].w);  // odd k
        }
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_CHUNKS; ++i) {
        int id = t + Cfg::THREADS * i;
        int row = id >> 3, c = id & 7;
        float* p = Bs + row * LDK + (c >> 1) * 8 + (c & 1) * 2;
        *reinterpret_cast<float2*>(p) = make_float2(r.b[i].x, r.b[i].z);
        *reinterpret_cast<float2*>(p + 4) = make_float2(r.b[i].y, r.b[i].w);
    }
}

// One slab of MFMAs for this wave's TM x TN block of 32x32 tiles.
template <class Cfg>
__device__ __forceinline__ void tile_mma(f32x16 (&acc)[Cfg::TM][Cfg::TN], const float* As, const float* Bs,
                                         int wm, int wn, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        float4 af[Cfg::TM], bf[Cfg::TN];
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
            af[i] = *reinterpret_cast<const float4*>(As + ((wm * Cfg::TM + i) * 32 + r) * LDK + kq * 8 + h * 4);
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
            bf[j] = *reinterpret_cast<const float4*>(Bs + ((wn * Cfg::TN + j) * 32 + r) * LDK + kq * 8 + h * 4);
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
            }
    }
}

// Full K loop for one output tile: single LDS buffer, register prefetch of the next slab
// under the current slab's MFMAs.  `pre` must already hold slab 0 on entry when
// `preloaded` is true (lets a caller overlap the first loads with its own epilogue).
template <class Cfg, bool A16 = false>
__device__ __forceinline__ void tile_gemm(f32x16 (&acc)[Cfg::TM][Cfg::TN], const void* __restrict__ A,
                                          int64_t a_row0, int64_t a_rows, const float* __restrict__ B,
                                          int64_t b_row0, int64_t b_rows, int K, float* As, float* Bs,
                                          TileRegs<Cfg, A16>& pre, bool preloaded) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    const int nslab = K / BK;
    if (!preloaded) tile_load<Cfg, A16>(pre, A, a_row0, a_rows, B, b_row0, b_rows, K, 0);
    for (int s = 0; s < nslab; ++s) {
        __syncthreads();  // previous slab's LDS reads are done
        tile_store_lds<Cfg, A16>(pre, As, Bs);
        __syncthreads();
        if (s + 1 < nslab) tile_load<Cfg, A16>(pre, A, a_row0, a_rows, B, b_row0, b_rows, K, s + 1);
        tile_mma<Cfg>(acc, As, Bs, wm, wn, lane);
    }
}

// Row of C held in acc[i][j][e] for this lane: (C/D map of the 32x32 f32 MFMA)
__device__ __forceinline__ int acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each
// XCD a contiguous range of logical ids.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nblocks >> 3, r = nblocks & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

#endif  // __HIPCC__

}  // namespace icrec
