// comm.hip — the multi-GPU exchange step of the sharded search, behind the C ABI.
//
// The reference has no distributed code (src/inference/serve_recommendations.py:172-181 only picks a device);
// north_star asks for a row-sharded catalog with per-shard partial top-k lists merged through an RCCL
// all-gather over xGMI.  One process per GPU:
//
//     ncclAllGather(query embeddings)   [Q/W, d] fp32 per rank -> [Q, d]          (786 KB per rank at configs[3])
//     icrec_search_partial              every query against this rank's shard -> keys [Q, k]
//     ncclAllGather(partial keys)       [Q, k] u64 per rank    -> [W, Q, k]       (655 KB per rank at configs[3])
//     icrec_merge_topk                  -> global top-k, bit-identical to the unsharded search
//
// all on the caller's stream, no host synchronisation, no torch.  librccl is bound at run time (dlopen of
// librccl.so.1 on the first icrec_comm_* call): a process that already loaded an RCCL build with that soname
// (PyTorch ships one) shares it instead of mapping a second copy, and a single-GPU user of libicrec.so never
// loads RCCL at all.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace icrec {

// Types and prototypes come from the RCCL header; the library itself is bound with dlopen/dlsym below.
struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.so) return ICREC_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* so = nullptr;
    for (const char* n : names) {
        so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (so) break;
    }
    if (!so) {
        set_error("icrec_comm: cannot load librccl.so.1 (%s)", dlerror());
        return ICREC_ENODEV;
    }
    Rccl r;
    r.so = so;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(so, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(so, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(so, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(so, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(so, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
        set_error("icrec_comm: librccl lacks a required symbol");
        dlclose(so);
        return ICREC_ENODEV;
    }
    g_rccl = r;
    return ICREC_OK;
}

#define ICREC_NCCL(call)                                                                              \
    do {                                                                                              \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) {                                                                     \
            ::icrec::set_error("%s failed: %s", #call,                                                \
                               g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error");     \
            return ICREC_EHIP;                                                                        \
        }                                                                                             \
    } while (0)

struct Comm {
    ncclComm_t comm = nullptr;  // NULL when world == 1 (no exchange to make)
    int rank = 0, world = 1, device = 0;
};

struct ShardWs {
    size_t q_all, keys_local, keys_all, search, total;
};
static ShardWs shard_ws(const icrec_index* idx, int n_local, int world, int k) {
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t Q = (size_t)n_local * world;
    ShardWs w;
    size_t o = 0;
    w.q_all = o;      o += al(Q * (size_t)icrec_index_dim(idx) * 4);
    w.keys_local = o; o += al(Q * (size_t)k * 8);
    w.keys_all = o;   o += al((size_t)world * Q * (size_t)k * 8);
    w.search = o;
    const size_t s = icrec_search_workspace_bytes(idx, (int32_t)Q, k);
    if (s == 0) { w.total = 0; return w; }
    o += al(s);
    w.total = o;
    return w;
}

// ---- per-rank exclusion lists -> the shard-local CSR icrec_search_partial takes (icrec_search_sharded_excl)
struct ExclWs {
    size_t off_all, rows_all, cnt, csr_off, csr_idx, inner, total;
};
static ExclWs excl_ws(const icrec_index* idx, int n_local, int world, int k, int cap) {
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t Q = (size_t)n_local * world;
    ExclWs w;
    size_t o = 0;
    w.off_all = o;  o += al((size_t)world * (n_local + 1) * 4);
    w.rows_all = o; o += al((size_t)world * cap * 4);
    w.cnt = o;      o += al(Q * 4);
    w.csr_off = o;  o += al((Q + 1) * 4);
    w.csr_idx = o;  o += al((size_t)world * cap * 4);
    w.inner = o;
    const size_t s = shard_ws(idx, n_local, world, k).total;
    if (s == 0) { w.total = 0; return w; }
    o += al(s);
    w.total = o;
    return w;
}

// Offsets arrive from other ranks over the all-gather: nothing a rank checks locally covers them.  One workgroup per
// rank rewrites its row of off_all in place as the running maximum of the offsets clamped to [0, cap]: the sequence
// is then non-decreasing inside [0, cap], every query's segment [off[i], off[i+1]) is well-formed, the segments of a
// rank are disjoint and together hold at most cap ids - so excl_fill_kernel can never write past csr_idx[world * cap]
// whatever the input was (a malformed list excludes less, it never reads or writes out of bounds: icrec.h).
__global__ __launch_bounds__(1024) void excl_sanitize_kernel(int32_t* __restrict__ off_all, int n_local, int cap) {
    __shared__ int part[16];
    __shared__ int carry_s;
    int32_t* off = off_all + (size_t)blockIdx.x * (n_local + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base <= n_local; base += 1024) {
        const int i = base + tid;
        int x = i <= n_local ? off[i] : 0;
        x = x < 0 ? 0 : (x > cap ? cap : x);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {  // inclusive max-scan inside the wave
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x = x > y ? x : y;
        }
        if (lane == 63) part[wave] = x;
        __syncthreads();
        int m = carry_s;
        for (int w = 0; w < wave; ++w) m = m > part[w] ? m : part[w];
        x = x > m ? x : m;
        if (i <= n_local) off[i] = x;
        __syncthreads();
        if (tid == 1023) carry_s = x;
        __syncthreads();
    }
}

// gathered query g = (rank r, local query i): its ids are rows_all[r][lo .. hi), lo / hi from the sanitised
// off_all[r][i], [i + 1] (non-decreasing, inside [0, cap])
__device__ __forceinline__ void excl_segment(const int32_t* __restrict__ off_all, int n_local, int cap, int g, int& r,
                                             int& lo, int& hi) {
    r = g / n_local;
    const int i = g - r * n_local;
    lo = off_all[(size_t)r * (n_local + 1) + i];
    hi = off_all[(size_t)r * (n_local + 1) + i + 1];
    lo = lo < 0 ? 0 : (lo > cap ? cap : lo);
    hi = hi < lo ? lo : (hi > cap ? cap : hi);
}

__global__ __launch_bounds__(256) void excl_count_kernel(const int32_t* __restrict__ off_all,
                                                         const int32_t* __restrict__ rows_all, int n_local, int cap,
                                                         int Q, int64_t row_lo, int64_t row_hi, int32_t* __restrict__ cnt) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= Q) return;
    int r, lo, hi;
    excl_segment(off_all, n_local, cap, g, r, lo, hi);
    int c = 0;
    for (int j = lo; j < hi; ++j) {
        const int64_t v = rows_all[(size_t)r * cap + j];
        c += (v >= row_lo && v < row_hi) ? 1 : 0;
    }
    cnt[g] = c;
}

// exclusive prefix sum of cnt[0 .. Q) -> off[0 .. Q]; one 1,024-thread workgroup, chunks of 1,024 with a carry
__global__ __launch_bounds__(1024) void excl_scan_kernel(const int32_t* __restrict__ cnt, int Q, int32_t* __restrict__ off) {
    __shared__ int part[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < Q; base += 1024) {
        const int g = base + tid;
        const int v = g < Q ? cnt[g] : 0;
        int x = v;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) part[wave] = x;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += part[w];
        const int carry = carry_s;
        if (g < Q) off[g] = carry + wbase + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + wbase + x;
        __syncthreads();
    }
    if (tid == 0) off[Q] = carry_s;
}

__global__ __launch_bounds__(256) void excl_fill_kernel(const int32_t* __restrict__ off_all,
                                                        const int32_t* __restrict__ rows_all, int n_local, int cap,
                                                        int Q, int64_t row_lo, int64_t row_hi,
                                                        const int32_t* __restrict__ off, int32_t* __restrict__ idx) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= Q) return;
    int r, lo, hi;
    excl_segment(off_all, n_local, cap, g, r, lo, hi);
    int o = off[g];
    for (int j = lo; j < hi; ++j) {  // ascending global rows in -> ascending local rows out (the search kernel bisects)
        const int64_t v = rows_all[(size_t)r * cap + j];
        if (v >= row_lo && v < row_hi) idx[o++] = (int32_t)(v - row_lo);
    }
}

// gathered (rank-major) lists -> the CSR of THIS shard: sanitise the offsets in place, count, scan, fill
static int build_local_csr(int32_t* off_all, const int32_t* rows_all, int world, int n_local, int cap, int64_t row_lo,
                           int64_t row_hi, int32_t* cnt, int32_t* csr_off, int32_t* csr_idx, hipStream_t st) {
    const int Q = n_local * world;
    hipLaunchKernelGGL(excl_sanitize_kernel, dim3(world), dim3(1024), 0, st, off_all, n_local, cap);
    hipLaunchKernelGGL(excl_count_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, (const int32_t*)off_all, rows_all,
                       n_local, cap, Q, row_lo, row_hi, cnt);
    hipLaunchKernelGGL(excl_scan_kernel, dim3(1), dim3(1024), 0, st, (const int32_t*)cnt, Q, csr_off);
    hipLaunchKernelGGL(excl_fill_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, (const int32_t*)off_all, rows_all,
                       n_local, cap, Q, row_lo, row_hi, (const int32_t*)csr_off, csr_idx);
    ICREC_HIP(hipGetLastError());
    return ICREC_OK;
}

}  // namespace icrec

using namespace icrec;

extern "C" {

int icrec_comm_unique_id(void* id_out) {
    ICREC_REQUIRE(id_out, "icrec_comm_unique_id: NULL argument");
    if (int rc = load_rccl()) return rc;
    static_assert(sizeof(ncclUniqueId) == ICREC_COMM_ID_BYTES, "ICREC_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    ICREC_NCCL(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)));
    return ICREC_OK;
}

int icrec_comm_init(const void* unique_id, int rank, int world, int device, icrec_comm** out) {
    ICREC_REQUIRE(out, "icrec_comm_init: NULL argument");
    ICREC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "icrec_comm_init: bad rank/world (%d/%d)", rank, world);
    ICREC_REQUIRE(world == 1 || unique_id, "icrec_comm_init: unique_id is required when world > 1");
    ICREC_HIP(hipSetDevice(device));
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    if (world > 1 || unique_id) {  // world == 1 with an id still initialises RCCL (used by the one-GPU test)
        if (int rc = load_rccl()) { delete c; return rc; }
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof id);
        const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
        if (r != ncclSuccess) {
            set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device,
                      g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
            delete c;
            return ICREC_EHIP;
        }
    }
    *out = reinterpret_cast<icrec_comm*>(c);
    return ICREC_OK;
}

int icrec_comm_destroy(icrec_comm* h) {
    Comm* c = reinterpret_cast<Comm*>(h);
    if (!c) return ICREC_OK;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)g_rccl.CommDestroy(c->comm);
    }
    delete c;
    return ICREC_OK;
}

int32_t icrec_comm_rank(const icrec_comm* h) { return h ? reinterpret_cast<const Comm*>(h)->rank : -1; }
int32_t icrec_comm_world(const icrec_comm* h) { return h ? reinterpret_cast<const Comm*>(h)->world : 0; }

size_t icrec_search_sharded_workspace_bytes(const icrec_index* idx, const icrec_comm* h, int32_t n_local, int32_t k) {
    const Comm* c = reinterpret_cast<const Comm*>(h);
    if (!idx || !c || n_local < 1 || k < 1) return 0;
    return shard_ws(idx, n_local, c->world, k).total;
}

int icrec_search_sharded(icrec_index* idx, icrec_comm* h, const float* q_local_dev, int32_t n_local, int32_t k,
                         const int32_t* excl_idx_dev, const int32_t* excl_off_dev, int64_t* out_idx_dev,
                         float* out_score_dev, void* ws, size_t ws_bytes, void* stream) {
    Comm* c = reinterpret_cast<Comm*>(h);
    ICREC_REQUIRE(idx && c && q_local_dev && out_idx_dev && out_score_dev, "icrec_search_sharded: NULL argument");
    ICREC_REQUIRE(n_local >= 1 && k >= 1 && k <= ICREC_MAX_K, "icrec_search_sharded: bad n_local/k (%d, %d)", n_local, k);
    ICREC_REQUIRE((int64_t)n_local * c->world < (1ll << 31), "icrec_search_sharded: too many queries");
    const ShardWs w = shard_ws(idx, n_local, c->world, k);
    if (w.total == 0 || !ws || ws_bytes < w.total) {
        set_error("icrec_search_sharded: workspace too small (%zu < %zu)", ws_bytes, w.total);
        return ICREC_ENOMEM;
    }
    ICREC_REQUIRE(icrec_index_device(idx) == c->device, "icrec_search_sharded: index is on device %d, communicator on %d",
                  icrec_index_device(idx), c->device);
    ICREC_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const int dim = icrec_index_dim(idx);
    const int32_t Q = n_local * c->world;
    char* base = reinterpret_cast<char*>(ws);
    float* q_all = reinterpret_cast<float*>(base + w.q_all);
    uint64_t* keys_local = reinterpret_cast<uint64_t*>(base + w.keys_local);
    uint64_t* keys_all = reinterpret_cast<uint64_t*>(base + w.keys_all);
    const float* q_use = q_local_dev;
    if (c->comm) {
        ICREC_NCCL(g_rccl.AllGather(q_local_dev, q_all, (size_t)n_local * dim, ncclFloat32, c->comm, st));
        q_use = q_all;
    }
    if (int rc = icrec_search_partial(idx, q_use, Q, k, excl_idx_dev, excl_off_dev, keys_local, base + w.search,
                                      w.total - w.search, stream))
        return rc;
    const uint64_t* merged_from = keys_local;
    if (c->comm) {
        ICREC_NCCL(g_rccl.AllGather(keys_local, keys_all, (size_t)Q * k, ncclUint64, c->comm, st));
        merged_from = keys_all;
    }
    return icrec_merge_topk(merged_from, c->comm ? c->world : 1, Q, k, out_idx_dev, out_score_dev, c->device, stream);
}

int icrec_exclusions_to_shard_csr(const int32_t* off_all_dev, const int32_t* rows_all_dev, int32_t world, int32_t n_local,
                                  int32_t excl_cap, int64_t row_lo, int64_t row_hi, int32_t* csr_off_dev,
                                  int32_t* csr_idx_dev, void* ws, size_t ws_bytes, int device, void* stream) {
    ICREC_REQUIRE(off_all_dev && rows_all_dev && csr_off_dev && csr_idx_dev, "icrec_exclusions_to_shard_csr: NULL argument");
    ICREC_REQUIRE(world >= 1 && n_local >= 1 && excl_cap >= 1 && row_hi >= row_lo,
                  "icrec_exclusions_to_shard_csr: bad world/n_local/excl_cap/rows (%d, %d, %d)", world, n_local, excl_cap);
    ICREC_REQUIRE((int64_t)n_local * world < (1ll << 31) && (int64_t)excl_cap * world < (1ll << 31),
                  "icrec_exclusions_to_shard_csr: too many queries / exclusions");
    const size_t need = icrec_exclusions_to_shard_csr_workspace_bytes(world, n_local);
    if (!ws || ws_bytes < need) {
        set_error("icrec_exclusions_to_shard_csr: workspace too small (%zu < %zu)", ws_bytes, need);
        return ICREC_ENOMEM;
    }
    ICREC_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const size_t Q = (size_t)n_local * world;
    int32_t* off_copy = reinterpret_cast<int32_t*>(ws);
    int32_t* cnt = off_copy + (((size_t)world * (n_local + 1) + 63) & ~(size_t)63);
    ICREC_HIP(hipMemcpyAsync(off_copy, off_all_dev, (size_t)world * (n_local + 1) * 4, hipMemcpyDeviceToDevice, st));
    (void)Q;
    return build_local_csr(off_copy, rows_all_dev, world, n_local, excl_cap, row_lo, row_hi, cnt, csr_off_dev, csr_idx_dev, st);
}

size_t icrec_exclusions_to_shard_csr_workspace_bytes(int32_t world, int32_t n_local) {
    if (world < 1 || n_local < 1) return 0;
    return ((((size_t)world * (n_local + 1) + 63) & ~(size_t)63) + (size_t)world * n_local) * 4;
}

size_t icrec_search_sharded_excl_workspace_bytes(const icrec_index* idx, const icrec_comm* h, int32_t n_local, int32_t k,
                                                 int32_t excl_cap) {
    const Comm* c = reinterpret_cast<const Comm*>(h);
    if (!idx || !c || n_local < 1 || k < 1 || excl_cap < 1) return 0;
    return excl_ws(idx, n_local, c->world, k, excl_cap).total;
}

int icrec_search_sharded_excl(icrec_index* idx, icrec_comm* h, const float* q_local_dev, int32_t n_local, int32_t k,
                              const int32_t* excl_rows_dev, const int32_t* excl_off_dev, int32_t excl_cap,
                              int64_t* out_idx_dev, float* out_score_dev, void* ws, size_t ws_bytes, void* stream) {
    Comm* c = reinterpret_cast<Comm*>(h);
    ICREC_REQUIRE(idx && c && q_local_dev && out_idx_dev && out_score_dev && excl_rows_dev && excl_off_dev,
                  "icrec_search_sharded_excl: NULL argument");
    ICREC_REQUIRE(n_local >= 1 && k >= 1 && k <= ICREC_MAX_K && excl_cap >= 1, "icrec_search_sharded_excl: bad n_local/k/excl_cap (%d, %d, %d)", n_local, k, excl_cap);
    ICREC_REQUIRE((int64_t)n_local * c->world < (1ll << 31) && (int64_t)excl_cap * c->world < (1ll << 31),
                  "icrec_search_sharded_excl: too many queries / exclusions");
    const ExclWs w = excl_ws(idx, n_local, c->world, k, excl_cap);
    if (w.total == 0 || !ws || ws_bytes < w.total) {
        set_error("icrec_search_sharded_excl: workspace too small (%zu < %zu)", ws_bytes, w.total);
        return ICREC_ENOMEM;
    }
    ICREC_REQUIRE(icrec_index_device(idx) == c->device, "icrec_search_sharded_excl: index is on device %d, communicator on %d",
                  icrec_index_device(idx), c->device);
    ICREC_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    char* base = reinterpret_cast<char*>(ws);
    int32_t* off_all = reinterpret_cast<int32_t*>(base + w.off_all);
    int32_t* rows_all = reinterpret_cast<int32_t*>(base + w.rows_all);
    int32_t* cnt = reinterpret_cast<int32_t*>(base + w.cnt);
    int32_t* csr_off = reinterpret_cast<int32_t*>(base + w.csr_off);
    int32_t* csr_idx = reinterpret_cast<int32_t*>(base + w.csr_idx);
    // the offsets always go through the workspace copy: they are sanitised in place (excl_sanitize_kernel)
    const int32_t* rows_use = excl_rows_dev;
    if (c->comm) {
        ICREC_NCCL(g_rccl.AllGather(excl_off_dev, off_all, (size_t)n_local + 1, ncclInt32, c->comm, st));
        ICREC_NCCL(g_rccl.AllGather(excl_rows_dev, rows_all, (size_t)excl_cap, ncclInt32, c->comm, st));
        rows_use = rows_all;
    } else {
        ICREC_HIP(hipMemcpyAsync(off_all, excl_off_dev, ((size_t)n_local + 1) * 4, hipMemcpyDeviceToDevice, st));
    }
    const int world = c->comm ? c->world : 1;
    const int64_t row_lo = icrec_index_row_offset(idx), row_hi = row_lo + icrec_index_rows(idx);
    if (int rc = build_local_csr(off_all, rows_use, world, n_local, excl_cap, row_lo, row_hi, cnt, csr_off, csr_idx, st))
        return rc;
    return icrec_search_sharded(idx, h, q_local_dev, n_local, k, csr_idx, csr_off, out_idx_dev, out_score_dev,
                                base + w.inner, w.total - w.inner, stream);
}

}  // extern "C"
