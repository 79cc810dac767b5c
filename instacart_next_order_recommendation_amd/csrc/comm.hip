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

}  // extern "C"
