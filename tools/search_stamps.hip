// Phase stamps of the resident filter pass (search_kernel<CfgRes, false, 3>): s_memtime at the start of every round's
// scoring and selection, for catalogs of a few rounds per block - where the 1,024 x 49,688 search spends its time.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DICREC_STAMPS tools/search_stamps.hip \
//        instacart_next_order_recommendation_amd/csrc/api.hip -o tools/_search_stamps
#include "../instacart_next_order_recommendation_amd/csrc/search.hip"

#include <algorithm>
#include <vector>

int main(int argc, char** argv) {
    const int Q = argc > 1 ? atoi(argv[1]) : 1024, k = 20;
    const int64_t big = argc > 2 ? atoll(argv[2]) : 49688;
    for (int64_t rows : {(int64_t)8192, big}) {
        std::vector<float> h((size_t)rows * 384), q((size_t)Q * 384);
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((double)(st >> 11) / 9007199254740992.0 - 0.5); };
        for (auto& v : h) v = rnd();
        for (auto& v : q) v = rnd();
        float *d_rows, *d_q, *d_sc; int64_t* d_idx; void* ws;
        hipMalloc(&d_rows, h.size() * 4); hipMalloc(&d_q, q.size() * 4); hipMalloc(&d_sc, (size_t)Q * k * 4); hipMalloc(&d_idx, (size_t)Q * k * 8);
        hipMemcpy(d_rows, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_q, q.data(), q.size() * 4, hipMemcpyHostToDevice);
        icrec_index* ix;
        if (icrec_index_create_ex(d_rows, rows, 384, 0, 0, ICREC_ROWS_F32_FILTER, &ix)) { printf("create: %s\n", icrec_last_error()); return 1; }
        const size_t wsb = icrec_search_workspace_bytes(ix, Q, k);
        hipMalloc(&ws, wsb);
        for (int rep = 0; rep < 3; ++rep)
            if (icrec_search(ix, d_q, Q, k, nullptr, nullptr, d_idx, d_sc, ws, wsb, nullptr)) { printf("search: %s\n", icrec_last_error()); return 1; }
        hipDeviceSynchronize();
        const int nblk = 512;
        std::vector<unsigned long long> hs((size_t)nblk * 128);
        hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(icrec::g_stamps), hs.size() * 8);
        auto med = [&](int a, int b) {
            std::vector<long long> d;
            for (int blk = 0; blk < nblk; ++blk) {
                const unsigned long long t0 = hs[(size_t)blk * 128 + a], t1 = hs[(size_t)blk * 128 + b];
                if (t0 && t1 && t1 > t0) d.push_back((long long)(t1 - t0));
            }
            if (d.empty()) return -1ll;
            std::sort(d.begin(), d.end());
            return d[d.size() / 2];
        };
        printf("rows %lld, Q %d: init %lld, query planes -> LDS %lld;", (long long)rows, Q, med(60, 61), med(61, 0));
        for (int r = 0; r < 14; ++r) {
            const long long g = med(2 * r, 2 * r + 1), sel = r < 13 ? med(2 * r + 1, 2 * r + 2) : -1;
            if (g < 0) break;
            printf(" round %d: score %lld select %lld;", r, g, sel);
        }
        printf(" lists out %lld; block life %lld\n", med(62, 63), med(60, 63));
        printf("   round 11 in detail: thresholds + pend %lld;", med(23, 30));
        for (int it = 0; it < 6; ++it) {
            const long long ex = med(30 + 3 * it, 31 + 3 * it), bar = med(31 + 3 * it, 32 + 3 * it);
            const long long mg = it < 5 ? med(32 + 3 * it, 33 + 3 * it) : -1;
            if (ex < 0) break;
            printf(" iteration %d: offer %lld barrier %lld merge.. %lld;", it, ex, bar, mg);
        }
        printf("\n");
        icrec_index_destroy(ix);
        hipFree(d_rows); hipFree(d_q); hipFree(d_sc); hipFree(d_idx); hipFree(ws);
    }
    return 0;
}
