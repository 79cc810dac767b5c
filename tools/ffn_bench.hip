// Standalone timing harness for the weights-direct encoder kernels (fused FFN, QKV, attention-out + LN) on
// random data at the bench shape, with the fused kernel's timing ablations (VAR bits, see encoder.hip).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/ffn_bench.hip \
//        instacart_next_order_recommendation_amd/csrc/api.hip -o tools/_ffn_bench
// Not part of the product; numbers from ablated variants are for attribution only.
#include "../instacart_next_order_recommendation_amd/csrc/encoder.hip"

#include <unistd.h>

#include <algorithm>
#include <functional>
#include <map>
#include <vector>

using namespace icrec;

// the six attention-output arguments of ffn_fused2_kernel, unused when AO = false
#define NOQKV (const _Float16*)nullptr, (const float*)nullptr, (float*)nullptr, 0
#define NOAO (const _Float16*)nullptr, (const _Float16*)nullptr, (const _Float16*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, NOQKV

__global__ void fill_half(_Float16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        p[i] = (_Float16)(((float)(z & 0xFFFF) / 65536.0f - 0.5f) * scale);
    }
}
__global__ void fill_float(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        p[i] = ((float)(z & 0xFFFF) / 65536.0f - 0.5f) * scale;
    }
}

template <class F>
static void timeit(const char* name, F&& launch, double flops) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) launch();
    hipDeviceSynchronize();
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    hipError_t e = hipGetLastError();
    printf("%-46s min %8.1f us  median %8.1f us  %7.1f TF (algorithmic, median)%s\n", name, ms[0] * 1e3, ms[3] * 1e3,
           flops / (ms[3] * 1e-3) / 1e12, e == hipSuccess ? "" : "  [HIP ERROR]");
}

// Round-robin comparison: clocks drift with temperature over a run, so variants are timed interleaved (R rounds, one
// launch of each per round) and compared by their medians.
struct Cand { const char* name; std::function<void()> launch; std::vector<float> ms; };
static void compare(std::vector<Cand>& cs, double flops, int rounds = 15) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (auto& c : cs) { c.launch(); c.launch(); }
    hipDeviceSynchronize();
    for (int r = 0; r < rounds; ++r)
        for (auto& c : cs) {
            hipEventRecord(a);
            c.launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float t; hipEventElapsedTime(&t, a, b);
            c.ms.push_back(t);
        }
    for (auto& c : cs) {
        std::sort(c.ms.begin(), c.ms.end());
        printf("  [interleaved] %-44s min %8.1f us  median %8.1f us  %7.1f TF\n", c.name, c.ms[0] * 1e3, c.ms[c.ms.size() / 2] * 1e3,
               flops / (c.ms[c.ms.size() / 2] * 1e-3) / 1e12);
    }
}

int main(int argc, char** argv) {
    const int T0 = argc > 1 ? atoi(argv[1]) : 131150;
    const int H = 384, I = 1536;
    float *x, *b1, *b2, *g, *bn, *qkv, *bq;
    _Float16 *xh, *xl, *W1p, *W2p, *Wqp, *Wop, *ch, *cl;
    const size_t Tmax = 140000;
    hipMalloc(&x, Tmax * H * 4); hipMalloc(&xh, Tmax * H * 2); hipMalloc(&xl, Tmax * H * 2);
    hipMalloc(&ch, Tmax * H * 2); hipMalloc(&cl, Tmax * H * 2);
    hipMalloc(&qkv, Tmax * 3 * H * 4);
    hipMalloc(&W1p, (size_t)2 * I * H * 2); hipMalloc(&W2p, (size_t)2 * I * H * 2);
    hipMalloc(&Wqp, (size_t)2 * 3 * H * H * 2); hipMalloc(&Wop, (size_t)2 * H * H * 2);
    hipMalloc(&b1, I * 4); hipMalloc(&b2, H * 4); hipMalloc(&g, H * 4); hipMalloc(&bn, H * 4); hipMalloc(&bq, 3 * H * 4);
    auto reinit = [&]() {
        fill_float<<<1024, 256>>>(x, Tmax * H, 1, 2.0f);
        fill_half<<<1024, 256>>>(xh, Tmax * H, 2, 32.0f);   // ~ 16 * activation
        fill_half<<<1024, 256>>>(xl, Tmax * H, 3, 0.02f);
        fill_half<<<1024, 256>>>(ch, Tmax * H, 12, 32.0f);
        fill_half<<<1024, 256>>>(cl, Tmax * H, 13, 0.02f);
    };
    reinit();
    fill_half<<<1024, 256>>>(W1p, (size_t)2 * I * H, 4, 100.0f);  // ~ 1024 * weight
    fill_half<<<1024, 256>>>(W2p, (size_t)2 * I * H, 5, 100.0f);
    fill_half<<<1024, 256>>>(Wqp, (size_t)2 * 3 * H * H, 6, 100.0f);
    fill_half<<<1024, 256>>>(Wop, (size_t)2 * H * H, 7, 100.0f);
    fill_float<<<8, 256>>>(b1, I, 8, 0.04f); fill_float<<<8, 256>>>(b2, H, 9, 0.04f); fill_float<<<8, 256>>>(bq, 3 * H, 10, 0.04f);
    fill_float<<<8, 256>>>(g, H, 11, 0.1f); fill_float<<<8, 256>>>(bn, H, 12, 0.04f);
    hipDeviceSynchronize();

#ifdef ICREC_STAMPS
    // phase stamps (s_memtime shader-clock ticks): median over blocks of each interval
    auto dump = [&](const char* name, int nblocks, int which, std::vector<int> ks) {
        std::vector<unsigned long long> hs((size_t)nblocks * 128);
        hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(g_stamps), hs.size() * 8);
        printf("%s: median cycles between stamps", name);
        for (size_t a = 0; a + 1 < ks.size(); ++a) {
            std::vector<long long> d;
            for (int b = 0; b < nblocks; ++b) {
                const unsigned long long t0 = hs[((size_t)b * 2 + which) * 64 + ks[a]], t1 = hs[((size_t)b * 2 + which) * 64 + ks[a + 1]];
                d.push_back((long long)(t1 - t0));
            }
            std::sort(d.begin(), d.end());
            printf(" [%d->%d] %lld", ks[a], ks[a + 1], d[d.size() / 2]);
        }
        std::vector<long long> dur; unsigned long long mn = ~0ull, mx = 0;
        for (int b = 0; b < nblocks; ++b) { unsigned long long t0 = hs[((size_t)b * 2 + which) * 64 + ks.front()], t1 = hs[((size_t)b * 2 + which) * 64 + ks.back()]; dur.push_back((long long)(t1 - t0)); mn = std::min(mn, t0); mx = std::max(mx, t1); }
        std::sort(dur.begin(), dur.end());
        printf("  | block median %lld p90 %lld, kernel span %llu\n", dur[dur.size() / 2], dur[dur.size() * 9 / 10], mx - mn);
    };
    for (int T : {2048, 16384, 131072}) {
        const int nb = T / 64;
        printf("==== T = %d (%d blocks of 64 tokens)\n", T, nb);
        auto kern = ffn_fused2_kernel<0>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FFN2_LDS);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(nb), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f, NOAO);
        hipDeviceSynchronize();
        dump("ffn2 producer wave0", nb, 0, {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 22, 23, 24, 25, 26, 27, 32, 33, 34, 35, 37, 38, 39, 40, 41, 42, 36, 30});
        dump("ffn2 consumer wave4", nb, 1, {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 22, 23, 24, 25, 27, 32, 33, 34, 35, 37, 38, 39, 40, 41, 42, 36, 30});
        if (T == 131072) {
            // shader clock under sustained load: s_memtime ticks between the first workgroup's start and the last one's end
            // of one launch, against that launch's duration by events; `burst` launches back to back before it
            for (int burst : {0, 40}) {
                hipEvent_t ea, eb;
                hipEventCreate(&ea); hipEventCreate(&eb);
                hipDeviceSynchronize();
                if (burst == 0) usleep(300000);
                for (int rep = 0; rep < burst; ++rep) hipLaunchKernelGGL(kern, dim3(nb), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f, NOAO);
                hipEventRecord(ea);
                hipLaunchKernelGGL(kern, dim3(nb), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f, NOAO);
                hipEventRecord(eb);
                hipEventSynchronize(eb);
                float ms; hipEventElapsedTime(&ms, ea, eb);
                std::vector<unsigned long long> hs((size_t)nb * 128);
                hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(g_stamps), hs.size() * 8);
                // the XCDs' counters are not synchronised, so no global first-start / last-end: every workgroup's own life
                // (start and end stamped on the same CU), summed and divided by the 256 CUs that were busy throughout
                double busy = 0;
                for (int b = 0; b < nb; ++b) busy += (double)(hs[(size_t)b * 128 + 30] - hs[(size_t)b * 128 + 0]);
                printf("fused FFN launch after %2d back-to-back launches: %.1f us by events, %.0f s_memtime ticks of workgroup life per CU -> %.2f GHz\n", burst, ms * 1e3, busy / 256, busy / 256 / (ms * 1e6));
            }
        }
        reinit();
        hipLaunchKernelGGL((wt_linear_kernel<3, 2, 1, 0>), dim3(nb * 3), dim3(256), 0, 0, xh, xl, T, H, Wqp, 3 * H, bq, qkv, nullptr, nullptr, 3);
        hipDeviceSynchronize();
        dump("qkv wave0", nb * 3, 0, {0, 1, 2, 3, 4, 5, 6, 7, 30});
        {
#ifndef STAMP_VAR
#define STAMP_VAR 0
#endif
            auto kao = ffn_fused2_kernel<STAMP_VAR, true>;
            hipFuncSetAttribute(reinterpret_cast<const void*>(kao), hipFuncAttributeMaxDynamicSharedMemorySize, FFN2_LDS);
            for (int rep = 0; rep < 2; ++rep)
                hipLaunchKernelGGL(kao, dim3(nb), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f,
                                   (const _Float16*)ch, (const _Float16*)cl, (const _Float16*)Wop, (const float*)b2, (const float*)g, (const float*)bn, NOQKV);
            hipDeviceSynchronize();
            dump("AO+ffn2 producer wave0", nb, 0, {0, 44, 45, 28, 29, 31, 1, 2, 3, 4, 5, 26, 27, 30, 43});
            dump("AO+ffn2 consumer wave4", nb, 1, {0, 44, 45, 46, 28, 29, 31, 1, 2, 3, 4, 5, 27, 30, 43});
        }
        if (T == 131072) {
            // In-kernel clock of the LAYER kernel (guide, DVFS item 6): d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup,
            // median over the workgroups of one launch that follows >= 2 s of back-to-back launches; the launch's wall time by
            // events beside it, and the workgroup life in both units.
            auto kl = ffn_fused2_kernel<0, true>;
            auto launch = [&] {
                hipLaunchKernelGGL(kl, dim3(nb), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f,
                                   (const _Float16*)ch, (const _Float16*)cl, (const _Float16*)Wop, (const float*)b2, (const float*)g, (const float*)bn,
                                   (const _Float16*)Wqp, (const float*)bq, qkv, 3 * H);
            };
            for (int rep = 0; rep < 1800; ++rep) launch();  // ~2.1 s
            hipEvent_t ea, eb;
            hipEventCreate(&ea); hipEventCreate(&eb);
            hipEventRecord(ea);
            launch();
            hipEventRecord(eb);
            hipEventSynchronize(eb);
            float ms; hipEventElapsedTime(&ms, ea, eb);
            std::vector<unsigned long long> hs((size_t)nb * 128);
            hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(g_stamps), hs.size() * 8);
            std::vector<double> clk, life_c, life_us;
            for (int b = 0; b < nb; ++b) {
                const double dc = (double)(hs[(size_t)b * 128 + 43] - hs[(size_t)b * 128 + 0]);
                const double dr = (double)(hs[(size_t)b * 128 + 63] - hs[(size_t)b * 128 + 62]);
                if (dr > 0) { clk.push_back(dc / dr * 0.1); life_c.push_back(dc); life_us.push_back(dr * 0.01); }
            }
            std::sort(clk.begin(), clk.end()); std::sort(life_c.begin(), life_c.end()); std::sort(life_us.begin(), life_us.end());
            double sum_us = 0; for (double v : life_us) sum_us += v;
            printf("layer kernel after 2 s of back-to-back launches: %.1f us by events; in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f); "
                   "workgroup life median %.0f shader cycles = %.1f us; sum of workgroup lives / 256 CUs = %.1f us\n", ms * 1e3,
                   clk[clk.size() / 2], clk[clk.size() / 10], clk[clk.size() * 9 / 10], life_c[life_c.size() / 2],
                   life_us[life_us.size() / 2], sum_us / 256);
        }
        reinit();
        {   // attention, long bucket: 512 sequences of 200 tokens
            const int nseq = 512, Ls = 200;
            std::vector<int> cuh(nseq + 1);
            for (int i = 0; i <= nseq; ++i) cuh[i] = i * Ls;
            int* cud; hipMalloc(&cud, (nseq + 1) * 4); hipMemcpy(cud, cuh.data(), (nseq + 1) * 4, hipMemcpyHostToDevice);
            fill_float<<<1024, 256>>>(qkv, (size_t)nseq * Ls * 3 * H, 21, 2.0f);
            const float sl2e = (1.0f / sqrtf(32.0f)) * 1.44269504088896340736f;
            for (int rep = 0; rep < 2; ++rep)
                hipLaunchKernelGGL((attention_x3_kernel<8, 8, true>), dim3(nseq * 12, 1), dim3(512), 0, 0, qkv, cud, 12, H, sl2e, (float*)nullptr, ch, cl, (const int32_t*)nullptr, 4);
            hipDeviceSynchronize();
            dump("attention long (L=200) wave0: start | Q split | K/V arrived | planes in LDS | barrier | maxima pass | exp + PV pass | barrier | out", nseq * 12, 0, {0, 6, 7, 1, 2, 9, 3, 4, 5});
            {   // per-CU timeline: how much of a CU's span is covered by 0 / 1 / 2 resident workgroups
                const int nblk = nseq * 12;
                std::vector<unsigned long long> hs((size_t)nblk * 128);
                hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(g_stamps), hs.size() * 8);
                std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
                for (int bI = 0; bI < nblk; ++bI) {
                    const unsigned long long id = hs[(size_t)bI * 128 + 8], cu = ((id >> 32) << 16) | (id & 0xFF00);
                    ev[cu].push_back({hs[(size_t)bI * 128 + 0], +1});
                    ev[cu].push_back({hs[(size_t)bI * 128 + 5], -1});
                }
                double cov[4] = {0, 0, 0, 0}, span = 0, life = 0;
                for (auto& kv : ev) {
                    auto& v = kv.second;
                    std::sort(v.begin(), v.end());
                    int c = 0;
                    for (size_t i = 0; i + 1 < v.size(); ++i) {
                        c += v[i].second;
                        cov[c > 3 ? 3 : c] += (double)(v[i + 1].first - v[i].first);
                    }
                    span += (double)(v.back().first - v.front().first);
                }
                for (int bI = 0; bI < nblk; ++bI) life += (double)(hs[(size_t)bI * 128 + 5] - hs[(size_t)bI * 128 + 0]);
                printf("attention per-CU timeline: %zu CUs, mean span %.0f ticks, resident workgroups 0: %.1f%%  1: %.1f%%  2: %.1f%%  3+: %.1f%%, "
                       "workgroups per CU %.1f, mean life %.0f ticks\n", ev.size(), span / ev.size(), 100 * cov[0] / span, 100 * cov[1] / span,
                       100 * cov[2] / span, 100 * cov[3] / span, (double)nblk / ev.size(), life / nblk);
            }
            timeit("attention_x3<8,8> 512 seq x 200 tok", [&] {
                hipLaunchKernelGGL((attention_x3_kernel<8, 8, true>), dim3(nseq * 12, 1), dim3(512), 0, 0, qkv, cud, 12, H, sl2e, (float*)nullptr, ch, cl, (const int32_t*)nullptr, 4);
            }, 4.0 * nseq * 12 * Ls * Ls * 32);
            hipFree(cud);
        }
    }
    return 0;
#endif
    if (T0 > 0)  // `tools/_ffn_bench 0`: only the attention lines below
    for (int T : {T0, 131072, 16384}) {
        const double ffn_flops = 4.0 * T * H * I;
        printf("---- T = %d tokens (%d blocks of 64)\n", T, (T + 63) / 64);
#define FFN2(V)                                                                                                         \
    timeit("ffn_fused2 (producer/consumer) VAR=" #V, [&] {                                                               \
        auto kern = ffn_fused2_kernel<V>;                                                                                \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FFN2_LDS);  \
        hipLaunchKernelGGL(kern, dim3((T + 63) / 64), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f, NOAO); \
    }, ffn_flops)
        FFN2(0);
        if (T == T0 || T == 131072) {
            FFN2(1); FFN2(4); FFN2(5); FFN2(13); FFN2(32);
        }
        reinit();
        timeit("qkv wt_linear<3,2,1,0> N=1152", [&] {
            hipLaunchKernelGGL((wt_linear_kernel<3, 2, 1, 0>), dim3(((T + 63) / 64) * 3), dim3(256), 0, 0, xh, xl, T, H, Wqp, 3 * H, bq, qkv, nullptr, nullptr, 3);
        }, 2.0 * T * H * 3 * H);
        timeit("qkv_resident_kernel N=1152", [&] {
            hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, QKVR_LDS);
            hipLaunchKernelGGL(qkv_resident_kernel, dim3((T + 63) / 64), dim3(512), QKVR_LDS, 0, xh, xl, T, Wqp, bq, qkv, 3 * H);
        }, 2.0 * T * H * 3 * H);
        timeit("attn-out + LN + FFN in one kernel (AO)", [&] {
            auto kern = ffn_fused2_kernel<0, true>;
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FFN2_LDS);
            hipLaunchKernelGGL(kern, dim3((T + 63) / 64), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f,
                               (const _Float16*)ch, (const _Float16*)cl, (const _Float16*)Wop, (const float*)b2, (const float*)g, (const float*)bn, NOQKV);
        }, ffn_flops + 2.0 * T * H * H);
        timeit("attn-out + LN + FFN + next layer's QKV in one kernel", [&] {
            auto kern = ffn_fused2_kernel<0, true>;
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FFN2_LDS);
            hipLaunchKernelGGL(kern, dim3((T + 63) / 64), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f,
                               (const _Float16*)ch, (const _Float16*)cl, (const _Float16*)Wop, (const float*)b2, (const float*)g, (const float*)bn,
                               (const _Float16*)Wqp, (const float*)bq, qkv, 3 * H);
        }, ffn_flops + 2.0 * T * H * H + 2.0 * T * H * 3 * H);
        if (T == 131072 && getenv("ICREC_GAPS")) {
            // Does the kernel's duration depend on what the chip did in the milliseconds before it?  The layer kernel timed by
            // events (and its shader clock, stamps build only) when launched back to back, and with the device left idle for
            // `gap` microseconds before every launch - the situation of a counter-collection pass, which serialises kernels.
            auto kern = ffn_fused2_kernel<0, true>;
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FFN2_LDS);
            auto launch = [&] {
                hipLaunchKernelGGL(kern, dim3((T + 63) / 64), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f,
                                   (const _Float16*)ch, (const _Float16*)cl, (const _Float16*)Wop, (const float*)b2, (const float*)g, (const float*)bn,
                                   (const _Float16*)Wqp, (const float*)bq, qkv, 3 * H);
            };
            hipEvent_t ea, eb;
            hipEventCreate(&ea); hipEventCreate(&eb);
            for (int gap_us : {0, 100, 300, 1000, 3000, 10000, 0}) {
                for (int i = 0; i < 30; ++i) launch();  // the same thermal state in front of every series
                hipDeviceSynchronize();
                std::vector<float> ms;
                for (int rep = 0; rep < 25; ++rep) {
                    if (gap_us) { hipDeviceSynchronize(); usleep(gap_us); }
                    hipEventRecord(ea);
                    launch();
                    hipEventRecord(eb);
                    if (gap_us) { hipEventSynchronize(eb); }
                    else if (rep == 24) hipEventSynchronize(eb);
                    if (gap_us || rep == 24) { float t; hipEventElapsedTime(&t, ea, eb); ms.push_back(t); }
                }
                std::sort(ms.begin(), ms.end());
                printf("  layer kernel, device idle %5d us before each launch: median %7.1f us  min %7.1f  max %7.1f  (%zu timed)\n", gap_us,
                       ms[ms.size() / 2] * 1e3, ms.front() * 1e3, ms.back() * 1e3, ms.size());
            }
            // duty cycle: the kernel alternating with a sleeping kernel of about its own length
        }
        if (T == 131072 && getenv("ICREC_VARS")) {  // round-robin A/B of layer-kernel variants (VAR bits): ICREC_VARS=1
            std::vector<Cand> cs;
#define LAYER_CAND(V)                                                                                                     \
            {                                                                                                             \
                auto kern = ffn_fused2_kernel<V, true>;                                                                   \
                hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FFN2_LDS); \
                cs.push_back({"layer kernel VAR=" #V, [=] {                                                              \
                    hipLaunchKernelGGL(kern, dim3((T + 63) / 64), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f, \
                                       (const _Float16*)ch, (const _Float16*)cl, (const _Float16*)Wop, (const float*)b2, (const float*)g, (const float*)bn, \
                                       (const _Float16*)Wqp, (const float*)bq, qkv, 3 * H); }, {}});                     \
            }
            LAYER_CAND(0) LAYER_CAND(64)
            compare(cs, ffn_flops + 2.0 * T * H * H + 2.0 * T * H * 3 * H, 25);
        }
        if (T == 131072) {  // does the kernel's time follow its cycles (a sleeping workgroup costs time) or the power cap (it does not)?
            for (int rep = 0; rep < 2; ++rep) {
                timeit("  [interleaved] layer kernel", [&] {
                    auto kern = ffn_fused2_kernel<0, true>;
                    hipLaunchKernelGGL(kern, dim3((T + 63) / 64), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f,
                                       (const _Float16*)ch, (const _Float16*)cl, (const _Float16*)Wop, (const float*)b2, (const float*)g, (const float*)bn,
                                       (const _Float16*)Wqp, (const float*)bq, qkv, 3 * H);
                }, ffn_flops + 2.0 * T * H * H + 2.0 * T * H * 3 * H);
                timeit("  [interleaved] layer kernel + ~16 k idle cycles per workgroup (of ~250 k)", [&] {
                    auto kern = ffn_fused2_kernel<64, true>;
                    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FFN2_LDS);
                    hipLaunchKernelGGL(kern, dim3((T + 63) / 64), dim3(512), FFN2_LDS, 0, xh, xl, T, I, W1p, b1, W2p, b2, g, bn, 1e-12f,
                                       (const _Float16*)ch, (const _Float16*)cl, (const _Float16*)Wop, (const float*)b2, (const float*)g, (const float*)bn,
                                       (const _Float16*)Wqp, (const float*)bq, qkv, 3 * H);
                }, ffn_flops + 2.0 * T * H * H + 2.0 * T * H * 3 * H);
            }
        }
        reinit();
        hipDeviceSynchronize();
    }
    {   // attention buckets on uniform-length batches (the product's launch shapes): 4-, 6- and 8-tile kernels
        const int nseq = 512;
        const float sl2e = (1.0f / sqrtf(32.0f)) * 1.44269504088896340736f;
        int* cud; hipMalloc(&cud, (nseq + 1) * 4);
        auto run = [&](const char* name, int Ls, auto launch) {
            std::vector<int> cuh(nseq + 1);
            for (int i = 0; i <= nseq; ++i) cuh[i] = i * Ls;
            hipMemcpy(cud, cuh.data(), (nseq + 1) * 4, hipMemcpyHostToDevice);
            fill_float<<<1024, 256>>>(qkv, (size_t)nseq * Ls * 3 * H, 21, 2.0f);
            timeit(name, launch, 4.0 * nseq * 12 * Ls * Ls * 32);
        };
        run("attention_x3<4,4> 512 seq x 110 tok", 110, [&] {
            hipLaunchKernelGGL((attention_x3_kernel<4, 4, true>), dim3(nseq * 12, 1), dim3(256), 0, 0, qkv, cud, 12, H, sl2e, (float*)nullptr, ch, cl, (const int32_t*)nullptr, 2);
        });
        run("attention_x3<6,6> 512 seq x 176 tok", 176, [&] {
            hipLaunchKernelGGL((attention_x3_kernel<6, 6, true>), dim3(nseq * 12, 1), dim3(384), 0, 0, qkv, cud, 12, H, sl2e, (float*)nullptr, ch, cl, (const int32_t*)nullptr, 4);
        });
        run("attention_x3<1,1> launched over 512 x 12 workgroups that all exit (230-token sequences): dispatch cost", 230, [&] {
            hipLaunchKernelGGL((attention_x3_kernel<1, 1, true>), dim3(nseq * 12, 1), dim3(64), 0, 0, qkv, cud, 12, H, sl2e, (float*)nullptr, ch, cl, (const int32_t*)nullptr, 0);
        });
        run("attention_x3<8,8> 512 seq x 230 tok", 230, [&] {
            hipLaunchKernelGGL((attention_x3_kernel<8, 8, true>), dim3(nseq * 12, 1), dim3(512), 0, 0, qkv, cud, 12, H, sl2e, (float*)nullptr, ch, cl, (const int32_t*)nullptr, 6);
        });
        hipFree(cud);
    }
    return 0;
}
