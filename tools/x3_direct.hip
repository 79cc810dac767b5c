// Launch the library's own linear_x3_kernel directly on synthetic planes (diagnosis only).
#include "../instacart_next_order_recommendation_amd/csrc/encoder.hip"
#include <vector>
int main() {
    using namespace icrec;
    const int M = 131072, K = 384, N = 1536;
    _Float16 *Ah, *Al, *Wh, *Wl, *oh, *ol; float *out, *bias;
    hipMalloc(&Ah, (size_t)M * K * 2); hipMalloc(&Al, (size_t)M * K * 2); hipMalloc(&Wh, (size_t)N * K * 2); hipMalloc(&Wl, (size_t)N * K * 2);
    hipMalloc(&out, (size_t)M * N * 4); hipMalloc(&oh, (size_t)M * N * 2); hipMalloc(&ol, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
    std::vector<_Float16> g((size_t)M * K);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    for (auto& v : g) v = (_Float16)(float)((rnd() + rnd() + rnd() + rnd() - 2.0) * 1.7);
    hipMemcpy(Ah, g.data(), g.size() * 2, hipMemcpyHostToDevice);
    for (auto& v : g) v = (_Float16)(float)(rnd() - 0.5);
    hipMemcpy(Al, g.data(), g.size() * 2, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)N * K; ++i) g[i] = (_Float16)(float)((rnd() - 0.5) * 0.2);
    hipMemcpy(Wh, g.data(), (size_t)N * K * 2, hipMemcpyHostToDevice); hipMemcpy(Wl, g.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, N * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int epi = 0; epi < 2; ++epi) {
        for (int r = 0; r < 7; ++r) {
            if (r == 2) hipEventRecord(a);
            if (epi == 0) launch_linear_x3<0>(Ah, Al, M, K, Wh, Wl, N, bias, out, nullptr, nullptr, 0);
            else launch_linear_x3<1>(Ah, Al, M, K, Wh, Wl, N, bias, nullptr, oh, ol, 0);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("library kernel EPI=%d direct: %.3f ms\n", epi, ms / 5);
    }
    {   // real pipeline data: run one encode, then time FFN-up directly on ITS planes / weights
        icrec_bert_cfg c = {30522, 384, 6, 12, 1536, 512, 2, 1e-12f, 2, 1};
        size_t n = icrec_encoder_weight_count(&c);
        std::vector<float> w(n);
        for (size_t i = 0; i < n; ++i) w[i] = (float)(rnd() - 0.5) * 0.15f;
        icrec_encoder* eh; icrec_encoder_create(w.data(), n, &c, 0, &eh);
        Encoder* e = (Encoder*)eh;
        const int B = 1024, L = 128;
        std::vector<int> ids(M), cu(B + 1);
        for (int i = 0; i < M; ++i) ids[i] = 1000 + (int)(rnd() * 20000);
        for (int i = 0; i <= B; ++i) cu[i] = i * L;
        int *d_ids, *d_cu; float* d_out; char* ws; size_t wsb = icrec_encode_workspace_bytes(eh, M, B);
        hipMalloc(&d_ids, M * 4); hipMalloc(&d_cu, (B + 1) * 4); hipMalloc(&d_out, B * 384 * 4); hipMalloc(&ws, wsb);
        hipMemcpy(d_ids, ids.data(), M * 4, hipMemcpyHostToDevice); hipMemcpy(d_cu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice);
        icrec_encode(eh, d_ids, d_cu, B, M, L, d_out, ws, wsb, 0); hipDeviceSynchronize();
        EncWs lay = enc_ws(c, M);
        _Float16* xh = (_Float16*)(ws + lay.xs); _Float16* xl = xh + (size_t)M * K;
        const LayerW& Lw = e->layers[5];
        struct { const char* name; const _Float16 *ah, *al, *wh, *wl; } cases[] = {
            {"real A, real W", xh, xl, Lw.W1_h, Lw.W1_l}, {"real A, synthetic W", xh, xl, Wh, Wl},
            {"synthetic A, real W", Ah, Al, Lw.W1_h, Lw.W1_l}, {"real A_hi only (lo synthetic), synthetic W", xh, Al, Wh, Wl},
            {"real A_lo only (hi synthetic), synthetic W", Ah, xl, Wh, Wl}};
        for (auto& cs : cases) {
            for (int r = 0; r < 7; ++r) { if (r == 2) hipEventRecord(a); launch_linear_x3<1>(cs.ah, cs.al, M, K, cs.wh, cs.wl, N, bias, nullptr, oh, ol, 0); }
            hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            printf("%-45s %.3f ms\n", cs.name, ms / 5);
        }
        std::vector<_Float16> hx((size_t)1 << 20); hipMemcpy(hx.data(), xl, hx.size() * 2, hipMemcpyDeviceToHost);
        size_t sub = 0, zero = 0, big = 0; double mx = 0;
        for (auto v : hx) { float f = fabsf((float)v); if (f == 0) zero++; else if (f < 6.1e-5f) sub++; if (f > 1000) big++; if (f > mx) mx = f; }
        printf("x_lo plane sample: zeros %zu, subnormal %zu, >1000: %zu, max %.3f of %zu\n", zero, sub, big, mx, hx.size());
        hipMemcpy(hx.data(), xh, hx.size() * 2, hipMemcpyDeviceToHost); mx = 0; zero = sub = 0;
        for (auto v : hx) { float f = fabsf((float)v); if (f == 0) zero++; else if (f < 6.1e-5f) sub++; if (f > mx) mx = f; }
        printf("x_hi plane sample: zeros %zu, subnormal %zu, max %.3f\n", zero, sub, mx);
    }
    {   // sustained: 300 back-to-back launches, timed in groups of 50
        for (int grp = 0; grp < 6; ++grp) {
            hipEventRecord(a);
            for (int r = 0; r < 50; ++r) launch_linear_x3<1>(Ah, Al, M, K, Wh, Wl, N, bias, nullptr, oh, ol, 0);
            hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            printf("sustained group %d: %.3f ms/launch\n", grp, ms / 50);
        }
    }
    {   // same kernel, operands laid out inside ONE allocation exactly as icrec_encode's workspace does
        icrec_bert_cfg c = {30522, 384, 6, 12, 1536, 512, 2, 1e-12f, 2, 1};
        EncWs w = enc_ws(c, M);
        char* base; hipMalloc(&base, w.total);
        _Float16* xh = (_Float16*)(base + w.xs); _Float16* xl = xh + (size_t)M * K;
        _Float16* hh = (_Float16*)(base + w.h); _Float16* hl = hh + (size_t)M * N;
        hipMemcpy(xh, Ah, (size_t)M * K * 2, hipMemcpyDeviceToDevice); hipMemcpy(xl, Al, (size_t)M * K * 2, hipMemcpyDeviceToDevice);
        for (int r = 0; r < 7; ++r) { if (r == 2) hipEventRecord(a); launch_linear_x3<1>(xh, xl, M, K, Wh, Wl, N, bias, nullptr, hh, hl, 0); }
        hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
        printf("EPI=1, operands inside one workspace allocation: %.3f ms\n", ms / 5);
        for (int r = 0; r < 7; ++r) { if (r == 2) hipEventRecord(a); launch_linear_x3<1>(xh, xl, M, K, Wh, Wl, N, bias, nullptr, oh, ol, 0); }
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("EPI=1, A in workspace, outputs separate: %.3f ms\n", ms / 5);
        for (int r = 0; r < 7; ++r) { if (r == 2) hipEventRecord(a); launch_linear_x3<1>(Ah, Al, M, K, Wh, Wl, N, bias, nullptr, hh, hl, 0); }
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("EPI=1, A separate, outputs in workspace: %.3f ms\n", ms / 5);
        // weights contiguous hi|lo like the encoder's plane pool
        _Float16* wp; hipMalloc(&wp, (size_t)N * K * 4);
        hipMemcpy(wp, Wh, (size_t)N * K * 2, hipMemcpyDeviceToDevice); hipMemcpy(wp + (size_t)N * K, Wl, (size_t)N * K * 2, hipMemcpyDeviceToDevice);
        for (int r = 0; r < 7; ++r) { if (r == 2) hipEventRecord(a); launch_linear_x3<1>(Ah, Al, M, K, wp, wp + (size_t)N * K, N, bias, nullptr, oh, ol, 0); }
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("EPI=1, W planes contiguous: %.3f ms\n", ms / 5);
    }
    return 0;
}
