// Drives libicrec's encoder from C++ (no Python) to time kernels in the real pipeline.
#include "../include/icrec.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main(int argc, char** argv) {
    int mode = argc > 1 ? atoi(argv[1]) : 1;
    icrec_bert_cfg cfg = {30522, 384, 6, 12, 1536, 512, 2, 1e-12f, 2, mode};
    size_t n = icrec_encoder_weight_count(&cfg);
    std::vector<float> w(n);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((double)(st >> 11) / 9007199254740992.0 - 0.5); };
    for (size_t i = 0; i < n; ++i) w[i] = rnd() * 0.15f;
    icrec_encoder* enc;
    if (icrec_encoder_create(w.data(), n, &cfg, 0, &enc)) { printf("create: %s\n", icrec_last_error()); return 1; }
    const int B = 1024, L = 128, T = B * L;
    std::vector<int> ids(T), cu(B + 1);
    for (int i = 0; i < T; ++i) ids[i] = 1000 + (int)((rnd() + 0.5f) * 20000);
    for (int i = 0; i <= B; ++i) cu[i] = i * L;
    int *d_ids, *d_cu; float* d_out; void* ws;
    size_t wsb = icrec_encode_workspace_bytes(enc, T, B);
    hipMalloc(&d_ids, T * 4); hipMalloc(&d_cu, (B + 1) * 4); hipMalloc(&d_out, B * 384 * 4); hipMalloc(&ws, wsb);
    hipMemcpy(d_ids, ids.data(), T * 4, hipMemcpyHostToDevice); hipMemcpy(d_cu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice);
    for (int it = 0; it < 2; ++it) icrec_encode(enc, d_ids, d_cu, B, T, L, d_out, ws, wsb, 0);
    hipDeviceSynchronize();
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            for (int it = 0; it < 5; ++it) icrec_encode(enc, d_ids, d_cu, B, T, L, d_out, ws, wsb, 0);
            hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            printf("mode %d: timing OFF, 5 encodes back to back: %.3f ms per encode\n", mode, ms / 5);
        }
    }
    icrec_timing_enable(1);
    for (int it = 0; it < 3; ++it) if (icrec_encode(enc, d_ids, d_cu, B, T, L, d_out, ws, wsb, 0)) { printf("encode: %s\n", icrec_last_error()); return 1; }
    hipDeviceSynchronize();
    double ms; long long cnt;
    icrec_timing_query(1, &ms, (int64_t*)&cnt); printf("mode %d: FFN-up avg %.3f ms over %lld launches\n", mode, ms, cnt);
    icrec_timing_query(2, &ms, (int64_t*)&cnt); printf("mode %d: encode avg %.3f ms\n", mode, ms);
    return 0;
}
