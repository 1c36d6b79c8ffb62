// Stand-alone C++ host program over the C ABI of librlvi_gfx950.so (include/rlvi_hip.h): no
// Python, no torch.  One M-step over a mini-batch (train_rlvi.py:85-96), the epoch end
// (train_rlvi.py:99-105: E-step + the epoch's scalars), the type-II threshold (:41-49), and a
// small-loss selection -- each checked against a few lines of host arithmetic.
//
//   hipcc --offload-arch=gfx950 -Iinclude examples/capi_smoke.cpp -Lrlvi_amd -lrlvi_gfx950 \
//         -Wl,-rpath,$PWD/rlvi_amd -o /tmp/capi_smoke && /tmp/capi_smoke
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rlvi_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define RLVI_OK(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "%s -> %d (%s)\n", #x, r_, rlvi_error_string(r_)); return 3; } } while (0)

template <typename T>
static T *to_device(const std::vector<T> &h) {
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main() {
    const int64_t B = 4096, C = 100, N = 8192;
    std::vector<float> logits(B * C), weights(N), residuals(N, 0.0f);
    std::vector<int64_t> labels(B), idx(B);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (auto &z : logits) z = (float)(6.0 * rnd() - 3.0);
    for (int64_t i = 0; i < B; ++i) {
        labels[i] = (int64_t)(rnd() * C) % C;
        idx[i] = 2 * i + 1;                                   // every second sample of the population
        if (rnd() < 0.6) logits[i * C + labels[i]] += 9.0f;   // "clean" rows: small loss
    }
    for (auto &w : weights) w = (float)rnd();

    // host arithmetic for the M-step scalars
    double ref_loss = 0.0, ref_hits = 0.0;
    std::vector<float> ref_rows(B);
    for (int64_t i = 0; i < B; ++i) {
        const float *z = &logits[i * C];
        float m = z[0];
        for (int64_t c = 1; c < C; ++c) m = std::fmax(m, z[c]);
        double se = 0.0;
        for (int64_t c = 0; c < C; ++c) se += std::exp((double)z[c] - m);
        const double li = std::log(se) - ((double)z[labels[i]] - m);
        ref_rows[i] = (float)li;
        ref_loss += (double)weights[idx[i]] * li / (double)B;
        ref_hits += z[labels[i]] == m ? 1.0 : 0.0;
    }

    float *d_logits = to_device(logits), *d_w = to_device(weights), *d_res = to_device(residuals);
    int64_t *d_labels = to_device(labels), *d_idx = to_device(idx);
    float *d_grad = nullptr, *d_out = nullptr, *d_thr = nullptr, *d_mask = nullptr;
    int32_t *d_iters = nullptr;
    void *ws = nullptr;
    const size_t ws_bytes = rlvi_workspace_bytes(N, B);
    HIP_OK(hipMalloc(&d_grad, B * C * sizeof(float)));
    HIP_OK(hipMalloc(&d_out, 4 * sizeof(float)));
    HIP_OK(hipMalloc(&d_thr, sizeof(float)));
    HIP_OK(hipMalloc(&d_mask, N * sizeof(float)));
    HIP_OK(hipMalloc(&d_iters, sizeof(int32_t)));
    HIP_OK(hipMalloc(&ws, ws_bytes));
    if (!d_logits || !d_w || !d_res || !d_labels || !d_idx) return 2;
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    if (rlvi_abi_version() != RLVI_ABI_VERSION) return 4;
    RLVI_OK(rlvi_workspace_init(ws, ws_bytes, st));

    // per mini-batch (accumulate mode: no scalars now) ...
    RLVI_OK(rlvi_mstep_fwd_bwd_f32(d_logits, C, d_labels, d_idx, d_w, d_res, N, B, C, 1.0f / B, d_grad, C,
                                   nullptr, ws, st));
    // ... per epoch: E-step over the population + the epoch's scalars
    RLVI_OK(rlvi_epoch_end_f32(d_res, d_w, N, 1e-3f, 40, 0, 0.05f, nullptr, 1, d_out, d_iters, ws, st));
    RLVI_OK(rlvi_fn_threshold_f32(d_w, N, 0.05f, d_thr, ws, st));
    RLVI_OK(rlvi_select_smallest_f32(d_res, N, N / 4, d_mask, st));
    int32_t status = -1;
    RLVI_OK(rlvi_workspace_status(ws, &status, st));
    HIP_OK(hipStreamSynchronize(st));

    float out[4], thr;
    int32_t iters;
    std::vector<float> w_new(N), res_new(N), mask(N);
    HIP_OK(hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&thr, d_thr, sizeof(thr), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&iters, d_iters, sizeof(iters), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(w_new.data(), d_w, N * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(res_new.data(), d_res, N * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(mask.data(), d_mask, N * sizeof(float), hipMemcpyDeviceToHost));

    int bad = 0;
    if (status != 0) { std::fprintf(stderr, "device status %d\n", status); ++bad; }
    if (std::fabs(out[0] - ref_loss) > 1e-5 * std::fabs(ref_loss)) { std::fprintf(stderr, "loss %g vs %g\n", out[0], ref_loss); ++bad; }
    if (std::fabs(out[1] - 100.0 * ref_hits / B) > 1e-3) { std::fprintf(stderr, "top-1 %g vs %g\n", out[1], 100.0 * ref_hits / B); ++bad; }
    // E-step invariants (train_rlvi.py:27,38): min-shifted residuals, max pi == 1, pi decreasing in l
    float rmin = INFINITY, wmax = 0.0f;
    int64_t kept = 0;
    for (int64_t i = 0; i < N; ++i) { rmin = std::fmin(rmin, res_new[i]); wmax = std::fmax(wmax, w_new[i]); kept += mask[i] == 1.0f; }
    if (rmin != 0.0f || wmax != 1.0f) { std::fprintf(stderr, "min residual %g, max pi %g\n", rmin, wmax); ++bad; }
    if (iters < 1 || iters > 40) { std::fprintf(stderr, "iterations %d\n", iters); ++bad; }
    if (!(thr >= 0.0f && thr <= 1.0f)) { std::fprintf(stderr, "threshold %g\n", thr); ++bad; }
    if (kept != N / 4) { std::fprintf(stderr, "selected %lld of %lld\n", (long long)kept, (long long)(N / 4)); ++bad; }
    // the fixed point itself: pi_i / pi_j = f(r e_i) / f(r e_j) with r = avg/(1-avg) of the LAST iteration;
    // check the weaker, exact property that equal residuals got equal weights and order is reversed
    for (int64_t i = 1; i < N && !bad; ++i)
        if ((res_new[i] < res_new[i - 1]) != (w_new[i] > w_new[i - 1]) && res_new[i] != res_new[i - 1] &&
            w_new[i] != w_new[i - 1]) { std::fprintf(stderr, "order broken at %lld\n", (long long)i); ++bad; }
    std::printf("capi_smoke: loss %.6f (host %.6f) top-1 %.3f%% iters %d thr %.6f kept %lld -> %s\n", out[0],
                ref_loss, out[1], iters, thr, (long long)kept, bad ? "FAIL" : "ok");
    return bad ? 1 : 0;
}
