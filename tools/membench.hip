// Memory-system probe for the streaming M-step at its real size (65536 x 100 fp32 = 26.2 MB in,
// 26.2 MB out, 12 rotating buffer pairs).  Build: hipcc --offload-arch=gfx950 -O3 -o membench membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int B = 65536, C = 100, ROT = 12;
constexpr size_t NELEM = (size_t)B * C;

// flat: lane i of the grid handles float4 i, i+stride, ... ; UNR loads in flight
template <int UNR, bool WRITE, bool NT>
__global__ __launch_bounds__(256) void k_flat(const float4* __restrict__ in, float4* __restrict__ out, float* sink, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * UNR;
    float acc = 0.f;
    for (; i < n4; i += stride) {
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            size_t j = i + (size_t)u * 256;
            if (j < n4) { if (NT) { vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(&in[j])); v[u] = make_float4(t.x,t.y,t.z,t.w); } else v[u] = in[j]; } else v[u] = make_float4(0,0,0,0);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            size_t j = i + (size_t)u * 256;
            if (WRITE) { if (j < n4) { float4 o = make_float4(v[u].x*2.f, v[u].y*2.f, v[u].z*2.f, v[u].w*2.f); if (NT) { vf4 t = {o.x,o.y,o.z,o.w}; __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(&out[j])); } else out[j] = o; } }
            else acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (!WRITE && acc == 123.456f) *sink = acc;
}

// row pattern: G lanes per row, K float4 per lane (as the M-step kernel), one wave-iteration per 64/G rows
template <int G, int K, bool WRITE>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ in, float* __restrict__ out, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & (G - 1), sub = lane / G;
    constexpr int R = 64 / G;
    float acc = 0.f;
    for (size_t row0 = ((size_t)blockIdx.x * 4 + wave) * R; row0 < B; row0 += (size_t)gridDim.x * 4 * R) {
        const size_t row = row0 + sub;
        float4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int col = (k * G + g) * 4;
            v[k] = col < C ? *reinterpret_cast<const float4*>(in + row * C + col) : make_float4(0,0,0,0);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int col = (k * G + g) * 4;
            if (WRITE) { if (col < C) *reinterpret_cast<float4*>(out + row * C + col) = make_float4(v[k].x*2.f, v[k].y*2.f, v[k].z*2.f, v[k].w*2.f); }
            else acc += v[k].x + v[k].y + v[k].z + v[k].w;
        }
    }
    if (!WRITE && acc == 123.456f) *sink = acc;
}

template <class F>
float time_it(F launch, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 24; ++i) launch(i);
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch(i);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best / iters * 1e3f;
}

int main() {
    std::vector<float*> in(ROT), out(ROT);
    for (int r = 0; r < ROT; ++r) { CK(hipMalloc(&in[r], NELEM * 4)); CK(hipMalloc(&out[r], NELEM * 4)); CK(hipMemset(in[r], 1, NELEM * 4)); CK(hipMemset(out[r], 0, NELEM * 4)); }
    float* sink; CK(hipMalloc(&sink, 4));
    const size_t n4 = NELEM / 4;
    const int iters = 240;
    auto rep = [&](const char* name, float us, double bytes) { printf("%-44s %7.2f us  %7.0f GB/s\n", name, us, bytes / us / 1e3); };
#define FLAT(UNR, WRITE, NT, NB, label) { float us = time_it([&](int i) { hipLaunchKernelGGL((k_flat<UNR, WRITE, NT>), dim3(NB), dim3(256), 0, 0, (const float4*)in[i % ROT], (float4*)out[i % ROT], sink, n4); }, iters); rep(label, us, (WRITE ? 2.0 : 1.0) * NELEM * 4); }
    FLAT(1, true, false, 6400, "copy flat unr1 grid=full(6400)")
    FLAT(2, true, false, 3200, "copy flat unr2 grid=3200")
    FLAT(4, true, false, 1600, "copy flat unr4 grid=1600")
    FLAT(4, true, false, 1024, "copy flat unr4 grid=1024 (loop)")
    FLAT(4, true, false, 512, "copy flat unr4 grid=512 (loop)")
    FLAT(8, true, false, 800, "copy flat unr8 grid=800")
    FLAT(4, true, true, 1600, "copy flat unr4 grid=1600 nontemporal")
    FLAT(1, false, false, 6400, "read flat unr1 grid=6400")
    FLAT(4, false, false, 1600, "read flat unr4 grid=1600")
    FLAT(8, false, false, 800, "read flat unr8 grid=800")
    FLAT(4, false, true, 1600, "read flat unr4 grid=1600 nontemporal")
#define ROWS(G, K, WRITE, NB, label) { float us = time_it([&](int i) { hipLaunchKernelGGL((k_rows<G, K, WRITE>), dim3(NB), dim3(256), 0, 0, in[i % ROT], out[i % ROT], sink); }, iters); rep(label, us, (WRITE ? 2.0 : 1.0) * NELEM * 4); }
    ROWS(4, 7, false, 1024, "read rows G=4 K=7 grid=1024")
    ROWS(8, 4, false, 2048, "read rows G=8 K=4 grid=2048")
    ROWS(32, 1, false, 2048, "read rows G=32 K=1 grid=2048")
    ROWS(32, 1, false, 8192, "read rows G=32 K=1 grid=8192")
    ROWS(4, 7, true, 1024, "copy rows G=4 K=7 grid=1024")
    ROWS(8, 4, true, 2048, "copy rows G=8 K=4 grid=2048")
    ROWS(32, 1, true, 8192, "copy rows G=32 K=1 grid=8192")
    // hipMemcpyAsync D2D for reference
    { float us = time_it([&](int i) { CK(hipMemcpyAsync(out[i % ROT], in[i % ROT], NELEM * 4, hipMemcpyDeviceToDevice, 0)); }, iters); rep("hipMemcpyAsync D2D", us, 2.0 * NELEM * 4); }
    return 0;
}
