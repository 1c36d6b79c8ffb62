// Lab: issue rate of fp32 VALU operations for ONE wave per SIMD and for two (MI355X).  Prints cycles per
// instruction for plain v_fma_f32, packed v_pk_fma_f32 and v_rcp_f32, with 8 independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = (f2){a[i], a[i] + 1.0f}; }
    const float m = 1.0001f, c = 1e-4f;
    const f2 m2 = {m, m}, c2 = {c, c};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = fmaf(a[i], m, c);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
            if (MODE == 2) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (MODE == 3) a[i] = a[i] * m;
            if (MODE == 4) p[i] = p[i] * m2;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char *names[5] = {"v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_mul_f32", "v_pk_mul_f32"};
    for (int threads = 256; threads <= 512; threads *= 2)
        for (int mode = 0; mode < 5; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) k<0><<<256, threads>>>(out, cyc, iters);
                if (mode == 1) k<1><<<256, threads>>>(out, cyc, iters);
                if (mode == 2) k<2><<<256, threads>>>(out, cyc, iters);
                if (mode == 3) k<3><<<256, threads>>>(out, cyc, iters);
                if (mode == 4) k<4><<<256, threads>>>(out, cyc, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d waves/SIMD  %-14s %.2f cycles per instruction (8 independent)\n", threads / 256, names[mode],
                   (double)h / (iters * 8.0));
        }
    return 0;
}
