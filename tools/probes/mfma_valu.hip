// Do fp32 MFMAs and VALU work of ANOTHER wave on the same SIMD overlap?  512-thread workgroups, one per CU: waves 0-3 run a
// dependent v_mfma_f32_32x32x2_f32 chain (or bf16 32x32x16), waves 4-7 a chain of v_pk_fma_f32 / v_fma_f32 / v_mov_dpp.
// Times: MFMA waves alone, VALU waves alone, both.  Build: hipcc -O3 --offload-arch=gfx950 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0 fp32 mfma, 1 bf16 mfma
__device__ float mfma_work(int iters, float a0) {
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    if (MODE == 0) {
        float a = a0, b = 1e-3f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    } else {
        bf16x8 a, b;
        for (int r = 0; r < 8; ++r) { a[r] = (__bf16)a0; b[r] = (__bf16)1e-3f; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c[r];
    return s;
}
template <int VK>     // 0 v_pk_fma_f32, 1 v_fma_f32, 2 dpp mov + add
__device__ float valu_work(int iters, float a0) {
    f32x2 x[8];
    for (int r = 0; r < 8; ++r) x[r] = f32x2{a0 + r, a0 - r};
    const f32x2 m = {1.0001f, 0.9999f}, d = {1e-3f, -1e-3f};
    float y[8];
    for (int r = 0; r < 8; ++r) y[r] = a0 + r;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (VK == 0) x[r] = __builtin_elementwise_fma(x[r], m, d);
                else if (VK == 1) y[r] = __builtin_fmaf(y[r], 1.0001f, 1e-3f);
                else y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x150 + 3, 0xF, 0xF, true));
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 8; ++r) s += x[r][0] + x[r][1] + y[r];
    return s;
}
template <int MODE, int VK>
__global__ __launch_bounds__(512, 2) void k(float* out, int mi, int vi, float a0, int prio) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) { if (prio == 2) __builtin_amdgcn_s_setprio(3); if (mi) s = mfma_work<MODE>(mi, a0 + threadIdx.x); }
    else          { if (prio == 1) __builtin_amdgcn_s_setprio(3); if (vi) s = valu_work<VK>(vi, a0 + threadIdx.x); }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE, int VK>
static float run(float* out, int mi, int vi, int prio = 0) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, VK><<<256, 512>>>(out, mi, vi, 1.f, prio); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE, VK><<<256, 512>>>(out, mi, vi, 1.f, prio);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
template <int MODE, int VK>
static void trio(float* out, const char* name, int mi, int vi) {
    for (int w = 0; w < 3; ++w) run<MODE, VK>(out, mi, vi);      // clocks
    const float m = run<MODE, VK>(out, mi, 0), v = run<MODE, VK>(out, 0, vi), b = run<MODE, VK>(out, mi, vi);
    const float b1 = run<MODE, VK>(out, mi, vi, 1), b2 = run<MODE, VK>(out, mi, vi, 2);
    printf("%-36s mfma alone %7.1f | valu alone %7.1f | both %7.1f (valu waves prio 3: %7.1f, mfma waves prio 3: %7.1f) sum %7.1f max %7.1f us\n", name, m, v, b, b1, b2, m + v, m > v ? m : v);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    trio<0, 0>(out, "fp32 mfma 32x32x2 + v_pk_fma_f32", 1000, 4000);
    trio<0, 1>(out, "fp32 mfma 32x32x2 + v_fma_f32", 1000, 4000);
    trio<0, 2>(out, "fp32 mfma 32x32x2 + dpp mov/add", 1000, 2000);
    trio<1, 0>(out, "bf16 mfma 32x32x16 + v_pk_fma_f32", 2000, 4000);
    trio<1, 1>(out, "bf16 mfma 32x32x16 + v_fma_f32", 2000, 4000);
    return 0;
}
