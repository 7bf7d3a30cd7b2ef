// Can VALU instructions of the SAME wave run in the shadow of its own MFMAs?  One wave per SIMD (256 threads, one workgroup
// per CU); per iteration 16 x [1 dependent v_mfma_f32_32x32x2_f32 + NV independent v_fma_f32 / v_pk_fma_f32].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NV, int PK, int MF>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a0) {
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    float y[16];
    f32x2 x[16];
    for (int r = 0; r < 16; ++r) { y[r] = a0 + r + threadIdx.x; x[r] = f32x2{a0 + r, a0 - r}; }
    const f32x2 m = {1.0001f, 0.9999f}, d = {1e-3f, -1e-3f};
    float a = a0 + threadIdx.x, b = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MF) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (PK) x[v & 15] = __builtin_elementwise_fma(x[v & 15], m, d);
                else y[v & 15] = __builtin_fmaf(y[v & 15], 1.0001f, 1e-3f);
            }
            if (MF) __builtin_amdgcn_sched_barrier(0);       // keep the groups where they are written
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c[r] + y[r] + x[r][0] + x[r][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int PK, int MF>
static float run(float* out, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<NV, PK, MF><<<256, 256>>>(out, iters, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NV, PK, MF><<<256, 256>>>(out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
template <int NV, int PK>
static void row(float* out) {
    const int iters = 1000;
    const float both = run<NV, PK, 1>(out, iters), valu = run<NV, PK, 0>(out, iters), mf = run<0, 0, 1>(out, iters);
    printf("per MFMA: %2d %-12s  mfma alone %6.1f us | valu alone %6.1f us | interleaved %6.1f us  (sum %6.1f)\n", NV,
           PK ? "v_pk_fma_f32" : "v_fma_f32", mf, valu, both, mf + valu);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    row<4, 0>(out); row<8, 0>(out); row<16, 0>(out); row<24, 0>(out);
    row<4, 1>(out); row<8, 1>(out); row<12, 1>(out);
    return 0;
}
