// Dependent-chain throughput of v_mfma_f32_32x32x2_f32: one wave per SIMD (256 threads) or two (two workgroups per CU),
// ONE accumulator chain per wave or two interleaved ones.  Build: hipcc -O3 --offload-arch=gfx950 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void chain(float* out, int iters, float a0, float b0) {
    f32x16 c[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) c[n][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += c[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
static void run(int wg_per_cu, float* out) {
    const int iters = 2000 / NACC, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain<NACC><<<grid, 256>>>(out, iters, 1.f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<NACC><<<grid, 256>>>(out, iters, 1.f, 1e-3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 16 * NACC * wg_per_cu;
    printf("acc=%d wg/cu=%d: %.1f us, %.1f ns per MFMA per SIMD = %.1f cycles at 2.4 GHz, %.1f TFLOP/s\n", NACC, wg_per_cu,
           ms * 1e3, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4,
           mfma_per_simd * 1024 * 4096.0 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; hipMalloc(&out, 512 * 256 * 4);
    run<1>(1, out); run<2>(1, out); run<1>(2, out); run<2>(2, out); run<4>(1, out);
    return 0;
}
