// v_mfma_f32_4x4x1_16b_f32 as a rank-1 ("outer product") accumulator: semantics with the A block broadcast (cbsz = 4, abid = 0)
// and cycles per instruction (one wave per SIMD, three independent accumulators as the gather would use them).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void sem(const float* h, const float* x, float* out) {      // one wave: out[k][c] = h[k] * x[c], k < 4, c < 64
    const int lane = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const float a = lane < 4 ? h[lane] : -1000.f;                        // only block 0's A values may matter
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x[lane], c, 4, 0, 0);
    for (int i = 0; i < 4; ++i) out[i * 64 + lane] = c[i];
}
__global__ __launch_bounds__(256, 1) void thr(float* out, int iters, float a0) {
    f32x4 c[3];
    for (int g = 0; g < 3; ++g) c[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = 1e-3f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int g = 0; g < 3; ++g) c[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[g], 4, 0, 0);
    float s = 0.f;
    for (int g = 0; g < 3; ++g) s += c[g][0] + c[g][1] + c[g][2] + c[g][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float hh[4] = {1.f, 2.f, 3.f, 4.f}, hx[64], *dh, *dx, *dout, ho[256];
    for (int i = 0; i < 64; ++i) hx[i] = 0.5f + i;
    (void)hipMalloc(&dh, 16); (void)hipMalloc(&dx, 256); (void)hipMalloc(&dout, 256 * 256 * 4);
    (void)hipMemcpy(dh, hh, 16, hipMemcpyHostToDevice); (void)hipMemcpy(dx, hx, 256, hipMemcpyHostToDevice);
    sem<<<1, 64>>>(dh, dx, dout); (void)hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int k = 0; k < 4; ++k) for (int c = 0; c < 64; ++c) if (ho[k * 64 + c] != hh[k] * hx[c]) ++bad;
    printf("semantics: reg i of lane c = h[i] * x[c] with cbsz=4/abid=0: %s (%d mismatches; out[1][5] = %g, expected %g)\n", bad ? "NO" : "yes", bad, ho[64 + 5], hh[1] * hx[5]);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) thr<<<256, 256>>>(dout, iters, 1.f);
    (void)hipEventRecord(e0); thr<<<256, 256>>>(dout, iters, 1.f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 48;
    printf("throughput: %.1f us for %.0f MFMA 4x4x1 per SIMD: %.2f ns = %.1f cycles at 2.4 GHz each; %.1f TFLOP/s\n", ms * 1e3, n, ms * 1e6 / n,
           ms * 1e6 / n * 2.4, n * 1024 * 512.0 / (ms * 1e-3) / 1e12);
    return 0;
}
