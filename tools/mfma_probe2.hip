// Probe 2: LDS-fed v_mfma_f32_32x32x2_f32 variants (persistent loops, 512 blocks of 256 threads, 2 WG/CU).
//  V0: regs only            V1: ds_read_b32 + swizzle per MFMA (old)     V2: ds_read_b128 per 4 MFMAs
//  V3: ds_read_b128 per 8 MFMAs (A fragment reused for two accumulators = both column halves)
//  V4: as V2 with B fragments from global (L2)    V5: as V3 with B from global
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int slot(int g, int hi, int i) { return (2 * g + hi) * 32 + (i ^ (g & 7)); }
template <int V>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ Wp, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float At[576 * 32];
    for (int t = threadIdx.x; t < 576 * 32; t += 256) At[t] = (float)(t % 7) * 0.125f;
    __syncthreads();
    const float4* A4 = reinterpret_cast<const float4*>(At);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r31 = lane & 31, hi = lane >> 5;
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    float a = 1.0f + lane * 0.001f, b = 0.5f - lane * 0.002f;
    const float4* wp = reinterpret_cast<const float4*>(Wp) + lane + wave * 36 * 64;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
#pragma unroll
            for (int s = 0; s < 144; ++s) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        } else if (V == 2 || V == 4) {
#pragma unroll 1
            for (int ch = 0; ch < 12; ++ch) {
                float4 bb[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) bb[u] = (V == 4) ? wp[(ch * 3 + u) * 64] : make_float4(b, b, b, b);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const float4 av = A4[slot((wave >> 1) * 36 + ch * 3 + u, hi, r31)];
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bb[u].x, c0, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bb[u].y, c0, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bb[u].z, c0, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bb[u].w, c0, 0, 0, 0);
                }
            }
        } else if (V == 3 || V == 5) {
#pragma unroll 1
            for (int ch = 0; ch < 6; ++ch) {      // 18 groups x 8 MFMAs = 144
                float4 bb[3], bd[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    bb[u] = (V == 5) ? wp[(ch * 3 + u) * 64] : make_float4(b, b, b, b);
                    bd[u] = (V == 5) ? wp[(18 + ch * 3 + u) * 64] : make_float4(a, a, a, a);
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const float4 av = A4[slot(wave * 18 + ch * 3 + u, hi, r31)];
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bb[u].x, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bd[u].x, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bb[u].y, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bd[u].y, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bb[u].z, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bd[u].z, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bb[u].w, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bd[u].w, c1, 0, 0, 0);
                }
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V>
void run(const char* name, float* Wp, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 512, iters = 40;
    probe<V><<<blocks, 256>>>(Wp, out, iters); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0); probe<V><<<blocks, 256>>>(Wp, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double flops = (double)blocks * 4 * iters * 144 * 4096.0;
    printf("%-62s %8.1f us  %7.1f TFLOP/s\n", name, best * 1e3, flops / best / 1e9);
}
int main() {
    float *Wp, *out; (void)hipMalloc(&Wp, 640 * 64 * 4 * 4); (void)hipMalloc(&out, 8192 * 256 * 4);
    std::vector<float> h(640 * 64 * 4); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    (void)hipMemcpy(Wp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("V0 regs only", Wp, out);
    run<2>("V2 ds_read_b128 per 4 MFMA, B in regs", Wp, out);
    run<3>("V3 ds_read_b128 per 8 MFMA (2 accumulators), B in regs", Wp, out);
    run<4>("V4 ds_read_b128 per 4 MFMA, B from L2", Wp, out);
    run<5>("V5 ds_read_b128 per 8 MFMA (2 accumulators), B from L2", Wp, out);
    return 0;
}
