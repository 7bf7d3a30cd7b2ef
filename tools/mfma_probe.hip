// Probe: achievable v_mfma_f32_32x32x2_f32 rate + in-kernel clock, with/without the LDS A-fragment reads.
// Build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int swz(int k) { return (k + (k >> 5)) & 31; }

// MODE 0: regs only; 1: ds_read_b32 A fragments (swizzled); 2: + B fragments from global (L2)
template <int MODE, int SHAPE>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ Wp, float* __restrict__ out, int iters,
                                                unsigned long long* clk) {
    __shared__ float At[576 * 32];
    for (int t = threadIdx.x; t < 576 * 32; t += 256) At[t] = (float)(t % 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r31 = lane & 31, hi = lane >> 5;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    f32x4 c4[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) c4[q][r] = 0.f;
    float a = 1.0f + lane * 0.001f, b = 0.5f - lane * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const float4* wp = reinterpret_cast<const float4*>(Wp) + lane + wave * 36 * 64;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 144; ++s) {
                if (SHAPE == 32) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
                else c4[s & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4[s & 3], 0, 0, 0);
            }
        } else {
#pragma unroll 1
            for (int ch = 0; ch < 6; ++ch) {
                float4 bb[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) bb[u] = (MODE == 2) ? wp[(ch * 6 + u) * 64] : make_float4(b, b, b, b);
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const float bv[4] = {bb[u].x, bb[u].y, bb[u].z, bb[u].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = (wave >> 1) * 288 + (ch * 6 + u) * 8 + hi + 2 * r;
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(At[k * 32 + (r31 ^ swz(k))], bv[r], c, 0, 0, 0);
                    }
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c[r];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) s += c4[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int SHAPE>
void run(const char* name, int blocks, int iters, float* Wp, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, SHAPE><<<blocks, 256>>>(Wp, out, iters, clk);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        probe<MODE, SHAPE><<<blocks, 256>>>(Wp, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 144 * (SHAPE == 32 ? 4096.0 : 2048.0);
    printf("%-44s blocks=%5d  %8.1f us  %7.1f TFLOP/s  clock %.2f GHz\n", name, blocks, best * 1e3, flops / best / 1e9,
           (double)h[0] / (double)h[1] * 0.1);
}

int main() {
    float *Wp, *out;
    unsigned long long* clk;
    hipMalloc(&Wp, 640 * 64 * 4 * 2); hipMalloc(&out, 8192 * 256 * 4); hipMalloc(&clk, 16);
    std::vector<float> h(640 * 64 * 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(Wp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 32>("regs only 32x32x2, 1 WG/CU", 256, 40, Wp, out, clk);
    run<0, 32>("regs only 32x32x2, 2 WG/CU", 512, 40, Wp, out, clk);
    run<0, 16>("regs only 16x16x4 (4 acc), 2 WG/CU", 512, 40, Wp, out, clk);
    run<1, 32>("LDS A frags, 2 WG/CU", 512, 40, Wp, out, clk);
    run<2, 32>("LDS A frags + global B, 2 WG/CU", 512, 40, Wp, out, clk);
    run<2, 32>("LDS A + global B, 3200 blocks x 1 iter", 3200, 1, Wp, out, clk);
    run<2, 32>("LDS A + global B, 512 blocks x 6 iter", 512, 6, Wp, out, clk);
    return 0;
}
