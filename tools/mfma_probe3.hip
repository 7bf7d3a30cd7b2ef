// Probe 3: one wave per SIMD (256 threads, 1 workgroup per CU, the whole 512-register file per wave).
// Weight-stationary NNConv inner structure: wave = 16 channels, lane = (row, slab half); the operand values of
// tile t+1 (80 per lane: 5 slabs x 16 channels, formed from LDS rows + per-edge weights, 12 edge slots in 6
// steps of 2) are formed while the 160 MFMAs of tile t issue from the same instruction stream.
//  V0: MFMAs only (operand values constant)      V1: operand formation only
//  V2: both, formation first then MFMAs           V3: both, interleaved per step (6 x [2 edges, 26-27 MFMAs])
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SLOTS = 192, HW = 12;

__device__ __forceinline__ void agen_step(const float4* __restrict__ xb4, const float* __restrict__ hb, int beg, int deg,
                                          int ownslot, int d0, int cq, int hoff, float (&a)[5][16]) {
    float wv[2][5];
    float4 x[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int d = d0 + u;
        const bool live = d < deg;
        const int xslot = live ? beg + d : ownslot;
        const int wrow = live ? beg + d : SLOTS - 1;
        const float2* wp = reinterpret_cast<const float2*>(hb + wrow * HW + hoff);
        const float2 w01 = wp[0], w23 = wp[1];
        wv[u][0] = w01.x; wv[u][1] = w01.y; wv[u][2] = w23.x; wv[u][3] = w23.y;
        wv[u][4] = hb[wrow * HW + hoff + 4];
        const int sw = xslot & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[u][q] = xb4[xslot * 16 + ((cq + q) ^ sw)];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[j][4 * q + 0] = fmaf(wv[u][j], x[u][q].x, a[j][4 * q + 0]);
                a[j][4 * q + 1] = fmaf(wv[u][j], x[u][q].y, a[j][4 * q + 1]);
                a[j][4 * q + 2] = fmaf(wv[u][j], x[u][q].z, a[j][4 * q + 2]);
                a[j][4 * q + 3] = fmaf(wv[u][j], x[u][q].w, a[j][4 * q + 3]);
            }
}

struct EdgeRegs { float wv[2][5]; float4 x[2][4]; };
__device__ __forceinline__ void agen_load(const float4* __restrict__ xb4, const float* __restrict__ hb, int beg, int deg,
                                          int ownslot, int d0, int cq, int hoff, EdgeRegs& e) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int d = d0 + u;
        const bool live = d < deg;
        const int xslot = live ? beg + d : ownslot;
        const int wrow = live ? beg + d : SLOTS - 1;
        const float2* wp = reinterpret_cast<const float2*>(hb + wrow * HW + hoff);
        const float2 w01 = wp[0], w23 = wp[1];
        e.wv[u][0] = w01.x; e.wv[u][1] = w01.y; e.wv[u][2] = w23.x; e.wv[u][3] = w23.y;
        e.wv[u][4] = hb[wrow * HW + hoff + 4];
        const int sw = xslot & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) e.x[u][q] = xb4[xslot * 16 + ((cq + q) ^ sw)];
    }
}
__device__ __forceinline__ void agen_fma(const EdgeRegs& e, float (&a)[5][16]) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[j][4 * q + 0] = fmaf(e.wv[u][j], e.x[u][q].x, a[j][4 * q + 0]);
                a[j][4 * q + 1] = fmaf(e.wv[u][j], e.x[u][q].y, a[j][4 * q + 1]);
                a[j][4 * q + 2] = fmaf(e.wv[u][j], e.x[u][q].z, a[j][4 * q + 2]);
                a[j][4 * q + 3] = fmaf(e.wv[u][j], e.x[u][q].w, a[j][4 * q + 3]);
            }
}
// 13 x [2 MFMAs (both column halves of one operand value), NV VALU] in issue order
template <int NV>
__device__ __forceinline__ void sched_interleave13() {
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
}

template <int LO, int HI>
__device__ __forceinline__ void mfma_range(const float (&a)[5][16], const float (&w)[5][16][2], f32x16& c0, f32x16& c1) {
#pragma unroll
    for (int s = LO; s < HI; ++s) {
        const int j = s / 16, i = s % 16;
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][i], w[j][i][0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][i], w[j][i][1], c1, 0, 0, 0);
    }
}

template <int V>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ Wp, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float xbuf[SLOTS * 64];
    __shared__ __attribute__((aligned(16))) float hbuf[SLOTS * HW];
    for (int t = threadIdx.x; t < SLOTS * 64; t += 256) xbuf[t] = (float)(t % 7) * 0.125f - 0.3f;
    for (int t = threadIdx.x; t < SLOTS * HW; t += 256) hbuf[t] = (t >= (SLOTS - 1) * HW) ? 0.f : (float)(t % 5) * 0.25f - 0.4f;
    __syncthreads();
    const float4* xb4 = reinterpret_cast<const float4*>(xbuf);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int cq = 4 * wave, hoff = 6 * h;
    float wreg[5][16][2];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) wreg[j][i][nh] = Wp[(((h * 5 + j) * 64 + 16 * wave + i) * 64) + nh * 32 + r];
    f32x16 c0, c1;
    for (int q = 0; q < 16; ++q) { c0[q] = 0.f; c1[q] = 0.f; }
    float acur[5][16], anext[5][16];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acur[j][i] = 0.01f * (j + i) + lane * 0.001f; anext[j][i] = 0.f; }
    const int deg = 3 + (r % 5), ownslot = 160 + r;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const int beg = (4 * r + (r >> 2) + 7 * it) % 150;      // varies per tile: nothing is loop invariant
        if (V == 0) {
            mfma_range<0, 80>(acur, wreg, c0, c1);
        } else if (V == 1) {
#pragma unroll
            for (int s = 0; s < 6; ++s) agen_step(xb4, hbuf, beg, deg, ownslot, 2 * s, cq, hoff, anext);
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) { acur[j][i] += anext[j][i]; anext[j][i] = 0.f; }
        } else if (V == 2) {
#pragma unroll
            for (int s = 0; s < 6; ++s) agen_step(xb4, hbuf, beg, deg, ownslot, 2 * s, cq, hoff, anext);
            mfma_range<0, 80>(acur, wreg, c0, c1);
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) { acur[j][i] = anext[j][i] * 0.125f; anext[j][i] = 0.f; }
        } else if (V == 4 || V == 5) {
            // next step's LDS reads first, this step's FMAs spread between the MFMAs
            EdgeRegs e0, e1;
            agen_load(xb4, hbuf, beg, deg, ownslot, 0, cq, hoff, e0);
#define STEP(S, LO, HI, EC, EN)                                                                     \
            if (S < 5) agen_load(xb4, hbuf, beg, deg, ownslot, 2 * (S + 1), cq, hoff, EN);         \
            agen_fma(EC, anext);                                                                    \
            mfma_range<LO, HI>(acur, wreg, c0, c1);                                                 \
            if (V == 5) {                                                                           \
                __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);                                 \
                __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);                                 \
                sched_interleave13<4>();                                                            \
            }
            STEP(0, 0, 13, e0, e1) STEP(1, 13, 26, e1, e0) STEP(2, 26, 39, e0, e1)
            STEP(3, 39, 52, e1, e0) STEP(4, 52, 65, e0, e1) STEP(5, 65, 80, e1, e0)
#undef STEP
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) { acur[j][i] = anext[j][i] * 0.125f; anext[j][i] = 0.f; }
        } else {
            agen_step(xb4, hbuf, beg, deg, ownslot, 0, cq, hoff, anext);  mfma_range<0, 13>(acur, wreg, c0, c1);
            agen_step(xb4, hbuf, beg, deg, ownslot, 2, cq, hoff, anext);  mfma_range<13, 26>(acur, wreg, c0, c1);
            agen_step(xb4, hbuf, beg, deg, ownslot, 4, cq, hoff, anext);  mfma_range<26, 39>(acur, wreg, c0, c1);
            agen_step(xb4, hbuf, beg, deg, ownslot, 6, cq, hoff, anext);  mfma_range<39, 52>(acur, wreg, c0, c1);
            agen_step(xb4, hbuf, beg, deg, ownslot, 8, cq, hoff, anext);  mfma_range<52, 65>(acur, wreg, c0, c1);
            agen_step(xb4, hbuf, beg, deg, ownslot, 10, cq, hoff, anext); mfma_range<65, 80>(acur, wreg, c0, c1);
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) { acur[j][i] = anext[j][i] * 0.125f; anext[j][i] = 0.f; }
        }
    }
    float s = 0;
    for (int q = 0; q < 16; ++q) s += c0[q] + c1[q];
    for (int j = 0; j < 5; ++j) for (int i = 0; i < 16; ++i) s += acur[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V>
void run(const char* name, float* Wp, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256, iters = 50;
    probe<V><<<blocks, 256>>>(Wp, out, iters); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0); probe<V><<<blocks, 256>>>(Wp, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double flops = (double)blocks * 4 * iters * 160 * 4096.0;
    printf("%-62s %8.1f us  %7.1f TFLOP/s (if the 160 MFMAs per tile count)  %6.2f us/tile\n", name, best * 1e3, flops / best / 1e9,
           best * 1e3 / iters);
}
int main() {
    float *Wp, *out; (void)hipMalloc(&Wp, 640 * 64 * 4); (void)hipMalloc(&out, 8192 * 256 * 4);
    std::vector<float> hv(640 * 64); for (size_t i = 0; i < hv.size(); ++i) hv[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    (void)hipMemcpy(Wp, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
    run<0>("V0 160 MFMAs per tile, operands constant", Wp, out);
    run<1>("V1 operand formation only (12 edge slots)", Wp, out);
    run<2>("V2 formation, then MFMAs", Wp, out);
    run<3>("V3 interleaved: 6 x [2 edge slots, 26 MFMAs]", Wp, out);
    run<4>("V4 as V3, LDS reads one step ahead", Wp, out);
    run<5>("V5 as V4 + sched_group_barrier: 13 x [2 MFMA, 4 VALU]", Wp, out);
    return 0;
}
