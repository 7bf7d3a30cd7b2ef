// Census: are 2 workgroups with L bytes of LDS (and V-ish VGPRs) co-resident on one CU?
// Each block sleeps 100 us of wall clock; 512 blocks take ~100 us if 2/CU are resident, ~200 us if not.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_FLOATS>
__global__ __launch_bounds__(256, 2) void sleeper(float* out) {
    __shared__ float buf[LDS_FLOATS];
    buf[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 10000ull) __builtin_amdgcn_s_sleep(32);
    out[blockIdx.x * 256 + threadIdx.x] = buf[(threadIdx.x * 7) % LDS_FLOATS];
}
template <int LDS_FLOATS>
void run(float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {256, 512, 768}) {
        sleeper<LDS_FLOATS><<<blocks, 256>>>(out); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); sleeper<LDS_FLOATS><<<blocks, 256>>>(out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        int occ = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sleeper<LDS_FLOATS>, 256, 0);
        printf("LDS %6d B  blocks %4d  %7.1f us   (occupancy API: %d blocks/CU)\n", LDS_FLOATS * 4, blocks, ms * 1e3, occ);
    }
}
int main() {
    float* out; (void)hipMalloc(&out, 1024 * 256 * 4);
    run<18432>(out);   // 73,728 B  (the fused NNConv tile)
    run<20480>(out);   // 81,920 B
    run<16384>(out);   // 65,536 B
    run<16000>(out);   // 64,000 B
    run<12288>(out);   // 49,152 B
    return 0;
}
