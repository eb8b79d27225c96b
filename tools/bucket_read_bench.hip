// bucket_read_bench.hip — how should a wave read 64 random 128-byte buckets?
// L lanes per bucket, each lane reading 8/L pieces of 16 B (piece i of lane l = cell l + i*L, so the L lanes of one
// load instruction cover L*16 contiguous bytes): L = 8 is one load per lane and 8 buckets per instruction (what the
// search kernel does), L = 1 is eight loads per lane into the lane's own bucket (64 different lines per instruction,
// each line touched by 8 consecutive instructions).  Prints buckets/s per L.
//   hipcc --offload-arch=gfx950 -O3 -o build/bucket_read_bench tools/bucket_read_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

template <int L>
__global__ __launch_bounds__(256) void br_kernel(const uint4 *table, uint64_t n_buckets, uint32_t iters, uint32_t *sink)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / L, sub = tid % L;
    constexpr int P = 8 / L;          // pieces per lane per bucket
    constexpr int R = L;              // buckets per lane group per iteration, so that 8 loads per lane are in flight
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[R * P];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t h = mix32(grp * 0x9E3779B1u + (it * R + r) * 0x85ebca6bu + 12345u);
            const uint64_t b = ((uint64_t)h * n_buckets) >> 32;
#pragma unroll
            for (int i = 0; i < P; i++) v[r * P + i] = table[b * 8 + sub + i * L];
        }
#pragma unroll
        for (int j = 0; j < R * P; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int L> static void run(const uint4 *d, uint64_t bytes, uint32_t *sink)
{
    const uint64_t n_buckets = bytes / 128;
    const int blocks = 256 * 8, threads = 256;
    const uint32_t iters = 32;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(br_kernel<L>, dim3(blocks), dim3(threads), 0, 0, d, n_buckets, 2u, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(br_kernel<L>, dim3(blocks), dim3(threads), 0, 0, d, n_buckets, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double buckets = (double)blocks * threads / L * iters * L;
    printf("lanes per bucket %d (%d loads per lane per bucket): %.2f G buckets/s, %.1f GB/s\n", L, 8 / L, buckets / ms / 1e6, buckets * 128 / ms / 1e6);
}

int main()
{
    const uint64_t bytes = 4ull << 30;
    uint4 *d = nullptr;
    uint32_t *sink = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    run<8>(d, bytes, sink);
    run<4>(d, bytes, sink);
    run<2>(d, bytes, sink);
    run<1>(d, bytes, sink);
    return 0;
}
