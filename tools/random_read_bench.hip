// random_read_bench.hip — what the MI355X memory system delivers for the access pattern of
// probe_kernel: independent random reads of S bytes (S = 16, 32, 64, 128) from a table much
// larger than the Infinity Cache, G = S/16 adjacent lanes reading one record (16 B each),
// four records in flight per lane group.  Prints requested-bytes GB/s per record size.
//   hipcc --offload-arch=gfx950 -O3 -o random_read_bench tools/random_read_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

template <int G>  // lanes per record
__global__ __launch_bounds__(256) void rr_kernel(const uint4 *table, uint64_t n_records, uint32_t iters, uint32_t *sink)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / G, sub = tid % G;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t h = mix32(grp * 0x9E3779B1u + (it * 4 + j) * 0x85ebca6bu + 12345u);
            const uint64_t rec = ((uint64_t)h * n_records) >> 32;
            v[j] = table[rec * G + sub];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// 64-B records read in PAIRS whose addresses differ by `stride` bytes (same aligned 2*stride block):
// if two requests close in address and time cost less than two independent ones, the ceiling is
// DRAM row activation (and the stride where the gain stops is the channel interleave), not the
// request path.
__global__ __launch_bounds__(256) void rr_pair_kernel(const uint4 *table, uint64_t n_records, uint32_t iters, uint32_t stride_recs,
                                                      uint32_t *sink)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / 4, sub = tid % 4;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t h = mix32(grp * 0x9E3779B1u + (it * 2 + j) * 0x85ebca6bu + 12345u);
            const uint64_t rec = (((uint64_t)h * n_records) >> 32) & ~(uint64_t)(2 * stride_recs - 1);
            v[2 * j] = table[rec * 4 + sub];
            v[2 * j + 1] = table[(rec + stride_recs) * 4 + sub];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static void run_pair(const uint4 *d, uint64_t bytes, uint32_t stride_bytes, uint32_t *sink)
{
    const uint64_t n_records = bytes / 64ull;
    const int blocks = 256 * 8, threads = 256;
    const uint32_t iters = 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(rr_pair_kernel, dim3(blocks), dim3(threads), 0, 0, d, n_records, 4u, stride_bytes / 64, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(rr_pair_kernel, dim3(blocks), dim3(threads), 0, 0, d, n_records, iters, stride_bytes / 64, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double recs = (double)blocks * threads / 4 * iters * 4;
    printf("64-B records in pairs %6u B apart, table %5.0f MiB: %.2f G records/s\n", stride_bytes, bytes / 1048576.0, recs / ms / 1e6);
}

template <int G> static void run(const uint4 *d, uint64_t bytes, uint32_t *sink)
{
    const uint64_t n_records = bytes / (16ull * G);
    const int blocks = 256 * 8, threads = 256;
    const uint32_t iters = 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(rr_kernel<G>, dim3(blocks), dim3(threads), 0, 0, d, n_records, 4u, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(rr_kernel<G>, dim3(blocks), dim3(threads), 0, 0, d, n_records, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double recs = (double)blocks * threads / G * iters * 4;
    printf("record %4d B: %8.1f M records/ms... %.2f G records/s, %.1f GB/s requested\n", 16 * G, recs / ms / 1e6 * 1e3 / 1e3,
           recs / ms / 1e6, recs * 16.0 * G / ms / 1e6);
}

int main()
{
    const uint64_t bytes = 3ull << 30;  // 3 GiB table
    uint4 *d = nullptr;
    uint32_t *sink = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    run<1>(d, bytes, sink);
    run<2>(d, bytes, sink);
    run<4>(d, bytes, sink);
    run<8>(d, bytes, sink);
    // footprint: does the request rate depend on the table size (Infinity Cache 256 MiB, TLB reach)?
    for (uint64_t mb : {32ull, 128ull, 512ull, 1024ull}) { printf("table %4llu MiB: ", (unsigned long long)mb); run<4>(d, mb << 20, sink); }
    for (uint32_t st : {64u, 128u, 256u, 512u, 1024u, 2048u, 4096u, 65536u}) run_pair(d, bytes, st, sink);
    // streaming reference: copy
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        uint4 *e = nullptr; hipMalloc(&e, 1ull << 30);
        hipMemcpy(e, d, 1ull << 30, hipMemcpyDeviceToDevice);
        hipEventRecord(a);
        for (int i = 0; i < 5; i++) hipMemcpy(e, d + (i & 1) * (1ull << 26), 1ull << 30, hipMemcpyDeviceToDevice);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("D2D copy 1 GiB: %.1f GB/s (read+write %.1f GB/s)\n", 5.0 * 1.0737 / ms * 1e3, 2 * 5.0 * 1.0737 / ms * 1e3);
    }
    return 0;
}
