// Microbenchmark: how fast can one wave per chunk read the first 4.5 KB of slots laid out at a stride?
// usage: sparse_read  (prints GB/s for several strides and depths)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int DEPTH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_read(const uint32_t* base, uint64_t stride_words, uint32_t n_chunks,
                                                      uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t gid = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (gid >= n_chunks) return;
    const uint32_t* S = base + gid * stride_words;
    uint4 w[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) w[u] = *reinterpret_cast<const uint4*>(S + 256 * u + 4 * lane);
    uint32_t t = 0;
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) t += w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
    if (lane == 0) out[gid] = t;
}

template <int DEPTH, int WAVES>
float run(const uint32_t* buf, uint64_t stride_words, uint32_t n, uint32_t* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_read<DEPTH, WAVES>), dim3((n + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, 0, buf, stride_words, n, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const uint32_t n = 107484;  // 212 frames x 507 chunks
    const uint64_t big = 9984;
    uint32_t *buf, *out;
    hipMalloc(&buf, (uint64_t)(n + 1) * big * 4);
    hipMalloc(&out, n * 4);
    hipMemset(buf, 1, (uint64_t)(n + 1) * big * 4);
    const uint64_t strides[3] = {9984, 2560, 1280};
    for (uint64_t st : strides) {
        float t1 = run<5, 1>(buf, st, n, out), t4 = run<5, 4>(buf, st, n, out), t2 = run<2, 4>(buf, st, n, out);
        printf("stride %5llu words: depth5 w1 %.3f ms (%.0f GB/s)  depth5 w4 %.3f ms (%.0f GB/s)  depth2 w4 %.3f ms (%.0f GB/s)\n",
               (unsigned long long)st, t1, n * 5120.0 / t1 / 1e6, t4, n * 5120.0 / t4 / 1e6, t2, n * 2048.0 / t2 / 1e6);
    }
    return 0;
}
