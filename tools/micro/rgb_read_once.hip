// Calibration for rocprofv3's FETCH_SIZE on the encode kernel's pixel-load pattern: one wave per chunk of
// 64 MCUs, lane = MCU, 8 rows x 24 bytes per lane as three 8-byte loads per row, every byte of the
// frames read exactly once (the halo and the per-component re-reads of the real kernel are left out).
// usage: rocprofv3 --pmc FETCH_SIZE -- rgb_read_once [frames]   (known bytes = frames * 1920*1080*3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ __launch_bounds__(64) void k_rgb_once(const uint8_t* rgb, int64_t stride, int64_t image_stride,
                                                 uint32_t mcu_w, uint32_t n_mcu, uint32_t cpi, uint32_t* out) {
    const int lane = threadIdx.x;
    const uint32_t gid = blockIdx.x, img = gid / cpi, c = gid - img * cpi;
    const uint32_t b = c * 64 + lane;
    const uint32_t bb = b < n_mcu ? b : n_mcu - 1;
    const uint32_t by = bb / mcu_w, bx = bb - by * mcu_w;
    const uint8_t* src = rgb + (int64_t)img * image_stride + (int64_t)by * 8 * stride + (int64_t)bx * 24;
    uint32_t t = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint2* q = reinterpret_cast<const uint2*>(src + (int64_t)r * stride);
        const uint2 a = q[0], b2 = q[1], c2 = q[2];
        t ^= a.x ^ a.y ^ b2.x ^ b2.y ^ c2.x ^ c2.y;
    }
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
    if (lane == 0) out[gid] = t;
}

int main(int argc, char** argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 64;
    const int w = 1920, h = 1080;
    const uint32_t mcu_w = w / 8, n_mcu = mcu_w * (h / 8), cpi = (n_mcu + 63) / 64;
    uint8_t* rgb; uint32_t* out;
    const size_t bytes = (size_t)frames * w * h * 3;
    if (hipMalloc(&rgb, bytes + 4096) != hipSuccess || hipMalloc(&out, (size_t)frames * cpi * 4) != hipSuccess) return 1;
    hipMemset(rgb, 7, bytes + 4096);
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL(k_rgb_once, dim3(frames * cpi), dim3(64), 0, 0, rgb, (int64_t)w * 3, (int64_t)w * h * 3, mcu_w, n_mcu, cpi, out);
    hipDeviceSynchronize();
    printf("frames %d known_bytes_per_launch %zu\n", frames, bytes);
    return 0;
}
