// Does a different vector instruction that reads the result of a dot instruction need wait states in between?
// Each kernel issues the pair from one inline-asm block (so the compiler inserts nothing) with 0 to 3 no-ops
// or independent vector instructions in between and compares with the host.
// Result on MI355X: v_dot4_i32_i8 needs three instructions of any kind in between (adjacent, one or two apart:
// most lanes read the old register contents); v_dot2_i32_i16 is interlocked (adjacent is right).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void k(const int* in, int* out, int b, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int a = in[i], r, x = a * 3;
    if constexpr (MODE == 0) asm volatile("v_dot4_i32_i8 %0, %1, %2, %3\n v_ashrrev_i32 %0, 3, %0" : "=&v"(r) : "v"(a), "s"(b), "v"(c));
    if constexpr (MODE == 1) asm volatile("v_dot4_i32_i8 %0, %1, %2, %3\n s_nop 0\n v_ashrrev_i32 %0, 3, %0" : "=&v"(r) : "v"(a), "s"(b), "v"(c));
    if constexpr (MODE == 2) asm volatile("v_dot4_i32_i8 %0, %1, %2, %3\n s_nop 1\n v_ashrrev_i32 %0, 3, %0" : "=&v"(r) : "v"(a), "s"(b), "v"(c));
    if constexpr (MODE == 3) asm volatile("v_dot4_i32_i8 %0, %1, %2, %3\n s_nop 2\n v_ashrrev_i32 %0, 3, %0" : "=&v"(r) : "v"(a), "s"(b), "v"(c));
    if constexpr (MODE == 4) asm volatile("v_dot4_i32_i8 %0, %2, %3, %4\n v_add_u32 %1, %1, %1\n v_ashrrev_i32 %0, 3, %0" : "=&v"(r), "+v"(x) : "v"(a), "s"(b), "v"(c));
    if constexpr (MODE == 5) asm volatile("v_dot2_i32_i16 %0, %1, %2, %3\n v_ashrrev_i32 %0, 3, %0" : "=&v"(r) : "v"(a), "s"(b), "v"(c));
    if constexpr (MODE == 6) asm volatile("v_dot4_i32_i8 %0, %2, %3, %4\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_ashrrev_i32 %0, 3, %0" : "=&v"(r), "+v"(x) : "v"(a), "s"(b), "v"(c));
    if constexpr (MODE == 7) asm volatile("v_dot2_i32_i16 %0, %1, %2, %3\n s_nop 2\n v_ashrrev_i32 %0, 3, %0" : "=&v"(r) : "v"(a), "s"(b), "v"(c));
    if constexpr (MODE == 8) asm volatile("v_dot4_i32_i8 %0, %2, %3, %4\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n v_ashrrev_i32 %0, 3, %0" : "=&v"(r), "+v"(x) : "v"(a), "s"(b), "v"(c));
    out[i] = r + (x & 0);
}
static int dot4(int a, int b, int c) { int s = c; for (int i = 0; i < 4; ++i) s += (int)(int8_t)(a >> (8 * i)) * (int)(int8_t)(b >> (8 * i)); return s; }
static int dot2(int a, int b, int c) { return c + (int)(int16_t)a * (int)(int16_t)b + (int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16); }
template <int MODE> void run(const char* nm, int* din, int* dout, const int* h, int n, int b, int c, bool d2) {
    k<MODE><<<n / 256, 256>>>(din, dout, b, c);
    int* o = new int[n]; hipMemcpy(o, dout, n * 4, hipMemcpyDeviceToHost);
    long bad = 0; for (int i = 0; i < n; ++i) { int e = (d2 ? dot2(h[i], b, c) : dot4(h[i], b, c)) >> 3; bad += (o[i] != e); }
    printf("%-28s wrong %ld of %d\n", nm, bad, n); delete[] o;
}
int main() {
    const int n = 1 << 22; int* h = new int[n]; uint32_t s = 1; for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)s; }
    int *din, *dout; hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4); hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
    const int b = 0x13fe2791, c = 123456;
    run<0>("dot4 adjacent", din, dout, h, n, b, c, false);
    run<1>("dot4 s_nop 0", din, dout, h, n, b, c, false);
    run<2>("dot4 s_nop 1", din, dout, h, n, b, c, false);
    run<3>("dot4 s_nop 2", din, dout, h, n, b, c, false);
    run<4>("dot4 one valu between", din, dout, h, n, b, c, false);
    run<8>("dot4 two valu between", din, dout, h, n, b, c, false);
    run<5>("dot2 adjacent", din, dout, h, n, b, c, true);
    run<6>("dot4 three valu between", din, dout, h, n, b, c, false);
    run<7>("dot2 s_nop 2", din, dout, h, n, b, c, true);
}
