// Issue cost of the vector instructions the kernels are made of: a loop of 64 independent instances per
// iteration (8 accumulators x 8), timed with s_memtime, one workgroup of 4, 8, 12 or 16 waves per CU (1 to 4
// waves per SIMD).  Prints shader cycles per wave-instruction per SIMD (4.0 = 16 lanes/clock, 2.0 = 32).
// Result on MI355X (the output is kept under profiles/): with two or more waves per SIMD, add/sub, and/or/xor/
// not, v_bitop3_b32, right shifts, v_mov and f32 add/mul/fma on register or literal operands issue at ~2.4
// cycles; everything else (left shifts, 24/32-bit multiplies, dot2/dot4, packed 16-bit, bfe, alignbit, perm,
// ffbh, min/max, the other three-operand integer forms, cvt, DPP moves, and any instruction with a scalar
// register operand) at ~4.3; in a mixed stream the cheap ones only partly keep their advantage.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(1024) void k_cost(uint64_t* out, int iters, uint32_t seed) {
    uint32_t a[8], b = seed ^ (threadIdx.x & 63), c = seed * 3 + 7;
    uint64_t w[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i * 77 + threadIdx.x; w[i] = a[i]; }
    __shared__ uint32_t lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i * seed;
    __syncthreads();
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[0]), "v"(b) : "vcc");
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#define ONE(i) \
    if constexpr (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 1) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 2) asm volatile("v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 3) asm volatile("v_add_u32_sdwa %0, %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(a[i])); \
    else if constexpr (OP == 4) asm volatile("v_dot2_i32_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 5) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 6) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(w[i]) : "v"(b)); \
    else if constexpr (OP == 7) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 8) asm volatile("v_bfe_u32 %0, %0, 0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 9) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 10) asm volatile("v_ffbh_i32 %0, %0" : "+v"(a[i])); \
    else if constexpr (OP == 11) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 12) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 13) asm volatile("v_sub_u32_e64 %0, 32, %0 clamp" : "+v"(a[i])); \
    else if constexpr (OP == 14) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 15) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 16) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 17) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : ); \
    else if constexpr (OP == 19) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 20) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 21) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i])); \
    else if constexpr (OP == 22) asm volatile("v_dot4_i32_i8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 23) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 25) asm volatile("v_pk_lshlrev_b16 %0, 4, %0" : "+v"(a[i])); \
    else if constexpr (OP == 26) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 30) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 31) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 33) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 34) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i])); \
    else if constexpr (OP == 35) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[i])); \
    else if constexpr (OP == 36) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 37) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 38) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 39) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 40) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 41) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xfe" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 42) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf8" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 43) asm volatile("v_and_b32 %0, 0x1ffffffc, %0" : "+v"(a[i])); \
    else if constexpr (OP == 44) asm volatile("v_add_u32 %0, 32, %0" : "+v"(a[i])); \
    else if constexpr (OP == 45) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 46) asm volatile("v_not_b32 %0, %0" : "+v"(a[i])); \
    else if constexpr (OP == 47) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 48) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 49) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc"); \
    else if constexpr (OP == 50) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(c)); \
    else if constexpr (OP == 51) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 52) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(w[i])); \
    else if constexpr (OP == 53) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc"); \
    else if constexpr (OP == 54) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"); \
    else if constexpr (OP == 55) asm volatile("v_sub_u32_e64 %0, 32, %0" : "+v"(a[i])); \
    else if constexpr (OP == 56) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 57) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(c)); \
    else if constexpr (OP == 90) { if constexpr (true) { if ((i & 1) == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } } \
    else if constexpr (OP == 91) { if ((i & 2) == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } \
    else if constexpr (OP == 92) { if ((i & 3) != 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } \
    else if constexpr (OP == 93) { if ((i & 3) == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } \
    else if constexpr (OP == 60) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[i])); \
    else if constexpr (OP == 61) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(a[i]) : "s"(c), "v"(b)); \
    else if constexpr (OP == 62) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 63) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[i])); \
    else if constexpr (OP == 64) asm volatile("v_bfe_i32 %0, %0, 12, 11" : "+v"(a[i])); \
    else if constexpr (OP == 65) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 66) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 67) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 68) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[i])); \
    else if constexpr (OP == 69) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[i]) : "s"(c)); \
    else if constexpr (OP == 70) asm volatile("v_sub_u32 %0, 32, %0" : "+v"(a[i])); \
    else if constexpr (OP == 71) asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(a[i])); \
    else if constexpr (OP == 72) asm volatile("v_max_u32 %0, 5, %0" : "+v"(a[i])); \
    else if constexpr (OP == 73) asm volatile("v_and_b32 %0, 31, %0" : "+v"(a[i])); \
    else if constexpr (OP == 74) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 75) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 76) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i])); \
    else if constexpr (OP == 77) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if constexpr (OP == 78) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 79) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i])); \
    else if constexpr (OP == 80) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if constexpr (OP == 81) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[i])); \
    else if constexpr (OP == 82) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP8(ONE)
#undef ONE
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)w[i];
    if (s == 0x12345678u) out[1] = s;  // keep the chain alive
    // every wave of workgroup 0 reports; the host takes the longest
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[2 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, uint64_t* d_out) {
    const int iters = 2000;
    double res[4];
    for (int wps = 1; wps <= 4; ++wps) {
        // wps waves per SIMD on every CU: 256 CUs x 4 SIMDs x wps single-wave workgroups
        // one workgroup of 4 * wps waves per CU: the waves of a workgroup are spread over the CU's 4 SIMDs
        const int grid = 256;
        hipLaunchKernelGGL((k_cost<OP>), dim3(grid), dim3(256 * wps), 0, 0, d_out, iters, 12345u);
        hipLaunchKernelGGL((k_cost<OP>), dim3(grid), dim3(256 * wps), 0, 0, d_out, iters, 12345u);
        hipDeviceSynchronize();
        uint64_t h[32];
        hipMemcpy(h, d_out, 32 * 8, hipMemcpyDeviceToHost);
        uint64_t mx = 0;
        for (int i = 0; i < 4 * wps; ++i) mx = h[2 + i] > mx ? h[2 + i] : mx;
        res[wps - 1] = (double)mx / (iters * 64.0) / wps;  // cycles per instruction per SIMD if waves share it evenly
    }
    printf("%-22s 1 wave/SIMD %.2f   2 waves %.2f   3 waves %.2f   4 waves %.2f\n", name, res[0], res[1], res[2], res[3]);
}

int main() {
    uint64_t* d_out;
    hipMalloc(&d_out, 512);
    hipMemset(d_out, 0, 512);
    run<0>("v_add_u32", d_out);
    run<1>("v_mad_i32_i24", d_out);
    run<26>("v_mad_u32_u24", d_out);
    run<24>("v_mul_u32_u24", d_out);
    run<2>("v_mul_i32_i24_sdwa", d_out);
    run<3>("v_add_u32_sdwa sext", d_out);
    run<4>("v_dot2_i32_i16", d_out);
    run<22>("v_dot4_i32_i8", d_out);
    run<5>("v_pk_add_i16", d_out);
    run<23>("v_pk_mad_i16", d_out);
    run<25>("v_pk_lshlrev_b16", d_out);
    run<6>("v_lshlrev_b64", d_out);
    run<7>("v_alignbit_b32", d_out);
    run<8>("v_bfe_u32", d_out);
    run<9>("v_perm_b32", d_out);
    run<10>("v_ffbh_i32", d_out);
    run<11>("v_or3_b32", d_out);
    run<12>("v_bitop3_b32", d_out);
    run<13>("v_sub_u32 clamp", d_out);
    run<14>("v_add3_u32", d_out);
    run<15>("v_lshl_or_b32", d_out);
    run<17>("v_lshl_add_u32", d_out);
    run<20>("v_and_or_b32", d_out);
    run<16>("v_mul_lo_u32", d_out);
    run<18>("v_cndmask_b32", d_out);
    run<19>("v_bcnt_u32_b32", d_out);
    run<21>("v_mov_b32_dpp row_shr", d_out);
    run<30>("v_and_b32", d_out);
    run<31>("v_or_b32", d_out);
    run<32>("v_xor_b32", d_out);
    run<33>("v_lshlrev_b32", d_out);
    run<34>("v_lshrrev_b32 imm", d_out);
    run<35>("v_ashrrev_i32 imm", d_out);
    run<36>("v_min_u32", d_out);
    run<37>("v_max_i32", d_out);
    run<38>("v_mov_b32", d_out);
    run<39>("v_sub_u32 e32", d_out);
    run<40>("v_add_u32 e64", d_out);
    run<41>("v_bitop3 (or3)", d_out);
    run<42>("v_bitop3 (and_or)", d_out);
    run<43>("v_and_b32 literal", d_out);
    run<44>("v_add_u32 inline const", d_out);
    run<45>("v_bfi_b32", d_out);
    run<46>("v_not_b32", d_out);
    run<47>("v_add_u16", d_out);
    run<48>("v_pk_add_u16", d_out);
    run<49>("v_cmp_lt_u32 vcc", d_out);
    run<50>("v_add_u32 sgpr src", d_out);
    run<51>("v_xad_u32", d_out);
    run<52>("v_lshl_add_u64", d_out);
    run<53>("v_add_co_u32", d_out);
    run<54>("v_cndmask_b32 (vcc set)", d_out);
    run<55>("v_sub_u32 e64 noclamp", d_out);
    run<56>("v_subrev_u32", d_out);
    run<57>("v_and_b32 sgpr", d_out);
    run<60>("v_lshlrev_b32 imm", d_out);
    run<61>("v_bitop3 sgpr src", d_out);
    run<62>("v_lshrrev_b32 vgpr", d_out);
    run<63>("v_bfe_u32 imm", d_out);
    run<64>("v_bfe_i32 imm", d_out);
    run<65>("v_alignbit imm", d_out);
    run<66>("v_mul_i32_i24 vop2", d_out);
    run<67>("v_lshl_or imm", d_out);
    run<68>("v_add_u32 literal", d_out);
    run<69>("v_or_b32 sgpr", d_out);
    run<70>("v_sub_u32 inline const", d_out);
    run<71>("v_pk_ashrrev_i16 imm", d_out);
    run<72>("v_max_u32 imm", d_out);
    run<73>("v_and_b32 inline const", d_out);
    run<74>("v_mbcnt_lo", d_out);
    run<75>("v_sad_u32", d_out);
    run<76>("v_cvt_f32_ubyte0", d_out);
    run<77>("v_fma_f32", d_out);
    run<78>("v_add_f32", d_out);
    run<79>("v_cvt_i32_f32", d_out);
    run<80>("v_mul_f32", d_out);
    run<81>("v_pk_fma_f32", d_out);
    run<82>("v_pk_mul_lo_u16", d_out);
    run<0>("all add (fast)", d_out);
    run<1>("all mad (slow)", d_out);
    run<90>("add,mad alternating", d_out);
    run<91>("add,add,mad,mad", d_out);
    run<92>("3 add : 1 mad", d_out);
    run<93>("1 add : 3 mad", d_out);
    return 0;
}
