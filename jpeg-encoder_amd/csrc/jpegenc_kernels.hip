// jpegenc_kernels.hip -- gfx950 kernels of the jpeg_top data path.
//
// Pipeline per launch group (a sub-batch of images):
//   k_encode_chunks  RGB -> YCbCr -> 8x8 DCT -> quantise -> Huffman bits.  One wavefront per chunk of
//                    64 consecutive MCUs, one lane per MCU; everything between the pixel load and the
//                    lane's bit string stays in registers, the bit strings are concatenated in LDS and
//                    written once, as the chunk's bit stream, to its workspace slot.
//   k_scan           exclusive prefix of the per-chunk bit counts (and later of the 0xFF counts) per image.
//   k_stuff_count    counts the complete 0xFF bytes each chunk owns once its global bit phase is known.
//   k_stuff_write    shifts every chunk stream to its global position, inserts 0x00 after 0xFF, writes
//                    the 32-bit words and the end-of-file result.
//   k_stuff_image(_flags)  both stuffing steps for batches of many images, one workgroup per image: no prefix
//                    arrays, no scan launches, every chunk stream read once.
// Calls of a few frames run as one launch of k_encode_chunks whose waves place, stuff and write their own chunks.
//
// Reference arithmetic reproduced here (paths relative to the reference tree):
//   colour      rtl/1_rbg2ycrcb/rgb2ycrcb.sv:29-39, 77-91, 105-135
//   DCT         rtl/2_dct/cb_dct.sv:141-157, 198-231, 265-281, 308-375, 392-475, 544-610, 613-808, 826-865
//               rtl/2_dct/y_dct.sv:41-45, 122, 266, 549-551
//   quantiser   rtl/3_quantization/y_quantizer.sv:37-46, 58, 74, 106
//   transpose   rtl/d_q_h/yd_q_h.sv:120-135
//   entropy     rtl/4_huffman/cr_huff.sv:176-231 (end-of-block), 518-519, 558, 591-596, 629-656, 675-716,
//               1053-1304 (zig-zag), 1317-1486 (sizes), 1553-2082 (tables); y_huff.sv:184-193; cb_huff.sv:230-243
//   interleave  rtl/5_fifo/fifo_out.sv:195-197, 222-243, 257-266
//   stuffing    rtl/5_fifo/ff_checker.sv:96-179, 265-360
#include <initializer_list>
#include <type_traits>

#include "jpegenc_internal.h"
#include "jpegenc_tables.h"

namespace jpegenc {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// Orders one wave's LDS traffic for a single-wave workgroup: the compiler may not move LDS accesses
// across it, and the hardware serves a wave's LDS instructions in order.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive prefix sum over the wave with DPP moves: shifts by 1, 2, 4, 8 inside each row of 16 lanes
// (lanes without a source add 0), then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into rows
// 2 and 3.
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return (uint32_t)x;
}

__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t* total) {
    const uint32_t x = wave_incl_scan_dpp(v);
    *total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}

// Maximum over the wave, the same way (0 is the identity for the lanes without a source).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    int x = (int)v;
    x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false));
    x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false));
    x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false));
    x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false));
    x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false));
    x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false));
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// Value of lane (lane ^ M).  The library shuffles clamp their lane index with a compare and a v_cndmask
// each (about 20 cycles of the vector ALU per wave); ds_swizzle takes the pattern as an immediate.
template <int M>
__device__ __forceinline__ int xor_lane(int v, int lane) {
    if constexpr (M < 32)
        return __builtin_amdgcn_ds_swizzle(v, 0x1F | (M << 10));  // bit mode: and 0x1F, or 0, xor M
    else
        return __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, v);
}

// Three-input logic in one v_bitop3_b32 with register operands only.  On gfx950 this instruction, like
// add/sub, and/or/xor, right shifts and moves, issues at twice the rate of v_or3_b32, v_and_or_b32,
// left shifts, multiplies and anything with a scalar-register operand (measured with tools/micro/valu_cost.hip),
// and the compiler does not know the difference.
__device__ __forceinline__ uint32_t or3_fast(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE);
}
__device__ __forceinline__ uint32_t and3_fast(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x80);
}

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

// zig-zag position (row*8+col, row = vertical frequency) of the k-th coefficient
__device__ constexpr uint8_t kZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                        12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                        58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ------------------------------------------------------------------------------------------------
// 8-point integer DCT-II, y_k = sum_c x_c * T[k][c] with T = round(2^14 * DCT-II).  The rows of T are
// exactly symmetric (k even) or antisymmetric (k odd), so folding the inputs first changes no integer:
// every product and sum below is the same integer the RTL's multiply-accumulate produces
// (cb_dct.sv:433-439, 467-473 for pass 1, :198-231, :265-281 for pass 2).  `bias0` is added to y_0 and
// `bias` to y_1..7 so that the rounding "bits above n plus the bit below" becomes one arithmetic shift.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sext11(int v) { return (v << 21) >> 21; }

// 2-D transform of one component block held in registers.  In: x[r*8+c] samples 0..255.
// Out: x[a*8+b] = coefficient (vertical frequency a, horizontal frequency b), i.e. the Huffman coder's
// input order after the wrapper transpose (yd_q_h.sv:120-135).
// Luma keeps 13-bit intermediates (round at bit 12, final at bit 16: y_dct.sv:71, 122, 549-551); chroma
// keeps 11-bit intermediates (round at bit 14 twice: cb_dct.sv:311, 826).  The intermediate wraps of
// the RTL (25-bit pass-1 snapshots, 13/11-bit B) cannot trigger for 8-bit inputs: |A| < 2^24 and
// |B| <= 1448 resp. 363; the 25-/32-bit Z registers are taken modulo 2^32 here and only bits below
// bit 27 are used.
template <bool LUMA>
struct DctK {
    // compile-time shifts: luma rounds at bit 12 then 16, chroma at bit 14 twice
    static constexpr int sh1 = LUMA ? 12 : 14;
    static constexpr int sh2 = LUMA ? 16 : 14;
    static constexpr int b1 = 1 << (sh1 - 1);
    static constexpr int b2 = 1 << (sh2 - 1);
};

// The 11-bit output ports of pass 2 (cb_dct.sv:20-35) cannot wrap for 8-bit samples: DC lies in
// [-1024, 1016] and |AC| <= 1021 (|Z| <= 1448 * 46344 for luma, 363 * 46344 for chroma, before the final
// shift).

// ---- the transform on packed operands ------------------------------------------------------------
// Two 16-bit values per register: v_pk_add/sub_i16 fold two pairs at once and v_dot2_i32_i16 does two
// multiply-adds per instruction, exactly (16 x 16 -> 32-bit products, 32-bit accumulate; every operand
// fits 16 bits: |fold| <= 5792, |T| <= 8035): 18 instructions per 8-point transform plus the shifts.
__device__ __forceinline__ int pk_add(int a, int b) {
    int r;
    asm("v_pk_add_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int pk_sub(int a, int b) {
    int r;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a.lo + b.hi, a.hi + b.lo) and (a.lo - b.hi, a.hi - b.lo)
__device__ __forceinline__ int pk_add_x(int a, int b) {
    int r;
    asm("v_pk_add_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int pk_sub_x(int a, int b) {
    int r;
    asm("v_pk_sub_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// acc + a.lo * c.lo + a.hi * c.hi.  Written as inline asm, so the compiler adds no wait states behind it:
// v_dot2_i32_i16 needs none (a different instruction may read the result at once, measured with
// tools/micro/dot_hazard.hip), unlike v_dot4_i32_i8, whose result is only there three instructions later.
__device__ __forceinline__ int dot2(int a, int c, int acc) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(c), "v"(acc));
    return r;
}
// the same with the coefficient pair in a vector register (a value per lane)
__device__ __forceinline__ int dot2v(int a, int c, int acc) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(c), "v"(acc));
    return r;
}
__host__ __device__ constexpr int cpair(int lo, int hi) { return (int)(((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16)); }

// Inputs as pairs p01 = (x0,x1), p23 = (x2,x3), p76 = (x7,x6), p54 = (x5,x4); a[k] = S * sum + bias.  S = 4 (T
// times four still fits 16 bits: 4 * 8035 = 32140) moves a rounding position from bit 14 to bit 16, where the
// rounded value is the upper half of the register.
template <int S = 1>
__device__ __forceinline__ void dct8p_acc(int p01, int p23, int p76, int p54, int bias0, int bias, int (&a)[8]) {
    const int s01 = pk_add(p01, p76), s23 = pk_add(p23, p54);  // (s0,s1), (s2,s3)
    const int d01 = pk_sub(p01, p76), d23 = pk_sub(p23, p54);  // (d0,d1), (d2,d3)
    const int e = pk_add_x(s01, s23);                           // (s0+s3, s1+s2)
    const int f = pk_sub_x(s01, s23);                           // (s0-s3, s1-s2)
    a[0] = dot2(e, cpair(S * 5793, S * 5793), bias0);
    a[4] = dot2(e, cpair(S * 5793, S * -5793), bias);
    a[2] = dot2(f, cpair(S * 7568, S * 3135), bias);
    a[6] = dot2(f, cpair(S * 3135, S * -7568), bias);
    a[1] = dot2(d23, cpair(S * 4551, S * 1598), dot2(d01, cpair(S * 8035, S * 6811), bias));
    a[3] = dot2(d23, cpair(S * -8035, S * -4551), dot2(d01, cpair(S * 6811, S * -1598), bias));
    a[5] = dot2(d23, cpair(S * 1598, S * 6811), dot2(d01, cpair(S * 4551, S * -8035), bias));
    a[7] = dot2(d23, cpair(S * 6811, S * -8035), dot2(d01, cpair(S * 1598, S * -4551), bias));
}
// y[k] = (sum + bias) >> sh
__device__ __forceinline__ void dct8p(int p01, int p23, int p76, int p54, int bias0, int bias, int sh, int (&y)[8]) {
    dct8p_acc(p01, p23, p76, p54, bias0, bias, y);
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] >>= sh;
}

// (lo >> sh) in the low half, (hi >> sh) in the high half: the second shift writes its 16 result bits
// straight into the upper half of the first one's register (SDWA), so no separate packing instruction.
__device__ __forceinline__ int shift_pack(int lo, int hi, int sh) {
    int r = lo >> sh;
    asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(r)
        : "v"(sh), "v"(hi));
    return r;
}

// Pass 1 on two consecutive rows of samples (p0: row 2i, p1: row 2i+1, values 0..255, packed as dct8p_acc
// wants them) and packing of the results for pass 2: col[k] = (B[lo_row][k], B[hi_row][k]) with lo_row = 2i
// for the upper half of the block and 2i+1 for the lower half (pass 2 wants the pairs (B0,B1), (B2,B3),
// (B5,B4), (B7,B6)).
template <bool LUMA>
__device__ __forceinline__ void dct_row_pair(const int (&p0)[4], const int (&p1)[4], bool lower_half, int (&col)[8]) {
    int a0[8], a1[8];
    if constexpr (LUMA) {
        dct8p_acc(p0[0], p0[1], p0[2], p0[3], DctK<LUMA>::b1 - JPEGENC_DCT_LEVEL_SHIFT, DctK<LUMA>::b1, a0);
        dct8p_acc(p1[0], p1[1], p1[2], p1[3], DctK<LUMA>::b1 - JPEGENC_DCT_LEVEL_SHIFT, DctK<LUMA>::b1, a1);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            col[k] = lower_half ? shift_pack(a1[k], a0[k], DctK<LUMA>::sh1) : shift_pack(a0[k], a1[k], DctK<LUMA>::sh1);
    } else {
        // Chroma rounds at bit 14: with 4 T (and four times the level shift) the rounded value is the upper half of
        // the sum, (A + 2^13) >> 14 == (4 A + 2^15) >> 16 with |4 A| < 2^27, and one v_perm_b32 packs two of them
        // where two shifts did.
        dct8p_acc<4>(p0[0], p0[1], p0[2], p0[3], (1 << 15) - 4 * JPEGENC_DCT_LEVEL_SHIFT, 1 << 15, a0);
        dct8p_acc<4>(p1[0], p1[1], p1[2], p1[3], (1 << 15) - 4 * JPEGENC_DCT_LEVEL_SHIFT, 1 << 15, a1);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            col[k] = lower_half ? (int)__builtin_amdgcn_perm((uint32_t)a0[k], (uint32_t)a1[k], 0x07060302u)
                                : (int)__builtin_amdgcn_perm((uint32_t)a1[k], (uint32_t)a0[k], 0x07060302u);
    }
}

// The two passes with the rounding positions as values (one component per lane: luma and chroma lanes side by side).
__device__ __forceinline__ void dct_row_pair_v(const int (&p0)[4], const int (&p1)[4], bool lower_half, int b1, int sh1,
                                               int (&col)[8]) {
    int a0[8], a1[8];
    dct8p_acc(p0[0], p0[1], p0[2], p0[3], b1 - JPEGENC_DCT_LEVEL_SHIFT, b1, a0);
    dct8p_acc(p1[0], p1[1], p1[2], p1[3], b1 - JPEGENC_DCT_LEVEL_SHIFT, b1, a1);
#pragma unroll
    for (int k = 0; k < 8; ++k) col[k] = lower_half ? shift_pack(a1[k], a0[k], sh1) : shift_pack(a0[k], a1[k], sh1);
}
__device__ __forceinline__ void dct_cols_packed_v(const int (&pc)[4][8], int b2, int sh2, int (&x)[64]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int y[8];
        dct8p(pc[0][k], pc[1][k], pc[3][k], pc[2][k], b2, b2, sh2, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j * 8 + k] = y[j];
    }
}

// Pass 2: column k from its four row pairs -> x[j*8+k], j = vertical frequency.
template <bool LUMA>
__device__ __forceinline__ void dct_cols_packed(const int (&pc)[4][8], int (&x)[64]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int y[8];
        dct8p(pc[0][k], pc[1][k], pc[3][k], pc[2][k], DctK<LUMA>::b2, DctK<LUMA>::b2, DctK<LUMA>::sh2, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j * 8 + k] = y[j];
    }
}

// Pass 2 for the coder that reads pairs.  The sums are left with their rounding position at bit 16 (luma's own;
// chroma's bit 14 by transforming with 4 T: (Z + 2^13) >> 14 == (4 Z + 2^15) >> 16, |4 Z| < 2^27), so every
// coefficient is the upper half of its register and needs no shift.  One v_perm_b32 puts the coefficients of two
// vertical frequencies of a column into the halves of a register, and two packed instructions turn both into
// the coder's amplitude source t = v - (v < 0) (zero exactly when v is): tp[4 k + i] = (t of frequency (2 i, k),
// t of (2 i + 1, k)).  Pairs within a column keep only the column's eight sums alive.  dc = the DC coefficient.
__device__ __forceinline__ int pk_amp_src(int p) {
    int s, t;
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(s) : "v"(p));  // (both halves shift by the constant's low half)
    asm("v_pk_add_i16 %0, %1, %2" : "=v"(t) : "v"(p), "v"(s));
    return t;
}
template <bool LUMA>
__device__ __forceinline__ void dct_cols_pairs(const int (&pc)[4][8], int (&tp)[32], int& dc) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int y[8];
        dct8p_acc<LUMA ? 1 : 4>(pc[0][k], pc[1][k], pc[3][k], pc[2][k], 1 << 15, 1 << 15, y);
        if (k == 0) dc = y[0] >> 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            tp[4 * k + i] = pk_amp_src((int)__builtin_amdgcn_perm((uint32_t)y[2 * i + 1], (uint32_t)y[2 * i], 0x07060302u));
    }
}
// where the coefficient of zig-zag position z sits in tp
__host__ __device__ constexpr int pair_reg(int z) { return (kZZ[z] & 7) * 4 + (kZZ[z] >> 4); }
__host__ __device__ constexpr int pair_half(int z) { return (kZZ[z] >> 3) & 1; }
// zig-zag position of the coefficient at row * 8 + col
__host__ __device__ constexpr int inv_zz(int pos) {
    for (int z = 0; z < 64; ++z)
        if (kZZ[z] == pos) return z;
    return -1;
}
// The ranges of zig-zag positions that the end-of-block tests tell apart: below 16, 16..31, 32..47, 48..63.  True
// if exactly the listed registers of tp hold a position of range `lower` in the lower half and one of range
// `upper` in the upper half.
__host__ __device__ constexpr int zz_range(int z) { return z < 16 ? 0 : z < 32 ? 1 : z < 48 ? 2 : 3; }
__host__ __device__ constexpr bool pairs_of_kind(int lower, int upper, std::initializer_list<int> regs) {
    int found = 0;
    for (int z = 0; z < 64; z += 1) {
        if (pair_half(z) != 0 || zz_range(z) != lower) continue;
        int up = -1;  // the position in the upper half of the same register
        for (int y = 0; y < 64; ++y)
            if (pair_reg(y) == pair_reg(z) && pair_half(y) == 1) up = y;
        if (zz_range(up) != upper) continue;
        bool listed = false;
        for (int r : regs) listed = listed || r == pair_reg(z);
        if (!listed) return false;
        ++found;
    }
    return found == (int)regs.size();
}

// ------------------------------------------------------------------------------------------------
// Per-lane bit writer: appends (bits,len) MSB-first into the lane's private LDS region.
// `win` holds the last 64 bits of the string.  Every append stores the most recently completed 32-bit
// word again (the same value until the next word completes), so there is no "did a word complete"
// decision: the word ending at the last word boundary is win >> (T & 31), and it belongs at word index
// (T >> 5) - 1 of the string.  T is kept one word low instead (string word w is priv word w + 1 of the
// region), so the store goes to priv word T >> 5; before the first word completes it lands in the
// region's word 0, which is never read.
// ------------------------------------------------------------------------------------------------
struct BitWriter {
    uint64_t win;   // the last 64 bits appended (older bits shift out)
    uint32_t T;     // bit position: priv word index of the last completed word * 32 + pending bits
    uint32_t wlim;  // GUARD only: byte address that takes every store beyond it (one minimum, no select):
                    // the word behind the string's room, rewritten by the flush; the region's word 0 for a
                    // lane that only counts its bits
};

// Branch-free append of (bits,len), len <= 26 (one append completes at most one word).  GUARD = false
// is for the first tier, where every lane owns its region and a lane that overruns it merely forces the
// chunk onto the next tier (the overrun lands in the neighbour's region or in the padding behind the
// last one, and everything is recomputed); GUARD = true diverts such stores to the region's word 0.
template <bool GUARD>
__device__ __forceinline__ void bw_emit(BitWriter& s, uint32_t bits, uint32_t amp, uint32_t len, uint32_t* priv) {
    s.T += len;
    s.win <<= len;
    const uint32_t lo = or3_fast((uint32_t)s.win, bits, amp);
    s.win = (s.win & 0xFFFFFFFF00000000ull) | lo;
    const uint32_t word = __builtin_amdgcn_alignbit((uint32_t)(s.win >> 32), lo, s.T);
    uint32_t at = (s.T >> 3) & ~3u;
    if constexpr (GUARD) at = min(at, s.wlim);
    *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(priv) + at) = word;
}

// Huffman-codes one component block (coefficients in Huffman input order) into the lane's writer.
// AC algorithm, lane-serial over the zig-zag order (cr_huff.sv:675-685 run counter, :656 forced run 0
// at the last slot, :558 ZRL only when end_of_block is low, :518 the last slot always emits):
//   non-zero v : symbol (run<<4)|size(v), then the low size bits of v (v>=0) or v-1 (v<0)
//   zero v     : ZRL iff the 4-bit run counter already holds 15 and end_of_block(k) is low
//   k == 63    : zero -> EOB, non-zero -> ordinary symbol and no EOB
// size(v) = bit_length(|v|): the RTL takes the highest set bit of v[9:0] for v >= 0 and the highest
// clear bit of (v-1)[9:0] for v < 0 (cr_huff.sv:1411-1486); ~(v-1) == -v makes the two cases one
// expression, and the restriction to bits [9:0] (resp. [10:0] for DC) only matters for values that
// 8-bit pixels cannot produce (see build_htab).  The table lookup does the rest: the index carries
// run, size and the end_of_block flag, and empty entries make a zero coefficient emit nothing.
// t = v for v >= 0, v - 1 for v < 0: its low `size` bits are the amplitude, and the number of leading
// bits equal to its sign bit is 32 - size (v = -4: t = ...11011, 29 leading ones, size 3).
// Two instructions of the double-rate kind (shift right by a constant, add): cheaper on gfx950 than one
// SDWA add.
__device__ __forceinline__ int amp_src(int v) { return v + (v >> 31); }
// v_ffbh_i32: number of leading bits equal to the sign bit, -1 when all 32 are (t = 0; t = -1 cannot
// occur since v = 0 maps to t = 0).
__device__ __forceinline__ int lead_bits(int t) {
    int lead;
    asm("v_ffbh_i32 %0, %1" : "=v"(lead) : "v"(t));
    return lead;
}
__device__ __forceinline__ uint32_t mag_bits(int lead) { return 32u - min((uint32_t)lead, 32u); }
__device__ __forceinline__ uint32_t amp_bits(int t, uint32_t size) {
    return __builtin_amdgcn_ubfe((uint32_t)t, 0u, size);
}

// ---- several symbols per append --------------------------------------------------------------------
// An append costs six instructions and an LDS store whatever it carries, so consecutive symbols are first
// joined in a register, (earlier << length of later) | later, pairs into fours into eights, and go out in one
// append when no lane's joined string exceeds 32 bits (then at most one word completes, as bw_emit needs).
// Joining costs 2.5 instructions per symbol.  Whether a group fits is one compare and a wave-uniform branch;
// a group that does not fit in some lane goes out as its two halves, each tried the same way, down to
// single symbols.  Which positions are tried in groups of which size is fixed (see encode_block): the low
// frequencies of photographic content hold the long codes.
struct AcSym {
    uint32_t h, a, l;  // code << size, amplitude bits, code length + size (all 0: nothing to emit)
};
struct Joined {
    uint32_t m, l;  // bits and their number (m is only meaningful while l <= 32)
};
// x << n for n <= 32 where n == 32 only meets x == 0 (an empty left part): the hardware takes n modulo 32
__device__ __forceinline__ uint32_t shl_mod(uint32_t x, uint32_t n) { return x << (n & 31u); }
__device__ __forceinline__ Joined join_syms(const AcSym& x, const AcSym& y) {
    return {or3_fast(shl_mod(x.h | x.a, y.l), y.h, y.a), x.l + y.l};
}
__device__ __forceinline__ Joined join(const Joined& x, const Joined& y) { return {shl_mod(x.m, y.l) | y.m, x.l + y.l}; }
__device__ __forceinline__ bool fits_all(const Joined& j) { return !__any((int)(j.l > 32u)); }

template <bool GUARD>
__device__ __forceinline__ void emit_pair(BitWriter& bw, const AcSym& s0, const AcSym& s1, const Joined& p, uint32_t* priv) {
    if (fits_all(p)) {
        bw_emit<GUARD>(bw, p.m, 0u, p.l, priv);
    } else {
        bw_emit<GUARD>(bw, s0.h, s0.a, s0.l, priv);
        bw_emit<GUARD>(bw, s1.h, s1.a, s1.l, priv);
    }
}
template <bool GUARD>
__device__ __forceinline__ void emit_quad(BitWriter& bw, const AcSym* s, const Joined& p0, const Joined& p1, const Joined& q,
                                          uint32_t* priv) {
    if (fits_all(q)) {
        bw_emit<GUARD>(bw, q.m, 0u, q.l, priv);
    } else {
        emit_pair<GUARD>(bw, s[0], s[1], p0, priv);
        emit_pair<GUARD>(bw, s[2], s[3], p1, priv);
    }
}
// N = 2, 4 or 8 symbols, tried as one group first
template <bool GUARD, int N>
__device__ __forceinline__ void emit_group(BitWriter& bw, const AcSym (&s)[N], uint32_t* priv) {
    static_assert(N == 2 || N == 4 || N == 8, "pairs, fours, eights");
    Joined p[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) p[i] = join_syms(s[2 * i], s[2 * i + 1]);
    if constexpr (N == 2) {
        emit_pair<GUARD>(bw, s[0], s[1], p[0], priv);
    } else if constexpr (N == 4) {
        emit_quad<GUARD>(bw, s, p[0], p[1], join(p[0], p[1]), priv);
    } else {
        const Joined q0 = join(p[0], p[1]), q1 = join(p[2], p[3]);
        const Joined o = join(q0, q1);
        if (fits_all(o)) {
            bw_emit<GUARD>(bw, o.m, 0u, o.l, priv);
        } else {
            emit_quad<GUARD>(bw, s, p[0], p[1], q0, priv);
            emit_quad<GUARD>(bw, s + 4, p[2], p[3], q1, priv);
        }
    }
}

template <bool GUARD, bool SKIP, bool MERGE = false>
__device__ __forceinline__ void encode_block(const int (&blk)[64], int dc_pred, bool cb_eob_quirk,
                                             const uint32_t* htab, const uint8_t* hlen, uint32_t* priv,
                                             BitWriter& bw) {
    static_assert(!(MERGE && SKIP), "the position-skipping loop appends symbol by symbol");
    {   // DC difference, 12 bit (cr_huff.sv:1317): |d| <= 2047 for 11-bit inputs, no wrap possible
        const int t = amp_src(blk[0] - dc_pred);
        const uint32_t size = mag_bits(lead_bits(t));
        const uint32_t at = ((uint32_t)size << 5) | kHtabDcSize;
        bw_emit<GUARD>(bw, htab[at], amp_bits(t, size), hlen[at], priv);
    }
    // group-of-eight all-zero tests over zig-zag indices 16..63 (cr_huff.sv:190-204)
    uint32_t o2 = 0, o3 = 0, o4 = 0, o5 = 0, o6 = 0, o7 = 0;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {  // two coefficients per three-input OR
        o2 = or3_fast(o2, (uint32_t)blk[kZZ[16 + i]], (uint32_t)blk[kZZ[17 + i]]);
        o3 = or3_fast(o3, (uint32_t)blk[kZZ[24 + i]], (uint32_t)blk[kZZ[25 + i]]);
        o4 = or3_fast(o4, (uint32_t)blk[kZZ[32 + i]], (uint32_t)blk[kZZ[33 + i]]);
        o5 = or3_fast(o5, (uint32_t)blk[kZZ[40 + i]], (uint32_t)blk[kZZ[41 + i]]);
        o6 = or3_fast(o6, (uint32_t)blk[kZZ[48 + i]], (uint32_t)blk[kZZ[49 + i]]);
        o7 = or3_fast(o7, (uint32_t)blk[kZZ[56 + i]], (uint32_t)blk[kZZ[57 + i]]);
    }
    const uint32_t t67 = o6 | o7;
    const uint32_t t47 = or3_fast(o4, o5, t67);
    // end_of_block as a table-index bit (256 = set), per block_counter range (cr_huff.sv:213-231)
    const uint32_t eob_a = (or3_fast(o2, o3, t47) == 0) ? 16u : 0u;  // block_counter < 32
    const uint32_t eob_b = (t47 == 0) ? 16u : 0u;              // block_counter < 48
    // block_counter <= 64: groups 7,8 -- except in cb_huff, where `6'd64` is 0 and the branch that
    // sets end_of_block to 1 always wins (cb_huff.sv:238-243)
    const uint32_t eob_c = (cb_eob_quirk || t67 == 0) ? 16u : 0u;

    // a run of 15 zeros cannot complete before k = 16, so the flag only matters from there on
    auto eob_of = [&](int k) { return (k < 16) ? 0u : (k + 2 < 32) ? eob_a : (k + 2 < 48) ? eob_b : eob_c; };
    uint32_t r5 = 0;  // run << 5
    uint32_t run_mask = 0x1E0u;
    asm volatile("" : "+v"(run_mask));  // kept in a vector register: a scalar operand would halve the issue rate
    if constexpr (MERGE) {
        // table entry and amplitude of the coefficient at zig-zag position k; moves the run counter
        auto sym_at = [&](int k) {
            const int t = amp_src(blk[kZZ[k]]);
            const int lead = lead_bits(t);
            const uint32_t size = mag_bits(lead);
            const uint32_t at = or3_fast(r5, size, eob_of(k));
            r5 = and3_fast(r5 + 32u, (uint32_t)lead, run_mask);
            return AcSym{htab[at], amp_bits(t, size), hlen[at]};
        };
        // Group sizes by position, from the longest joined strings per wave of photographic content at the
        // all-ones tables (1 byte per pixel): pairs always fit, fours from position 5 on (3 % of the waves fail
        // at 5..8, none later), eights from 25 on (13 % fail at 25..32, 1 % at 33..40).
        auto group = [&](auto first, auto count) {
            constexpr int K0 = decltype(first)::value, N = decltype(count)::value;
            AcSym s[N];
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] = sym_at(K0 + i);
            emit_group<GUARD, N>(bw, s, priv);
        };
        using std::integral_constant;
#define JPEGENC_GROUP(K0, N) group(integral_constant<int, K0>{}, integral_constant<int, N>{})
        JPEGENC_GROUP(1, 2);
        JPEGENC_GROUP(3, 2);
        JPEGENC_GROUP(5, 4);
        JPEGENC_GROUP(9, 4);
        JPEGENC_GROUP(13, 4);
        JPEGENC_GROUP(17, 4);
        JPEGENC_GROUP(21, 4);
        JPEGENC_GROUP(25, 8);
        JPEGENC_GROUP(33, 8);
        JPEGENC_GROUP(41, 8);
        JPEGENC_GROUP(49, 8);
        JPEGENC_GROUP(57, 4);
        JPEGENC_GROUP(61, 2);
#undef JPEGENC_GROUP
    } else {
#pragma unroll
    for (int k = 1; k <= 62; ++k) {
        const int v = blk[kZZ[k]];
        if constexpr (SKIP) {
            // Quantised content: most of the time no lane of the wave has anything to emit at this
            // position (a zero emits only as the 16th of a run outside end_of_block) -- then only the
            // run counters move.  With all-ones tables the test would almost never succeed, so that
            // variant does not make it.
            const bool emits = (v != 0) || ((r5 | eob_of(k)) == 0x1E0u);
            if (!__any((int)emits)) {
                r5 = (r5 + 32u) & 0x1E0u;
                continue;
            }
        }
        const int t = amp_src(v);
        const int lead = lead_bits(t);
        const uint32_t size = mag_bits(lead);
        const uint32_t eob = eob_of(k);
        const uint32_t at = or3_fast(r5, size, eob);
        bw_emit<GUARD>(bw, htab[at], amp_bits(t, size), hlen[at], priv);
        // zero coefficient (lead == -1, all ones): run + 1 modulo 16; otherwise the run restarts (lead is
        // at most 31 then and has no bit in common with the run field)
        r5 = and3_fast(r5 + 32u, (uint32_t)lead, run_mask);
    }
    }
    {
        const int v = blk[63];
        const int t = amp_src(v);
        const int lead = lead_bits(t);
        const uint32_t size = mag_bits(lead);
        // lead is -1 for a zero: its sign as a mask picks the EOB entry
        const uint32_t zmask = (uint32_t)(lead >> 31);
        const uint32_t at = ((r5 | size) & ~zmask) | ((uint32_t)kHtabEob & zmask);
        bw_emit<GUARD>(bw, htab[at], amp_bits(t, size), hlen[at], priv);
    }
}

// ---- the coder on pairs of coefficients ------------------------------------------------------------------
// tp = the amplitude sources as 16-bit halves (dct_cols_pairs).  The leading-bit count reads its half through an
// SDWA operand select and the amplitude's bit field starts at bit 0 or 16: nothing is unpacked.  Same symbols, same
// order as encode_block.
template <int HALF>
__device__ __forceinline__ int lead_bits_half(int t2) {
    int lead;
    if constexpr (HALF == 0)
        asm("v_ffbh_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(lead) : "v"(t2));
    else
        asm("v_ffbh_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(lead) : "v"(t2));
    return lead;
}

// EXACT: the size rule on bits [9:0] as written (cr_huff.sv:1411-1486), for coefficients that 8-bit pixels cannot
// produce: an AC value of -1024 has no clear bit among (v - 1)[9:0], so its size is 0 although it is not zero, and
// the coder emits the table entry of (run, size 0) with no amplitude bits.  The leading-bit count gives such a value
// size 11, which no other AC value has: the table holds those entries at size 11 (build_htab), and only the
// amplitude has to go.  Used by the coefficient-fed variant; the variants fed from pixels never meet the value.
// TAB8: the table as eight-byte entries (code << size, length) at htab, one LDS read per lookup; else a word array at
// htab and a byte array at hlen.
template <bool GUARD, bool MERGE, bool EXACT = false, bool TAB8 = false>
__device__ __forceinline__ void encode_block_pairs(const int (&tp)[32], int dc, int dc_pred, bool cb_eob_quirk,
                                                   const uint32_t* htab, const uint8_t* hlen, uint32_t* priv,
                                                   BitWriter& bw) {
    struct Entry {
        uint32_t h, l;
    };
    auto entry = [&](uint32_t at) -> Entry {
        if constexpr (TAB8) {
            const uint2 e = reinterpret_cast<const uint2*>(htab)[at];
            return {e.x, e.y};
        } else {
            return {htab[at], hlen[at]};
        }
    };
    {   // DC difference, 12 bit (cr_huff.sv:1317)
        const int t = amp_src(dc - dc_pred);
        const uint32_t size = mag_bits(lead_bits(t));
        const Entry e = entry(((uint32_t)size << 5) | kHtabDcSize);
        bw_emit<GUARD>(bw, e.h, amp_bits(t, size), e.l, priv);
    }
    // The group-of-eight all-zero tests (cr_huff.sv:190-204) only ever ask about zig-zag positions 16..31, 32..47
    // and 48..63 together with everything behind them.  Of the registers of tp, some hold two positions of one of
    // these ranges, some a position of one range in the lower half and of the next in the upper half, some a
    // position below 16 in the lower half: ORed by kind, then the halves that count are picked.
    const uint32_t ga_both = or3_fast((uint32_t)tp[6], (uint32_t)tp[13], (uint32_t)tp[17]) | (uint32_t)tp[24];
    const uint32_t ga_hi = or3_fast((uint32_t)tp[2], (uint32_t)tp[9], (uint32_t)tp[16]) | (uint32_t)tp[20];
    const uint32_t gab = or3_fast((uint32_t)tp[3], (uint32_t)tp[10], (uint32_t)tp[21]) | (uint32_t)tp[28];   // lower: 16..31, upper: 32..47
    const uint32_t gb_both = or3_fast((uint32_t)tp[7], (uint32_t)tp[14], (uint32_t)tp[18]) | (uint32_t)tp[25];
    const uint32_t gbc = or3_fast((uint32_t)tp[11], (uint32_t)tp[15], (uint32_t)tp[22]) | (uint32_t)tp[29];  // lower: 32..47, upper: 48..63
    const uint32_t gc_both = or3_fast(or3_fast((uint32_t)tp[19], (uint32_t)tp[23], (uint32_t)tp[26]), (uint32_t)tp[27], (uint32_t)tp[30]) | (uint32_t)tp[31];
    static_assert(pairs_of_kind(1, 1, {6, 13, 17, 24}) && pairs_of_kind(0, 1, {2, 9, 16, 20}) && pairs_of_kind(1, 2, {3, 10, 21, 28}) &&
                      pairs_of_kind(2, 2, {7, 14, 18, 25}) && pairs_of_kind(2, 3, {11, 15, 22, 29}) &&
                      pairs_of_kind(3, 3, {19, 23, 26, 27, 30, 31}) && pairs_of_kind(0, 0, {0, 1, 4, 5, 8, 12}),
                  "every register of tp is in the list of its kind");
    const uint32_t t67 = gc_both | (gbc & 0xFFFF0000u);                                  // positions 48..63
    const uint32_t t47 = or3_fast(t67, gb_both, (gbc & 0xFFFFu) | (gab & 0xFFFF0000u));  // 32..63
    const uint32_t t27 = or3_fast(t47, ga_both, (gab & 0xFFFFu) | (ga_hi & 0xFFFF0000u));  // 16..63
    const uint32_t eob_a = (t27 == 0) ? 16u : 0u;                    // block_counter < 32 (cr_huff.sv:213-231)
    const uint32_t eob_b = (t47 == 0) ? 16u : 0u;                    // block_counter < 48
    const uint32_t eob_c = (cb_eob_quirk || t67 == 0) ? 16u : 0u;    // block_counter <= 64; cb_huff.sv:238-243
    auto eob_of = [&](int k) { return (k < 16) ? 0u : (k + 2 < 32) ? eob_a : (k + 2 < 48) ? eob_b : eob_c; };
    uint32_t r5 = 0;  // run << 5
    uint32_t run_mask = 0x1E0u;
    asm volatile("" : "+v"(run_mask));  // (a vector register: see encode_block)
    // lead, size and amplitude of the coefficient at zig-zag position k
    auto coef_at = [&](auto kc, int& lead, uint32_t& size, uint32_t& amp) {
        constexpr int k = decltype(kc)::value;
        const int t2 = tp[pair_reg(k)];
        lead = lead_bits_half<pair_half(k)>(t2);
        size = mag_bits(lead);
        amp = __builtin_amdgcn_ubfe((uint32_t)t2, pair_half(k) ? 16u : 0u, size);
        if constexpr (EXACT) amp = size > 10u ? 0u : amp;
    };
    auto sym_at = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        int lead;
        uint32_t size, amp;
        coef_at(kc, lead, size, amp);
        const uint32_t at = or3_fast(r5, size, eob_of(k));
        r5 = and3_fast(r5 + 32u, (uint32_t)lead, run_mask);
        const Entry e = entry(at);
        return AcSym{e.h, amp, e.l};
    };
    using std::integral_constant;
    if constexpr (MERGE) {
        auto group = [&](auto first, auto count) {
            constexpr int K0 = decltype(first)::value, N = decltype(count)::value;
            AcSym s[N];
            s[0] = sym_at(integral_constant<int, K0>{});
            s[1] = sym_at(integral_constant<int, K0 + 1>{});
            if constexpr (N >= 4) {
                s[2] = sym_at(integral_constant<int, K0 + 2>{});
                s[3] = sym_at(integral_constant<int, K0 + 3>{});
            }
            if constexpr (N >= 8) {
                s[4] = sym_at(integral_constant<int, K0 + 4>{});
                s[5] = sym_at(integral_constant<int, K0 + 5>{});
                s[6] = sym_at(integral_constant<int, K0 + 6>{});
                s[7] = sym_at(integral_constant<int, K0 + 7>{});
            }
            emit_group<GUARD, N>(bw, s, priv);
        };
#define JPEGENC_GROUP(K0, N) group(integral_constant<int, K0>{}, integral_constant<int, N>{})
        JPEGENC_GROUP(1, 2);  // (the same groups as in encode_block)
        JPEGENC_GROUP(3, 2);
        JPEGENC_GROUP(5, 4);
        JPEGENC_GROUP(9, 4);
        JPEGENC_GROUP(13, 4);
        JPEGENC_GROUP(17, 4);
        JPEGENC_GROUP(21, 4);
        JPEGENC_GROUP(25, 8);
        JPEGENC_GROUP(33, 8);
        JPEGENC_GROUP(41, 8);
        JPEGENC_GROUP(49, 8);
        JPEGENC_GROUP(57, 4);
        JPEGENC_GROUP(61, 2);
#undef JPEGENC_GROUP
    } else {
        auto single = [&](auto kc) {
            const AcSym s = sym_at(kc);
            bw_emit<GUARD>(bw, s.h, s.a, s.l, priv);
        };
        auto eight = [&](auto first) {
            constexpr int K0 = decltype(first)::value;
            single(integral_constant<int, K0>{});
            single(integral_constant<int, K0 + 1>{});
            single(integral_constant<int, K0 + 2>{});
            single(integral_constant<int, K0 + 3>{});
            single(integral_constant<int, K0 + 4>{});
            single(integral_constant<int, K0 + 5>{});
            if constexpr (K0 + 6 <= 62) single(integral_constant<int, K0 + 6>{});
            if constexpr (K0 + 7 <= 62) single(integral_constant<int, K0 + 7>{});
        };
        eight(integral_constant<int, 1>{});
        eight(integral_constant<int, 9>{});
        eight(integral_constant<int, 17>{});
        eight(integral_constant<int, 25>{});
        eight(integral_constant<int, 33>{});
        eight(integral_constant<int, 41>{});
        eight(integral_constant<int, 49>{});
        eight(integral_constant<int, 57>{});
    }
    {   // the last slot always emits: EOB for a zero, else the ordinary symbol (cr_huff.sv:518, :656)
        int lead;
        uint32_t size, amp;
        coef_at(integral_constant<int, 63>{}, lead, size, amp);
        const uint32_t zmask = (uint32_t)(lead >> 31);
        const Entry e = entry(((r5 | size) & ~zmask) | ((uint32_t)kHtabEob & zmask));
        bw_emit<GUARD>(bw, e.h, amp, e.l, priv);
    }
}

// ------------------------------------------------------------------------------------------------
// pixel loads: 8 rows x 24 bytes of one MCU into 48 dwords (row r -> raw[6r..6r+5])
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int ALIGN>
__device__ __forceinline__ void load_mcu(const uint8_t* src, int64_t stride, uint32_t (&raw)[48]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint8_t* row = src + (int64_t)r * stride;
        if constexpr (ALIGN == 8) {
            const uint2* q = reinterpret_cast<const uint2*>(row);
            const uint2 a = q[0], b = q[1], c = q[2];
            raw[r * 6 + 0] = a.x; raw[r * 6 + 1] = a.y; raw[r * 6 + 2] = b.x;
            raw[r * 6 + 3] = b.y; raw[r * 6 + 4] = c.x; raw[r * 6 + 5] = c.y;
        } else if constexpr (ALIGN == 4) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(row);
#pragma unroll
            for (int i = 0; i < 6; ++i) raw[r * 6 + i] = q[i];
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                raw[r * 6 + i] = (uint32_t)row[4 * i] | ((uint32_t)row[4 * i + 1] << 8) |
                                 ((uint32_t)row[4 * i + 2] << 16) | ((uint32_t)row[4 * i + 3] << 24);
        }
    }
}

// The same through buffer loads (8-byte-aligned rows, all offsets of an image below 2^31): one lane offset for
// the whole MCU, the row's offset r * stride in a scalar register, so no 64-bit address per row.  LAST marks an
// MCU's last fetch: non-temporal, its lines may leave the L2 ahead of lines that other waves still need.  (A
// non-temporal global load would be folded into the identical plain load of the other components.)
template <bool LAST>
__device__ __forceinline__ void load_mcu_buf(__amdgpu_buffer_rsrc_t img, uint32_t lane_off, int32_t stride,
                                             uint32_t (&raw)[48]) {
    constexpr int aux = LAST ? 2 : 0;  // cache policy bit 1: nt
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(img, (int)lane_off, r * stride, aux);
        const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(img, (int)lane_off + 16, r * stride, aux);
        raw[r * 6 + 0] = a.x; raw[r * 6 + 1] = a.y; raw[r * 6 + 2] = a.z;
        raw[r * 6 + 3] = a.w; raw[r * 6 + 4] = b.x; raw[r * 6 + 5] = b.y;
    }
}

// byte j (0..23) of row r
__device__ __forceinline__ uint32_t raw_byte(const uint32_t (&raw)[48], int r, int j) {
    return (raw[r * 6 + (j >> 2)] >> (8 * (j & 3))) & 0xFFu;
}

// One colour component of one pixel (rgb2ycrcb.sv:77-91 sums, :105-135 rounding).  k0 carries the
// +2^21 offset of Cb/Cr and the rounding bit 2^13.  The 22-bit sums are never negative, so the
// arithmetic shift is the RTL's bit slice.  The rounded value is 0..256; 256 is reachable only by Cb and
// Cr (pure blue / pure red).  MODE 0: luma, no fix-up needed; MODE 1: chroma as written, the 8-bit
// rounder wraps 256 to 0 and the clamp is dead code; MODE 2: chroma with the clamp working.
__device__ __forceinline__ int colour_sum(uint32_t R, uint32_t G, uint32_t B, int k0, int cr, int cg, int cb) {
    return k0 + __mul24(cr, (int)R) + __mul24(cg, (int)G) + __mul24(cb, (int)B);
}
template <int MODE>
__device__ __forceinline__ int colour(uint32_t R, uint32_t G, uint32_t B, int k0, int cr, int cg, int cb) {
    const int v = colour_sum(R, G, B, k0, cr, cg, cb) >> 14;
    if constexpr (MODE == 0) return v;
    if constexpr (MODE == 1) return v & 0xFF;
    return min(v, 255);
}
// Two samples from their sums, packed (lo, hi) for the transform.  The second sample's last instruction
// writes into the upper half of the first one's register (SDWA): byte 2 of a register whose upper half is
// zero for MODE 0 (luma never reaches 256) and MODE 1 (the wrap is the byte select), the upper word for
// the clamp of MODE 2.
template <int MODE>
__device__ __forceinline__ int colour_pair(int s_lo, int s_hi) {
    int r = s_lo >> 14;
    if constexpr (MODE == 1) r &= 0xFF;
    if constexpr (MODE == 2) r = min(r, 255);
    if constexpr (MODE == 2) {
        const int t = s_hi >> 14;
        asm("v_min_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
            : "+v"(r)
            : "v"(255), "v"(t));
    } else {
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
            : "+v"(r)
            : "v"(14), "v"(s_hi));
    }
    return r;
}

// Measured switches (encode kernel per 768 frames, one box): symbols joined in the exact-fit passes too -- 6.16
// against 6.23 ms with 6 % of the chunks redone, 8.54 against 8.85 ms with 34 %, nothing where none is; the blue
// term's multiply picking its byte itself -- 3.249 against 3.270 ms.
#ifndef JPEGENC_FIT_MERGE
#define JPEGENC_FIT_MERGE 1
#endif
#ifndef JPEGENC_BLUE_SDWA
#define JPEGENC_BLUE_SDWA 1
#endif
struct CompCoef {
    int k0, cr, cg, cb;
};
__device__ __forceinline__ CompCoef comp_coef(int comp) {
    CompCoef c;
    c.k0 = (comp == 0) ? 8192 : (JPEGENC_C_OFFSET + 8192);
    c.cr = (comp == 0) ? JPEGENC_Y_R : (comp == 1) ? -JPEGENC_CB_R : JPEGENC_CR_R;
    c.cg = (comp == 0) ? JPEGENC_Y_G : (comp == 1) ? -JPEGENC_CB_G : -JPEGENC_CR_G;
    c.cb = (comp == 0) ? JPEGENC_Y_B : (comp == 1) ? JPEGENC_CB_B : -JPEGENC_CR_B;
    return c;
}

// The 8 pixels of row r of the MCU as the four pairs (x0,x1), (x2,x3), (x7,x6), (x5,x4) of one component.
template <int MODE>
__device__ __forceinline__ void colour_row(const uint32_t (&raw)[48], int r, const CompCoef& cc, int (&p)[4]) {
    // R and G of a pixel are gathered into the two halves of a register (v_perm over the one or two dwords
    // they sit in) and multiplied and summed with k0 by one v_dot2_i32_i16; B comes in by a byte-select multiply.
    const int crg = cpair(cc.cr, cc.cg);
    int s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = 3 * q, d = j >> 2, b = j & 3;
        const uint32_t lo = raw[r * 6 + d], hi = raw[r * 6 + (b == 3 ? d + 1 : d)];
        const uint32_t sel = 0x0C000C00u | (uint32_t)b | ((uint32_t)(b == 3 ? 4 : b + 1) << 16);
        const int rg = (int)__builtin_amdgcn_perm(hi, lo, sel);
#if JPEGENC_BLUE_SDWA
        // B sits in byte (j + 2) & 3 of its dword: in the middle bytes the multiply picks it itself (SDWA operand
        // select) and a plain add follows, instead of a bit-field extract and a multiply-add
        const int jb = j + 2;
        if ((jb & 3) == 1 || (jb & 3) == 2) {
            int prod;
            if ((jb & 3) == 1)
                asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
                    : "=v"(prod) : "v"(cc.cb), "v"(raw[r * 6 + (jb >> 2)]));
            else
                asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2"
                    : "=v"(prod) : "v"(cc.cb), "v"(raw[r * 6 + (jb >> 2)]));
            s[q] = dot2(rg, crg, cc.k0) + prod;
        } else {
            s[q] = dot2(rg, crg, cc.k0) + __mul24(cc.cb, (int)raw_byte(raw, r, jb));
        }
#else
        s[q] = dot2(rg, crg, cc.k0) + __mul24(cc.cb, (int)raw_byte(raw, r, j + 2));
#endif
    }
    p[0] = colour_pair<MODE>(s[0], s[1]);
    p[1] = colour_pair<MODE>(s[2], s[3]);
    p[2] = colour_pair<MODE>(s[7], s[6]);
    p[3] = colour_pair<MODE>(s[5], s[4]);
}

// The same with the component's coefficients as values per lane (crg = the R and G coefficients as a 16-bit pair).
template <int MODE>
__device__ __forceinline__ void colour_row_v(const uint32_t (&raw)[48], int r, int k0, int crg, int cb, int (&p)[4]) {
    int s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = 3 * q, d = j >> 2, b = j & 3;
        const uint32_t lo = raw[r * 6 + d], hi = raw[r * 6 + (b == 3 ? d + 1 : d)];
        const uint32_t sel = 0x0C000C00u | (uint32_t)b | ((uint32_t)(b == 3 ? 4 : b + 1) << 16);
        const int rg = (int)__builtin_amdgcn_perm(hi, lo, sel);
        s[q] = dot2v(rg, crg, k0) + __mul24(cb, (int)raw_byte(raw, r, j + 2));
    }
    p[0] = colour_pair<MODE>(s[0], s[1]);
    p[1] = colour_pair<MODE>(s[2], s[3]);
    p[2] = colour_pair<MODE>(s[7], s[6]);
    p[3] = colour_pair<MODE>(s[5], s[4]);
}

// quantiser (y_quantizer.sv:58, 74, 106): 11-bit result of (F * (4096/Q) + 2048) >> 12; the 23-bit
// product register only drops bits that the 11-bit result does not see.
__device__ __forceinline__ int quant(int f, int qm) { return sext11((__mul24(f, qm) + 2048) >> 12); }

// DC coefficient of one component of the MCU *before* this chunk, computed by the whole wave (lane =
// pixel): needed as the DC predictor of lane 0.  Same arithmetic as dct8x8 restricted to k = j = 0.
__device__ __forceinline__ int halo_dc(int sample, bool luma, int qm0, bool do_quant, int lane) {
    int s = sample;
    s += xor_lane<1>(s, lane);
    s += xor_lane<2>(s, lane);
    s += xor_lane<4>(s, lane);  // row sum, identical in the 8 lanes of a row
    const int sh1 = luma ? 12 : 14, sh2 = luma ? 16 : 14;
    int b = (__mul24(5793, s) - JPEGENC_DCT_LEVEL_SHIFT + (1 << (sh1 - 1))) >> sh1;
    b += xor_lane<8>(b, lane);
    b += xor_lane<16>(b, lane);
    b += xor_lane<32>(b, lane);  // sum over the 8 rows
    int f = sext11((__mul24(5793, b) + (1 << (sh2 - 1))) >> sh2);
    if (do_quant) f = quant(f, qm0);
    return f;
}

// ------------------------------------------------------------------------------------------------
// Stuffing helpers, shared by the stuffing kernels and by the single-launch form of the encode kernel.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kCountMask = 0xFFFFFFu;  // counts live in the low 24 bits of their words
[[maybe_unused]] constexpr int kStageBytes = 2048 + 64;  // one step: <= 1024 bytes, doubled by stuffing, + residual

[[maybe_unused]] constexpr int kStuffWaves = 4;  // independent waves per workgroup, one chunk each
constexpr int kStuffDepth = 5;  // steps whose loads are in flight together

// bit 7 of every byte of the result is set iff that byte of v is 0xFF: the low seven bits plus one carry
// into bit 7 exactly when they are all ones, and never beyond the byte
__device__ __forceinline__ uint32_t ff_flags(uint32_t v) {
    const uint32_t t = (v & 0x7F7F7F7Fu) + 0x01010101u;
    return t & v & 0x80808080u;
}

// Requests the first SPEC 1 KB steps of a chunk stream: 16 bytes per lane and step plus the word behind them.
template <int SPEC>
__device__ __forceinline__ void request_chunk(const uint32_t* S, int lane, uint4 (&w)[kStuffDepth], uint32_t (&w4)[kStuffDepth]) {
#pragma unroll
    for (int u = 0; u < kStuffDepth; ++u) {
        if (u < SPEC) {
            const uint32_t* q = S + 256 * u + 4 * lane;
            w[u] = *reinterpret_cast<const uint4*>(q);
            w4[u] = q[4];
        }
    }
}

// The 0xFF bytes owned by the chunk whose stream S (n bits) starts at bit o of the image's stream; pred8 = the
// last 8 bits before it.  w, w4: the first SPEC steps as requested by request_chunk (left untouched when the
// chunk has no more than kStuffDepth steps).  The same value in every lane.
template <int SPEC>
__device__ __forceinline__ uint32_t count_owned_ff(const uint32_t* S, uint4 (&w)[kStuffDepth], uint32_t (&w4)[kStuffDepth],
                                                   uint32_t n, uint64_t o, uint32_t pred8, int lane) {
    const uint32_t sbits = (uint32_t)(o & 7u);  // bits of the first owned byte that lie in the predecessor
    const uint32_t A = (8u - sbits) & 7u;
    const uint32_t nbytes = (uint32_t)(((o + n) >> 3) - (o >> 3));
    const uint32_t nm = nbytes - (sbits ? 1u : 0u);  // owned bytes that lie entirely in this chunk
    uint32_t total = 0;
    if (sbits != 0 && lane == 0) {
        const uint32_t pw = pred8;
        const uint32_t sb = ((pw << A) | (readlane_u32(w[0].x, 0) >> (32u - A))) & 0xFFu;
        total = (sb == 0xFFu) ? 1u : 0u;
    }
    for (uint32_t pass = 0; pass < nm; pass += 1024 * kStuffDepth) {
#pragma unroll
        for (int u = 0; u < kStuffDepth; ++u) {  // what was not requested ahead
            const uint32_t* q = S + ((pass >> 2) + 256 * u) + 4 * lane;
            if ((pass != 0 || u >= SPEC) && pass + 1024u * u < nm) {  // uniform
                w[u] = *reinterpret_cast<const uint4*>(q);
                w4[u] = q[4];
            }
        }
#pragma unroll
        for (int u = 0; u < kStuffDepth; ++u) {
            const uint32_t base = pass + 1024u * u;
            if (base < nm) {  // uniform
                uint32_t v0 = w[u].x, v1 = w[u].y, v2 = w[u].z, v3 = w[u].w;
                if (A != 0) {  // uniform
                    v0 = __builtin_amdgcn_alignbit(w[u].x, w[u].y, 32u - A);
                    v1 = __builtin_amdgcn_alignbit(w[u].y, w[u].z, 32u - A);
                    v2 = __builtin_amdgcn_alignbit(w[u].z, w[u].w, 32u - A);
                    v3 = __builtin_amdgcn_alignbit(w[u].w, w4[u], 32u - A);
                }
                uint32_t m0 = ff_flags(v0), m1 = ff_flags(v1), m2 = ff_flags(v2), m3 = ff_flags(v3);
                if (base + 1024 > nm) {  // uniform: the last step, some bytes lie behind the chunk's last whole byte
                    const uint32_t j = base + 16u * (uint32_t)lane;
                    const int nv = (j < nm) ? (int)min(16u, nm - j) : 0;
                    auto keep = [&](int nb) { return nb >= 4 ? 0xFFFFFFFFu : nb <= 0 ? 0u : ~(0xFFFFFFFFu >> (8 * nb)); };
                    m0 &= keep(nv); m1 &= keep(nv - 4); m2 &= keep(nv - 8); m3 &= keep(nv - 12);
                }
                total += __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3);
            }
        }
    }
    return readlane_u32(wave_incl_scan_dpp(total), 63);  // the wave's sum
}

// Writes the stuffed bytes of one chunk (stream S of n bits at bit o of the image's stream, ff_before 0xFF bytes
// owned by the chunks before it, pred8 = the last 8 bits before it) and, for the image's last chunk at end of
// file, the result record (prev_known: the previous chunk's bit count and last 32 bits come as arguments, else
// they are read from its slot when the tail word needs them).  w, w4: the first SPEC steps as requested by request_chunk when have_first (uniform),
// else they are requested here.  `stage`: this wave's kStageBytes of LDS.
template <int SPEC>
__device__ __forceinline__ void write_chunk(const StuffParams& p, const uint32_t* S, uint4 (&w)[kStuffDepth],
                                            uint32_t (&w4)[kStuffDepth], bool have_first, uint8_t* const stage, uint32_t img,
                                            uint32_t c, uint32_t gid, uint32_t n, uint64_t o, uint64_t ff_before,
                                            uint32_t pred8, int lane, bool prev_known = false, uint32_t prev_bits = 0,
                                            uint32_t prev_tail32 = 0) {
    if (!have_first) request_chunk<SPEC>(S, lane, w, w4);  // (uniform)
    const uint64_t q_first = o >> 3;
    const uint32_t sbits = (uint32_t)(o & 7u);
    const uint32_t A = (8u - sbits) & 7u;
    const uint32_t nbytes = (uint32_t)(((o + n) >> 3) - q_first);
    const uint32_t nm = nbytes - (sbits ? 1u : 0u);

    uint32_t* const out_words = p.words_out + (uint64_t)img * p.out_stride;
    uint8_t* const out_bytes = reinterpret_cast<uint8_t*>(out_words);
    const uint64_t cap_bytes = p.cap_words * 4;
    const uint64_t pos0 = q_first + ff_before + p.ff_base;  // stuffed byte offset of the first owned byte
    uint64_t wbase = (pos0 >> 2) - p.out_word_base;             // index in out_words of stage[0]
    uint32_t head = (uint32_t)(pos0 & 3u);          // leading bytes of the stage's first word that belong to the previous chunk
    uint32_t fill = head;                           // bytes in the stage (uniform)

    if (sbits != 0) {  // the byte that begins in the predecessor: its last 8 bits travel with its bit count
        const uint32_t pw = pred8;
        const uint32_t sb = ((pw << A) | (readlane_u32(w[0].x, 0) >> (32u - A))) & 0xFFu;
        if (lane == 0) {
            stage[fill] = (uint8_t)sb;
            stage[fill + 1] = 0;
        }
        fill += (sb == 0xFFu) ? 2u : 1u;
    }

    auto flush = [&](bool final_pad, bool quads_only = false) {
        // stores the complete words of the stage, four per lane and turn; keeps the residual bytes for
        // the next step.  quads_only (between the steps of a chunk): only whole groups of four words, so that every
        // lane takes the 16-byte store and up to 15 bytes stay behind, not 3
        const uint32_t nwords = final_pad ? ((fill + 3u) >> 2) : quads_only ? ((fill >> 4) << 2) : (fill >> 2);
        for (uint32_t i = 4u * lane; i < nwords; i += 256) {
            const uint4 raw = *reinterpret_cast<const uint4*>(stage + 4 * i);
            const uint64_t gw = wbase + i;
            const uint32_t t[4] = {__builtin_bswap32(raw.x), __builtin_bswap32(raw.y), __builtin_bswap32(raw.z),
                                   __builtin_bswap32(raw.w)};
            if (i + 4 <= nwords && gw + 4 <= p.cap_words && !(i == 0 && head)) {
                __builtin_memcpy(out_words + gw, t, 16);  // word-aligned 16-byte store
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (i + k >= nwords) break;
                    if (i + k == 0 && head) {  // word shared with the previous chunk: only our bytes
                        for (uint32_t b = head; b < 4; ++b) {
                            const uint64_t ga = gw * 4 + b;
                            if (ga < cap_bytes) out_bytes[ga ^ 3] = stage[b];
                        }
                    } else if (gw + k < p.cap_words) {
                        out_words[gw + k] = t[k];
                    }
                }
            }
        }
        wave_sync();
        if (nwords) {
            const uint32_t rest = final_pad ? 0u : fill - 4u * nwords;  // at most 3 bytes, 15 with quads_only (nwords >= 4 then)
            if ((uint32_t)lane < rest) stage[lane] = stage[nwords * 4 + lane];  // reads and writes are disjoint words
            wbase += nwords;
            fill = rest;
            head = 0;
        }
        wave_sync();
    };

    // One word holding at least one 0xFF (flags m), all four bytes valid, to stage offset `at`.  Every byte
    // goes out as the pair (byte, 0x00): the zero is the stuffing when the byte is 0xFF and is
    // overwritten by the next byte's pair otherwise.  The last byte of a lane's last word must not
    // write beyond the lane's bytes: it is stored alone unless it is 0xFF.
    auto ff_word = [&](uint32_t at, uint32_t v, uint32_t m, bool lane_end) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint16_t pair = (uint16_t)((v >> (24 - 8 * b)) & 0xFFu);
            const uint32_t f = (m >> (31 - 8 * b)) & 1u;
            if (b < 3 || !lane_end || f)
                __builtin_memcpy(&stage[at], &pair, 2);
            else
                stage[at] = (uint8_t)pair;
            at += 1u + f;
        }
    };

    for (uint32_t pass = 0; pass < nm; pass += 1024 * kStuffDepth) {
#pragma unroll
        for (int u = 0; u < kStuffDepth; ++u) {  // what was not requested ahead
            const uint32_t* q = S + ((pass >> 2) + 256 * u) + 4 * lane;
            if ((pass != 0 || u >= SPEC) && pass + 1024u * u < nm) {  // uniform
                w[u] = *reinterpret_cast<const uint4*>(q);
                w4[u] = q[4];
            }
        }
#pragma unroll
        for (int u = 0; u < kStuffDepth; ++u) {
            const uint32_t base = pass + 1024u * u;
            if (base >= nm) break;  // uniform
            const uint32_t j = base + 16u * (uint32_t)lane;
            const int nvalid = (j < nm) ? (int)min(16u, nm - j) : 0;
            uint32_t v[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
            if (A != 0) {  // uniform
                v[0] = __builtin_amdgcn_alignbit(w[u].x, w[u].y, 32u - A);
                v[1] = __builtin_amdgcn_alignbit(w[u].y, w[u].z, 32u - A);
                v[2] = __builtin_amdgcn_alignbit(w[u].z, w[u].w, 32u - A);
                v[3] = __builtin_amdgcn_alignbit(w[u].w, w4[u], 32u - A);
            }
            uint32_t m[4] = {ff_flags(v[0]), ff_flags(v[1]), ff_flags(v[2]), ff_flags(v[3])};
            const bool partial = base + 1024 > nm;  // uniform: the chunk's last step, some lanes hold fewer than 16 bytes
            if (partial) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int nb = nvalid - 4 * i;  // valid bytes of word i
                    m[i] &= nb >= 4 ? 0xFFFFFFFFu : nb <= 0 ? 0u : ~(0xFFFFFFFFu >> (8 * nb));
                }
            }
            const uint32_t cnt = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
            const uint32_t incl = wave_incl_scan_dpp(cnt);
            const uint32_t it_ff = readlane_u32(incl, 63);
            uint32_t a = fill + 16u * (uint32_t)lane + (incl - cnt);  // stage offset of this lane's first byte
            if (cnt == 0) {
                // (A lane with fewer than 16 bytes left -- the chunk's last, and the lanes behind it -- stores 16
                // all the same: what it adds lies behind the chunk's last byte, inside the stage, and is never used.)
                uint32_t t[4] = {__builtin_bswap32(v[0]), __builtin_bswap32(v[1]), __builtin_bswap32(v[2]),
                                 __builtin_bswap32(v[3])};
                __builtin_memcpy(&stage[a], t, 16);  // byte-aligned 16-byte store
            } else if (nvalid == 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (m[i] == 0) {
                        const uint32_t t = __builtin_bswap32(v[i]);
                        __builtin_memcpy(&stage[a], &t, 4);
                    } else {
                        ff_word(a, v[i], m[i], i == 3);
                    }
                    a += 4u + __popc(m[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int nb = nvalid - 4 * i;
                    for (int b = 0; b < 4 && b < nb; ++b) {
                        stage[a++] = (uint8_t)(v[i] >> (24 - 8 * b));
                        if (m[i] & (0x80000000u >> (8 * b))) stage[a++] = 0;
                    }
                }
            }
            fill += min(1024u, nm - base) + it_ff;
            wave_sync();
            // (the chunk's last step leaves everything to the flush behind the loop: one pass over the stage, not two)
            if (base + 1024 < nm) flush(false, true);
        }
    }

    const bool last = (c == p.cpi - 1) && p.eof;
    if (!last) {
        flush(false);  // the whole words still in the stage
        // bytes after the last whole word belong to a word shared with the next chunk
        if ((uint32_t)lane >= head && (uint32_t)lane < fill) {
            const uint64_t ga = wbase * 4 + lane;
            if (ga < cap_bytes) out_bytes[ga ^ 3] = stage[lane];
        }
    } else {
        // end of file: the incomplete last byte is never stuffed; the residual bits are presented
        // left-aligned in one more word with their count (ff_checker.sv:289-360, functional)
        const uint64_t total_bits = o + n;
        const uint32_t rem = (uint32_t)(total_bits & 7u);
        const uint64_t stuffed_bytes = (wbase + p.out_word_base) * 4 + fill;
        if (lane == 0) {
            if (head != 0 && c > 0) {
                // Nothing has left the stage yet and its first word is shared with the previous chunk, so
                // the word reported as tail_word may be that one: fetch the previous chunk's last `head`
                // stuffed bytes (its last three owned bytes with their stuffing suffice) into the stage.
                // They are only read here; in memory that chunk writes them itself.
                // (bits [np - sbits - 24, np - sbits) of the previous chunk's np bits, in the top 24 bits of w32: from
                // its slot, or from its last 32 bits when the caller has them at hand)
                uint32_t w32;
                if (prev_known) {
                    w32 = (prev_tail32 >> sbits) << 8;
                } else {
                    const uint32_t np = p.chunk_bits[gid - 1] & kCountMask;
                    const uint32_t* Sp = S - p.slot_words;
                    const uint32_t l = np - sbits - 24u, wi = l >> 5, sh = l & 31u;
                    w32 = sh ? __builtin_amdgcn_alignbit(Sp[wi], Sp[wi + 1], 32u - sh) : Sp[wi];
                }
                uint32_t seq = 0;  // the last stuffed bytes, newest in the low byte
                for (int k = 2; k >= 0; --k) {
                    const uint32_t b = (w32 >> (8 + 8 * k)) & 0xFFu;
                    seq = (seq << 8) | b;
                    if (b == 0xFFu) seq <<= 8;
                }
                for (uint32_t k = 0; k < head; ++k) stage[k] = (uint8_t)(seq >> (8 * (head - 1 - k)));
            } else if (head != 0) {
                // a one-chunk segment whose first word begins in the segment before it: those bytes are the
                // neighbour's and read as zero here, like the ones it leaves zero in memory
                for (uint32_t k = 0; k < head; ++k) stage[k] = 0;
            }
            uint32_t f = fill;
            if (rem) {
                // the last rem bits of the chunk stream, left-aligned in a byte
                const uint32_t l = n - rem, wi = l >> 5, sh = l & 31u;
                const uint32_t w32 = sh ? __builtin_amdgcn_alignbit(S[wi], S[wi + 1], 32u - sh) : S[wi];
                const uint32_t vb = w32 >> 24;
                stage[f++] = (uint8_t)(vb & (0xFFu << (8u - rem)));
            }
            while (f & 3u) stage[f++] = 0;
        }
        fill = (fill + (rem ? 1u : 0u) + 3u) & ~3u;
        wave_sync();
        const uint32_t tail_idx = (uint32_t)(((stuffed_bytes * 8 + rem) >> 5) - p.out_word_base - wbase);  // word holding the residual
        const uint32_t tail_raw = (tail_idx * 4 < fill) ? reinterpret_cast<const uint32_t*>(stage)[tail_idx] : 0u;
        flush(true);
        if (lane == 0) {
            const uint64_t sbits_total = stuffed_bytes * 8 + rem;
            jpegenc_result r;
            r.n_full_words = sbits_total >> 5;
            r.total_bits_unstuffed = total_bits;
            r.total_bytes_stuffed = stuffed_bytes;
            r.tail_bits = (uint8_t)(sbits_total & 31u);
            r.eof_partial_ready = r.tail_bits ? 1 : 0;
            r.tail_word = r.tail_bits ? __builtin_bswap32(tail_raw) : 0u;
            const uint64_t need = r.n_full_words + (r.tail_bits ? 1 : 0) - p.out_word_base;
            r.status = (need > p.cap_words) ? (int16_t)JPEGENC_E_TOO_SMALL : (int16_t)JPEGENC_OK;
            // a chunk of the image did not fit its workspace slot (slots sized from earlier calls): nothing here is valid
            if (p.img_flags && p.img_flags[img]) r.status = (int16_t)JPEGENC_E_AGAIN;
            p.results[img] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Hand-off words of the single-launch form: one 64-bit word per chunk and chain, tag << 32 | payload, written
// once by an 8-byte device-scope store and polled by 8-byte device-scope loads (the word is the whole message:
// no other memory has to be ordered with it).  Payload bit 31 = some chunk up to here gave up waiting.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void publish_state(uint64_t* st, uint32_t tag, uint32_t payload, int lane) {
    if (lane == 0) __hip_atomic_store(st, ((uint64_t)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Gathered {
    uint64_t sum;   // of the payloads' field over the c words
    uint32_t last;  // payload of word c - 1
    uint32_t err;   // 0x80000000 if a payload carried the error bit or the wait ran out
};
// Waits for st[0 .. c) to carry `tag` (512 words at a time, eight per lane, all requested together: the
// chunks of a launch finish at about the same time, so one round trip usually finds them all) and sums `field`
// of their payloads.  The wait is bounded: a chunk whose predecessors do not arrive within about a tenth of a
// second goes on with what it has and raises the error bit, which travels down the chain into the image's result.
__device__ __forceinline__ Gathered gather_states(const uint64_t* st, uint32_t c, uint32_t tag, uint32_t field, int lane,
                                                  bool error_bit = true) {
    Gathered g{0, 0, 0};
    for (uint32_t j0 = 0; j0 < c; j0 += 512) {
        uint64_t v[8];
        for (uint32_t spins = 0;; ++spins) {
            bool ready = true;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t j = j0 + 64u * u + (uint32_t)lane;
                v[u] = (j < c) ? __hip_atomic_load(st + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((uint64_t)tag << 32);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) ready = ready && (uint32_t)(v[u] >> 32) == tag;
            if (__all((int)ready)) break;
            if (g.err || spins >= (1u << 16)) {  // (after one wait has run out no further one is tried)
                g.err = 0x80000000u;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        uint32_t part = 0, any_err = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t pay = ((uint32_t)(v[u] >> 32) == tag) ? (uint32_t)v[u] : 0u;
            part += pay & field;  // (eight fields of at most 31 bits: the bit counts are 19 bits, the 0xFF counts 16)
            any_err |= pay;
            const uint32_t jl = c - 1u - j0;  // where word c - 1 sits in this round, if it does
            if (jl < 512u && (jl >> 6) == (uint32_t)u) g.last = readlane_u32(pay, (int)(jl & 63u));
        }
        if (error_bit && __any((int)(any_err >> 31))) g.err = 0x80000000u;  // (error_bit false: payloads of 32 data bits)
        g.sum += (uint64_t)readlane_u32(wave_incl_scan_dpp(part), 63);
    }
    return g;
}

// ------------------------------------------------------------------------------------------------
// k_encode_chunks
// ------------------------------------------------------------------------------------------------
// LDS budget: per wave 64 regions of REGION words (one per lane, for the first pass), per workgroup one Huffman
// table.  The first pass gives every lane one region; if any lane overflows, the chunk is redone with regions cut
// to the lengths the first pass has counted.  Four sizes, measured on 1080p batches:
//   short  34 words (32 usable = 1024 bits per MCU, 2.0 bytes per pixel), four waves per workgroup, four workgroups =
//          sixteen waves per CU (four per SIMD, 128 registers: eight values that are only needed at the wave's start
//          and end live in scratch).  The add / logic / right-shift kind of instruction issues faster with four
//          waves on a SIMD than with three: 1.0 to 1.4 % faster than small regions where no MCU exceeds 1024 bits
//          (content up to about 1.5 bytes per pixel), 15 % slower with 6 % of the chunks redone (1.8 bytes per
//          pixel), three times slower when every chunk is.  (33 words: the same; eight waves per workgroup with 36
//          words: 4 % slower than small.)
//   small  46 words (44 usable = 1408 bits per MCU, 2.75 bytes per pixel), four waves per workgroup sharing the
//          table (eight-byte entries): three workgroups = twelve waves per CU (three per SIMD).  Enough for
//          photographic content.  (47 words with the table split into a word and a byte array: 2 % slower; 41 words
//          with one wave per workgroup: 3 % slower and overflows from 2.0 bytes per pixel on; 49 words no longer fit
//          three workgroups, 20 % slower.)
//   large  73 words (71 usable = 2272 bits, 4.4 bytes per pixel), two waves per workgroup, four workgroups = eight
//          waves per CU: 15 % slower where small regions suffice, up to twice as fast where they do not (uniform
//          noise, 2150 bits per MCU on average, is done in one pass: 181 against 85 Gpixel/s with 65 words).
//          (Four waves per workgroup: 8 % slower; one wave, seven per CU: 7 % slower.)
//   single-launch form: 65 words, one wave per workgroup, eight per CU, which is what makes 2048 chunks resident.
// The host picks short, small or large per launch from the overflow counts of its previous launches (see
// EncodeParams::stats).  The variants off the common path (quantising: short streams; rows not 8-byte aligned)
// keep 41 words and one wave per workgroup, twelve per CU: with 47 words they would spill registers.  Odd region
// strides keep lanes on different LDS banks.
// JPEGENC_HTAB8 = words per small region when the common variant keeps its Huffman table as eight-byte entries (one
// LDS read per lookup instead of two: encode kernel 3.20 against 3.26 ms per 768 frames), 0 = the split table with
// 47-word regions.  46 words and the 4 KB table are what three workgroups per CU have room for (52 992 of 53 248
// bytes); the even stride costs nothing measurable (45 words: 3.19 / 3.21 against 3.20 / 3.21 ms).
#ifndef JPEGENC_HTAB8
#define JPEGENC_HTAB8 46
#endif
[[maybe_unused]] constexpr int kRegionShort = 34, kRegionBasic = 41, kRegionSmall = JPEGENC_HTAB8 ? JPEGENC_HTAB8 : 47, kRegionFused = 65, kRegionLarge = 73;
[[maybe_unused]] constexpr int kSmallWaves = 4, kLargeWaves = 2;  // per workgroup (short regions: as small ones)
constexpr uint32_t kShortRegionBits = (kRegionShort - 2) * 32;  // what a lane may emit in a short region
constexpr uint32_t kSmallRegionBits = (kRegionSmall - 2) * 32;
constexpr uint32_t kLargeRegionBits = (kRegionLarge - 2) * 32;

// End of the range of lanes [lo, hi) that one exact-fit pass can hold: `need_incl` is the inclusive prefix
// sum of the lanes' region sizes, `before_lo` the sum over the lanes before `lo`.
__device__ __forceinline__ uint32_t fit_range_end(uint32_t need_incl, uint32_t lo, uint32_t before_lo, uint32_t kFitWords) {
    // the sums grow with the lane, so the lanes that fit are lo, lo + 1, ... up to the first that does not
    // (for the lanes before lo the difference wraps to a huge number; they are shifted out)
    const uint64_t fits = __ballot((int)(need_incl - before_lo <= kFitWords)) >> lo;
    return lo + (uint32_t)__popcll(fits);
}

// LPM (lanes per MCU) = 3, single-launch form of calls of at most 2048 x 21 MCUs (one 1080p frame): a lane codes one
// component block, lane = 3 * MCU + component, 21 MCUs per chunk -- a third of the work per wave and three times
// the waves, for calls that leave most of the GPU idle otherwise.  The 63 strings concatenate in lane order as ever.
// COEFFS: the checking variant that takes the coefficients at the coders' inputs (EncodeParams::coeffs) instead of
// transforming pixels: everything from the coders on is the code of the variant it is built from.
template <int ALIGN, bool QUANT, bool WRAP, int REGION, bool FUSED, int WAVES, int LPM = 1, bool COEFFS = false>
__global__ __launch_bounds__(64 * WAVES, REGION >= kRegionFused ? 2 : REGION == kRegionShort ? 4 : 3) void k_encode_chunks(EncodeParams p) {
    static_assert(LPM == 1 || (LPM == 3 && FUSED && !QUANT && ALIGN == 8), "one lane per block exists for the single-launch form only");
    static_assert(!COEFFS || (!QUANT && !FUSED && LPM == 1), "coefficients feed the coder that reads pairs");
    constexpr int kMcus = LPM == 3 ? 21 : kChunkMcus;  // MCUs per chunk
    constexpr int kRegionWords = REGION;
    constexpr int kOwnWords = kRegionWords - 1;  // first pass: REGION - 2 usable words and one zero word behind them
    constexpr int kOwnRegs = (kOwnWords + 7) & ~7;  // the in-register concatenation works in blocks of 8 words
    static_assert(!FUSED || WAVES == 1, "the single-launch form turns the table's place into its wave's stage");
    // behind the regions: room for the unguarded overrun of the last lane in the first pass (kMaxMcuWords from its
    // base), which later holds the 64 region sizes of the exact-fit passes
    constexpr int kPadWords = ((kMaxMcuWords - REGION > 64 ? kMaxMcuWords - REGION : 64) + 3) & ~3;
    constexpr int kPrivWords = 64 * kRegionWords + kPadWords;
    static_assert(kPrivWords % 4 == 0, "every wave's part starts on a 16-byte boundary");
    static_assert(63 * kRegionWords + kMaxMcuWords <= kPrivWords, "first-pass overrun of the last lane");
    constexpr uint32_t kFitWords = 64 * kRegionWords;  // what the exact regions of one pass may add up to
    static_assert(kFitWords >= (REGION == kRegionShort ? 13 : 16) * (kMaxMcuWords + 2), "a pass holds at least 16 (short regions: 13) worst-case MCUs");
    // One wave codes one chunk in LDS of its own; the waves of a workgroup share the Huffman table only.  `flat`
    // is all of it (the bit writer's addresses count from there: a constant base costs no instruction per store),
    // `priv` this wave's part.
    __shared__ __attribute__((aligned(16))) uint32_t flat[WAVES * kPrivWords];
    const uint32_t wave = WAVES > 1 ? threadIdx.x >> 6 : 0u;
    const uint32_t woff = wave * (uint32_t)kPrivWords;
    uint32_t* const priv = flat + woff;
    // the Huffman table: code << size as words, then code length + size as bytes (one array: the single-launch
    // form reuses it as the stage of the stuffing step)
    constexpr bool kTab8 = (JPEGENC_HTAB8 != 0) && (REGION == kRegionSmall || REGION == kRegionShort) && !QUANT && LPM == 1 && WAVES == kSmallWaves;
    __shared__ __attribute__((aligned(16))) uint32_t tables[kTab8 ? 2 * kHtabEntries : kHtabEntries + kHtabEntries / 4];
    uint32_t* const htab = tables;
    uint8_t* const hlen = reinterpret_cast<uint8_t*>(tables + kHtabEntries);
    static_assert(!FUSED || sizeof(tables) >= kStageBytes, "the stage fits the table's place");
    // Single-launch form, stream still in the LDS: the stuffing steps read it 1 KB at a time plus one word, also
    // behind its end -- never behind this wave's part of the LDS.
    static_assert(!FUSED || ((64 * (REGION - 2) - 1) / 256) * 256 + 4 * 63 + 4 < kPrivWords,
                  "the last stuffing step's reads stay inside the wave's LDS");
    // A lane per component block: a block's longest string fits its first-pass region, so such a chunk never takes
    // the exact-fit passes and never touches its workspace slot (the host reserves none for these chunks).
    static_assert(LPM != 3 || (kMaxMcuBits / 3) <= (REGION - 2) * 32, "one block always fits its region");
    const int lane = threadIdx.x & 63;
    // grid: x = chunk of the image (WAVES of them per workgroup), y and z = image (no division)
    const uint32_t c = blockIdx.x * WAVES + wave;
    const uint32_t img = blockIdx.z * 65535u + blockIdx.y;
    if (img >= p.n_images) return;  // (the whole workgroup)
    const uint32_t gid = img * p.cpi + c;
    // within one wave the LDS needs no barrier, only program order
    auto sync_wave = [] {
        if constexpr (WAVES == 1)
            __syncthreads();
        else
            wave_sync();
    };
    const bool cb_quirk = (p.flags & 2u) != 0;
    // Symbols are joined before they are appended (emit_group) except where that loses: with quantisation tables the
    // coder skips whole positions instead, and the large regions are chosen for content whose codes are long
    // (uniform noise: three pairs in ten exceed 32 bits in some lane, and the encode kernel is 10 % slower with joining).
    constexpr bool kJoin = !QUANT && REGION != kRegionLarge;
    // The coder reads the coefficients as 16-bit pairs (dct_cols_pairs, encode_block_pairs) where there is no
    // quantiser between transform and coder and a lane codes all three components.
    constexpr bool kPairs = !QUANT && LPM == 1;

    {
        constexpr int kThreads = 64 * WAVES;
        static_assert(kHtabEntries % kThreads == 0, "whole rounds of the workgroup");
        uint32_t e[kHtabEntries / kThreads];  // all loads in flight at once
#pragma unroll
        for (int j = 0; j < kHtabEntries / kThreads; ++j) e[j] = p.htab[threadIdx.x + kThreads * j];
#pragma unroll
        for (int j = 0; j < kHtabEntries / kThreads; ++j) {
            if constexpr (kTab8) {
                reinterpret_cast<uint2*>(tables)[threadIdx.x + kThreads * j] = make_uint2(e[j] & 0x7FFFFFFu, e[j] >> 27);
            } else {
                htab[threadIdx.x + kThreads * j] = e[j] & 0x7FFFFFFu;
                hlen[threadIdx.x + kThreads * j] = (uint8_t)(e[j] >> 27);
            }
        }
    }
    if (WAVES > 1 && c >= p.cpi) {  // a wave behind the image's last chunk: only the table's barrier is its business
        __syncthreads();
        return;
    }

    // MCU coordinates: one division for the chunk's first MCU (wave-uniform); a lane's own follow from its
    // offset with at most one step to the next block row when the image is at least 64 MCUs wide.  Lanes
    // behind the image's last MCU read that last MCU.
    const uint32_t b0 = c * kMcus;
    const uint32_t by0 = b0 / p.mcu_w, bx0 = b0 - by0 * p.mcu_w;
    // this lane's MCU within the chunk and (LPM == 3) its component; lane 63 has none there
    const uint32_t ml = LPM == 3 ? min((uint32_t)lane / 3u, (uint32_t)kMcus - 1u) : (uint32_t)lane;
    [[maybe_unused]] const uint32_t lcomp = LPM == 3 ? ((uint32_t)lane - 3u * ((uint32_t)lane / 3u)) : 0u;
    const uint32_t b = b0 + ml;
    const bool valid = b < p.n_mcu && (LPM == 1 || lane < 3 * kMcus);
    const uint32_t off = min(ml, p.n_mcu - 1u - b0);
    uint32_t by, bx;
    if (p.mcu_w >= (uint32_t)kMcus) {
        const uint32_t x = bx0 + off;
        const uint32_t wrap = (uint32_t)((int32_t)(p.mcu_w - 1u - x) >> 31);  // all ones when x >= mcu_w
        by = by0 - wrap;
        bx = x - (wrap & p.mcu_w);
    } else {
        const uint32_t bb = b0 + off;
        by = bb / p.mcu_w;
        bx = bb - by * p.mcu_w;
    }
    const uint8_t* img_base = p.rgb + (int64_t)img * p.image_stride;
    const uint8_t* src = img_base + (int64_t)by * 8 * p.stride + (int64_t)bx * 24;
    // ALIGN == 8 (the host grants it only when every offset inside an image is below 2^31): the image as a
    // raw buffer and the MCU's first byte as a 32-bit offset into it
    const __amdgpu_buffer_rsrc_t img_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img_base), 0, -1, 0x00020000);
    const uint32_t lane_off = by * 8u * (uint32_t)p.stride + bx * 24u;

    // DC predictors for lane 0: the MCU just before this chunk (0 at the start of an image: rst)
    // (a segment that continues an image, flags bit 2, takes them from the last MCU of the block row above
    // its first row)
    int halo[3] = {0, 0, 0};
    if constexpr (COEFFS) {
        if (c > 0) {
            const int16_t* hq = p.coeffs + ((uint64_t)img * p.n_mcu + (b0 - 1u)) * 192u;
#pragma unroll
            for (int comp = 0; comp < 3; ++comp) halo[comp] = __builtin_amdgcn_readfirstlane(sext11((int)hq[comp * 64]));
        }
    } else if (c > 0 || (p.flags & 4u)) {
        // the MCU before the chunk's first one: left of it, or the last one of the block row above
        const int64_t hy = bx0 ? (int64_t)by0 : (int64_t)by0 - 1;
        const int64_t hx = bx0 ? (int64_t)bx0 - 1 : (int64_t)p.mcu_w - 1;
        const uint8_t* hp = img_base + (hy * 8 + (lane >> 3)) * p.stride + hx * 24 + (lane & 7) * 3;
        const uint32_t R = hp[0], G = hp[1], B = hp[2];
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) {
            const CompCoef cc = comp_coef(comp);
            int smp = colour<0>(R, G, B, cc.k0, cc.cr, cc.cg, cc.cb);
            if (comp != 0) smp = WRAP ? (smp & 0xFF) : min(smp, 255);
            // the same value in every lane: kept in scalar registers, so that picking one per component is
            // scalar work too
            halo[comp] = __builtin_amdgcn_readfirstlane(halo_dc(smp, comp == 0, QUANT ? p.qmul[comp * 64] : 4096, QUANT, lane));
        }
    }
    __syncthreads();  // htab visible

    uint32_t* slot = p.slots + (uint64_t)gid * p.slot_words;
    uint32_t chunk_total = 0;  // bits written to the slot so far
    uint32_t carry = 0;        // partial last word of the slot (left-aligned), when chunk_total % 32 != 0
    uint32_t tail8 = 0;        // last 8 bits of the stream so far (uniform)
    uint32_t tail32 = 0;       // FUSED: its last 32 bits
    bool unstored = false;     // the chunk does not fit its slot

    // Pass loop.  The first pass codes all 64 lanes at once, every lane into its own region (see kRegionWords).
    // It counts every lane's bits even when a lane cannot store them, so if any lane overflowed its region
    // the lengths of all 64 strings are known, and the chunk is redone with every lane given exactly the
    // words it needs (`fit`): in one more pass when the whole chunk fits the LDS, else in as few passes over
    // consecutive lane ranges [lo, hi) as the LDS allows (uniform noise: two ranges; the worst case of 156
    // words per lane: four, five with short regions).  Lanes outside the range of a pass compute along and store nothing.
    bool fit = false;
    uint32_t lo = 0, hi = 64;             // lanes coded by this pass (uniform)
    // fit: the words of every lane's region wait in the LDS behind the regions (the room of the first pass's
    // overrun, which the guarded passes never touch), not in a register across the passes
    uint32_t* const need_lds = priv + 64 * kRegionWords;
    static_assert(kPrivWords - 64 * kRegionWords >= 64, "room for 64 region sizes");
    for (;;) {
        const bool active = valid && (uint32_t)lane >= lo && (uint32_t)lane < hi;
        // region: word 0 takes the stores of the bit writer that are not meant to land, words [1, 1+cap)
        // the string, one more word the zero behind a full string; the lanes outside the range send all their
        // stores to word 0 of the first region
        uint32_t base = woff + (uint32_t)lane * kRegionWords;  // (index into flat)
        BitWriter bw;
        bw.wlim = active ? (base + (uint32_t)kRegionWords - 1u) * 4u : base * 4u;
        if (fit) {
            const uint32_t need = need_lds[lane];
            uint32_t need_total;
            const uint32_t need_before = wave_excl_scan_u32(need, &need_total);
            base = woff + (active ? need_before - readlane_u32(need_before, (int)lo) : 0u);
            bw.wlim = active ? (base + need - 1u) * 4u : base * 4u;
        }
        bw.win = 0;
        bw.T = base * 32u;

        if constexpr (LPM == 3) {
            // one block per lane: the component's constants are values of the lane
            const bool luma = lcomp == 0;
            const int k0 = luma ? 8192 : (JPEGENC_C_OFFSET + 8192);
            const int cr = luma ? JPEGENC_Y_R : (lcomp == 1) ? -JPEGENC_CB_R : JPEGENC_CR_R;
            const int cg = luma ? JPEGENC_Y_G : (lcomp == 1) ? -JPEGENC_CB_G : -JPEGENC_CR_G;
            const int cb = luma ? JPEGENC_Y_B : (lcomp == 1) ? JPEGENC_CB_B : -JPEGENC_CR_B;
            const int crg = (int)(((uint32_t)cr & 0xFFFFu) | ((uint32_t)cg << 16));
            const int sh1 = luma ? 12 : 14, sh2 = luma ? 16 : 14;
            const int b1 = 1 << (sh1 - 1), b2 = 1 << (sh2 - 1);
            int x[64];
            uint32_t raw[48];
            load_mcu_buf<true>(img_rsrc, lane_off, (int32_t)p.stride, raw);
            int pc[4][8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int p0[4], p1[4];
                // (luma never reaches 256, so the chroma form of the rounder's last step serves every lane)
                colour_row_v<WRAP ? 1 : 2>(raw, 2 * i, k0, crg, cb, p0);
                colour_row_v<WRAP ? 1 : 2>(raw, 2 * i + 1, k0, crg, cb, p1);
                dct_row_pair_v(p0, p1, i >= 2, b1, sh1, pc[i]);
            }
            dct_cols_packed_v(pc, b2, sh2, x);
            // DC of the same component one MCU earlier = three lanes down; the first MCU takes the halo's
            const int h = luma ? halo[0] : (lcomp == 1) ? halo[1] : halo[2];
            const int before = __builtin_amdgcn_ds_bpermute((lane - 3) << 2, x[0]);
            const int pred = lane < 3 ? h : before;
            if (!fit)
                encode_block<false, QUANT, kJoin>(x, pred, lcomp == 1 && cb_quirk, htab, hlen, flat, bw);
            else
                encode_block<true, QUANT>(x, pred, lcomp == 1 && cb_quirk, htab, hlen, flat, bw);
        }
#pragma unroll 1
        for (int comp = 0; LPM == 1 && comp < 3; ++comp) {
            const CompCoef cc = comp_coef(comp);
            int x[kPairs ? 1 : 64];
            [[maybe_unused]] int tp[kPairs ? 32 : 1], dc_pair = 0;
            // The MCU's 192 bytes are fetched again for every component (mostly from L2): holding them in 48
            // registers across the components spills at the register budget of three waves per SIMD,
            // and the scratch traffic then exceeds the saved fetches; with large regions and two waves per SIMD
            // they fit (255 registers), and the kernel is 3 to 12 % slower for it (both measured).
            uint32_t raw[COEFFS ? 1 : 48];
            if constexpr (COEFFS) {
                // the block's 64 values, 11-bit ports (cr_huff.sv:18-25), in the coder's reading order -> the pairs
                // of dct_cols_pairs: tp[4 k + i] = (t of (2 i, k), t of (2 i + 1, k)), k = horizontal frequency
                const u32x4* q = reinterpret_cast<const u32x4*>(
                    p.coeffs + ((uint64_t)img * p.n_mcu + min(b, p.n_mcu - 1u)) * 192u + (uint32_t)comp * 64u);
                uint32_t cw[32];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const u32x4 v = q[i];
                    cw[4 * i] = v.x; cw[4 * i + 1] = v.y; cw[4 * i + 2] = v.z; cw[4 * i + 3] = v.w;
                }
                auto at = [&](int pos) {
                    const int z = inv_zz(pos);
                    return sext11((int)(cw[z >> 1] >> (16 * (z & 1))));
                };
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        tp[4 * k + i] = pk_amp_src((at(16 * i + k) & 0xFFFF) | (at(16 * i + 8 + k) << 16));
                dc_pair = at(0);
            } else if constexpr (ALIGN == 8) {
                if (comp == 2)
                    load_mcu_buf<true>(img_rsrc, lane_off, (int32_t)p.stride, raw);  // the last fetch of these pixels
                else
                    load_mcu_buf<false>(img_rsrc, lane_off, (int32_t)p.stride, raw);
            } else {
                load_mcu<ALIGN>(src, p.stride, raw);
            }
            // Two separately compiled bodies (shifts are immediates).  The differing asm comments keep
            // the optimiser from folding them back into one body with run-time shift counts.  Each row
            // is converted and transformed before the next one is touched, so the raw pixels die early.
            if constexpr (COEFFS) {
            } else if (comp == 0) {
                int pc[4][8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int p0[4], p1[4];
                    colour_row<0>(raw, 2 * i, cc, p0);
                    colour_row<0>(raw, 2 * i + 1, cc, p1);
                    dct_row_pair<true>(p0, p1, i >= 2, pc[i]);
                }
                if constexpr (kPairs)
                    dct_cols_pairs<true>(pc, tp, dc_pair);
                else
                    dct_cols_packed<true>(pc, x);
                asm volatile("; luma dct done" ::: "memory");
            } else {
                int pc[4][8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int p0[4], p1[4];
                    colour_row<WRAP ? 1 : 2>(raw, 2 * i, cc, p0);
                    colour_row<WRAP ? 1 : 2>(raw, 2 * i + 1, cc, p1);
                    dct_row_pair<false>(p0, p1, i >= 2, pc[i]);
                }
                if constexpr (kPairs)
                    dct_cols_pairs<false>(pc, tp, dc_pair);
                else
                    dct_cols_packed<false>(pc, x);
                asm volatile("; chroma dct done" ::: "memory");
            }
            if constexpr (QUANT) {
                const int32_t* qm = p.qmul + comp * 64;
#pragma unroll
                for (int i = 0; i < 64; ++i) x[i] = quant(x[i], qm[i]);
            }
            const int h = (comp == 0) ? halo[0] : (comp == 1) ? halo[1] : halo[2];
            const int dc = kPairs ? dc_pair : x[0];
            // DC of the lane before; lane 0 has no source and keeps h (wave_shr:1)
            const int pred = __builtin_amdgcn_update_dpp(h, dc, 0x138, 0xF, 0xF, false);
            if constexpr (kPairs) {
                if (!fit)
                    encode_block_pairs<false, kJoin, COEFFS, kTab8>(tp, dc, pred, comp == 1 && cb_quirk, htab, hlen, flat, bw);
                else
                    encode_block_pairs<true, JPEGENC_FIT_MERGE && kJoin, COEFFS, kTab8>(tp, dc, pred, comp == 1 && cb_quirk, htab, hlen, flat, bw);
            } else {
                if (!fit)
                    encode_block<false, QUANT, kJoin>(x, pred, comp == 1 && cb_quirk, htab, hlen, flat, bw);
                else
                    encode_block<true, QUANT>(x, pred, comp == 1 && cb_quirk, htab, hlen, flat, bw);
            }
        }
        const uint32_t Lall = valid ? (bw.T - base * 32u) : 0u;
        const uint32_t cap = fit ? need_lds[lane] - 2u : (uint32_t)kRegionWords - 2u;
        // flush the residual bits, left-aligned and zero-filled
        if (active && Lall <= cap * 32u) {
            if ((bw.T & 31u) != 0) flat[(bw.T >> 5) + 1u] = (uint32_t)bw.win << ((0u - bw.T) & 31u);
            flat[((bw.T + 31u) >> 5) + 1u] = 0u;  // one zero word behind the string
        }
        if (!fit && p.stats) {
            // for the host's choice of REGION: chunks with an MCU beyond the short, the small and the large region,
            // spread over 16 counters each on lines of their own, and the launch's chunk count beside them
            if (__any((int)(Lall > kShortRegionBits))) {
                if (lane == 0)
                    __hip_atomic_fetch_add(&p.stats[(gid & 15u) * 16u + 4u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__any((int)(Lall > kSmallRegionBits))) {
                    if (lane == 0)
                        __hip_atomic_fetch_add(&p.stats[(gid & 15u) * 16u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__any((int)(Lall > kLargeRegionBits)) && lane == 0)  // ... and even for a large one
                        __hip_atomic_fetch_add(&p.stats[(gid & 15u) * 16u + 1u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (gid == 0 && lane == 0) {
                p.stats[16 * 16] = p.n_images * p.cpi;
                p.stats[16 * 16 + 1] = p.report_bits;
                p.stats[16 * 16 + 2] = p.slot_words;
                p.stats[16 * 16 + 3] = p.stats_seq;
            }
        }
        if (!fit && __any((int)(Lall > cap * 32u))) {
            // words per lane: the unread word 0, the string, the zero word behind it
            const uint32_t need = ((Lall + 31u) >> 5) + 2u;
            need_lds[lane] = need;
            fit = true;
            lo = 0;
            hi = fit_range_end(wave_incl_scan_dpp(need), 0u, 0u, kFitWords);
            sync_wave();
            continue;  // restart the chunk with exact regions
        }
        const uint32_t L = active ? Lall : 0u;  // bits of this lane's string in this pass
        sync_wave();  // strings visible to all lanes

        // ---- concatenate the strings of this pass in place, in lane order, at LDS word 0 ----------
        // Destination word d0+t is (src[t-1]:src[t]) >> s; the bits a string shares with its predecessor's
        // last word travel in `tailw` (wave-uniform).  In place is safe because a string never moves to a
        // higher address and all reads of a step precede its writes.
        uint32_t gtotal;
        const uint32_t off = wave_excl_scan_u32(L, &gtotal);
        if (!fit) {
            // First pass, every lane at once: each lane takes its whole string (at most kOwnWords words,
            // the last one zero) into registers, the destination range is cleared, and every lane ORs
            // its words, shifted to its bit offset, into place.  Words shared by neighbouring strings
            // get their parts from several lanes, hence the OR.  In place is safe because all reads
            // come first: the LDS serves one wave's instructions in order.
            const uint32_t s = off & 31u, j0 = off >> 5;
            const uint32_t nd = L ? ((s + L + 31u) >> 5) : 0u;  // destination words touched
            const uint32_t nmax = wave_max_u32(nd);             // uniform, <= kOwnWords
            uint32_t own[kOwnRegs];  // (the words past kOwnWords are read and never used)
#pragma unroll
            for (int t = 0; t < kOwnRegs; ++t) own[t] = flat[base + 1 + t];
            wave_sync();
            // (The four-word forms here and in the slot copy below are left out of the quantising variants:
            // their streams are short, and at the register limit the allocator then spills more.)
            if constexpr (QUANT) {
                for (uint32_t i = lane; i < ((gtotal + 31u) >> 5); i += 64) priv[i] = 0u;
            } else {
                for (uint32_t i = lane; i < ((gtotal + 127u) >> 7); i += 64)  // four words per lane and step
                    reinterpret_cast<uint4*>(priv)[i] = make_uint4(0u, 0u, 0u, 0u);
            }
            wave_sync();
#pragma unroll
            for (int t0 = 0; t0 < kOwnRegs; t0 += 8) {
                if ((uint32_t)t0 < nmax) {
#pragma unroll
                    for (int t = t0; t < t0 + 8; ++t) {
                        const uint32_t w = __builtin_amdgcn_alignbit(t ? own[t - 1] : 0u, own[t], s);
                        if ((uint32_t)t < nd)
                            __hip_atomic_fetch_or(&priv[j0 + t], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            wave_sync();
        }
        uint32_t tailw = 0;
        for (uint32_t M = lo; fit && M < hi; ++M) {  // one source string per step, all 64 lanes moving one word each
            const uint32_t LM = readlane_u32(L, (int)M);
            if (LM == 0) continue;  // uniform
            const uint32_t oM = readlane_u32(off, (int)M);
            const uint32_t bM = readlane_u32(base, (int)M) + 1u;
            const uint32_t s = oM & 31u, d0 = oM >> 5;
            const uint32_t nw = (LM + 31u) >> 5;      // source words
            const uint32_t nd = (s + LM + 31u) >> 5;  // destination words touched
            if (nd <= 64u) {
                const uint32_t t = lane;
                const uint32_t wlo = (t < nw) ? flat[bM + t] : 0u;
                uint32_t whi = __shfl_up(wlo, 1);
                if (t == 0) whi = 0u;
                uint32_t w = __builtin_amdgcn_alignbit(whi, wlo, s);
                if (t == 0) w |= tailw;
                sync_wave();  // all reads of this step before its writes
                if (t < nd) priv[d0 + t] = w;
                tailw = ((s + LM) & 31u) ? readlane_u32(w, (int)nd - 1) : 0u;
                sync_wave();
            } else {
                for (uint32_t t0 = 0; t0 < nd; t0 += 64) {
                    const uint32_t t = t0 + lane;
                    const uint32_t wlo = (t < nw) ? flat[bM + t] : 0u;
                    const uint32_t whi = (t >= 1 && t - 1 < nw) ? flat[bM + t - 1] : 0u;
                    uint32_t w = __builtin_amdgcn_alignbit(whi, wlo, s);
                    if (t == 0) w |= tailw;
                    sync_wave();
                    if (t < nd) priv[d0 + t] = w;
                    if (t0 + 64 >= nd) tailw = ((s + LM) & 31u) ? readlane_u32(w, (int)(nd - 1 - t0)) : 0u;
                    sync_wave();
                }
            }
        }

        // ---- append the stream of this pass to the slot at bit offset chunk_total -----------------
        // (a chunk that outgrows its slot -- a stride below the worst case, chosen by the host from earlier launches --
        // stores nothing from here on and only goes on counting)
        if (chunk_total + gtotal > p.slot_words * 32u) unstored = true;  // uniform
        if (unstored) {
            chunk_total += gtotal;
        } else if (gtotal != 0) {
            const uint32_t s = chunk_total & 31u, d0 = chunk_total >> 5;
            const uint32_t nw = (gtotal + 31u) >> 5;
            const uint32_t nd = (s + gtotal + 31u) >> 5;
            // The chunk's first pass starts the slot: whole groups of four words go out as they are, four per
            // lane and step; the general loop below is left with the last 0 to 3 words.  (Single-launch form:
            // a chunk done in its first pass stays in the LDS, where the stuffing step below reads it.)
            uint32_t t_first = lane;
            const bool to_slot = !FUSED || fit;
            if (!QUANT && chunk_total == 0 && to_slot) {
                const uint32_t nq = nd >> 2;
                for (uint32_t q = lane; q < nq; q += 64) {
                    if constexpr (FUSED)  // read back by this wave in a moment: kept in the L2
                        reinterpret_cast<u32x4*>(slot)[q] = reinterpret_cast<const u32x4*>(priv)[q];
                    else
                        __builtin_nontemporal_store(reinterpret_cast<const u32x4*>(priv)[q], reinterpret_cast<u32x4*>(slot) + q);
                }
                t_first = 4u * nq + lane;
            }
            for (uint32_t t = t_first; to_slot && t < nd; t += 64) {
                const uint32_t wlo = (t < nw) ? priv[t] : 0u;
                const uint32_t whi = (t >= 1 && t - 1 < nw) ? priv[t - 1] : 0u;
                uint32_t w = __builtin_amdgcn_alignbit(whi, wlo, s);
                if (t == 0) w |= carry;
                if constexpr (FUSED)
                    slot[d0 + t] = w;
                else
                    __builtin_nontemporal_store(w, &slot[d0 + t]);  // streamed: keeps the pixels in L2 for the next component
            }
            const uint32_t nt = chunk_total + gtotal;
            {   // the last 8 bits of the stream so far (a non-empty pass has at least 18 bits)
                const uint32_t l = gtotal - 8u, wi = l >> 5, sh = l & 31u;
                tail8 = (sh ? ((priv[wi] << sh) | (priv[wi + 1] >> (32u - sh))) : priv[wi]) >> 24;
            }
            if (FUSED && gtotal >= 32u) {  // (a chunk that another one follows holds 64 MCUs: at least 1152 bits)
                const uint32_t l = gtotal - 32u, wi = l >> 5, sh = l & 31u;
                tail32 = sh ? ((priv[wi] << sh) | (priv[wi + 1] >> (32u - sh))) : priv[wi];
            }
            if (nt & 31u) {  // keep the partial last word for the next pass (uniform)
                const uint32_t t = nd - 1;
                const uint32_t wlo = (t < nw) ? priv[t] : 0u;
                const uint32_t whi = (t >= 1 && t - 1 < nw) ? priv[t - 1] : 0u;
                uint32_t w = s ? ((whi << (32u - s)) | (wlo >> s)) : wlo;
                if (t == 0) w |= carry;
                carry = w;
            } else {
                carry = 0;
            }
            chunk_total = nt;
        }
        if (hi >= 64u) break;
        sync_wave();  // LDS free for the next range of lanes
        lo = hi;
        {
            const uint32_t need_incl = wave_incl_scan_dpp(need_lds[lane]);
            hi = fit_range_end(need_incl, lo, readlane_u32(need_incl, (int)lo - 1), kFitWords);
        }
    }
    // bit count (at most 64 * 4980 < 2^24) and, for the successor's first byte, the stream's last 8 bits
    if (!FUSED && lane == 0) {
        p.chunk_bits[gid] = unstored ? kUnstoredChunkBits : (chunk_total | (tail8 << 24));
        if (p.stats && chunk_total > p.report_bits)  // (few chunks: the host keeps report_bits just below the longest)
            __hip_atomic_fetch_max(&p.stats[(gid & 15u) * 16u + 2u], chunk_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (unstored) {
            if (p.img_flags) __hip_atomic_store(&p.img_flags[img], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.stats) __hip_atomic_fetch_add(&p.stats[(gid & 15u) * 16u + 3u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    if constexpr (FUSED) {
        // ---- single-launch form: place, stuff and write this chunk's stream ----------------------------
        // Two chains of hand-off words run through the image's chunks: bit counts (with the stream's last 8
        // bits), then 0xFF counts, which need the bit offsets; a third word per chunk carries its last 32 bits,
        // which the image's last chunk may need for a tail word it shares with the chunk before it.  Nothing
        // else crosses between waves, so no fences.  Every chunk of the launch is resident (the host's limit),
        // so every word a chunk waits for is on its way.
        const uint32_t n_chunks = p.n_images * p.cpi;
        uint64_t* const st_bits = p.state + (uint64_t)img * p.cpi;
        uint64_t* const st_ff = st_bits + n_chunks;
        uint64_t* const st_tail = st_ff + n_chunks;
        // the chunk's stream: still in the LDS from word 0 on when the first pass sufficed, else in its slot
        const uint32_t* const stream = fit ? slot : priv;
        if (fit) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slot is written: this wave reads it back below
        if (c + 2 == p.cpi) publish_state(st_tail + c, p.generation, tail32, lane);
        publish_state(st_bits + c, p.generation, chunk_total | (tail8 << 23), lane);
        const Gathered gb = gather_states(st_bits, c, p.generation, 0x7FFFFu, lane);
        const uint64_t o = gb.sum;
        const uint32_t pred8 = c ? ((gb.last >> 23) & 0xFFu) : 0u;
        StuffParams sp;
        sp.slots = p.slots;
        sp.slot_words = p.slot_words;
        sp.img_flags = nullptr;
        sp.chunk_bits = p.chunk_bits;
        sp.bit_off = nullptr;
        sp.chunk_ff = nullptr;
        sp.ff_off = nullptr;
        sp.words_out = p.words_out;
        sp.out_stride = p.out_stride;
        sp.cap_words = p.cap_words;
        sp.results = p.results;
        sp.cpi = p.cpi;
        sp.pred_tail8 = 0;
        sp.bit_base = sp.ff_base = sp.out_word_base = 0;
        sp.eof = 1;
        sp.spec_depth = kStuffDepth;
        sp.inline_scan = 0;
        uint4 w[kStuffDepth];
        uint32_t w4[kStuffDepth];
        request_chunk<kStuffDepth>(stream, lane, w, w4);
        const bool one_pass = chunk_total <= 8u * 1024u * kStuffDepth - 64u;
        const uint32_t mine = count_owned_ff<kStuffDepth>(stream, w, w4, chunk_total, o, pred8, lane);
        publish_state(st_ff + c, p.generation, mine | gb.err, lane);
        const Gathered gf = gather_states(st_ff, c, p.generation, 0x7FFFFFFFu, lane);
        const bool last = c + 1 == p.cpi;
        uint32_t prev_tail32 = 0, tail_err = 0;
        if (last && c > 0) {
            const Gathered gt = gather_states(st_tail + (c - 1), 1u, p.generation, 0u, lane, false);
            prev_tail32 = gt.last;
            tail_err = gt.err;
        }
        wave_sync();  // the Huffman table has served: its place is this wave's stage
        write_chunk<kStuffDepth>(sp, stream, w, w4, one_pass, reinterpret_cast<uint8_t*>(tables), img, c, gid, chunk_total, o,
                                 gf.sum, pred8, lane, true, c ? (gb.last & 0x7FFFFu) : 0u, prev_tail32);
        if (last && (gb.err | gf.err | tail_err) && lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            p.results[img].status = (int16_t)JPEGENC_E_HIP;  // a wait ran out: the words are not to be trusted
        }
    }
}

#ifndef JPEGENC_TU_QUANT  // (the second translation unit holds the quantising encode variants only)

// ------------------------------------------------------------------------------------------------
// k_scan: exclusive prefix per image, one 256-thread workgroup per image.
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_scan(ScanParams p) {
    __shared__ uint64_t part[256];
    const uint32_t img = blockIdx.x;
    const uint32_t* in = p.counts + (uint64_t)img * p.cpi;
    uint64_t* out = p.prefix + (uint64_t)img * p.cpi;
    const uint32_t per = (p.cpi + 255u) / 256u;
    const uint32_t lo = min(threadIdx.x * per, p.cpi), hi = min(lo + per, p.cpi);
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += in[i] & kCountMask;
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        uint64_t v = (threadIdx.x >= (unsigned)d) ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t run = part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        out[i] = run;
        run += in[i] & kCountMask;
    }
    if (threadIdx.x == 255 && img == 0) {  // the last thread's range ends at cpi: its running sum is the total
        if (p.total_out) *p.total_out = run;
        if (p.last_hi8_out) *p.last_hi8_out = in[p.cpi - 1] >> 24;
    }
}

// The place of a band in the whole stream: see SpanParams.
__global__ void k_segment_span(SpanParams p) {
    const uint64_t o = *p.bit_base, end_bits = o + *p.bits;
    const uint64_t pos0 = (o >> 3) + *p.ff_base;
    const uint64_t end_pos = (end_bits >> 3) + *p.ff_base + *p.ff;  // stuffed bytes of the whole stream so far
    const uint64_t fw = pos0 >> 2;
    const uint64_t end_word = p.is_last ? ((end_pos * 8 + (end_bits & 7u) + 31u) >> 5) : ((end_pos + 3u) >> 2);
    const uint64_t n = end_word - fw;
    p.span[0] = fw;
    p.span[1] = n;
    if (p.cap_words) p.words[0] = 0u;
    if (n && n - 1 < p.cap_words) p.words[n - 1] = 0u;
}

hipError_t launch_segment_span(const SpanParams& p, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_segment_span, dim3(1), dim3(1), 0, s, p);
    return hipGetLastError();
}

// The stuffing parameters with a band's place taken from device memory where the host has left pointers.
__device__ __forceinline__ StuffParams with_device_bases(const StuffParams& p) {
    StuffParams q = p;
    if (p.bit_base_p) {
        q.bit_base = *p.bit_base_p;
        q.pred_tail8 = (uint32_t)*p.pred_tail8_p & 0xFFu;
    }
    if (p.ff_base_p) {
        q.ff_base = *p.ff_base_p;
        q.out_word_base = ((q.bit_base >> 3) + q.ff_base) >> 2;
    }
    return q;
}

// ------------------------------------------------------------------------------------------------
// Stuffing (k_stuff_count, k_stuff_write): one wavefront per chunk, 16 stream bytes per lane and step.
// The image's unstuffed stream is the concatenation of its chunk streams; chunk c covers global bits
// [o, o+n).  A byte of that stream is owned by the chunk holding its last bit, so chunk c owns bytes
// [o>>3, (o+n)>>3): its first byte may begin in the previous chunk (every chunk has >= 1152 bits, so
// at most one byte, and at most one predecessor, is involved).  Owned bytes equal to 0xFF get 0x00
// after them (ff_checker.sv:96-179).  Output byte i of the stuffed stream lives in word i/4 at bits
// 31-8*(i%4), i.e. at byte address i^3 of a little-endian word array.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// k_stuff_count: number of 0xFF bytes each chunk owns (ownership rule above).  One wave
// per chunk, 16 stream bytes per lane and step.  The byte that begins in the predecessor is taken on
// its own; the bytes after it start A = (8 - o % 8) % 8 bits into the chunk's stream, so lane t's four
// shifted words come from one aligned 16-byte load at word 256 * step + 4 * t and the word behind it.
// ------------------------------------------------------------------------------------------------
// Sum of counts[0 .. c) (low 24 bits of each word) by one wave: the exclusive prefix a chunk needs, for launches
// too small to be worth a scan kernel.
__device__ __forceinline__ uint32_t wave_sum_before(const uint32_t* counts, uint32_t c, int lane) {
    uint32_t s = 0;
    for (uint32_t j = lane; j < c; j += 64) s += counts[j] & kCountMask;
    return readlane_u32(wave_incl_scan_dpp(s), 63);
}

template <int SPEC>  // 1 KB steps requested before the chunk's length is known
__global__ __launch_bounds__(64 * kStuffWaves) void k_stuff_count(StuffParams p_in, uint32_t n_chunks) {
    const StuffParams p = with_device_bases(p_in);
    const int lane = threadIdx.x & 63;
    const uint32_t gid = blockIdx.x * kStuffWaves + (threadIdx.x >> 6);
    if (gid >= n_chunks) return;
    const uint32_t* S = p.slots + (uint64_t)gid * p.slot_words;
    // The first kStuffDepth steps' words are requested before the chunk's bit count and offset are known
    // (their addresses depend on neither), so a photographic chunk costs one memory round trip, not one
    // for the counts and one per step.  Words behind the chunk's end are read (the slot is always large
    // enough) and ignored.
    uint4 w[kStuffDepth];
    uint32_t w4[kStuffDepth];
    request_chunk<SPEC>(S, lane, w, w4);
    const uint32_t c = gid % p.cpi;
    const uint32_t n = p.chunk_bits[gid] & kCountMask;
    uint64_t o;
    if (p.inline_scan) {  // uniform
        o = wave_sum_before(p.chunk_bits + (gid - c), c, lane);
        if (lane == 0) p.bit_off[gid] = o;
        o += p.bit_base;
    } else {
        o = p.bit_off[gid] + p.bit_base;
    }
    const uint32_t pred8 = (c > 0) ? (p.chunk_bits[gid - 1] >> 24) : p.pred_tail8;
    const uint32_t total = count_owned_ff<SPEC>(S, w, w4, n, o, pred8, lane);
    if (lane == 0) p.chunk_ff[gid] = total;
}

// ------------------------------------------------------------------------------------------------
// k_stuff_write: shifts every chunk stream to its byte phase, inserts 0x00 after 0xFF and writes the
// 32-bit words; the last chunk of an image also writes the end-of-file result.  Same walk over the
// chunk as k_stuff_count.  The stuffed bytes of a step are laid out in stream order in an LDS stage
// (one per wave) and leave it as whole words; the bytes of a word shared with a neighbouring chunk are
// stored one by one.
// ------------------------------------------------------------------------------------------------
template <int SPEC>
__global__ __launch_bounds__(64 * kStuffWaves) void k_stuff_write(StuffParams p_in, uint32_t n_chunks) {
    const StuffParams p = with_device_bases(p_in);
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[kStuffWaves][kStageBytes];
    uint8_t* const stage = stage_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const uint32_t gid = blockIdx.x * kStuffWaves + (threadIdx.x >> 6);
    if (gid >= n_chunks) return;
    const uint32_t* S = p.slots + (uint64_t)gid * p.slot_words;
    uint4 w[kStuffDepth];
    uint32_t w4[kStuffDepth];
    request_chunk<SPEC>(S, lane, w, w4);  // ahead of the counts and offsets, as in k_stuff_count
    const uint32_t img = gid / p.cpi;
    const uint32_t c = gid - img * p.cpi;
    const uint32_t n = p.chunk_bits[gid] & kCountMask;
    const uint64_t o = p.bit_off[gid] + p.bit_base;
    const uint64_t ff_before = p.inline_scan ? (uint64_t)wave_sum_before(p.chunk_ff + (gid - c), c, lane) : p.ff_off[gid];
    const uint32_t pred8 = (c > 0) ? (p.chunk_bits[gid - 1] >> 24) : p.pred_tail8;
    write_chunk<SPEC>(p, S, w, w4, true, stage, img, c, gid, n, o, ff_before, pred8, lane);
}

// ------------------------------------------------------------------------------------------------
// k_stuff_image: both stuffing steps for whole images in one launch, one workgroup of kImageWaves waves per
// image (two such workgroups share a CU).  The workgroup walks the image's chunks in order, kImageWaves at a time (one per wave): every wave
// requests its chunk, works out its bit offset from the counts of the round's earlier chunks, counts its
// 0xFF bytes, and after one barrier -- the counts of the round stand in LDS -- writes its stuffed bytes.
// The running bit offset and 0xFF count stay in registers, so no prefix arrays, no scan launches, and each
// chunk stream is read from memory once (a chunk of more than kStuffDepth steps: twice).  For batches of
// many images (the host's threshold); fewer images keep k_stuff_count / k_stuff_write, whose grid covers the
// chunks.  Both stuffing steps are bound by vector instructions (87 % of the write step's time), not by
// memory, which is why reading the streams once buys so little.
// ------------------------------------------------------------------------------------------------
constexpr int kImageWaves = 8;  // measured on 1024 x 1080p: 16 waves 1.49 ms, 8 waves 1.41 ms, 4 waves 1.40 ms (and 1.10 against 0.73 ms at 512 frames)

template <int SPEC>
__global__ __launch_bounds__(64 * kImageWaves) void k_stuff_image(StuffParams p) {
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[kImageWaves][kStageBytes];
    __shared__ uint32_t round_ff[2][kImageWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t* const stage = stage_all[wave];
    const uint32_t img = blockIdx.x;
    const uint32_t first = img * p.cpi;
    uint64_t bits_before = p.bit_base, ff_before = 0;  // of the rounds done so far
    uint32_t tail8_before = p.pred_tail8;              // last 8 bits before this round
    for (uint32_t c0 = 0, r = 0; c0 < p.cpi; c0 += kImageWaves, ++r) {
        const uint32_t c = c0 + wave;
        const bool have = c < p.cpi;  // the image's last round may be short; its idle waves still meet the barrier
        const uint32_t gid = first + (have ? c : p.cpi - 1u);
        const uint32_t* S = p.slots + (uint64_t)gid * p.slot_words;
        uint4 w[kStuffDepth];
        uint32_t w4[kStuffDepth];
        if (have) request_chunk<SPEC>(S, lane, w, w4);
        // the round's bit counts: lane j holds chunk c0 + j
        const uint32_t word = (lane < kImageWaves && c0 + lane < p.cpi) ? p.chunk_bits[first + c0 + lane] : 0u;
        const uint32_t incl = wave_incl_scan_dpp(word & kCountMask);
        const uint32_t n = readlane_u32(word, wave) & kCountMask;
        const uint64_t o = bits_before + (readlane_u32(incl, wave) - n);
        const uint32_t pred8 = wave ? (readlane_u32(word, wave - 1) >> 24) : tail8_before;
        const bool one_pass = n <= 8u * 1024u * kStuffDepth - 64u;  // the requested steps cover it at any phase
        uint32_t mine = 0;
        if (have) mine = count_owned_ff<SPEC>(S, w, w4, n, o, pred8, lane);
        if (lane == 0) round_ff[r & 1][wave] = mine;
        __syncthreads();  // one barrier per round: the counts alternate between two rows
        const uint32_t ffw = (lane < kImageWaves) ? round_ff[r & 1][lane] : 0u;
        const uint32_t ffi = wave_incl_scan_dpp(ffw);
        if (have)
            write_chunk<SPEC>(p, S, w, w4, one_pass, stage, img, c, gid, n, o, ff_before + (readlane_u32(ffi, wave) - mine),
                              pred8, lane);
        bits_before += readlane_u32(incl, kImageWaves - 1);
        ff_before += readlane_u32(ffi, kImageWaves - 1);
        tail8_before = readlane_u32(word, kImageWaves - 1) >> 24;  // only read when another round follows: then this one was full
    }
}

// The same without a barrier per round, for images of at most kStuffImageLdsChunks chunks: every chunk's 0xFF count
// has a word of its own in the LDS (valid bit | count, all cleared before the first round), and a wave waits only
// for the seven counts between its previous chunk and this one.  A count is published before its wave asks for
// anything, so no wait depends on another wait; the waves drift apart, and one wave's loads and stores overlap
// another's arithmetic -- with a barrier per round all eight wait for memory together and compute together.
constexpr uint32_t kFfValid = 0x80000000u;
// WAVES per workgroup: 8 for 512 images and more (two workgroups per CU), 16 for 256 to 511 (one per CU).
template <int SPEC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_stuff_image_flags(StuffParams p) {
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[WAVES][kStageBytes];
    __shared__ uint32_t ff_pub[kStuffImageLdsChunks];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t* const stage = stage_all[wave];
    const uint32_t img = blockIdx.x;
    const uint32_t first = img * p.cpi;
    for (uint32_t i = threadIdx.x; i < p.cpi; i += 64 * WAVES) ff_pub[i] = 0u;
    __syncthreads();
    uint64_t bits_before = p.bit_base;      // of the rounds done so far
    uint64_t ff_upto = 0;                   // 0xFF bytes of the chunks up to and including this wave's previous one
    uint32_t tail8_before = p.pred_tail8;   // last 8 bits before this round
    for (uint32_t c0 = 0; c0 + wave < p.cpi; c0 += WAVES) {
        const uint32_t c = c0 + wave;
        const uint32_t gid = first + c;
        const uint32_t* S = p.slots + (uint64_t)gid * p.slot_words;
        uint4 w[kStuffDepth];
        uint32_t w4[kStuffDepth];
        request_chunk<SPEC>(S, lane, w, w4);
        // the round's bit counts: lane j holds chunk c0 + j
        const uint32_t word = (lane < WAVES && c0 + lane < p.cpi) ? p.chunk_bits[first + c0 + lane] : 0u;
        const uint32_t incl = wave_incl_scan_dpp(word & kCountMask);
        const uint32_t n = readlane_u32(word, wave) & kCountMask;
        const uint64_t o = bits_before + (readlane_u32(incl, wave) - n);
        const uint32_t pred8 = wave ? (readlane_u32(word, wave - 1) >> 24) : tail8_before;
        const bool one_pass = n <= 8u * 1024u * kStuffDepth - 64u;  // the requested steps cover it at any phase
        const uint32_t mine = count_owned_ff<SPEC>(S, w, w4, n, o, pred8, lane);
        if (lane == 0) __hip_atomic_store(&ff_pub[c], mine | kFfValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // the counts of the chunks between this wave's previous chunk and this one: lane j asks for chunk c - 7 + j
        uint32_t between = 0;
        {
            const bool asks = lane < WAVES - 1 && c + (uint32_t)lane >= (uint32_t)(WAVES - 1);
            const uint32_t at = asks ? c - (uint32_t)(WAVES - 1) + (uint32_t)lane : c;
            for (;;) {
                const uint32_t v = __hip_atomic_load(&ff_pub[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__all((int)((v & kFfValid) != 0))) {
                    between = asks ? (v & ~kFfValid) : 0u;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            between = readlane_u32(wave_incl_scan_dpp(between), 63);
        }
        write_chunk<SPEC>(p, S, w, w4, one_pass, stage, img, c, gid, n, o, ff_upto + between, pred8, lane);
        ff_upto += (uint64_t)between + mine;
        bits_before += readlane_u32(incl, WAVES - 1);
        tail8_before = readlane_u32(word, WAVES - 1) >> 24;  // only read when another round follows: then this one was full
    }
}

// ------------------------------------------------------------------------------------------------
// k_pad_edges: copies images of w x h pixels into images of w8 x h8 (the next multiples of 8), repeating
// the last column and the last row.  An extension: jpeg_top has no edge handling (a block is 64
// consecutive samples, cb_dct.sv:478-541) and the entry points reject such sizes unless asked to pad.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pad_edges(const uint8_t* src, int64_t stride, int64_t image_stride, int w, int h,
                                                   uint8_t* dst, int w8, int h8) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // pixel of one padded image
    const int img = blockIdx.y;
    if (idx >= (int64_t)w8 * h8) return;
    const int y = (int)(idx / w8), x = (int)(idx - (int64_t)y * w8);
    const uint8_t* q = src + (int64_t)img * image_stride + (int64_t)min(y, h - 1) * stride + (int64_t)min(x, w - 1) * 3;
    uint8_t* o = dst + ((int64_t)img * h8 * w8 + idx) * 3;
    o[0] = q[0];
    o[1] = q[1];
    o[2] = q[2];
}

hipError_t launch_pad_edges(const uint8_t* src, int64_t stride, int64_t image_stride, int w, int h, int n_images,
                            uint8_t* dst, hipStream_t s) {
    const int w8 = (w + 7) & ~7, h8 = (h + 7) & ~7;
    const int64_t px = (int64_t)w8 * h8;
    if ((px + 255) / 256 > 0x7FFFFFFFll) return hipErrorInvalidValue;
    (void)hipGetLastError();
    for (int first = 0; first < n_images; first += 65535) {  // the grid's second dimension ends at 65535
        const int n = min(65535, n_images - first);
        hipLaunchKernelGGL(k_pad_edges, dim3((unsigned)((px + 255) / 256), (unsigned)n), dim3(256), 0, s,
                           src + (int64_t)first * image_stride, stride, image_stride, w, h, dst + (int64_t)first * px * 3, w8, h8);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_coeffs: the coefficients as they enter the three Huffman coders (after colour conversion, the two DCT
// passes, the quantisers and the wrapper transpose: yd_q_h.sv:120-135 and its Cb/Cr twins), for checking that
// stage on its own.  Lane = MCU with the device functions of k_encode_chunks; out[mcu][component][k], k = the
// position in the coder's reading order (zig-zag, cr_huff.sv:1053-1304).  Not on the hot path: byte loads,
// strided 16-bit stores.
// ------------------------------------------------------------------------------------------------
template <bool WRAP>
__global__ __launch_bounds__(64) void k_coeffs(const uint8_t* rgb, int64_t stride, uint32_t mcu_w, uint32_t n_mcu,
                                               const int32_t* qmul, int16_t* out) {
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= n_mcu) return;
    const uint32_t by = b / mcu_w, bx = b - by * mcu_w;
    const uint8_t* src = rgb + (int64_t)by * 8 * stride + (int64_t)bx * 24;
    uint32_t raw[48];
    load_mcu<1>(src, stride, raw);
#pragma unroll 1
    for (int comp = 0; comp < 3; ++comp) {
        const CompCoef cc = comp_coef(comp);
        int x[64];
        int pc[4][8];
        if (comp == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int p0[4], p1[4];
                colour_row<0>(raw, 2 * i, cc, p0);
                colour_row<0>(raw, 2 * i + 1, cc, p1);
                dct_row_pair<true>(p0, p1, i >= 2, pc[i]);
            }
            dct_cols_packed<true>(pc, x);
            asm volatile("; luma dct done" ::: "memory");
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int p0[4], p1[4];
                colour_row<WRAP ? 1 : 2>(raw, 2 * i, cc, p0);
                colour_row<WRAP ? 1 : 2>(raw, 2 * i + 1, cc, p1);
                dct_row_pair<false>(p0, p1, i >= 2, pc[i]);
            }
            dct_cols_packed<false>(pc, x);
            asm volatile("; chroma dct done" ::: "memory");
        }
        int16_t* o = out + ((int64_t)b * 3 + comp) * 64;
#pragma unroll
        for (int k = 0; k < 64; ++k) o[k] = (int16_t)(qmul ? quant(x[kZZ[k]], qmul[comp * 64 + kZZ[k]]) : x[kZZ[k]]);
    }
}

hipError_t launch_coeffs(const uint8_t* rgb, int64_t stride, uint32_t mcu_w, uint32_t n_mcu, bool wrap,
                         const int32_t* qmul, int16_t* out, hipStream_t s) {
    (void)hipGetLastError();
    const dim3 grid((n_mcu + 63u) / 64u);
    if (wrap)
        hipLaunchKernelGGL(k_coeffs<true>, grid, dim3(64), 0, s, rgb, stride, mcu_w, n_mcu, qmul, out);
    else
        hipLaunchKernelGGL(k_coeffs<false>, grid, dim3(64), 0, s, rgb, stride, mcu_w, n_mcu, qmul, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Workspace of jpegenc_encode_batch_noalloc: one array per device, part of the code object, held for as long
// as the library stays loaded.  On the host the array's name is only a handle for hipGetSymbolAddress; what
// kernels, memsets and copies need is the device address that static_workspace() looks up.
// ------------------------------------------------------------------------------------------------
static __device__ __align__(256) uint8_t scratch[kStaticWsBytes];

hipError_t static_workspace(void** ptr) {
    return hipGetSymbolAddress(ptr, HIP_SYMBOL(scratch));  // the address on the current device
}

#endif  // !JPEGENC_TU_QUANT

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// This file is compiled twice.  The variants of k_encode_chunks with a quantiser are built in a translation
// unit of their own (-DJPEGENC_TU_QUANT) with the scheduler's max-ILP strategy: there it removes their spills
// (9 % faster); of the variants for the all-ones tables only the common one with small regions joins them there.
// (Every launcher first takes any error an earlier, unrelated HIP call has left pending: what it returns is
// then the outcome of its own launch.)
template <int ALIGN, bool QUANT, int REGION, bool FUSED = false, int WAVES = 1, int LPM = 1, bool COEFFS = false>
static hipError_t launch_encode_b(const EncodeParams& p, dim3 grid, hipStream_t s) {
    (void)hipGetLastError();
    grid.x = (grid.x + WAVES - 1) / WAVES;  // WAVES chunks per workgroup
    if (COEFFS || (p.flags & 1u))  // (the colour rounder's switch means nothing to the coefficient-fed variant)
        hipLaunchKernelGGL((k_encode_chunks<ALIGN, QUANT, true, REGION, FUSED, WAVES, LPM, COEFFS>), grid, dim3(64 * WAVES), 0, s, p);
    else
        hipLaunchKernelGGL((k_encode_chunks<ALIGN, QUANT, COEFFS, REGION, FUSED, WAVES, LPM, COEFFS>), grid, dim3(64 * WAVES), 0, s, p);
    return hipGetLastError();
}

#ifdef JPEGENC_TU_QUANT
hipError_t launch_encode_quant(const EncodeParams& p, dim3 grid, int align, hipStream_t s) {
    if (align >= 8) return launch_encode_b<8, true, kRegionBasic>(p, grid, s);
    if (align >= 4) return launch_encode_b<4, true, kRegionBasic>(p, grid, s);
    return launch_encode_b<1, true, kRegionBasic>(p, grid, s);
}
// The common variant with small regions is built here as well: since its coder joins symbols and reads coefficient
// pairs it has registers to spare, and the max-ILP strategy makes it 1.6 % faster (the large-region variant loses
// 15 % under it on uniform noise, the single-launch variants gain nothing: they stay with the default strategy).
hipError_t launch_encode_small(const EncodeParams& p, dim3 grid, hipStream_t s) {
    return launch_encode_b<8, false, kRegionSmall, false, kSmallWaves>(p, grid, s);
}
hipError_t launch_encode_short(const EncodeParams& p, dim3 grid, hipStream_t s) {
    return launch_encode_b<8, false, kRegionShort, false, kSmallWaves>(p, grid, s);
}
#else
hipError_t launch_encode_quant(const EncodeParams& p, dim3 grid, int align, hipStream_t s);
hipError_t launch_encode_small(const EncodeParams& p, dim3 grid, hipStream_t s);
hipError_t launch_encode_short(const EncodeParams& p, dim3 grid, hipStream_t s);

hipError_t launch_encode(const EncodeParams& p_in, uint32_t n_images, int align, hipStream_t s) {
    EncodeParams p = p_in;
    p.n_images = n_images;
    const uint64_t chunks = (uint64_t)n_images * p.cpi;
    if (chunks == 0 || chunks > 0x7FFFFFFFull) return hipErrorInvalidValue;  // chunk ids are 31-bit
    // x = chunk, y and z = image (the grid's second and third dimensions end at 65535)
    const uint32_t ny = n_images < 65535u ? n_images : 65535u;
    const uint32_t nz = (n_images + 65534u) / 65535u;
    if (nz > 65535u) return hipErrorInvalidValue;
    const dim3 grid(p.cpi, ny, nz);
    if (p.coeffs) {
        // coefficient-fed checking variants: the coders of the small-region variant (symbols joined) or of the
        // large-region one (symbol by symbol)
        if (p.state || p.qmul) return hipErrorInvalidValue;
        return p.regions == kRegionsLarge ? launch_encode_b<8, false, kRegionLarge, false, kLargeWaves, 1, true>(p, grid, s)
                                          : launch_encode_b<8, false, kRegionSmall, false, kSmallWaves, 1, true>(p, grid, s);
    }
    if (p.state) {
        // single-launch form: only the common variant (a launch this small leaves most of the LDS unused anyway)
        if (p.qmul || align < 8 || chunks > kSingleLaunchMaxChunks) return hipErrorInvalidValue;
        if (p.lanes_per_mcu == 3) return launch_encode_b<8, false, kRegionFused, true, 1, 3>(p, grid, s);
        return launch_encode_b<8, false, kRegionFused, true>(p, grid, s);
    }
    if (p.qmul) return launch_encode_quant(p, grid, align, s);
    // large regions exist for the common variant only (8-byte-aligned rows, all-ones tables: quantised streams are short)
    if (align >= 8)
        return p.regions == kRegionsLarge   ? launch_encode_b<8, false, kRegionLarge, false, kLargeWaves>(p, grid, s)
               : p.regions == kRegionsShort ? launch_encode_short(p, grid, s)
                                            : launch_encode_small(p, grid, s);
    if (align >= 4) return launch_encode_b<4, false, kRegionBasic>(p, grid, s);
    return launch_encode_b<1, false, kRegionBasic>(p, grid, s);
}

hipError_t launch_scan(const ScanParams& p, uint32_t n_images, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_scan, dim3(n_images), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_stuff_count(const StuffParams& p, uint32_t n_images, hipStream_t s) {
    (void)hipGetLastError();
    const uint32_t n_chunks = n_images * p.cpi;
    const dim3 grid((n_chunks + kStuffWaves - 1) / kStuffWaves), block(64 * kStuffWaves);
    if (p.spec_depth >= kStuffDepth)
        hipLaunchKernelGGL((k_stuff_count<kStuffDepth>), grid, block, 0, s, p, n_chunks);
    else
        hipLaunchKernelGGL((k_stuff_count<1>), grid, block, 0, s, p, n_chunks);
    return hipGetLastError();
}

hipError_t launch_stuff_image(const StuffParams& p, uint32_t n_images, hipStream_t s) {
    (void)hipGetLastError();
    const dim3 grid(n_images), block(64 * kImageWaves);
    if (p.cpi <= kStuffImageLdsChunks) {  // every chunk's count has its word in the LDS: no barrier per round
        // fewer than 512 images: 16 waves per workgroup, so that 256 images still give every CU 16 waves
        if (n_images < 512) {
            if (p.spec_depth >= kStuffDepth)
                hipLaunchKernelGGL((k_stuff_image_flags<kStuffDepth, 16>), grid, dim3(64 * 16), 0, s, p);
            else
                hipLaunchKernelGGL((k_stuff_image_flags<1, 16>), grid, dim3(64 * 16), 0, s, p);
        } else if (p.spec_depth >= kStuffDepth) {
            hipLaunchKernelGGL((k_stuff_image_flags<kStuffDepth, kImageWaves>), grid, block, 0, s, p);
        } else {
            hipLaunchKernelGGL((k_stuff_image_flags<1, kImageWaves>), grid, block, 0, s, p);
        }
        return hipGetLastError();
    }
    if (p.spec_depth >= kStuffDepth)
        hipLaunchKernelGGL((k_stuff_image<kStuffDepth>), grid, block, 0, s, p);
    else
        hipLaunchKernelGGL((k_stuff_image<1>), grid, block, 0, s, p);
    return hipGetLastError();
}

hipError_t launch_stuff_write(const StuffParams& p, uint32_t n_images, hipStream_t s) {
    (void)hipGetLastError();
    const uint32_t n_chunks = n_images * p.cpi;
    const dim3 grid((n_chunks + kStuffWaves - 1) / kStuffWaves), block(64 * kStuffWaves);
    if (p.spec_depth >= kStuffDepth)
        hipLaunchKernelGGL((k_stuff_write<kStuffDepth>), grid, block, 0, s, p, n_chunks);
    else
        hipLaunchKernelGGL((k_stuff_write<1>), grid, block, 0, s, p, n_chunks);
    return hipGetLastError();
}

#endif  // JPEGENC_TU_QUANT

}  // namespace jpegenc
