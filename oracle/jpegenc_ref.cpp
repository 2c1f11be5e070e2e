// jpegenc_ref.cpp -- CPU ORACLE for the jpeg_top data path.  TEST INFRASTRUCTURE ONLY.
//
// See jpegenc_ref.h for the policy (P1 as-written arithmetic, P2 evident-intent plumbing) and for
// the statement that parity is unpinned by the reference itself.  Every function cites the RTL
// text it follows (paths relative to the reference tree).
//
// Notation: ashr = arithmetic shift right on two's complement; rhu(x,n) = (x >> n) + bit (n-1) of
// x ("take the bits above n, add the bit just below") = floor((x + 2^(n-1)) / 2^n);
// wrapN(x) = low N bits of x reinterpreted as signed.
#include "jpegenc_ref.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

inline int64_t wrapN(int64_t x, int n) {
    const uint64_t m = (uint64_t(1) << n) - 1;
    uint64_t u = uint64_t(x) & m;
    if (u >> (n - 1)) u |= ~m;
    return int64_t(u);
}
inline int64_t rhu(int64_t x, int n) { return (x >> n) + ((x >> (n - 1)) & 1); }

// ---- tables, built canonically from BITS/HUFFVAL (T.81 Annex C).  The RTL hard-codes the result:
// DC rtl/4_huffman/cr_huff.sv:1553-1578, AC :1581-1906, run/size map :1907-2082 (y_huff.sv and
// cb_huff.sv carry identical copies). --------------------------------------------------------------
const uint8_t kDcBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kAcBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcVal[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct Tables {
    uint16_t dc_code[12];
    uint8_t dc_len[12];
    // AC symbol (run<<4|size) -> code/len after the RTL's two-level lookup
    // Cr_AC[Cr_AC_run_code[sym]].  len 0 = the RTL never assigns this map entry.
    uint16_t ac_code[256];
    uint8_t ac_len[256];
    Tables() {
        int k = 0;
        unsigned code = 0;
        for (int len = 1; len <= 16; ++len) {
            for (int i = 0; i < kDcBits[len - 1]; ++i, ++k, ++code) {
                dc_code[k] = uint16_t(code);  // HUFFVAL for DC is 0..11 in order
                dc_len[k] = uint8_t(len);
            }
            code <<= 1;
        }
        std::memset(ac_code, 0, sizeof ac_code);
        std::memset(ac_len, 0, sizeof ac_len);
        uint16_t first_code = 0;
        uint8_t first_len = 0;
        k = 0;
        code = 0;
        for (int len = 1; len <= 16; ++len) {
            for (int i = 0; i < kAcBits[len - 1]; ++i, ++k, ++code) {
                if (k == 0) { first_code = uint16_t(code); first_len = uint8_t(len); }
                ac_code[kAcVal[k]] = uint16_t(code);
                ac_len[kAcVal[k]] = uint8_t(len);
            }
            code <<= 1;
        }
        // cr_huff.sv:2069-2082: map entries (run 1..14, size 0) are assigned table index 0.
        for (int run = 1; run <= 14; ++run) {
            ac_code[run << 4] = first_code;
            ac_len[run << 4] = first_len;
        }
    }
};
const Tables& tables() {
    static const Tables t;
    return t;
}

// DCT-II matrix * 2^14: rtl/2_dct/cb_dct.sv:141-157 (constants), :613-709 (row select by count),
// :712-808 (transposed select by count_of_copy); rtl/2_dct/y_dct.sv:41-45, 339-532.
const int kT[8][8] = {{5793, 5793, 5793, 5793, 5793, 5793, 5793, 5793},
                      {8035, 6811, 4551, 1598, -1598, -4551, -6811, -8035},
                      {7568, 3135, -3135, -7568, -7568, -3135, 3135, 7568},
                      {6811, -1598, -8035, -4551, 4551, 8035, 1598, -6811},
                      {5793, -5793, -5793, 5793, 5793, -5793, -5793, 5793},
                      {4551, -8035, 1598, 6811, -6811, -1598, 8035, -4551},
                      {3135, -7568, 7568, -3135, -3135, 7568, -7568, 3135},
                      {1598, -4551, 6811, -8035, 8035, -6811, 4551, -1598}};

// Standard zig-zag; the *_huff shift chain (cr_huff.sv:1053-1304) walks Y{row}{col} in this order.
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- bit sinks ---------------------------------------------------------------------------------
struct BitVecSink {  // one bit per byte, for unit tests
    uint8_t* out;
    size_t cap, n;
    bool ok;
    void put(uint32_t bits, int len) {
        for (int i = len - 1; i >= 0; --i) {
            if (n < cap) out[n] = (bits >> i) & 1; else ok = false;
            ++n;
        }
    }
};
struct ByteSink {  // MSB-first packed bytes
    std::vector<uint8_t>* v;
    uint64_t acc = 0;
    int nacc = 0;
    uint64_t total = 0;
    void put(uint32_t bits, int len) {
        if (len == 0) return;
        acc = (acc << len) | (bits & ((len >= 32) ? 0xffffffffu : ((1u << len) - 1)));
        nacc += len;
        total += uint64_t(len);
        while (nacc >= 8) {
            v->push_back(uint8_t(acc >> (nacc - 8)));
            nacc -= 8;
        }
    }
    void finish() {  // left-align the residual bits, zero fill
        if (nacc > 0) v->push_back(uint8_t((acc << (8 - nacc)) & 0xff));
        nacc = 0;
    }
};

// size rule of the AC path: y_huff.sv:1318-1372 / cr_huff.sv:1411-1486 look only at bits [9:0] of
// v (v >= 0) or of v-1 (v < 0): highest set bit, resp. highest clear bit, plus one.
inline int size10(int v) {
    unsigned n = unsigned(v < 0 ? v - 1 : v) & 0x3ff;
    if (v < 0) n = ~n & 0x3ff;
    int s = 0;
    while (n) { ++s; n >>= 1; }
    return s;
}
// DC size: cr_huff.sv:1325-1407, bits [10:0] of the 12-bit difference d (d >= 0) or d-1 (d < 0).
inline int size11(int d) {
    unsigned n = unsigned(d < 0 ? d - 1 : d) & 0x7ff;
    if (d < 0) n = ~n & 0x7ff;
    int s = 0;
    while (n) { ++s; n >>= 1; }
    return s;
}

template <class Sink>
void huff_block(const ref_config* cfg, int comp, const int32_t blk[64], int32_t* pred, Sink& sink) {
    const Tables& T = tables();
    int z[64];
    for (int k = 0; k < 64; ++k) z[k] = int(wrapN(blk[kZigzag[k]], 11));  // 11-bit ports, cr_huff.sv:18-25

    // DC: d = {c0[10],c0} - previous (12 bit), previous <- sign-extended c0; cr_huff.sv:1317-1320,
    // :713-716.  The predictor is cleared only by rst (cr_huff.sv:710).
    int d = int(wrapN(int64_t(z[0]) - int64_t(*pred), 12));
    *pred = z[0];
    int s = size11(d);
    // amplitude = low `size` bits of d (d >= 0) or d-1 (d < 0): cr_huff.sv:670, :612-613, :572.
    uint32_t amp = uint32_t(d < 0 ? d - 1 : d) & ((1u << s) - 1);
    sink.put(T.dc_code[s], T.dc_len[s]);  // cr_huff.sv:610-611
    sink.put(amp, s);

    // group-of-8 all-zero flags over the zig-zag order: cr_huff.sv:190-204 (third..eighth = g 2..7)
    bool G[8];
    for (int g = 0; g < 8; ++g) {
        G[g] = true;
        for (int i = 0; i < 8; ++i) G[g] = G[g] && (z[8 * g + i] == 0);
    }
    const bool all2 = G[2] && G[3] && G[4] && G[5] && G[6] && G[7];
    const bool all4 = G[4] && G[5] && G[6] && G[7];
    const bool all6 = G[6] && G[7];
    auto eob = [&](int k) -> bool {
        // end_of_block as seen by the coefficient at zig-zag index k: it is sampled when
        // block_counter == k + 2 (zrl pipeline depth: cr_huff.sv:656, :634, :596, :595, :558;
        // end_of_block register :213-231).
        const int bc = k + 2;
        if (bc < 32) return all2;
        if (bc < 48) return all4;
        if (comp == 1 && cfg->quirk_cb_6d64) {
            // cb_huff.sv:238,241: `block_counter <= 6'd64` / `> 6'd64`; the 6-bit literal 64 is 0,
            // so for bc >= 48 the last branch always fires: end_of_block <= 1.
            return true;
        }
        if (bc <= 64) return all6;  // y_huff.sv:190, cr_huff.sv:225
        return true;                // y_huff.sv:192, cr_huff.sv:228
    };

    int run = 0;  // zero_run_length, 4 bits: cr_huff.sv:143, :675-685
    for (int k = 1; k <= 62; ++k) {
        const int v = z[k];
        if (v == 0) {
            // zero coefficient emits only the ZRL code, when the run register already holds 15 and
            // end_of_block is low: code_15_0, cr_huff.sv:558; output count :518-519.
            if (run == 15 && !eob(k)) sink.put(T.ac_code[0xF0], T.ac_len[0xF0]);
            run = (run + 1) & 15;
        } else {
            const int sz = size10(v);
            const int sym = (run << 4) | sz;  // code_index = {zrl_2, bits}: cr_huff.sv:176
            sink.put(T.ac_code[sym], T.ac_len[sym]);  // cr_huff.sv:629, :591-592
            sink.put(uint32_t(v < 0 ? v - 1 : v) & ((1u << sz) - 1), sz);  // cr_huff.sv:696, :651, :590
            run = 0;
        }
    }
    {
        // k = 63: the run is forced to 0 when the coefficient is zero (block_counter == 62,
        // cr_huff.sv:656) and the slot always emits (block_counter != 67, cr_huff.sv:518):
        // zero -> symbol 0x00 (EOB); non-zero -> ordinary (run,size) code and no EOB.
        const int v = z[63];
        if (v == 0) {
            sink.put(T.ac_code[0x00], T.ac_len[0x00]);
        } else {
            const int sz = size10(v);
            const int sym = (run << 4) | sz;
            sink.put(T.ac_code[sym], T.ac_len[sym]);
            sink.put(uint32_t(v < 0 ? v - 1 : v) & ((1u << sz) - 1), sz);
        }
    }
}

// Stuffing + word packing: the functional effect of ff_checker.sv (:96-112 FF detect, :115-179
// insertion, :265-360 EOF residual): 0x00 after every complete 0xFF byte, 32-bit words with the
// first byte in bits 31:24, residual left-aligned with its valid-bit count.
int stuff_and_pack(const std::vector<uint8_t>& bytes, uint64_t total_bits, uint32_t* words_out, size_t cap_words,
                   ref_result* res) {
    const uint64_t full_bytes = total_bits / 8;
    const int rem_bits = int(total_bits % 8);
    std::vector<uint8_t> st;
    st.reserve(bytes.size() + bytes.size() / 64 + 8);
    for (uint64_t i = 0; i < full_bytes; ++i) {
        st.push_back(bytes[i]);
        if (bytes[i] == 0xFF) st.push_back(0x00);
    }
    const uint64_t stuffed_bytes = st.size();
    if (rem_bits) st.push_back(bytes[full_bytes]);  // already left-aligned, zero-filled
    const uint64_t stuffed_bits = stuffed_bytes * 8 + uint64_t(rem_bits);
    const uint64_t n_full = stuffed_bits / 32;
    const int tail_bits = int(stuffed_bits % 32);
    while (st.size() % 4) st.push_back(0);
    const uint64_t n_words = n_full + (tail_bits ? 1 : 0);
    if (res) {
        std::memset(res, 0, sizeof *res);
        res->n_full_words = n_full;
        res->total_bits_unstuffed = total_bits;
        res->total_bytes_stuffed = stuffed_bytes;
        res->tail_bits = uint8_t(tail_bits);
        res->eof_partial_ready = tail_bits ? 1 : 0;
    }
    if (n_words > cap_words) return -2;
    for (uint64_t i = 0; i < n_words; ++i)
        words_out[i] = (uint32_t(st[4 * i]) << 24) | (uint32_t(st[4 * i + 1]) << 16) | (uint32_t(st[4 * i + 2]) << 8) |
                       uint32_t(st[4 * i + 3]);
    if (res) res->tail_word = tail_bits ? words_out[n_full] : 0;
    return 0;
}

void gather_block(const ref_config* cfg, const uint8_t* rgb, ptrdiff_t stride, int bx, int by, uint8_t comp[3][64]) {
    for (int r = 0; r < 8; ++r) {
        const uint8_t* row = rgb + ptrdiff_t(by * 8 + r) * stride + ptrdiff_t(bx) * 24;
        for (int c = 0; c < 8; ++c) {
            // jpeg_top.data_in = {B,G,R}: rgb2ycrcb.sv:21, :73-75
            const uint32_t px = uint32_t(row[3 * c]) | (uint32_t(row[3 * c + 1]) << 8) | (uint32_t(row[3 * c + 2]) << 16);
            uint8_t ycc[3];
            ref_rgb2ycbcr_px(cfg, px, ycc);
            comp[0][r * 8 + c] = ycc[0];
            comp[1][r * 8 + c] = ycc[1];
            comp[2][r * 8 + c] = ycc[2];
        }
    }
}

template <class Sink>
int encode_image(const ref_config* cfg, const uint8_t* rgb, int w, int h, ptrdiff_t stride, Sink& sink,
                 int16_t* coeffs_out) {
    if (w <= 0 || h <= 0 || (w % 8) || (h % 8)) return -1;
    int32_t pred[3] = {0, 0, 0};  // rst at the start of every image
    uint8_t comp[3][64];
    int32_t blk[64];
    size_t b = 0;
    // Feed order (P2, fifo_out.sv:195-197, :222-243, :257-266): blocks left->right, top->bottom,
    // and for each block the bits of Y, then Cb, then Cr.
    for (int by = 0; by < h / 8; ++by)
        for (int bx = 0; bx < w / 8; ++bx, ++b) {
            gather_block(cfg, rgb, stride, bx, by, comp);
            for (int c = 0; c < 3; ++c) {
                ref_block_coeffs(cfg, c, comp[c], blk);
                if (coeffs_out)
                    for (int k = 0; k < 64; ++k) coeffs_out[(b * 3 + c) * 64 + k] = int16_t(blk[kZigzag[k]]);
                huff_block(cfg, c, blk, &pred[c], sink);
            }
        }
    return 0;
}

struct NullSink {
    void put(uint32_t, int) {}
};

}  // namespace

extern "C" {

void ref_default_config(ref_config* cfg) {
    std::memset(cfg, 0, sizeof *cfg);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 64; ++i) cfg->qtab[c][i] = 1;  // y_quantizer.sv:16 `'{default: 64'd1}`
    cfg->quirk_rgb_wrap = 1;
    cfg->quirk_cb_6d64 = 1;
    cfg->quirk_ydct_zext = 0;
}

void ref_rgb2ycbcr_px(const ref_config* cfg, uint32_t px, uint8_t out[3]) {
    const uint32_t R = px & 0xff, G = (px >> 8) & 0xff, B = (px >> 16) & 0xff;  // rgb2ycrcb.sv:73-75
    // 22-bit sums, rgb2ycrcb.sv:77-91 (coefficients :29-39).  None can leave [0, 2^22).
    const uint32_t ys = 4899 * R + 9617 * G + 1868 * B;
    const uint32_t cbs = 2097152 - 2764 * R - 5428 * G + 8192 * B;
    const uint32_t crs = 2097152 + 8192 * R - 6860 * G - 1332 * B;
    const uint32_t sums[3] = {ys & 0x3fffff, cbs & 0x3fffff, crs & 0x3fffff};
    for (int i = 0; i < 3; ++i) {
        // rgb2ycrcb.sv:105-108: rounded = sum[21:14]; if (sum[13]) rounded = rounded + 1 -- in an
        // 8-bit variable (:53), so 255 + 1 wraps to 0 and the `> 255` clamp (:109-113) is dead (Q1).
        uint32_t r = (sums[i] >> 14) + ((sums[i] >> 13) & 1);
        if (cfg->quirk_rgb_wrap) r &= 0xff; else if (r > 255) r = 255;
        out[i] = uint8_t(r);
    }
}

void ref_dct8x8(const ref_config* cfg, const uint8_t x[64], int luma, int32_t F[64]) {
    // Pass 1 (per input row r, all k): A[r][k] = sum_c x[r][c]*T[k][c], minus 128*8*5793 for k = 0
    // only (cb_dct.sv:392-407, :544-553; y_dct.sv:142-161, :261-275); the k >= 1 accumulators run
    // across the whole block and the per-row value is the difference of successive 25-bit snapshots
    // (cb_dct.sv:556-610), which equals the row's own sum modulo 2^25.
    // Rounding: chroma B = wrap11(rhu(A,14)) (cb_dct.sv:826-835), luma B = wrap13(rhu(A,12))
    // (y_dct.sv:71, :549-551).
    int64_t Bm[8][8];
    for (int r = 0; r < 8; ++r)
        for (int k = 0; k < 8; ++k) {
            int64_t a = 0;
            for (int c = 0; c < 8; ++c) {
                int64_t t = kT[k][c];
                // Q6 (off by default): y_dct.sv:65-66,171-177 multiplies an unsigned byte by a
                // `logic signed [13:0]`; as written the operand would be zero-extended.
                if (luma && k >= 1 && cfg->quirk_ydct_zext && t < 0) t += 16384;
                a += int64_t(x[r * 8 + c]) * t;
            }
            if (k == 0) a -= 5932032;
            a = wrapN(a, 25);
            Bm[r][k] = luma ? wrapN(rhu(a, 12), 13) : wrapN(rhu(a, 14), 11);
        }
    // Pass 2: Z[k][j] = sum_r B[r][k]*T[j][r] (cb_dct.sv:198-231 products, :265-281 accumulate;
    // transposed select :712-808).  Chroma Z registers are 25 bits and F = Z[24:14] + Z[13]
    // (cb_dct.sv:73-80, :308-375); luma Z is 32 bits and F = Z[26:16] + Z[15] (y_dct.sv:81, :122).
    for (int k = 0; k < 8; ++k)
        for (int j = 0; j < 8; ++j) {
            int64_t zz = 0;
            for (int r = 0; r < 8; ++r) zz += Bm[r][k] * int64_t(kT[j][r]);
            if (luma) {
                zz = wrapN(zz, 32);
                F[k * 8 + j] = int32_t(wrapN(rhu(zz, 16), 11));
            } else {
                zz = wrapN(zz, 25);
                F[k * 8 + j] = int32_t(wrapN(rhu(zz, 14), 11));
            }
        }
}

void ref_quantize(const int32_t F[64], const uint16_t qtab[64], int32_t Q[64]) {
    for (int i = 0; i < 64; ++i) {
        // QM = 4096 / Q (0 when Q == 0): y_quantizer.sv:37-46.  Product kept in 23 bits (:27, :74),
        // Q = (Z_temp_1 + 2048) >> 12 truncated to 11 bits (:22, :106) -- bits [22:12] of the sum.
        const int64_t qm = qtab[i] ? 4096 / qtab[i] : 0;
        const int64_t prod = wrapN(int64_t(wrapN(F[i], 11)) * qm, 23);
        Q[i] = int32_t(wrapN((prod + 2048) >> 12, 11));
    }
}

void ref_block_coeffs(const ref_config* cfg, int comp, const uint8_t x[64], int32_t blk[64]) {
    int32_t F[64], Q[64];
    ref_dct8x8(cfg, x, comp == 0, F);
    if (cfg->quant_index_order) {
        uint16_t qt[64];  // F[k*8+j] has k horizontal: transpose the row-major (vertical, horizontal) table
        for (int k = 0; k < 8; ++k)
            for (int j = 0; j < 8; ++j) qt[k * 8 + j] = cfg->qtab[comp][j * 8 + k];
        ref_quantize(F, qt, Q);
    } else {
        ref_quantize(F, cfg->qtab[comp], Q);
    }
    // Wrapper transpose yd_q_h.sv:120-135 (cbd_q_h.sv:114-128, crd_q_h.sv:112-126):
    // huff.Y{a}{b} = Q{b}{a}, so blk[a][b] is (vertical a, horizontal b).
    for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b) blk[a * 8 + b] = Q[b * 8 + a];
}

long ref_huff_block(const ref_config* cfg, int comp, const int32_t blk[64], int32_t* pred, uint8_t* bits_out,
                    size_t cap) {
    BitVecSink s{bits_out, cap, 0, true};
    huff_block(cfg, comp, blk, pred, s);
    return s.ok ? long(s.n) : -1;
}

size_t ref_max_words(int w, int h) {
    // <= 22 + 63*26 = 1660 bits per component block, x3 per MCU, x2 for stuffing, +1 tail word.
    const uint64_t mcus = uint64_t(w / 8) * uint64_t(h / 8);
    return size_t((mcus * 4980 * 2 + 31) / 32 + 1);
}

int ref_encode_rgb8_unstuffed(const ref_config* cfg, const uint8_t* rgb, int w, int h, ptrdiff_t stride,
                              uint8_t* bytes_out, size_t cap_bytes, uint64_t* total_bits) {
    std::vector<uint8_t> bytes;
    ByteSink sink;
    sink.v = &bytes;
    const int rc = encode_image(cfg, rgb, w, h, stride, sink, nullptr);
    if (rc) return rc;
    sink.finish();
    if (total_bits) *total_bits = sink.total;
    if (bytes.size() > cap_bytes) return -2;
    std::memcpy(bytes_out, bytes.data(), bytes.size());
    return 0;
}

int ref_encode_rgb8(const ref_config* cfg, const uint8_t* rgb, int w, int h, ptrdiff_t stride, uint32_t* words_out,
                    size_t cap_words, ref_result* res) {
    std::vector<uint8_t> bytes;
    bytes.reserve(size_t(w) * size_t(h) * 2);
    ByteSink sink;
    sink.v = &bytes;
    const int rc = encode_image(cfg, rgb, w, h, stride, sink, nullptr);
    if (rc) return rc;
    sink.finish();
    return stuff_and_pack(bytes, sink.total, words_out, cap_words, res);
}

int ref_image_coeffs(const ref_config* cfg, const uint8_t* rgb, int w, int h, ptrdiff_t stride, int16_t* out) {
    NullSink s;
    return encode_image(cfg, rgb, w, h, stride, s, out);
}

int ref_encode_coeffs(const ref_config* cfg, const int16_t* coeffs, size_t n_mcu, uint32_t* words_out,
                      size_t cap_words, ref_result* res) {
    // The three *_huff modules fed directly at their coefficient ports (cr_huff.sv:18-25: 11-bit ports, the
    // values are wrapped to 11 bits), blocks in feed order, Y then Cb then Cr (fifo_out.sv:195-197, :222-243),
    // DC predictors cleared by rst before the first block (cr_huff.sv:710), then ff_checker.
    std::vector<uint8_t> bytes;
    ByteSink sink;
    sink.v = &bytes;
    int32_t pred[3] = {0, 0, 0};
    int32_t blk[64];
    for (size_t b = 0; b < n_mcu; ++b)
        for (int c = 0; c < 3; ++c) {
            for (int k = 0; k < 64; ++k) blk[kZigzag[k]] = coeffs[(b * 3 + size_t(c)) * 64 + size_t(k)];
            huff_block(cfg, c, blk, &pred[c], sink);
        }
    sink.finish();
    return stuff_and_pack(bytes, sink.total, words_out, cap_words, res);
}

// ---- port-level shim (jpeg_top.sv:18-28) --------------------------------------------------------
struct ref_port {
    ref_config cfg;
    std::vector<uint8_t> px;  // R,G,B per pushed pixel, feed order
    std::vector<uint32_t> words;
};

ref_port* ref_port_open(const ref_config* cfg) {
    ref_port* p = new ref_port;
    p->cfg = *cfg;
    return p;
}
void ref_port_reset(ref_port* p) {
    p->px.clear();
    p->words.clear();
}
void ref_port_push_pixel(ref_port* p, uint32_t d) {
    p->px.push_back(uint8_t(d & 0xff));
    p->px.push_back(uint8_t((d >> 8) & 0xff));
    p->px.push_back(uint8_t((d >> 16) & 0xff));
}
size_t ref_port_end_of_file(ref_port* p, ref_result* res) {
    // 64 consecutive samples form one 8x8 block, row-major (cb_dct.sv:478-541): the pushed stream
    // is a raster image 8 pixels wide and 8*nblocks tall.  A trailing partial block never reaches
    // the DCT's output_enable and is dropped.
    const size_t nblk = p->px.size() / (64 * 3);
    p->words.assign(ref_max_words(8, int(nblk * 8)) + 1, 0);
    ref_result r;
    std::memset(&r, 0, sizeof r);
    if (nblk) ref_encode_rgb8(&p->cfg, p->px.data(), 8, int(nblk * 8), 24, p->words.data(), p->words.size(), &r);
    if (res) *res = r;
    const size_t n = size_t(r.n_full_words) + (r.tail_bits ? 1 : 0);
    p->words.resize(n);
    p->px.clear();
    return n;
}
const uint32_t* ref_port_words(const ref_port* p) { return p->words.data(); }
void ref_port_close(ref_port* p) { delete p; }

}  // extern "C"
