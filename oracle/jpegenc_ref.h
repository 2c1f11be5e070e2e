/* jpegenc_ref.h -- CPU ORACLE for the jpeg_top data path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is an untimed CPU restatement of the reference RTL (navaalnoshi/JPEG-Encoder,
 * rtl/0_jpeg_top/jpeg_top.sv and everything below it), written from the RTL text.  It is the
 * checker the HIP path is compared against; it is never linked into, loaded by or called from
 * the product library (libjpegenc.so).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * PARITY UNPINNED: the reference ships no testbench, vectors or fixtures, the RTL as committed
 * does not elaborate, and no HDL simulator exists in the build image.  The oracle is therefore
 * pinned only by (a) the RTL text, (b) an independently written numpy restatement
 * (tests/ref_numpy.py) and (c) hand-derived known answers (tests/test_oracle_kat.py).
 *
 * Policy.  P1: arithmetic that is a pure function of the image follows the text as written,
 * quirks included (Q1 8-bit wrap in the colour rounder, Q2 coarse end-of-block test, Q3 the
 * 6'd64 literal in cb_huff, Q4 table pairing, Q5 size-0 symbol for -1024).  P2: behaviour that
 * depends on cycle timing or on text that cannot compile is replaced by its evident intent
 * (pipeline alignment, FIFO plumbing, byte stuffing, EOF tail; luma DCT uses signed products).
 */
#ifndef JPEGENC_REF_H
#define JPEGENC_REF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ref_config {
    /* Quantisation tables for Y, Cb, Cr.  Entry [i*8+j] scales DCT output Z{i+1}{j+1}
     * (i = horizontal frequency), as y_quantizer.sv:37-46 indexes Y_Q_MATRIX[i*8+j]. */
    uint16_t qtab[3][64];
    uint8_t quirk_rgb_wrap;  /* Q1, default 1 (as written) */
    uint8_t quirk_cb_6d64;   /* Q3, default 1 (as written) */
    uint8_t quirk_ydct_zext; /* Q6, default 0 (repaired)   */
    uint8_t quant_index_order; /* 0 (default): entry [i*8+j] scales Z{i+1}{j+1}, i horizontal; 1: entry
                                  [a*8+b] scales vertical frequency a, horizontal b (the hookup of the
                                  quantiser wrappers, yd_q_h.sv:90-117, leaves it open) */
} ref_config;

typedef struct ref_result {
    uint64_t n_full_words;         /* complete stuffed 32-bit words */
    uint64_t total_bits_unstuffed; /* Huffman bits before stuffing */
    uint64_t total_bytes_stuffed;  /* complete stuffed bytes (excludes a final partial byte) */
    uint32_t tail_word;            /* residual bits, left-aligned, zero-filled */
    uint8_t tail_bits;             /* end_of_file_bitstream_count, 0..31 */
    uint8_t eof_partial_ready;     /* tail_bits != 0 */
    uint8_t pad[2];
} ref_result;

void ref_default_config(ref_config* cfg);

/* rgb2ycrcb.sv: data_in24 = {B[23:16],G[15:8],R[7:0]} -> out = {Y, Cb, Cr}. */
void ref_rgb2ycbcr_px(const ref_config* cfg, uint32_t data_in24, uint8_t out[3]);

/* y_dct.sv / cb_dct.sv: x[r*8+c] -> F[k*8+j] = Z{k+1}{j+1}_final (k horizontal, j vertical). */
void ref_dct8x8(const ref_config* cfg, const uint8_t x[64], int luma, int32_t F[64]);

/* *_quantizer.sv: element-wise, Q[i] from F[i] and qtab[i]. */
void ref_quantize(const int32_t F[64], const uint16_t qtab[64], int32_t Q[64]);

/* Full per-block coefficient path: 64 samples of one component -> Huffman input order
 * blk[a*8+b] = Y{a+1}{b+1} (a = vertical frequency), i.e. after the wrapper transpose. */
void ref_block_coeffs(const ref_config* cfg, int comp, const uint8_t x[64], int32_t blk[64]);

/* *_huff.sv: codes one component block given in Huffman input order; appends bits (one bit per
 * byte, values 0/1) to bits_out (capacity cap) and returns the number of bits, or -1 if cap is
 * too small.  *pred is the DC predictor (sign-extended previous DC), updated. */
long ref_huff_block(const ref_config* cfg, int comp, const int32_t blk[64], int32_t* pred,
                    uint8_t* bits_out, size_t cap);

/* Worst-case stuffed 32-bit words for a w x h image (incl. tail word). */
size_t ref_max_words(int w, int h);

/* Whole image, raster R,G,B bytes, w and h multiples of 8, blocks fed left->right, top->bottom.
 * words_out receives the full words followed by the tail word if tail_bits != 0.
 * Returns 0, or -1 bad dims, -2 buffer too small. */
int ref_encode_rgb8(const ref_config* cfg, const uint8_t* rgb, int w, int h, ptrdiff_t stride,
                    uint32_t* words_out, size_t cap_words, ref_result* res);

/* Same stream, but also hands back the unstuffed bytes (for stage-level tests). */
int ref_encode_rgb8_unstuffed(const ref_config* cfg, const uint8_t* rgb, int w, int h,
                              ptrdiff_t stride, uint8_t* bytes_out, size_t cap_bytes,
                              uint64_t* total_bits);

/* Coefficients only: out[(b*3+comp)*64 + zz] in zig-zag order for every MCU b. */
int ref_image_coeffs(const ref_config* cfg, const uint8_t* rgb, int w, int h, ptrdiff_t stride,
                     int16_t* out);

/* The stream from coefficients: coeffs[(b*3+comp)*64 + zz] (the layout ref_image_coeffs writes) are put on
 * the coefficient ports of y_huff / cb_huff / cr_huff for MCUs b = 0 .. n_mcu-1 after rst, and the bits go
 * through interleave and stuffing as in ref_encode_rgb8.  Values are wrapped to the ports' 11 bits. */
int ref_encode_coeffs(const ref_config* cfg, const int16_t* coeffs, size_t n_mcu, uint32_t* words_out,
                      size_t cap_words, ref_result* res);

/* Port-level shim shaped like jpeg_top: one pixel per call, 64 consecutive pixels = one block. */
typedef struct ref_port ref_port;
ref_port* ref_port_open(const ref_config* cfg);
void ref_port_reset(ref_port* p);                         /* rst */
void ref_port_push_pixel(ref_port* p, uint32_t data_in24); /* enable + data_in */
/* end_of_file_signal: flush; returns number of words available (full + tail). */
size_t ref_port_end_of_file(ref_port* p, ref_result* res);
const uint32_t* ref_port_words(const ref_port* p);
void ref_port_close(ref_port* p);

#ifdef __cplusplus
}
#endif
#endif
