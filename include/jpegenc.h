/* jpegenc.h -- C ABI of libjpegenc.so: the jpeg_top data path on MI355X (gfx950).
 *
 * The reference design's only interface is the port list of its top module
 *   rtl/0_jpeg_top/jpeg_top.sv:18-28
 *     clk, rst, end_of_file_signal, enable, data_in[23:0]
 *     -> JPEG_bitstream[31:0], data_ready, end_of_file_bitstream_count[4:0], eof_data_partial_ready
 * Each entry point below names the port(s) it stands for.  The library is the GPU path only:
 * there is no CPU fallback, and every compute entry point fails with JPEGENC_E_HIP when no
 * gfx950 device is usable.
 *
 * Stream format (identical to what jpeg_top presents on JPEG_bitstream): for every 8x8 block in
 * feed order the Huffman bits of Y, Cb, Cr; MSB first; 0x00 stuffed after every complete 0xFF byte;
 * cut into 32-bit words whose bit 31 is the first bit.  No JPEG markers.  The residual bits after
 * the last full word are reported left-aligned in `tail_word` with their count in `tail_bits`.
 *
 * Threading: a context is bound to one device and one stream and is not thread-safe; distinct
 * contexts are independent.  All calls return 0 or a negative JPEGENC_E_* code; nothing throws.
 */
#ifndef JPEGENC_H
#define JPEGENC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JPEGENC_OK 0
#define JPEGENC_E_BAD_ARG (-1)   /* null pointer, w/h not positive multiples of 8, bad stride */
#define JPEGENC_E_TOO_SMALL (-2) /* output capacity too small; result reports the size needed */
#define JPEGENC_E_HIP (-3)       /* HIP runtime error or no usable GPU; see jpegenc_last_error */
#define JPEGENC_E_UNSUPPORTED (-4)
#define JPEGENC_E_NOMEM (-5)
#define JPEGENC_E_STATE (-6) /* calls out of order: push/end_of_file protocol, segment calls */
#define JPEGENC_E_LIMIT (-7) /* jpegenc_encode_batch_noalloc: the call exceeds JPEGENC_NOALLOC_MAX_CHUNKS */
#define JPEGENC_E_AGAIN (-8) /* per-image status under JPEGENC_SLOTS_ADAPTIVE: repeat the call (see jpegenc_set_slot_mode) */

/* flags for the encode calls */
#define JPEGENC_HOST_PTRS 0u   /* rgb / words_out / results are host memory (copied, call blocks) */
#define JPEGENC_PAD_EDGES 4u   /* extension, not in jpeg_top (which has no edge handling): w and h need not be
                                  multiples of 8; the last column and row are repeated up to the next multiples
                                  and the padded image is coded (jpegenc_encode_rgb8 / _batch only) */
#define JPEGENC_DEVICE_PTRS 1u /* all three are device memory on the context's device; the call
                                  only enqueues work on the context's stream -- except when the
                                  context's workspace has to grow for it: then it allocates, which
                                  synchronises the device (see jpegenc_reserve) */

typedef struct jpegenc_ctx jpegenc_ctx;

/* Configuration = the SystemVerilog parameters of the design plus the as-written/repaired
 * switches for the arithmetic accidents in the RTL text. */
typedef struct jpegenc_config {
    /* Y_Q_MATRIX / CB_Q_MATRIX / CR_Q_MATRIX (rtl/3_quantization/y_quantizer.sv:15-17,
     * cb_quantizer.sv:19-21, cr_quantizer.sv:11-13).  Entry [i*8+j] scales DCT output Z{i+1}{j+1}
     * (i = horizontal frequency).  Default: all ones. */
    uint16_t qtab[3][64];
    uint8_t quirk_rgb_wrap;  /* 1 = 8-bit wrap in the colour rounder as written (rgb2ycrcb.sv:105-113) */
    uint8_t quirk_cb_6d64;   /* 1 = cb_huff's 6'd64 literal as written (cb_huff.sv:238,241) */
    uint8_t quirk_ydct_zext; /* must be 0: the luma zero-extension accident (y_dct.sv:65-66,171-177)
                                is repaired; 1 is rejected with JPEGENC_E_UNSUPPORTED */
    uint8_t quant_index_order; /* the quantiser wrappers hook scalar ports to array ports (yd_q_h.sv:90-117 vs
                                  y_quantizer.sv:21-22), so which table entry meets which frequency is
                                  not fixed by the text.  0 (default) = the index the quantiser itself
                                  uses, entry [i*8+j] for Z{i+1}{j+1}, i horizontal; 1 = the conventional
                                  row-major order, entry [a*8+b] for vertical frequency a, horizontal b.
                                  Same stream for symmetric tables, and for the all-ones default. */
    int32_t device;          /* HIP device ordinal */
} jpegenc_config;

/* What jpeg_top signals at end of file, per image (jpeg_top.sv:24-27), plus sizes. */
typedef struct jpegenc_result {
    uint64_t n_full_words;         /* number of data_ready pulses: complete 32-bit words */
    uint64_t total_bits_unstuffed; /* Huffman bits before stuffing */
    uint64_t total_bytes_stuffed;  /* complete bytes after stuffing */
    uint32_t tail_word;            /* JPEG_bitstream at eof_data_partial_ready: residual, left-aligned */
    uint8_t tail_bits;             /* end_of_file_bitstream_count[4:0] */
    uint8_t eof_partial_ready;     /* eof_data_partial_ready (tail_bits != 0) */
    int16_t status;                /* JPEGENC_OK or JPEGENC_E_TOO_SMALL for this image */
} jpegenc_result;

/* Fills cfg with the RTL defaults (all-ones tables, quirks as written, device 0). */
int jpegenc_default_config(jpegenc_config* cfg);

/* Creates a context on cfg->device with its own stream.  Stands for instantiating jpeg_top. */
int jpegenc_create(jpegenc_ctx** out, const jpegenc_config* cfg);
int jpegenc_destroy(jpegenc_ctx* ctx);

/* Makes the context launch on a caller-owned hipStream_t (pass NULL to go back to its own, which is a
 * non-blocking stream: not ordered with the device's default stream.  To run on the default stream pass
 * hipStreamLegacy, not NULL).  All calls of a context share its workspace, so work it has queued on the
 * stream it leaves is ordered before anything later queued on the new one (the new stream is made to
 * wait for an event); calls of one context therefore never overlap, whatever streams they run on --
 * use one context per stream for concurrency. */
int jpegenc_set_stream(jpegenc_ctx* ctx, void* hip_stream);
/* Blocks until everything enqueued by this context has finished. */
int jpegenc_synchronize(jpegenc_ctx* ctx);

/* Upper bound on words (full words + tail word) for a w x h image:
 * <= 22 + 63*26 bits per component block, doubled for stuffing. */
size_t jpegenc_max_words(int w, int h);

/* One image.  Replaces: rst, then `enable` with one pixel per clock on data_in[23:0] =
 * {B,G,R} (rgb2ycrcb.sv:21) for every 8x8 block left->right, top->bottom, then
 * end_of_file_signal (jpeg_top.sv:20-23).  `rgb` is a raster of R,G,B bytes, `stride` bytes per
 * row; w and h must be multiples of 8 (the design has no edge handling: a block is 64 consecutive
 * samples, cb_dct.sv:478-541).  words_out receives the JPEG_bitstream words in data_ready order,
 * followed by the tail word when tail_bits != 0 (jpeg_top.sv:24-27). */
int jpegenc_encode_rgb8(jpegenc_ctx* ctx, const uint8_t* rgb, int w, int h, ptrdiff_t stride, uint32_t flags,
                        uint32_t* words_out, size_t cap_words, jpegenc_result* result);

/* n independent images of equal size (each starts from rst).  Image i starts at
 * rgb + i*image_stride and writes to words_out + i*cap_words_per_image; results[i] is filled.
 * An image whose stream does not fit gets status JPEGENC_E_TOO_SMALL (sizes still reported, no
 * write beyond its slot) and the call returns JPEGENC_E_TOO_SMALL after finishing the others
 * (host-pointer mode) or JPEGENC_OK (device-pointer mode: check results[i].status). */
int jpegenc_encode_batch(jpegenc_ctx* ctx, const uint8_t* rgb, int n_images, int w, int h, ptrdiff_t stride,
                         ptrdiff_t image_stride, uint32_t flags, uint32_t* words_out, size_t cap_words_per_image,
                         jpegenc_result* results);

/* jpegenc_encode_batch for callers that cannot allocate device memory: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference (hardware has no
 * allocator).  Device pointers only (rgb, words_out and results are device memory on the context's device);
 * same output and the same per-image status as jpegenc_encode_batch with JPEGENC_DEVICE_PTRS; only
 * enqueues work on the context's stream, always as the separate kernels (JPEGENC_PIPELINE_SPLIT: the
 * single-launch form tags every launch, so a captured launch could not be replayed; these can).  It calls no hipMalloc/hipFree and no
 * other allocating routine (kernel timing, which creates events, is off for these calls): the intermediate
 * chunk streams live in a fixed array inside the library's code object, JPEGENC_NOALLOC_MAX_CHUNKS chunks
 * of 64 blocks (81.9 MB that every GPU the library runs on holds until the process ends).  A call needs
 * n_images * ceil((w/8) * (h/8) / 64) chunks -- one 3840x2160 frame is 2025, four 1080p frames are 2028 --
 * and is refused with JPEGENC_E_LIMIT (nothing enqueued) beyond that.  w and h must be multiples of 8
 * (no JPEGENC_PAD_EDGES: padding needs a staging buffer).
 * All calls on one GPU, from every context, share that GPU's array: calls that overlap in time on one GPU
 * (two streams, two contexts, two threads) race.  The array keeps the last call's contents; the kernels
 * rewrite every part they later read, so nothing has to be cleared between calls.
 * Captured into a HIP graph the call leaves kernel launches only (no event, no copy): the graph refers to the
 * context's tables, so it has to be destroyed before jpegenc_destroy, and it keeps the staging-region size the
 * context stood at when it was captured (jpegenc_set_region_mode; no statistics are collected by these calls). */
#define JPEGENC_NOALLOC_MAX_CHUNKS 2048
int jpegenc_encode_batch_noalloc(jpegenc_ctx* ctx, const uint8_t* rgb, int n_images, int w, int h, ptrdiff_t stride,
                                 ptrdiff_t image_stride, uint32_t* words_out, size_t cap_words_per_image,
                                 jpegenc_result* results);

/* The stage before the entropy coders, for checking it on its own: the coefficients as the three Huffman
 * coders receive them -- after rgb2ycrcb.sv:77-135, the two DCT passes (y_dct.sv, cb_dct.sv, cr_dct.sv), the
 * quantisers (y_quantizer.sv:58-106 and twins) and the wrapper transpose (yd_q_h.sv:120-135), i.e. the nets
 * between *_quantizer and *_huff that a testbench would probe.  coeffs_out[((by * (w/8) + bx) * 3 + c) * 64 +
 * k]: 8x8 block (bx, by), component c (0 Y, 1 Cb, 2 Cr), k = position in the order the coder reads them
 * (zig-zag, cr_huff.sv:1053-1304; k = 0 is the DC value, not yet its difference), 11-bit signed values.  w and h multiples of 8; flags: JPEGENC_HOST_PTRS (blocks) or JPEGENC_DEVICE_PTRS
 * (enqueues on the context's stream). */
int jpegenc_image_coeffs(jpegenc_ctx* ctx, const uint8_t* rgb, int w, int h, ptrdiff_t stride, uint32_t flags,
                         int16_t* coeffs_out);

/* The other half of that probe: the stream from coefficients.  Puts coeffs (the layout jpegenc_image_coeffs
 * writes: [(by * (w/8) + bx) * 3 + c][k], k in the coders' reading order; values are taken modulo the ports'
 * 11 bits, cr_huff.sv:18-25) on the coefficient ports of y_huff / cb_huff / cr_huff for every 8x8 block of a
 * w x h image after rst, and runs everything from there on -- the three coders (cr_huff.sv:176-231, 518-716,
 * 1317-1486 and the y/cb twins), the interleave (fifo_out.sv:222-266) and the stuffing (ff_checker.sv:96-360)
 * -- as jpegenc_encode_rgb8 does, with the same device code from the coders on.  This is how the coders' rules
 * are checked on values that no 8-bit image produces: zero runs ending at positions 30/31 and 46/47
 * (cr_huff.sv:213-231, cb_huff.sv:238-241), a last coefficient that is not zero, DC differences of 12 bits, and
 * the AC value -1024, whose size by the rule on bits [9:0] is 0 (cr_huff.sv:1411-1486).  The quantisation
 * tables of the context play no part (the values are what the quantisers would deliver).  flags:
 * JPEGENC_HOST_PTRS (blocks) or JPEGENC_DEVICE_PTRS (coeffs 16-byte aligned; enqueues on the context's stream). */
int jpegenc_encode_coeffs(jpegenc_ctx* ctx, const int16_t* coeffs, int w, int h, uint32_t flags, uint32_t* words_out,
                          size_t cap_words, jpegenc_result* result);

/* One image cut into n_stripes horizontal stripes of stripe_rows rows (a multiple of 8;
 * n_stripes*stripe_rows == h).  Every stripe is encoded exactly as jpeg_top would encode that
 * w x stripe_rows sub-image after rst -- the restart-interval view of a huge image.  Equivalent to
 * jpegenc_encode_batch with image_stride = stripe_rows*stride. */
int jpegenc_encode_stripes(jpegenc_ctx* ctx, const uint8_t* rgb, int w, int h, ptrdiff_t stride, int n_stripes,
                           uint32_t flags, uint32_t* words_out, size_t cap_words_per_stripe,
                           jpegenc_result* results);

/* One stream from consecutive segments.  No counterpart in jpeg_top, which codes an image in one
 * run: these three calls let several GPUs each code a horizontal band (whole block rows) of one
 * image and still produce, once the bands' words are ORed together at their positions, exactly the
 * words jpegenc_encode_rgb8 gives for the whole image (jpeg_top.sv:24-27 for the whole feed).  What
 * couples the bands is what the RTL carries from block to block: the DC predictors
 * (cr_huff.sv:591-596), the running bit offset (fifo_out.sv:222-243) and the stuffed byte count
 * (ff_checker.sv:96-179).  Per band, in this order:
 *   1. jpegenc_segment_encode: codes the band.  With JPEGENC_SEG_CONTINUES the 8 pixel rows above
 *      `rgb` (rgb - 8*stride ...) must be readable: their last 8x8 block supplies the DC
 *      predictors.  Returns the band's bit count and its last 8 bits.
 *   2. jpegenc_segment_count(bit_offset = bits of all bands before this one, prev_tail8 = the last
 *      8 bits of the band before it, 0 for the first): 0xFF bytes this band owns at that phase.
 *   3. jpegenc_segment_write(ff_offset = 0xFF bytes of all bands before this one, is_last): writes
 *      the band's words; words_out[i] is word first_word + i of the whole stream, n_words of them;
 *      bytes of the first and last word that belong to the neighbouring bands are zero.  The last
 *      band also fills `result` (host memory) with the whole stream's figures (its tail_word holds
 *      this band's bytes only: if the band before it shares that word, take it from the ORed words).
 * `flags`: JPEGENC_HOST_PTRS/JPEGENC_DEVICE_PTRS for rgb resp. words_out, ORed with
 * JPEGENC_SEG_CONTINUES in step 1 for every band but the first.  Every one of the three calls ends with
 * a stream synchronisation and small device-to-host copies (the counts it returns), and step 1 may
 * allocate.  The band's chunk streams stay in the context's workspace between the steps: any other
 * encode call on the context in between discards them, and steps 2 and 3 then return JPEGENC_E_STATE. */
#define JPEGENC_SEG_CONTINUES 2u
int jpegenc_segment_encode(jpegenc_ctx* ctx, const uint8_t* rgb, int w, int rows, ptrdiff_t stride, uint32_t flags,
                           uint64_t* bits, uint32_t* tail8);
int jpegenc_segment_count(jpegenc_ctx* ctx, uint64_t bit_offset, uint32_t prev_tail8, uint64_t* ff_bytes);
int jpegenc_segment_write(jpegenc_ctx* ctx, uint64_t ff_offset, int is_last, uint32_t flags, uint32_t* words_out,
                          size_t cap_words, uint64_t* first_word, uint64_t* n_words, jpegenc_result* result);

/* The same three steps for callers that keep the bands' figures on the GPU (several GPUs exchanging them with
 * device-side collectives): nothing is copied to the host, nothing synchronises, every call only enqueues work on
 * the context's stream (step 1 may allocate, as above).  All d_* pointers are device memory on the context's
 * device and must stay valid, and the words the calls read unchanged, until step 3 has run.
 *   1. jpegenc_segment_encode_dev: d_bits_tail8[0] = the band's bit count, [1] = its last 8 bits.  `flags` as for
 *      jpegenc_segment_encode (they describe rgb).
 *   2. jpegenc_segment_count_dev: reads *d_bit_offset and *d_prev_tail8 (the sums and the neighbour's word that
 *      the caller has worked out on the device), leaves the band's 0xFF count in *d_ff_bytes.
 *   3. jpegenc_segment_write_dev: reads *d_ff_offset; writes the words to d_words (room for cap_words), d_span[0]
 *      = first_word, d_span[1] = n_words, and for the last band *d_result.  n_words > cap_words means the band
 *      did not fit (nothing is written beyond cap_words; the last band's status says so too).
 * No counterpart in jpeg_top, like the three calls above. */
int jpegenc_segment_encode_dev(jpegenc_ctx* ctx, const uint8_t* rgb, int w, int rows, ptrdiff_t stride, uint32_t flags,
                               uint64_t* d_bits_tail8);
int jpegenc_segment_count_dev(jpegenc_ctx* ctx, const uint64_t* d_bit_offset, const uint64_t* d_prev_tail8,
                              uint64_t* d_ff_bytes);
int jpegenc_segment_write_dev(jpegenc_ctx* ctx, const uint64_t* d_ff_offset, int is_last, uint32_t* d_words,
                              size_t cap_words, uint64_t* d_span, jpegenc_result* d_result);

/* Port-level protocol, one call per clock edge of interest (jpeg_top.sv:20-27):
 *   jpegenc_reset        = rst
 *   jpegenc_push_pixel   = enable with data_in[23:0] = {B[23:16],G[15:8],R[7:0]}; 64 consecutive
 *                          pixels are one 8x8 block, row-major
 *   jpegenc_end_of_file  = end_of_file_signal: runs the GPU path over the pushed blocks
 *   jpegenc_read_words   = the JPEG_bitstream words in data_ready order (+ tail word)
 * Pixels are staged on the host and encoded on the device at end_of_file. */
int jpegenc_reset(jpegenc_ctx* ctx);
int jpegenc_push_pixel(jpegenc_ctx* ctx, uint32_t data_in24);
int jpegenc_push_pixels(jpegenc_ctx* ctx, const uint32_t* data_in24, size_t n);
int jpegenc_end_of_file(jpegenc_ctx* ctx, jpegenc_result* result);
int jpegenc_read_words(jpegenc_ctx* ctx, uint32_t* words_out, size_t cap_words, size_t* n_words);

/* Helper: words (value semantics, bit 31 first) -> byte stream; writes ceil(nbits/8) bytes where
 * nbits = 32*n_full_words + tail_bits.  Host memory only. */
int jpegenc_words_to_bytes(const uint32_t* words, const jpegenc_result* result, uint8_t* bytes_out, size_t cap_bytes,
                           size_t* n_bytes);

/* Optional per-kernel timing with HIP events on the context's stream.  When enabled every kernel
 * launch is bracketed by events; jpegenc_kernel_times returns, per kernel, the number of launches
 * and the summed duration in ms since the last reset (this call synchronises the stream). */
#define JPEGENC_N_KERNELS 5
int jpegenc_set_profiling(jpegenc_ctx* ctx, int enable);
int jpegenc_kernel_times(jpegenc_ctx* ctx, uint64_t launches[JPEGENC_N_KERNELS], double ms[JPEGENC_N_KERNELS],
                         int reset);
const char* jpegenc_kernel_name(int index);

/* Scratch memory budget in bytes for the intermediate chunk streams (default 32 GiB); a batch is
 * processed in sub-batches that fit (39 936 bytes per 64 blocks: 20.2 MB per 1080p frame, touched only
 * as far as the frame's stream is long).  Host-pointer batches are also staged in groups of images
 * whose input and output copies fit the budget.  When the device cannot provide the budget the
 * sub-batches are halved until they fit; JPEGENC_E_NOMEM when not even one image does. */
int jpegenc_set_workspace_limit(jpegenc_ctx* ctx, size_t bytes);

/* How much scratch a group of 64 blocks gets for its intermediate stream.  No counterpart in jpeg_top.
 * WORST_CASE (default): 39 936 bytes, what 64 blocks can need at most -- no call can ever outgrow it; 20.2 MB per
 * 1080p frame, of which photographic content touches about a tenth.
 * ADAPTIVE: as much as the longest group seen in the context's earlier large calls (more than 2048 groups) and a
 * quarter more, between 4 KB and the worst case: 1024 frames of 1080p photographs then take 3 to 4 GB instead of
 * 20.7.  The context's first large call sends a first few images through with worst-case scratch and waits, once,
 * for their statistics before it sizes the rest (so that call synchronises the stream once, even with device
 * pointers).  The price: when content turns up whose groups are longer than anything before by more than that
 * quarter, the groups that do not fit are not coded, and their images get status JPEGENC_E_AGAIN (nothing valid
 * in their words).  Host-pointer calls repeat such a call themselves (the second go cannot fail); a device-pointer
 * caller checks results[i].status as for JPEGENC_E_TOO_SMALL and repeats the call -- the statistics of the failed
 * call have told the context the size it needs by the time its results can be read.  The scratch never shrinks;
 * its size follows the longest group down by an eighth after four large calls in a row that stay below it. */
#define JPEGENC_SLOTS_WORST_CASE 0
#define JPEGENC_SLOTS_ADAPTIVE 1
int jpegenc_set_slot_mode(jpegenc_ctx* ctx, int mode);
/* Bytes of device memory the context's workspace holds at the moment. */
size_t jpegenc_workspace_bytes(const jpegenc_ctx* ctx);

/* Launch structure.  No counterpart in jpeg_top and no effect on the output.  AUTO (default): a call of at
 * most 2048 groups of 64 blocks (four 1080p frames) with all-ones tables and 8-byte-aligned rows runs as ONE
 * kernel launch -- every group's wavefront waits for the bit and 0xFF counts of the groups before it, then
 * stuffs and writes its own part of the stream (a wavefront that waits in vain for a tenth of a second gives
 * up and the image's status becomes JPEGENC_E_HIP); everything else runs as separate kernels (encode, prefix
 * sums, 0xFF count, stuff + write).  Within the single launch a wavefront lane codes one 8x8 block of Y, Cb and
 * Cr (groups of 64 blocks), or, for calls of at most 2048 x 21 blocks (one 1080p frame), one component of a
 * block (groups of 21 blocks: three times the wavefronts, a third of the work each).  SPLIT: always the
 * separate kernels.  SINGLE_MCU: as AUTO without the one-component-per-lane form. */
#define JPEGENC_PIPELINE_AUTO 0
#define JPEGENC_PIPELINE_SPLIT 1
#define JPEGENC_PIPELINE_SINGLE_MCU 2
int jpegenc_set_pipeline(jpegenc_ctx* ctx, int mode);

/* How the encode kernel sizes the per-block staging regions in LDS.  No counterpart in jpeg_top and no
 * effect on the output: only on speed.  Small regions (1408 bits per block of Y, Cb and Cr together) let
 * twelve wavefronts share a compute unit and are right for photographic content up to about 2.5 bytes per
 * pixel; a group of 64 blocks that holds a longer block is coded twice.  Large regions (2272 bits, more than
 * a block of uniform noise needs on average) leave room for eight wavefronts and cost 15 % where small ones
 * suffice.  Short regions (1024 bits) let sixteen wavefronts share a compute unit: 1 % faster than small ones
 * where no block is longer (content up to about 1.5 bytes per pixel).  AUTO (default): the kernel counts, in
 * every launch, the groups with a block too long for a short, a small and a large region, and the context uses
 * for the next launches whichever size those counts make cheaper (large from about 3 to 4 % of the groups too
 * long for small regions on, at once; back to small after four launches in a row whose counts favour it; short
 * after four launches in a row with fewer than one group in 1024 too long for it, back to small from one in
 * 256 on, at once). */
#define JPEGENC_REGIONS_AUTO 0
#define JPEGENC_REGIONS_SMALL 1
#define JPEGENC_REGIONS_LARGE 2
#define JPEGENC_REGIONS_SHORT 3
int jpegenc_set_region_mode(jpegenc_ctx* ctx, int mode);
/* The counts behind that choice, as the last finished launch that collected them left them (calls of at most
 * 2048 groups and calls with quantisation tables or unaligned rows collect none): counts[0] = groups of 64
 * blocks with a block too long for a small region, counts[1] = with one too long for a large region,
 * counts[2] = groups of that launch, counts[3] = groups with a block too long for a short region; counts[4] =
 * the size the context's next launch in AUTO mode starts from (JPEGENC_REGIONS_SMALL / _LARGE / _SHORT).  Read
 * after jpegenc_synchronize for figures of a particular call. */
int jpegenc_region_stats(jpegenc_ctx* ctx, uint64_t counts[5]);

/* Grows the workspace now to what jpegenc_encode_batch(n_images, w, h) will need (under the limit
 * above), so that later device-pointer calls of at most that size only enqueue work: no allocation, no
 * device synchronisation.  The workspace never shrinks before jpegenc_destroy.  No counterpart in
 * jpeg_top (hardware has no allocator). */
int jpegenc_reserve(jpegenc_ctx* ctx, int n_images, int w, int h);

const char* jpegenc_strerror(int code);
/* Text of the last HIP error seen by this context ("" if none). */
const char* jpegenc_last_error(const jpegenc_ctx* ctx);
/* "jpegenc-gfx950 <version> kernels <first 12 hex digits of the SHA-1 of the kernel sources it was built from>" */
const char* jpegenc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* JPEGENC_H */
