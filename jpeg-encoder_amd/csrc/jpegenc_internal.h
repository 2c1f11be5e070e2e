// jpegenc_internal.h -- shared between the HIP kernels and the C-ABI host code of libjpegenc.so.
#ifndef JPEGENC_INTERNAL_H
#define JPEGENC_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jpegenc.h"

namespace jpegenc {

// A chunk is 64 consecutive MCUs of one image, coded by one wavefront (lane = MCU).
constexpr int kChunkMcus = 64;
// Worst case bits per MCU: 3 component blocks x (22 DC + 63 x 26 AC) = 4980 bits = 156 words (rounded up).
constexpr int kMaxMcuBits = 4980;
constexpr int kMaxMcuWords = 156;
// Chunk stream slot in the workspace, in 32-bit words: worst case of 64 MCUs.  The stride between slots is a
// launch parameter (slot_words): this worst case, or less when the context sizes the slots from the chunk lengths
// it has seen (jpegenc_set_slot_mode); a chunk that does not fit a smaller slot is not stored and its image's
// status says so.
constexpr int kSlotWords = kChunkMcus * kMaxMcuWords;  // 9984 words = 39,936 B
constexpr uint32_t kMinSlotWords = 1024;               // smallest stride: the stand-in count below fits it
// What an unstored chunk reports as its length: any length from 1152 bits (what a chunk with a successor has at
// least) that fits the smallest slot keeps the stuffing kernels' reads and writes in bounds.
constexpr uint32_t kUnstoredChunkBits = 2048;

// The workspace of one launch group, sized for `chunks` chunks.
struct Workspace {
    uint32_t* slots;       // [chunks + 1][slot_words]: the stuffing kernels read up to 1 KB past a chunk's stream
    uint32_t* chunk_bits;  // [chunks]
    uint32_t* chunk_ff;    // [chunks]
    uint64_t* bit_off;     // [chunks]
    uint64_t* ff_off;      // [chunks]
    uint32_t slot_words = kSlotWords;  // stride between slots
    uint32_t* img_flags = nullptr;     // [images] or nullptr: non-zero = a chunk of the image was not stored
};

// Static workspace of jpegenc_encode_batch_noalloc: room for JPEGENC_NOALLOC_MAX_CHUNKS chunks, laid out as
// slots | bit_off | ff_off | chunk_bits | chunk_ff (every part starts a multiple of 256 bytes into it).
constexpr size_t kStaticChunks = JPEGENC_NOALLOC_MAX_CHUNKS;
constexpr size_t kStaticSlotBytes = (kStaticChunks + 1) * size_t(kSlotWords) * 4;
static_assert(kStaticSlotBytes % 256 == 0 && kStaticChunks % 64 == 0, "the parts stay 256-byte aligned");
constexpr size_t kStaticWsBytes = kStaticSlotBytes + kStaticChunks * (8 + 8 + 4 + 4);
// Device address, on the current device, of that workspace (defined beside the array in jpegenc_kernels.hip).
hipError_t static_workspace(void** ptr);

// Huffman table, one 32-bit word per entry in global memory: (total_length << 27) | (code << size) with
// total_length = code length + amplitude bits, 0 = nothing to emit.  The kernel keeps the two fields in
// two LDS arrays (a word and a byte per entry), so a lookup needs no unpacking.
// Index = run << 5 | eob << 4 | size:
//   size 0..10  AC.  (run <= 14, size 0) is a zero coefficient inside a run: nothing; (run 15, size 0) is
//               the 16th zero: ZRL in the eob = 0 entries, nothing in the eob = 1 entries
//               (rtl/4_huffman/cr_huff.sv:558).
//   size 11     AC value -1024, whose size by the rule on bits [9:0] is 0 (cr_huff.sv:1411-1486): the code the
//               run/size map gives (run, size 0), no amplitude bits -- EOB's code for run 0, the map's entry 0
//               for runs 1..14 (cr_huff.sv:2069-2082), ZRL's for run 15, whatever end_of_block says.  Only the
//               coefficient-fed variant reaches it.
//   size 14, run 0, eob 0   the EOB code (an AC size never exceeds 10, so sizes 11..15 are free)
//   size 15, eob 0          the DC code of size `run` (0..11)
constexpr int kHtabEntries = 512;
constexpr int kHtabEob = 14;
constexpr int kHtabDcSize = 15;

struct EncodeParams {
    const uint8_t* rgb;    // first image of this launch
    int64_t stride;        // bytes per pixel row
    int64_t image_stride;  // bytes between images
    uint32_t mcu_w;        // MCUs per block row (w / 8)
    uint32_t n_mcu;        // MCUs per image
    uint32_t cpi;          // chunks per image = ceil(n_mcu / 64), or ceil(n_mcu / 21) when lanes_per_mcu == 3
    uint32_t n_images;     // images of this launch (set by launch_encode)
    uint32_t flags;        // bit 0: rgb wrap quirk, bit 1: cb 6'd64 quirk, bit 2: the images continue a
                           // stream: the DC predictors of their first MCU come from the block row above
    const int16_t* coeffs;  // nullptr, or (coefficient-fed checking variant, rgb unused) [n_images][n_mcu][3][64]:
                            // the values at the coders' inputs in their reading order, 16-byte aligned
    uint32_t* slots;       // [n_images * cpi][slot_words]
    uint32_t slot_words;   // stride between slots, kMinSlotWords .. kSlotWords
    uint32_t report_bits;  // chunks longer than this report their length (stats word 2 of their line: maximum)
    uint32_t stats_seq;    // copied to stats word 259: which launch the block belongs to
    uint32_t* img_flags;   // [n_images] zeroed by the host, or nullptr (then slot_words is the worst case): set
                           // non-zero when a chunk of the image exceeds its slot and is left unstored
    uint32_t* chunk_bits;  // [n_images * cpi]
    const uint32_t* htab;  // [kHtabEntries]
    const int32_t* qmul;   // [3][64] multipliers 4096/Q in Huffman-input order, or nullptr
    uint32_t* stats;       // [kEncodeStatsWords] or nullptr: per launch, zeroed by the host before it; words
                           // 0, 16, ..., 240 count the chunks that hold an MCU too long for a small LDS region,
                           // words 1, 17, ..., 241 those with one too long even for a large region, words 2, 18,
                           // ... the longest chunk among those above report_bits (0: none), words 3, 19, ... the
                           // chunks left unstored, words 4, 20, ... the chunks with an MCU too long for a short
                           // region; word 256 receives the launch's chunk count, word 257 its
                           // report_bits, word 258 its slot_words, word 259
                           // its stats_seq
    uint32_t regions;      // kRegionsSmall / kRegionsLarge / kRegionsShort: which kernel variant (size of the LDS regions)
    // Single-launch form (small launches only, `state` != nullptr): every chunk's wave goes on to find its place
    // in the image's stream from the hand-off words of the chunks before it, stuffs its stream and writes the
    // words.  state = [3][n_images * cpi] words of tag << 32 | payload, tag = `generation` (never 0, never reused:
    // words of earlier launches read as "not there yet", so the array needs no clearing).
    uint64_t* state;
    uint32_t generation;
    uint32_t lanes_per_mcu;   // 1, or 3 (single-launch form only): a lane per component block, chunks of 21 MCUs
    uint32_t* words_out;      // as in StuffParams
    uint64_t out_stride;
    uint64_t cap_words;
    jpegenc_result* results;
};
// Largest launch that takes the single-launch form: all its chunks are resident at once (256 CUs x 8 workgroups).
constexpr uint32_t kSingleLaunchMaxChunks = 2048;
constexpr uint32_t kWideChunkMcus = 21;  // MCUs per chunk with a lane per component block
constexpr int kEncodeStatsWords = 16 * 16 + 16;
// EncodeParams::regions: small = 12 waves per CU, large = 8 (long strings), short = 16 (short strings)
constexpr uint32_t kRegionsSmall = 0, kRegionsLarge = 1, kRegionsShort = 2;

struct ScanParams {
    const uint32_t* counts;  // [n_images][cpi]
    uint64_t* prefix;        // [n_images][cpi] exclusive
    uint32_t cpi;
    // one-image launches of the segment calls: where to leave the sum of the counts and the high byte of the last
    // count word (a chunk's last 8 bits travel there); nullptr otherwise
    uint64_t* total_out = nullptr;
    uint64_t* last_hi8_out = nullptr;
};

struct StuffParams {
    const uint32_t* slots;
    uint32_t slot_words;        // stride between slots
    const uint32_t* img_flags;  // as in EncodeParams (first image of this launch), or nullptr
    const uint32_t* chunk_bits;
    uint64_t* bit_off;        // exclusive prefix of chunk_bits per image (written by the count pass when inline_scan)
    uint32_t* chunk_ff;       // count pass: number of 0xFF bytes owned by the chunk
    const uint64_t* ff_off;   // write pass: exclusive prefix of chunk_ff per image
    uint32_t* words_out;      // first image of this launch
    uint64_t out_stride;      // words between images
    uint64_t cap_words;       // capacity per image
    jpegenc_result* results;  // first image of this launch (device memory)
    uint32_t cpi;
    // A segment of a longer stream (all zero, eof = 1 for whole images): where the image's stream starts in
    // the whole one, the 0xFF bytes before it, the word of the whole stream that words_out[0] stands for,
    // the last 8 bits before the segment, and whether the image's end is the end of the file.
    uint32_t pred_tail8;
    uint64_t bit_base;
    uint64_t ff_base;
    uint64_t out_word_base;
    uint32_t eof;
    // The same three from device memory (k_stuff_count and k_stuff_write only; nullptr: the values above): a
    // band's place in the whole stream is known on the device first, when the ranks' counts have been gathered,
    // and need not travel through the host.  With ff_base_p the word that words_out[0] stands for follows too:
    // out_word_base = ((bit_base >> 3) + ff_base) >> 2.
    const uint64_t* bit_base_p = nullptr;
    const uint64_t* pred_tail8_p = nullptr;
    const uint64_t* ff_base_p = nullptr;
    // How many 1 KB steps of a chunk the stuffing kernels request before they know its length: 5 (all of a
    // photographic chunk at the all-ones tables, 4.4 KB) or 1 (quantised content); selects the kernel variant.
    uint32_t spec_depth;
    // Small launches (a frame or a few): no k_scan launches; every chunk's wave sums the counts of its image's
    // earlier chunks itself (k_stuff_count leaves the bit offsets in bit_off for k_stuff_write).  Needs
    // cpi * 318 720 bits < 2^32.
    uint32_t inline_scan;
};

// Launchers (defined in jpegenc_kernels.hip).  align = 8, 4 or 1: guaranteed alignment of every
// pixel-row address (rgb, stride, image_stride).
hipError_t launch_encode(const EncodeParams& p, uint32_t n_images, int align, hipStream_t s);
hipError_t launch_scan(const ScanParams& p, uint32_t n_images, hipStream_t s);
hipError_t launch_stuff_count(const StuffParams& p, uint32_t n_images, hipStream_t s);
hipError_t launch_stuff_write(const StuffParams& p, uint32_t n_images, hipStream_t s);
// Images of at most this many chunks take the per-image stuffing kernel without a barrier per round (every chunk's
// 0xFF count has a word in the LDS of the image's workgroup: 16 KB).
constexpr uint32_t kStuffImageLdsChunks = 4096;
// count + write of whole images in one launch (one workgroup per image; for batches of many images)
hipError_t launch_stuff_image(const StuffParams& p, uint32_t n_images, hipStream_t s);

// Where a band's words lie in the whole stream (first word, word count -> span[0], span[1]), from device-resident
// figures; also clears the band's first and last word, whose bytes outside the band the write step leaves alone.
struct SpanParams {
    const uint64_t* bit_base;  // bits of the stream before the band
    const uint64_t* bits;      // of the band
    const uint64_t* ff_base;   // 0xFF bytes before the band
    const uint64_t* ff;        // owned by the band
    uint32_t is_last;
    uint32_t* words;
    uint64_t cap_words;
    uint64_t* span;
};
hipError_t launch_segment_span(const SpanParams& p, hipStream_t s);

hipError_t launch_pad_edges(const uint8_t* src, int64_t stride, int64_t image_stride, int w, int h, int n_images,
                            uint8_t* dst, hipStream_t s);

hipError_t launch_coeffs(const uint8_t* rgb, int64_t stride, uint32_t mcu_w, uint32_t n_mcu, bool wrap,
                         const int32_t* qmul, int16_t* out, hipStream_t s);

// Host-side table construction (jpegenc_host.cpp).
void build_htab(uint32_t out[kHtabEntries]);

}  // namespace jpegenc
#endif
