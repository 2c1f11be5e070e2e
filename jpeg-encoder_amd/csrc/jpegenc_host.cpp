// jpegenc_host.cpp -- C ABI of libjpegenc.so (include/jpegenc.h): context, workspace, launch sequencing.
// The compute path is the HIP kernels in jpegenc_kernels.hip; there is no CPU fallback in this library.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "jpegenc_internal.h"
#include "jpegenc_tables.h"

using namespace jpegenc;

namespace jpegenc {

// Layout: see jpegenc_internal.h.  Code/length tables: T.81 K.4 (DC) and K.5 (AC) as hard-coded in
// rtl/4_huffman/cr_huff.sv:1553-2082.  The RTL's map entries (run 1..14, size 0) -> table entry 0 are
// reachable only by an AC coefficient of -1024 (size rule on bits [9:0]); coefficients computed from
// 8-bit pixels satisfy |AC| <= 1021 (|B| <= 1448 luma / 363 chroma, row sums of |T| <= 46344), so on
// the pixel path a size-0 symbol always is a zero coefficient and the (run <= 14, size 0) entries stay empty;
// what the RTL emits for -1024 stands at size 11, where the coefficient-fed entry point finds it.
void build_htab(uint32_t out[kHtabEntries]) {
    for (int i = 0; i < kHtabEntries; ++i) out[i] = 0;
    for (int eob = 0; eob < 2; ++eob)
        for (int sym = 0; sym < 256; ++sym) {
            const int idx = JPEGENC_AC_RUN_CODE[sym];
            const int size = sym & 15;
            if (idx == 0xFF || size > 10) continue;
            if (size == 0 && (sym != JPEGENC_SYM_ZRL || eob)) continue;
            const uint32_t code = JPEGENC_AC_CODE[idx];
            const uint32_t len = JPEGENC_AC_LEN[idx];
            out[((sym >> 4) << 5) | (eob << 4) | size] = ((len + size) << 27) | (code << size);
        }
    {
        const int idx = JPEGENC_AC_RUN_CODE[JPEGENC_SYM_EOB];
        out[kHtabEob] = (uint32_t(JPEGENC_AC_LEN[idx]) << 27) | JPEGENC_AC_CODE[idx];
    }
    for (int run = 0; run < 16; ++run) {  // -1024: symbol (run, size 0) through the map, no amplitude
        const int idx = JPEGENC_AC_RUN_CODE[run << 4];
        for (int eob = 0; eob < 2; ++eob)
            out[(run << 5) | (eob << 4) | 11] = (uint32_t(JPEGENC_AC_LEN[idx]) << 27) | JPEGENC_AC_CODE[idx];
    }
    for (int s = 0; s < 12; ++s) {
        const uint32_t code = JPEGENC_DC_CODE[s];
        const uint32_t len = JPEGENC_DC_LEN[s];
        out[(s << 5) | kHtabDcSize] = ((len + s) << 27) | (code << s);
    }
}

}  // namespace jpegenc

namespace {

enum SegWord { SEG_BITS = 0, SEG_TAIL8, SEG_BIT_OFFSET, SEG_PREV_TAIL8, SEG_FF, SEG_FF_OFFSET, SEG_FIRST_WORD, SEG_N_WORDS, kSegWords };

enum KernelId { K_ENCODE = 0, K_SCAN = 1, K_STUFF_COUNT = 2, K_STUFF_WRITE = 3, K_STUFF_IMAGE = 4 };
const char* const kKernelNames[JPEGENC_N_KERNELS] = {"k_encode_chunks", "k_scan", "k_stuff_count", "k_stuff_write",
                                                     "k_stuff_image"};

struct EventPair {
    int kernel;
    hipEvent_t a, b;
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;  // elements
    hipError_t reserve(size_t want) {
        if (want <= n) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            n = 0;
            if (e != hipSuccess) return e;
        }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
        if (e == hipSuccess) n = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace

struct jpegenc_ctx {
    jpegenc_config cfg;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    size_t ws_limit = size_t(32) << 30;
    bool quant = false;
    uint32_t flags = 0;

    DevBuf<uint32_t> htab;
    DevBuf<int32_t> qmul;
    // workspace, sized for ws_chunks chunks
    DevBuf<uint32_t> slots;
    DevBuf<uint32_t> chunk_bits, chunk_ff;
    DevBuf<uint64_t> bit_off, ff_off;
    // staging for host-pointer calls
    DevBuf<uint8_t> in_stage;
    DevBuf<uint8_t> pad_stage;  // JPEGENC_PAD_EDGES: the images padded to multiples of 8
    DevBuf<uint32_t> out_stage;
    DevBuf<jpegenc_result> res_stage;

    // Recorded on the stream after every group of launches that uses the workspace: a later call on another
    // stream (jpegenc_set_stream) waits for it before it touches the workspace again.
    hipEvent_t ws_event = nullptr;
    bool ws_pending = false;

    // Choice of the encode kernel's LDS region size (jpegenc_set_region_mode): the kernel counts, per launch,
    // the chunks that hold an MCU too long for a small region; the counts reach `stats_host` (pinned) by an
    // asynchronous copy behind the launch and steer the launches after it.
    int region_mode = JPEGENC_REGIONS_AUTO;
    uint32_t regions = jpegenc::kRegionsSmall;  // what AUTO stands at
    int small_votes = 0;  // launches in a row whose counts favoured small regions while large ones were in use
    int short_votes = 0;  // launches in a row without an MCU too long for a short region while small ones were in use
    DevBuf<uint32_t> stats;  // [kEncodeStatsWords] + one flag word per image of a launch group (img_flags)
    uint32_t* stats_host = nullptr;

    // Stride of the chunk slots (jpegenc_set_slot_mode).  ADAPTIVE: `est_bits` follows the longest chunk of the
    // launches whose statistics have arrived -- chunks longer than 7/8 of it report their length, so it is exact
    // while chunks reach that band, and it shrinks by an eighth per launch while none does -- and the slots get
    // that length and a quarter more.
    int slot_mode = JPEGENC_SLOTS_WORST_CASE;
    uint32_t est_bits = 0;      // 0: nothing seen yet (worst-case slots, and every chunk reports)
    int quiet_launches = 0;     // statistics blocks in a row in which no chunk reported
    uint64_t stats_seen = 0;    // sequence number of the last statistics block taken into est_bits
    uint64_t stats_issued = 0;  // launches that collect statistics, so far

    // Single-launch form of small launches (jpegenc_set_pipeline): hand-off words of the chunks, 3 per chunk
    int pipeline = JPEGENC_PIPELINE_AUTO;
    int n_cu = 0;  // compute units of the device
    DevBuf<uint64_t> state;

    // jpegenc_encode_batch_noalloc: this device's copy of the library's static workspace
    uint8_t* static_ws = nullptr;

    // segment API: what jpegenc_segment_encode left in the workspace, and where the band's figures stand in
    // device memory (the caller's words with the _dev calls, seg_dev with the calls that return them)
    bool seg_ready = false, seg_counted = false;
    uint32_t seg_cpi = 0;
    const uint64_t* seg_bits_p = nullptr;
    const uint64_t* seg_bit_offset_p = nullptr;
    const uint64_t* seg_prev_tail8_p = nullptr;
    const uint64_t* seg_ff_p = nullptr;
    DevBuf<uint64_t> seg_dev;  // [kSegWords]: bits, tail8, bit_offset, prev_tail8, ff_bytes, ff_offset, first_word, n_words

    // port-level protocol
    std::vector<uint8_t> port_px;
    std::vector<uint32_t> port_words;
    bool port_done = false;

    bool profiling = false;
    std::vector<EventPair> events;
    uint64_t k_launches[JPEGENC_N_KERNELS] = {};
    double k_ms[JPEGENC_N_KERNELS] = {};

    std::string last_error;
};

namespace {

int fail_hip(jpegenc_ctx* ctx, hipError_t e, const char* what) {
    if (ctx) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        ctx->last_error = buf;
    }
    return JPEGENC_E_HIP;
}

// Makes the context's device current for the duration of a call and puts the caller's back afterwards.
struct DevGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
    }
    ~DevGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};

#define HIPCHK(ctx, expr)                                       \
    do {                                                        \
        hipError_t e_ = (expr);                                 \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #expr);  \
    } while (0)

// For allocations: running out of device memory is JPEGENC_E_NOMEM, not a HIP failure.
#define MEMCHK(ctx, expr)                                                  \
    do {                                                                   \
        hipError_t e_ = (expr);                                            \
        if (e_ == hipErrorOutOfMemory) {                                   \
            (void)hipGetLastError();                                       \
            (ctx)->last_error = "device memory exhausted: " #expr;         \
            return JPEGENC_E_NOMEM;                                        \
        }                                                                  \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #expr);             \
    } while (0)

// Event pairs wait in the context until jpegenc_kernel_times collects them; beyond this many the launches
// simply go untimed.
constexpr size_t kMaxPendingEvents = 16384;

struct ProfScope {
    jpegenc_ctx* ctx;
    EventPair ep;
    bool on;
    ProfScope(jpegenc_ctx* c, int kernel) : ctx(c), on(c->profiling && c->events.size() < kMaxPendingEvents) {
        if (!on) return;
        ep.kernel = kernel;
        if (hipEventCreate(&ep.a) != hipSuccess) {
            on = false;
            return;
        }
        if (hipEventCreate(&ep.b) != hipSuccess) {
            (void)hipEventDestroy(ep.a);
            on = false;
            return;
        }
        (void)hipEventRecord(ep.a, ctx->stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(ep.b, ctx->stream);
        ctx->events.push_back(ep);
    }
};

// 8: the kernel variant that addresses an image's pixels with 32-bit offsets from its first byte (buffer
// loads), so every offset inside an image has to stay below 2^31 as well.
int pointer_align(const void* p, int64_t stride, int64_t image_stride, int n_images, int rows) {
    uint64_t bits = reinterpret_cast<uintptr_t>(p) | uint64_t(stride);
    if (n_images > 1) bits |= uint64_t(image_stride);
    const bool fits32 = stride > 0 && stride * int64_t(rows) < (int64_t(1) << 31);
    if ((bits & 7) == 0 && fits32) return 8;
    if ((bits & 3) == 0) return 4;
    return 1;
}

// Launches of at most this many chunks (16 frames of 1080p) skip the scan kernels: there a launch costs more
// than the few loads with which every chunk's wave sums the counts of its image's earlier chunks (4 us less
// per call up to 16 frames, nothing gained from 32 on).  The sums are 32-bit: cpi * 318 720 bits must stay
// below 2^32.
constexpr uint32_t kStuffImageMinImages = 512;
constexpr uint32_t kStuffImageMinChunks = 32;  // per image: its workgroup takes 8 chunks at a time
constexpr uint64_t kInlineScanChunks = 8192;
constexpr uint32_t kInlineScanMaxCpi = 8192;

// Geometry of a call and the number of images one launch group holds under the workspace budget.
struct Plan {
    uint32_t mcu_w, n_mcu, cpi;
    uint32_t sub;         // images per launch group
    uint32_t slot_words;  // stride of the chunk slots
};

// Takes the statistics that have arrived since the last look into the estimate of the longest chunk (see
// jpegenc_ctx::est_bits).  The words are written by asynchronous copies behind earlier launches; any mix of old and
// new words only makes the estimate a little stale.
void update_slot_estimate(jpegenc_ctx* ctx) {
    const volatile uint32_t* h = ctx->stats_host;
    const uint32_t seq = h[16 * 16 + 3];
    if (seq == 0 || seq == uint32_t(ctx->stats_seen)) return;
    ctx->stats_seen = seq;
    uint32_t longest = 0, unstored = 0;
    for (int i = 0; i < 16; ++i) {
        longest = std::max<uint32_t>(longest, uint32_t(h[16 * i + 2]));
        unstored += h[16 * i + 3];
    }
    const uint32_t report = h[16 * 16 + 1];
    const uint32_t worst = uint32_t(kSlotWords) * 32u;
    if (longest > worst) return;  // (not a length: torn or foreign words)
    if (longest) {
        // exact: every chunk above `report` took part in the maximum (an unstored chunk is above it by far)
        ctx->est_bits = unstored ? std::max(ctx->est_bits, longest) : longest;
        ctx->quiet_launches = 0;
    } else if (report != 0xFFFFFFFFu && report < ctx->est_bits && ++ctx->quiet_launches >= 4) {
        ctx->est_bits = std::max<uint32_t>(report, 1);  // four launches with nothing above `report`: down an eighth
        ctx->quiet_launches = 0;
    }
}

uint32_t current_slot_words(jpegenc_ctx* ctx, bool adaptive_allowed) {
    if (!adaptive_allowed || ctx->slot_mode != JPEGENC_SLOTS_ADAPTIVE) return uint32_t(kSlotWords);
    update_slot_estimate(ctx);
    if (ctx->est_bits == 0) return uint32_t(kSlotWords);
    uint64_t words = (uint64_t(ctx->est_bits) * 5 / 4 + 31) / 32 + 64;
    words = (words + 255) & ~uint64_t(255);
    return uint32_t(std::min<uint64_t>(std::max<uint64_t>(words, kMinSlotWords), uint64_t(kSlotWords)));
}

int make_plan(jpegenc_ctx* ctx, int n_images, int w, int h, Plan* pl, bool adaptive_allowed = true) {
    pl->mcu_w = uint32_t(w / 8);
    const uint64_t n_mcu64 = uint64_t(pl->mcu_w) * uint32_t(h / 8);
    if (n_mcu64 == 0 || n_mcu64 > 0x7FFFFFFFull) return JPEGENC_E_UNSUPPORTED;
    pl->n_mcu = uint32_t(n_mcu64);
    pl->cpi = (pl->n_mcu + kChunkMcus - 1) / kChunkMcus;
    // (small calls keep worst-case slots: nothing to save, and their kernels collect no statistics)
    pl->slot_words = uint64_t(n_images) * pl->cpi > kSingleLaunchMaxChunks ? current_slot_words(ctx, adaptive_allowed)
                                                                           : uint32_t(kSlotWords);
    const uint64_t per_image = uint64_t(pl->cpi) * pl->slot_words * 4;
    uint64_t sub = ctx->ws_limit / per_image;
    if (sub < 1) sub = 1;
    if (sub > uint64_t(n_images)) sub = uint64_t(n_images);
    while (sub * pl->cpi > 0x7FFFFFF0ull) sub /= 2;  // grid limit
    pl->sub = uint32_t(sub);
    return JPEGENC_OK;
}

hipError_t reserve_workspace(jpegenc_ctx* ctx, size_t chunks, uint32_t slot_words = uint32_t(kSlotWords)) {
    // behind the last slot: what the stuffing kernels may read past a chunk's end (five 1 KB steps and a word)
    hipError_t e = ctx->slots.reserve(chunks * size_t(slot_words) + std::max<size_t>(slot_words, 2048));
    if (e == hipSuccess) e = ctx->chunk_bits.reserve(chunks);
    if (e == hipSuccess) e = ctx->chunk_ff.reserve(chunks);
    if (e == hipSuccess) e = ctx->bit_off.reserve(chunks);
    if (e == hipSuccess) e = ctx->ff_off.reserve(chunks);
    return e;
}

// Grows the workspace for the plan.  When the device has no room for it (another allocator may hold most of
// the memory) the launch groups are halved until they fit, down to one image.
int reserve_for_plan(jpegenc_ctx* ctx, Plan* pl) {
    for (;;) {
        hipError_t e = reserve_workspace(ctx, size_t(pl->sub) * pl->cpi, pl->slot_words);
        if (e == hipSuccess && pl->slot_words != uint32_t(kSlotWords))
            e = ctx->stats.reserve(size_t(kEncodeStatsWords) + pl->sub);  // + the images' flag words
        if (e == hipSuccess) return JPEGENC_OK;
        if (e != hipErrorOutOfMemory) return fail_hip(ctx, e, "workspace allocation");
        (void)hipGetLastError();  // the failed allocation is not left behind as a pending error
        if (pl->sub == 1) {
            ctx->last_error = "device memory exhausted: the workspace for one image does not fit";
            return JPEGENC_E_NOMEM;
        }
        pl->sub = (pl->sub + 1) / 2;
    }
}

Workspace ctx_workspace(const jpegenc_ctx* ctx) {
    Workspace ws;
    ws.slots = ctx->slots.p;
    ws.chunk_bits = ctx->chunk_bits.p;
    ws.chunk_ff = ctx->chunk_ff.p;
    ws.bit_off = ctx->bit_off.p;
    ws.ff_off = ctx->ff_off.p;
    return ws;
}
Workspace ctx_workspace(const jpegenc_ctx* ctx, const Plan& pl) {
    Workspace ws = ctx_workspace(ctx);
    ws.slot_words = pl.slot_words;
    ws.img_flags = pl.slot_words != uint32_t(kSlotWords) ? ctx->stats.p + kEncodeStatsWords : nullptr;
    return ws;
}

// Marks the workspace as in use by what has just been enqueued on the context's stream.
int workspace_used(jpegenc_ctx* ctx) {
    HIPCHK(ctx, hipEventRecord(ctx->ws_event, ctx->stream));
    ctx->ws_pending = true;
    return JPEGENC_OK;
}

// Region size for the next launch of the encode kernel, and its statistics buffer (zeroed on the stream, together
// with the images' flag words when the slots are not the worst case).  Only the common kernel variant (8-byte-aligned
// rows, all-ones tables) exists with large and with short regions.
int choose_regions(jpegenc_ctx* ctx, EncodeParams* ep, int align, uint64_t chunks, uint32_t n_images) {
    ep->stats = nullptr;
    ep->regions = kRegionsSmall;
    ep->report_bits = 0xFFFFFFFFu;
    ep->stats_seq = 0;
    const bool common = !ctx->quant && align >= 8;
    const bool adaptive = ctx->slot_mode == JPEGENC_SLOTS_ADAPTIVE && chunks > kSingleLaunchMaxChunks;
    if (!common && !adaptive) return JPEGENC_OK;
    if (common && chunks <= kSingleLaunchMaxChunks && ctx->region_mode == JPEGENC_REGIONS_AUTO) {
        // every chunk is resident at once even with large regions: they cost nothing here, and a small call is
        // spared the two stream operations of the statistics
        ep->regions = kRegionsLarge;
        return JPEGENC_OK;
    }
    if (common && ctx->region_mode == JPEGENC_REGIONS_AUTO) {
        // whatever the last completed copy left in the pinned words: a heuristic, any value is safe
        const volatile uint32_t* h = ctx->stats_host;
        uint64_t over_short = 0, over_small = 0, over_large = 0;
        for (int i = 0; i < 16; ++i) {
            over_small += h[16 * i];
            over_large += h[16 * i + 1];
            over_short += h[16 * i + 4];
        }
        const uint64_t total = h[16 * 16];
        if (total >= 256 && over_large <= over_small && over_small <= over_short && over_short <= total) {
            // A chunk with an MCU too long for its region is coded again, and the other waves of its workgroup
            // (four with small regions, two with large ones) hold their LDS idle meanwhile: a workgroup with any
            // such chunk costs about 2.15 times one without.  Large regions cost every chunk about 15 % (two waves
            // per SIMD instead of three).  Measured break-even: 3 to 4 % of the chunks too long for a small
            // region.  The margin keeps a launch near the break-even from flipping back and forth.
            const double qs = 1.0 - double(over_small) / double(total), ql = 1.0 - double(over_large) / double(total);
            const double small_cost = 1.0 + 1.15 * (1.0 - qs * qs * qs * qs);
            const double large_cost = 1.15 * (1.0 + 1.15 * (1.0 - ql * ql));
            // The wrong choice is not equally bad (small regions on noise: 2.7 times the time; large ones on easy
            // content: 1.15 times), so the way up is taken at once and the way down only after four launches in a
            // row that ask for it: a caller that alternates between kinds of content stays on large regions.
            if (small_cost > large_cost * 1.02) {
                ctx->regions = kRegionsLarge;
                ctx->small_votes = ctx->short_votes = 0;
            } else if (ctx->regions == kRegionsLarge) {
                if (small_cost * 1.02 < large_cost && ++ctx->small_votes >= 4) {
                    ctx->regions = kRegionsSmall;
                    ctx->small_votes = ctx->short_votes = 0;
                }
            } else if (ctx->regions == kRegionsShort) {
                // Short regions gain 1 to 2.4 % and lose 15 % with 6 % of the chunks redone: they are left as soon as
                // one chunk in 256 holds an MCU too long for them ...
                if (over_short * 256 > total) {
                    ctx->regions = kRegionsSmall;
                    ctx->short_votes = 0;
                }
            } else {
                // ... and taken after four calls in a row that find fewer than one in 1024 in the latest counts
                ctx->short_votes = over_short * 1024 < total ? ctx->short_votes + 1 : 0;
                if (ctx->short_votes >= 4) {
                    ctx->regions = kRegionsShort;
                    ctx->short_votes = 0;
                }
            }
        }
    } else if (common) {
        ctx->regions = ctx->region_mode == JPEGENC_REGIONS_LARGE   ? kRegionsLarge
                       : ctx->region_mode == JPEGENC_REGIONS_SHORT ? kRegionsShort
                                                                   : kRegionsSmall;
    }
    ep->regions = common ? ctx->regions : kRegionsSmall;
    ep->stats = ctx->stats.p;
    if (++ctx->stats_issued == 0 || uint32_t(ctx->stats_issued) == 0) ctx->stats_issued = 1;
    ep->stats_seq = uint32_t(ctx->stats_issued);
    // chunks in the top eighth below the longest seen so far report their length (see jpegenc_ctx::est_bits)
    if (adaptive) ep->report_bits = ctx->est_bits - ctx->est_bits / 8;
    const size_t words = size_t(kEncodeStatsWords) + (ep->img_flags ? n_images : 0);
    HIPCHK(ctx, hipMemsetAsync(ctx->stats.p, 0, words * 4, ctx->stream));
    return JPEGENC_OK;
}

// Behind the encode launch: its statistics on their way to the host (no synchronisation; read by later calls).
int collect_stats(jpegenc_ctx* ctx, const EncodeParams& ep) {
    if (!ep.stats) return JPEGENC_OK;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stats_host, ctx->stats.p, kEncodeStatsWords * 4, hipMemcpyDeviceToHost, ctx->stream));
    return JPEGENC_OK;
}

// Tag of a single-launch call in its hand-off words: unique in the process (contexts may share the static
// workspace's neighbourhood in time, and a word left by any earlier launch must read as "not there yet").
uint32_t next_generation() {
    static std::atomic<uint32_t> counter{0};
    uint32_t g;
    do g = counter.fetch_add(1, std::memory_order_relaxed) + 1;
    while (g == 0);
    return g;
}

// The launches for n images (one launch group) with the workspace `ws`.  All pointers are device memory.
// `timed`: bracket the kernels with events when profiling is on (creates events) and let small calls take the
// single-launch form (whose launch carries a tag of its own, so that enqueued work cannot be replayed).
int run_group(jpegenc_ctx* ctx, const Workspace& ws, bool timed, const Plan& pl, const uint8_t* rgb, uint32_t n,
              ptrdiff_t stride, ptrdiff_t image_stride, int align, uint32_t* words, size_t cap_words,
              jpegenc_result* results, const int16_t* coeffs = nullptr) {
    const bool prof_was = ctx->profiling;
    if (!timed) ctx->profiling = false;
    struct Restore {
        jpegenc_ctx* c;
        bool v;
        ~Restore() { c->profiling = v; }
    } restore{ctx, prof_was};
    const uint32_t cpi = pl.cpi;
    EncodeParams ep;
    ep.rgb = rgb;
    ep.stride = stride;
    ep.image_stride = image_stride;
    ep.mcu_w = pl.mcu_w;
    ep.n_mcu = pl.n_mcu;
    ep.cpi = cpi;
    ep.flags = ctx->flags;
    ep.coeffs = coeffs;
    ep.slots = ws.slots;
    ep.slot_words = ws.slot_words;
    ep.report_bits = 0xFFFFFFFFu;
    ep.stats_seq = 0;
    ep.img_flags = ws.img_flags;
    ep.chunk_bits = ws.chunk_bits;
    ep.htab = ctx->htab.p;
    ep.qmul = (ctx->quant && !coeffs) ? ctx->qmul.p : nullptr;
    ep.state = nullptr;
    ep.generation = 0;
    ep.lanes_per_mcu = 1;
    ep.words_out = words;
    ep.out_stride = cap_words;
    ep.cap_words = cap_words;
    ep.results = results;
    // A frame or a few (every chunk resident at once), common kernel variant: one launch does it all -- each
    // chunk's wave waits for the counts of the chunks before it, then stuffs and writes its own stream.
    // (every chunk's wave has to be resident, eight one-wave workgroups per compute unit: a device with fewer
    // units than the 256 the limit is named for takes the form for correspondingly smaller calls)
    const uint64_t resident = std::min<uint64_t>(kSingleLaunchMaxChunks, 8u * uint64_t(std::max(ctx->n_cu, 0)));
    if (timed && !coeffs && ctx->pipeline != JPEGENC_PIPELINE_SPLIT && uint64_t(n) * cpi <= resident && !ctx->quant &&
        align >= 8) {
        ep.stats = nullptr;
        ep.regions = kRegionsLarge;
        ep.state = ctx->state.p;
        ep.generation = next_generation();
        // about a frame of 1080p or less: a lane per component block, chunks of 21 MCUs -- three times the waves
        // with a third of the work each, on a GPU that such a call leaves mostly idle (these chunks never use
        // their workspace slots)
        const uint32_t cpi_wide = (pl.n_mcu + kWideChunkMcus - 1) / kWideChunkMcus;
        if (ctx->pipeline == JPEGENC_PIPELINE_AUTO && uint64_t(n) * cpi_wide <= resident) {
            ep.cpi = cpi_wide;
            ep.lanes_per_mcu = 3;
        }
        ProfScope ps(ctx, K_ENCODE);
        HIPCHK(ctx, launch_encode(ep, n, align, ctx->stream));
        return JPEGENC_OK;
    }
    int rc_regions = JPEGENC_OK;
    if (coeffs) {  // the coders of the small-region variant unless large regions are asked for; no statistics
        ep.stats = nullptr;
        ep.regions = ctx->region_mode == JPEGENC_REGIONS_LARGE ? kRegionsLarge : kRegionsSmall;
    } else if (!timed) {
        // the call that allocates nothing (possibly under graph capture): the region size the context stands at,
        // no statistics -- their memset and copy would be replayed with the graph and steer nothing
        ep.stats = nullptr;
        ep.regions = kRegionsSmall;
        if (!ctx->quant && align >= 8)
            ep.regions = ctx->region_mode == JPEGENC_REGIONS_AUTO    ? ctx->regions
                         : ctx->region_mode == JPEGENC_REGIONS_LARGE ? kRegionsLarge
                         : ctx->region_mode == JPEGENC_REGIONS_SHORT ? kRegionsShort
                                                                     : kRegionsSmall;
    } else {
        rc_regions = choose_regions(ctx, &ep, align, uint64_t(n) * cpi, n);
    }
    if (rc_regions != JPEGENC_OK) return rc_regions;
    if (ep.img_flags && !ep.stats)  // (the flag words are zeroed with the statistics; without those, on their own)
        HIPCHK(ctx, hipMemsetAsync(ep.img_flags, 0, size_t(n) * 4, ctx->stream));
    {
        ProfScope ps(ctx, K_ENCODE);
        HIPCHK(ctx, launch_encode(ep, n, align, ctx->stream));
    }
    rc_regions = collect_stats(ctx, ep);
    if (rc_regions != JPEGENC_OK) return rc_regions;
    // Many images: one workgroup per image does both stuffing steps in one launch and reads every chunk stream
    // once.  Measured on 1080p batches against count + scan + write: 1.41 against 1.44 ms of kernel time per 1024
    // frames, 1.5 to 3 % less per step (no scan launches, 2.3 GB less read); 0.73 against 0.74 ms at 512 frames;
    // with too few workgroups it loses (16-wave workgroups: 8 % slower at 128 frames): used from 512 images on.
    // (from 256 images on when the image's chunk counts fit the LDS of its workgroup: the form without a barrier per
    // round, with 16 waves per workgroup below 512 images: 256 x 3840x2160 1.25 against 1.44 ms)
    bool per_image = cpi >= kStuffImageMinChunks && n >= kStuffImageMinImages;
    if (!per_image && cpi >= 128 && cpi <= kStuffImageLdsChunks && ctx->n_cu > 0 && n >= uint32_t(ctx->n_cu)) {
        // one 16-wave workgroup per CU: only when the images fill the CUs in whole turns or nearly (384 images on
        // 256 CUs: 0.63 against 0.55 ms)
        const uint32_t turns = (n + uint32_t(ctx->n_cu) - 1) / uint32_t(ctx->n_cu);
        per_image = uint64_t(n) * 10 >= uint64_t(turns) * uint32_t(ctx->n_cu) * 9;
    }
    // a frame or a few: the stuffing kernels work out the prefixes themselves, two launches fewer
    const bool inline_scan = !per_image && uint64_t(n) * cpi <= kInlineScanChunks && cpi <= kInlineScanMaxCpi;
    ScanParams sp;
    sp.counts = ws.chunk_bits;
    sp.prefix = ws.bit_off;
    sp.cpi = cpi;
    if (!inline_scan && !per_image) {
        ProfScope ps(ctx, K_SCAN);
        HIPCHK(ctx, launch_scan(sp, n, ctx->stream));
    }
    StuffParams tp;
    tp.slots = ws.slots;
    tp.slot_words = ws.slot_words;
    tp.img_flags = ws.img_flags;
    tp.chunk_bits = ws.chunk_bits;
    tp.bit_off = ws.bit_off;
    tp.chunk_ff = ws.chunk_ff;
    tp.ff_off = ws.ff_off;
    tp.words_out = words;
    tp.out_stride = cap_words;
    tp.cap_words = cap_words;
    tp.results = results;
    tp.cpi = cpi;
    tp.pred_tail8 = 0;
    tp.bit_base = tp.ff_base = tp.out_word_base = 0;
    tp.eof = 1;
    tp.spec_depth = (ctx->quant && !coeffs) ? 1u : 5u;
    tp.inline_scan = inline_scan ? 1u : 0u;
    if (per_image) {
        ProfScope ps(ctx, K_STUFF_IMAGE);
        HIPCHK(ctx, launch_stuff_image(tp, n, ctx->stream));
        return JPEGENC_OK;
    }
    {
        ProfScope ps(ctx, K_STUFF_COUNT);
        HIPCHK(ctx, launch_stuff_count(tp, n, ctx->stream));
    }
    sp.counts = ws.chunk_ff;
    sp.prefix = ws.ff_off;
    if (!inline_scan) {
        ProfScope ps(ctx, K_SCAN);
        HIPCHK(ctx, launch_scan(sp, n, ctx->stream));
    }
    {
        ProfScope ps(ctx, K_STUFF_WRITE);
        HIPCHK(ctx, launch_stuff_write(tp, n, ctx->stream));
    }
    return JPEGENC_OK;
}

// All pointers are device memory.  Enqueues the whole pipeline on ctx->stream; allocates (and thereby
// synchronises the device) only when the workspace has to grow.
int encode_device(jpegenc_ctx* ctx, const uint8_t* rgb, int n_images, int w, int h, ptrdiff_t stride,
                  ptrdiff_t image_stride, uint32_t* words, size_t cap_words, jpegenc_result* results) {
    Plan pl;
    int rc = make_plan(ctx, n_images, w, h, &pl);
    if (rc != JPEGENC_OK) return rc;
    ctx->seg_ready = ctx->seg_counted = false;  // whatever a segment call left in the workspace is gone
    const int align = pointer_align(rgb, stride, image_stride, n_images, h);
    int done = 0;
    if (ctx->slot_mode == JPEGENC_SLOTS_ADAPTIVE && ctx->est_bits == 0 && uint64_t(n_images) * pl.cpi > kSingleLaunchMaxChunks) {
        // The context's first large call with slots sized from experience, and there is none yet: a first group of
        // a few images goes through with worst-case slots, every chunk reporting its length, and the host waits
        // for that group's statistics -- once per context -- before it sizes the slots for the rest.
        Plan probe = pl;
        probe.slot_words = uint32_t(kSlotWords);
        probe.sub = uint32_t(std::min<uint64_t>(uint64_t(n_images), 2 * kSingleLaunchMaxChunks / pl.cpi + 1));  // > 2048 chunks: statistics are collected
        rc = reserve_for_plan(ctx, &probe);
        if (rc != JPEGENC_OK) return rc;
        rc = run_group(ctx, ctx_workspace(ctx, probe), true, probe, rgb, probe.sub, stride, image_stride, align, words,
                       cap_words, results);
        if (rc != JPEGENC_OK) return rc;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        done = int(probe.sub);
        if (done == n_images) return workspace_used(ctx);
        rc = make_plan(ctx, n_images - done, w, h, &pl);
        if (rc != JPEGENC_OK) return rc;
    }
    rc = reserve_for_plan(ctx, &pl);
    if (rc != JPEGENC_OK) return rc;
    // (The single-launch form with a lane per component block cuts an image into three times the chunks, and
    // needs no slots for them: such a chunk always fits its LDS regions -- a static_assert in the kernel.)
    const Workspace ws = ctx_workspace(ctx, pl);
    for (int first = done; first < n_images; first += int(pl.sub)) {
        const uint32_t n = uint32_t(std::min<int64_t>(pl.sub, n_images - first));
        rc = run_group(ctx, ws, true, pl, rgb + int64_t(first) * image_stride, n, stride, image_stride, align,
                       words + uint64_t(first) * cap_words, cap_words, results + first);
        if (rc != JPEGENC_OK) return rc;
    }
    return workspace_used(ctx);
}

// Device address of the library's static workspace on `device`, which must be the current device.  Looked up
// once per device, at the first jpegenc_create on it.
hipError_t static_workspace_of(int device, uint8_t** out) {
    constexpr int kMaxDevices = 64;
    static std::mutex mu;
    static uint8_t* known[kMaxDevices] = {};
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!known[device]) {
        void* p = nullptr;
        const hipError_t e = static_workspace(&p);
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        known[device] = static_cast<uint8_t*>(p);
    }
    *out = known[device];
    return hipSuccess;
}

int check_dims(const void* rgb, int n, int w, int h, ptrdiff_t stride, const void* words, const void* results,
               bool pad_edges = false) {
    if (!rgb || !words || !results) return JPEGENC_E_BAD_ARG;
    if (n <= 0 || w <= 0 || h <= 0 || (!pad_edges && ((w % 8) || (h % 8)))) return JPEGENC_E_BAD_ARG;
    if (stride < ptrdiff_t(w) * 3) return JPEGENC_E_BAD_ARG;
    return JPEGENC_OK;
}

int encode_staged(jpegenc_ctx* ctx, const uint8_t* d_rgb, int n_images, int w, int h, uint32_t* words, size_t cap_words,
                  jpegenc_result* results);

int encode_host(jpegenc_ctx* ctx, const uint8_t* rgb, int n_images, int w, int h, ptrdiff_t stride,
                ptrdiff_t image_stride, uint32_t* words, size_t cap_words, jpegenc_result* results) {
    const size_t row = size_t(w) * 3;
    const size_t img_bytes = row * size_t(h);
    MEMCHK(ctx, ctx->in_stage.reserve(img_bytes * size_t(n_images)));
    MEMCHK(ctx, ctx->out_stage.reserve(cap_words * size_t(n_images)));
    MEMCHK(ctx, ctx->res_stage.reserve(size_t(n_images)));
    for (int i = 0; i < n_images; ++i)
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->in_stage.p + img_bytes * size_t(i), row, rgb + int64_t(i) * image_stride,
                                     size_t(stride), row, size_t(h), hipMemcpyHostToDevice, ctx->stream));
    return encode_staged(ctx, ctx->in_stage.p, n_images, w, h, words, cap_words, results);
}

// Device-resident, densely packed input; output and results go back to host memory.
int encode_staged(jpegenc_ctx* ctx, const uint8_t* d_rgb, int n_images, int w, int h, uint32_t* words, size_t cap_words,
                  jpegenc_result* results) {
    const size_t row = size_t(w) * 3;
    const size_t img_bytes = row * size_t(h);
    MEMCHK(ctx, ctx->out_stage.reserve(cap_words * size_t(n_images)));
    MEMCHK(ctx, ctx->res_stage.reserve(size_t(n_images)));
    int rc = JPEGENC_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        rc = encode_device(ctx, d_rgb, n_images, w, h, ptrdiff_t(row), ptrdiff_t(img_bytes), ctx->out_stage.p, cap_words,
                           ctx->res_stage.p);
        if (rc != JPEGENC_OK) return rc;
        HIPCHK(ctx, hipMemcpyAsync(results, ctx->res_stage.p, sizeof(jpegenc_result) * size_t(n_images),
                                   hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        bool again = false;
        for (int i = 0; i < n_images; ++i) again = again || results[i].status == JPEGENC_E_AGAIN;
        if (!again) break;
        // a chunk outgrew the slots sized from earlier calls: the statistics of this call are in by now (the stream
        // has drained) and hold its longest chunk, so the second go fits
        update_slot_estimate(ctx);
    }
    for (int i = 0; i < n_images; ++i) {
        uint64_t nw = results[i].n_full_words + (results[i].tail_bits ? 1 : 0);
        if (nw > cap_words) {
            nw = cap_words;
            rc = JPEGENC_E_TOO_SMALL;
        }
        if (nw)
            HIPCHK(ctx, hipMemcpyAsync(words + size_t(i) * cap_words, ctx->out_stage.p + size_t(i) * cap_words,
                                       size_t(nw) * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

}  // namespace

extern "C" {

int jpegenc_default_config(jpegenc_config* cfg) {
    if (!cfg) return JPEGENC_E_BAD_ARG;
    std::memset(cfg, 0, sizeof *cfg);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 64; ++i) cfg->qtab[c][i] = 1;
    cfg->quirk_rgb_wrap = 1;
    cfg->quirk_cb_6d64 = 1;
    cfg->quirk_ydct_zext = 0;
    cfg->device = 0;
    return JPEGENC_OK;
}

int jpegenc_create(jpegenc_ctx** out, const jpegenc_config* cfg) {
    if (!out || !cfg) return JPEGENC_E_BAD_ARG;
    *out = nullptr;
    if (cfg->quirk_ydct_zext) return JPEGENC_E_UNSUPPORTED;
    if (cfg->quant_index_order > 1) return JPEGENC_E_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return JPEGENC_E_HIP;
    if (cfg->device < 0 || cfg->device >= ndev) return JPEGENC_E_BAD_ARG;
    jpegenc_ctx* ctx = new (std::nothrow) jpegenc_ctx;
    if (!ctx) return JPEGENC_E_NOMEM;
    ctx->cfg = *cfg;
    ctx->device = cfg->device;
    DevGuard guard(ctx->device);
    hipError_t e = guard.err;
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, ctx->device);
    if (e == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        delete ctx;
        return JPEGENC_E_HIP;  // the code object is gfx950-only
    }
    if (e == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ws_event, hipEventDisableTiming);
    // the address of the static workspace on this device (every device holds its own copy at its own address)
    if (e == hipSuccess) e = static_workspace_of(ctx->device, &ctx->static_ws);
    if (e != hipSuccess) {
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return JPEGENC_E_HIP;
    }
    ctx->stream = ctx->own_stream;
    ctx->flags = (cfg->quirk_rgb_wrap ? 1u : 0u) | (cfg->quirk_cb_6d64 ? 2u : 0u);

    uint32_t htab[kHtabEntries];
    build_htab(htab);
    int32_t qmul[3 * 64];
    bool all_ones = true;
    for (int c = 0; c < 3; ++c)
        for (int a = 0; a < 8; ++a)
            for (int b = 0; b < 8; ++b) {
                // Huffman-input position (a vertical, b horizontal) is DCT output Z{b+1}{a+1}, which the
                // quantiser scales with table entry [b*8+a] (y_quantizer.sv:40-43, yd_q_h.sv:120-135).
                const uint16_t q = cfg->quant_index_order ? cfg->qtab[c][a * 8 + b] : cfg->qtab[c][b * 8 + a];
                if (q != 1) all_ones = false;
                qmul[c * 64 + a * 8 + b] = q ? int32_t(4096 / q) : 0;
            }
    ctx->quant = !all_ones;
    e = ctx->htab.reserve(kHtabEntries);
    if (e == hipSuccess) e = ctx->qmul.reserve(3 * 64);
    if (e == hipSuccess) e = ctx->stats.reserve(kEncodeStatsWords);
    if (e == hipSuccess) e = ctx->seg_dev.reserve(kSegWords);
    if (e == hipSuccess) e = ctx->state.reserve(3 * size_t(kSingleLaunchMaxChunks));
    if (e == hipSuccess) e = hipMemset(ctx->state.p, 0, 3 * size_t(kSingleLaunchMaxChunks) * sizeof(uint64_t));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&ctx->stats_host), kEncodeStatsWords * 4, hipHostMallocDefault);
    if (e == hipSuccess) std::memset(ctx->stats_host, 0, kEncodeStatsWords * 4);
    if (e == hipSuccess) e = hipMemcpy(ctx->htab.p, htab, sizeof htab, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->qmul.p, qmul, sizeof qmul, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        jpegenc_destroy(ctx);
        return JPEGENC_E_HIP;
    }
    *out = ctx;
    return JPEGENC_OK;
}

int jpegenc_destroy(jpegenc_ctx* ctx) {
    if (!ctx) return JPEGENC_E_BAD_ARG;
    DevGuard guard(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws_event) (void)hipEventDestroy(ctx->ws_event);
    for (auto& ep : ctx->events) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    ctx->htab.release();
    ctx->qmul.release();
    ctx->stats.release();
    ctx->state.release();
    ctx->seg_dev.release();
    if (ctx->stats_host) (void)hipHostFree(ctx->stats_host);
    ctx->slots.release();
    ctx->chunk_bits.release();
    ctx->chunk_ff.release();
    ctx->bit_off.release();
    ctx->ff_off.release();
    ctx->in_stage.release();
    ctx->pad_stage.release();
    ctx->out_stage.release();
    ctx->res_stage.release();
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return JPEGENC_OK;
}

int jpegenc_set_stream(jpegenc_ctx* ctx, void* hip_stream) {
    if (!ctx) return JPEGENC_E_BAD_ARG;
    const hipStream_t next = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    if (next != ctx->stream && ctx->ws_pending) {
        // every call shares the context's workspace: what is still queued on the old stream has to finish
        // with it before anything on the new stream may overwrite it
        DevGuard guard(ctx->device);
        HIPCHK(ctx, guard.err);
        HIPCHK(ctx, hipStreamWaitEvent(next, ctx->ws_event, 0));
    }
    ctx->stream = next;
    return JPEGENC_OK;
}

int jpegenc_synchronize(jpegenc_ctx* ctx) {
    if (!ctx) return JPEGENC_E_BAD_ARG;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return JPEGENC_OK;
}

size_t jpegenc_max_words(int w, int h) {
    if (w <= 0 || h <= 0) return 0;
    const uint64_t mcus = uint64_t((w + 7) / 8) * uint64_t((h + 7) / 8);
    return size_t((mcus * uint64_t(kMaxMcuBits) * 2 + 31) / 32 + 1);
}

int jpegenc_encode_batch(jpegenc_ctx* ctx, const uint8_t* rgb, int n_images, int w, int h, ptrdiff_t stride,
                         ptrdiff_t image_stride, uint32_t flags, uint32_t* words_out, size_t cap_words_per_image,
                         jpegenc_result* results) {
    if (!ctx) return JPEGENC_E_BAD_ARG;
    const bool pad = (flags & JPEGENC_PAD_EDGES) && ((w % 8) || (h % 8));
    int rc = check_dims(rgb, n_images, w, h, stride, words_out, results, pad);
    if (rc != JPEGENC_OK) return rc;
    if (cap_words_per_image == 0) return JPEGENC_E_BAD_ARG;
    if (n_images > 1 && (image_stride < 0 ? -image_stride : image_stride) < stride * ptrdiff_t(h) && image_stride != 0)
        return JPEGENC_E_BAD_ARG;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    if (!(flags & JPEGENC_DEVICE_PTRS) && n_images > 1) {
        // host memory: the staging copies of input and output count against the workspace budget too, so a
        // large batch goes through in groups of images
        const uint64_t per_image = uint64_t(w) * 3 * uint64_t(h) * (pad ? 2 : 1) + uint64_t(cap_words_per_image) * 4;
        const uint64_t group = std::max<uint64_t>(1, ctx->ws_limit / per_image);
        if (group < uint64_t(n_images)) {
            int worst = JPEGENC_OK;
            for (int first = 0; first < n_images; first += int(group)) {
                const int n = int(std::min<uint64_t>(group, uint64_t(n_images - first)));
                rc = jpegenc_encode_batch(ctx, rgb + int64_t(first) * image_stride, n, w, h, stride, image_stride, flags,
                                          words_out + size_t(first) * cap_words_per_image, cap_words_per_image,
                                          results + first);
                if (rc == JPEGENC_E_TOO_SMALL)
                    worst = rc;  // reported per image; the other groups are still coded
                else if (rc != JPEGENC_OK)
                    return rc;
            }
            return worst;
        }
    }
    if (pad) {
        // extension (not in jpeg_top): replicate the right and bottom edges up to multiples of 8 on the
        // device, then code the padded images
        const int w8 = (w + 7) & ~7, h8 = (h + 7) & ~7;
        const size_t row = size_t(w) * 3, padded = size_t(w8) * 3 * size_t(h8);
        const uint8_t* d_src = rgb;
        ptrdiff_t s_stride = stride, s_istride = image_stride;
        if (!(flags & JPEGENC_DEVICE_PTRS)) {
            MEMCHK(ctx, ctx->in_stage.reserve(row * size_t(h) * size_t(n_images)));
            for (int i = 0; i < n_images; ++i)
                HIPCHK(ctx, hipMemcpy2DAsync(ctx->in_stage.p + row * size_t(h) * size_t(i), row,
                                             rgb + int64_t(i) * image_stride, size_t(stride), row, size_t(h),
                                             hipMemcpyHostToDevice, ctx->stream));
            d_src = ctx->in_stage.p;
            s_stride = ptrdiff_t(row);
            s_istride = ptrdiff_t(row * size_t(h));
        }
        MEMCHK(ctx, ctx->pad_stage.reserve(padded * size_t(n_images)));
        HIPCHK(ctx, launch_pad_edges(d_src, s_stride, s_istride, w, h, n_images, ctx->pad_stage.p, ctx->stream));
        if (flags & JPEGENC_DEVICE_PTRS)
            return encode_device(ctx, ctx->pad_stage.p, n_images, w8, h8, ptrdiff_t(w8) * 3, ptrdiff_t(padded), words_out,
                                 cap_words_per_image, results);
        return encode_staged(ctx, ctx->pad_stage.p, n_images, w8, h8, words_out, cap_words_per_image, results);
    }
    if (flags & JPEGENC_DEVICE_PTRS)
        return encode_device(ctx, rgb, n_images, w, h, stride, image_stride, words_out, cap_words_per_image, results);
    return encode_host(ctx, rgb, n_images, w, h, stride, image_stride, words_out, cap_words_per_image, results);
}

int jpegenc_reserve(jpegenc_ctx* ctx, int n_images, int w, int h) {
    if (!ctx || n_images <= 0 || w <= 0 || h <= 0) return JPEGENC_E_BAD_ARG;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    Plan pl;
    const int rc = make_plan(ctx, n_images, (w + 7) & ~7, (h + 7) & ~7, &pl);
    if (rc != JPEGENC_OK) return rc;
    ctx->seg_ready = ctx->seg_counted = false;  // growing the workspace discards a segment's chunk streams
    return reserve_for_plan(ctx, &pl);
}

int jpegenc_encode_batch_noalloc(jpegenc_ctx* ctx, const uint8_t* rgb, int n_images, int w, int h, ptrdiff_t stride,
                                 ptrdiff_t image_stride, uint32_t* words_out, size_t cap_words_per_image,
                                 jpegenc_result* results) {
    if (!ctx) return JPEGENC_E_BAD_ARG;
    int rc = check_dims(rgb, n_images, w, h, stride, words_out, results);
    if (rc != JPEGENC_OK) return rc;
    if (cap_words_per_image == 0) return JPEGENC_E_BAD_ARG;
    if (n_images > 1 && (image_stride < 0 ? -image_stride : image_stride) < stride * ptrdiff_t(h) && image_stride != 0)
        return JPEGENC_E_BAD_ARG;
    Plan pl;
    rc = make_plan(ctx, n_images, w, h, &pl, false);  // the static workspace has worst-case slots
    if (rc != JPEGENC_OK) return rc;
    if (uint64_t(n_images) * pl.cpi > kStaticChunks) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "jpegenc_encode_batch_noalloc: %llu chunks of 64 blocks needed, the static workspace holds %llu",
                      (unsigned long long)(uint64_t(n_images) * pl.cpi), (unsigned long long)kStaticChunks);
        ctx->last_error = buf;
        return JPEGENC_E_LIMIT;
    }
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    // the parts of the static workspace (device addresses derived from the looked-up base, never from the symbol)
    uint8_t* const base = ctx->static_ws;
    Workspace ws;
    ws.slots = reinterpret_cast<uint32_t*>(base);
    ws.bit_off = reinterpret_cast<uint64_t*>(base + kStaticSlotBytes);
    ws.ff_off = ws.bit_off + kStaticChunks;
    ws.chunk_bits = reinterpret_cast<uint32_t*>(ws.ff_off + kStaticChunks);
    ws.chunk_ff = ws.chunk_bits + kStaticChunks;
    // one launch group as separate kernels; no events are created (timed = false), nothing is allocated, staged
    // or synchronised, and no launch argument changes from call to call for the same arguments
    rc = run_group(ctx, ws, false, pl, rgb, uint32_t(n_images), stride, image_stride,
                   pointer_align(rgb, stride, image_stride, n_images, h), words_out, cap_words_per_image, results);
    if (rc != JPEGENC_OK) return rc;
    // Outside graph capture the call is ordered like any other (jpegenc_set_stream makes a new stream wait for the
    // event; recording it allocates nothing).  Under capture nothing is recorded: the event would become part of
    // the graph, be recorded again by every replay, and a later jpegenc_set_stream would wait on a captured
    // event.  What a replay touches is the device's static array, not this context's workspace; ordering replays
    // against other work on the GPU is the caller's business, as the header says.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    if (cap != hipStreamCaptureStatusNone) return JPEGENC_OK;
    return workspace_used(ctx);
}

int jpegenc_image_coeffs(jpegenc_ctx* ctx, const uint8_t* rgb, int w, int h, ptrdiff_t stride, uint32_t flags,
                         int16_t* coeffs_out) {
    if (!ctx || !rgb || !coeffs_out) return JPEGENC_E_BAD_ARG;
    if (w <= 0 || h <= 0 || (w % 8) || (h % 8) || stride < ptrdiff_t(w) * 3) return JPEGENC_E_BAD_ARG;
    const uint32_t mcu_w = uint32_t(w / 8);
    const uint64_t n_mcu = uint64_t(mcu_w) * uint32_t(h / 8);
    if (n_mcu > 0x7FFFFFFFull) return JPEGENC_E_UNSUPPORTED;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    const int32_t* qmul = ctx->quant ? ctx->qmul.p : nullptr;
    const bool wrap = (ctx->flags & 1u) != 0;
    if (flags & JPEGENC_DEVICE_PTRS) {
        HIPCHK(ctx, launch_coeffs(rgb, stride, mcu_w, uint32_t(n_mcu), wrap, qmul, coeffs_out, ctx->stream));
        return JPEGENC_OK;
    }
    // host memory: staged through buffers of this call's own (a checking aid, not a hot path)
    const size_t row = size_t(w) * 3, out_bytes = size_t(n_mcu) * 3 * 64 * sizeof(int16_t);
    DevBuf<uint8_t> in;
    DevBuf<int16_t> out;
    hipError_t e = in.reserve(row * size_t(h));
    if (e == hipSuccess) e = out.reserve(size_t(n_mcu) * 3 * 64);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(in.p, row, rgb, size_t(stride), row, size_t(h), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = launch_coeffs(in.p, int64_t(row), mcu_w, uint32_t(n_mcu), wrap, qmul, out.p, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(coeffs_out, out.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    in.release();
    out.release();
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        ctx->last_error = "device memory exhausted: jpegenc_image_coeffs staging";
        return JPEGENC_E_NOMEM;
    }
    HIPCHK(ctx, e);
    return JPEGENC_OK;
}

int jpegenc_encode_coeffs(jpegenc_ctx* ctx, const int16_t* coeffs, int w, int h, uint32_t flags, uint32_t* words_out,
                          size_t cap_words, jpegenc_result* result) {
    if (!ctx || !coeffs || !words_out || !result || cap_words == 0) return JPEGENC_E_BAD_ARG;
    if (w <= 0 || h <= 0 || (w % 8) || (h % 8)) return JPEGENC_E_BAD_ARG;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    Plan pl;
    int rc = make_plan(ctx, 1, w, h, &pl, false);  // (a checking aid: worst-case slots, no repeat to think of)
    if (rc != JPEGENC_OK) return rc;
    ctx->seg_ready = ctx->seg_counted = false;
    rc = reserve_for_plan(ctx, &pl);
    if (rc != JPEGENC_OK) return rc;
    const Workspace ws = ctx_workspace(ctx, pl);
    if (flags & JPEGENC_DEVICE_PTRS) {
        if (reinterpret_cast<uintptr_t>(coeffs) & 15u) return JPEGENC_E_BAD_ARG;
        rc = run_group(ctx, ws, true, pl, nullptr, 1, 0, 0, 8, words_out, cap_words, result, coeffs);
        if (rc != JPEGENC_OK) return rc;
        return workspace_used(ctx);
    }
    // host memory: staged through buffers of this call's own (a checking aid, not a hot path)
    const size_t in_bytes = size_t(pl.n_mcu) * 3 * 64 * sizeof(int16_t);
    DevBuf<uint8_t> in;
    hipError_t e = in.reserve(in_bytes);
    if (e == hipSuccess) e = ctx->out_stage.reserve(cap_words);
    if (e == hipSuccess) e = ctx->res_stage.reserve(1);
    if (e == hipSuccess) e = hipMemcpyAsync(in.p, coeffs, in_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        rc = run_group(ctx, ws, true, pl, nullptr, 1, 0, 0, 8, ctx->out_stage.p, cap_words, ctx->res_stage.p,
                       reinterpret_cast<const int16_t*>(in.p));
        if (rc == JPEGENC_OK) rc = workspace_used(ctx);
        if (rc == JPEGENC_OK) {
            e = hipMemcpyAsync(result, ctx->res_stage.p, sizeof(jpegenc_result), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess) {
                uint64_t nw = result->n_full_words + (result->tail_bits ? 1 : 0);
                if (nw > cap_words) {
                    nw = cap_words;
                    rc = JPEGENC_E_TOO_SMALL;
                }
                if (nw) e = hipMemcpy(words_out, ctx->out_stage.p, size_t(nw) * 4, hipMemcpyDeviceToHost);
            }
        }
    }
    (void)hipStreamSynchronize(ctx->stream);  // nothing in flight reads the staging buffer when it goes
    in.release();
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        ctx->last_error = "device memory exhausted: jpegenc_encode_coeffs staging";
        return JPEGENC_E_NOMEM;
    }
    HIPCHK(ctx, e);
    return rc;
}

int jpegenc_encode_rgb8(jpegenc_ctx* ctx, const uint8_t* rgb, int w, int h, ptrdiff_t stride, uint32_t flags,
                        uint32_t* words_out, size_t cap_words, jpegenc_result* result) {
    return jpegenc_encode_batch(ctx, rgb, 1, w, h, stride, stride * ptrdiff_t(h), flags, words_out, cap_words, result);
}

int jpegenc_encode_stripes(jpegenc_ctx* ctx, const uint8_t* rgb, int w, int h, ptrdiff_t stride, int n_stripes,
                           uint32_t flags, uint32_t* words_out, size_t cap_words_per_stripe,
                           jpegenc_result* results) {
    if (n_stripes <= 0 || h <= 0 || (h % n_stripes)) return JPEGENC_E_BAD_ARG;
    const int rows = h / n_stripes;
    if (rows % 8) return JPEGENC_E_BAD_ARG;
    return jpegenc_encode_batch(ctx, rgb, n_stripes, w, rows, stride, stride * ptrdiff_t(rows), flags, words_out,
                                cap_words_per_stripe, results);
}

// ---- one stream from consecutive segments -------------------------------------------------------------
// The _dev calls only enqueue work: a band's figures (bit count, last 8 bits, 0xFF count, place of its words) are
// left in device memory and its offsets are read from device memory.  The calls that return those figures are the
// same launches with the figures in the context's own words, copied back after a stream synchronisation.
int jpegenc_segment_encode_dev(jpegenc_ctx* ctx, const uint8_t* rgb, int w, int rows, ptrdiff_t stride, uint32_t flags,
                               uint64_t* d_bits_tail8) {
    if (!ctx || !rgb || !d_bits_tail8) return JPEGENC_E_BAD_ARG;
    if (w <= 0 || rows <= 0 || (w % 8) || (rows % 8) || stride < ptrdiff_t(w) * 3) return JPEGENC_E_BAD_ARG;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    ctx->seg_ready = ctx->seg_counted = false;
    const bool continues = (flags & JPEGENC_SEG_CONTINUES) != 0;
    const uint8_t* d_rgb = rgb;
    ptrdiff_t d_stride = stride;
    if (!(flags & JPEGENC_DEVICE_PTRS)) {
        const size_t row = size_t(w) * 3;
        const int halo = continues ? 8 : 0;
        MEMCHK(ctx, ctx->in_stage.reserve(row * size_t(rows + halo)));
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->in_stage.p, row, rgb - ptrdiff_t(halo) * stride, size_t(stride), row,
                                     size_t(rows + halo), hipMemcpyHostToDevice, ctx->stream));
        d_rgb = ctx->in_stage.p + row * size_t(halo);
        d_stride = ptrdiff_t(row);
    }
    const uint32_t mcu_w = uint32_t(w / 8);
    const uint64_t n_mcu64 = uint64_t(mcu_w) * uint32_t(rows / 8);
    if (n_mcu64 > 0x7FFFFFFFull) return JPEGENC_E_UNSUPPORTED;
    const uint32_t cpi = uint32_t((n_mcu64 + kChunkMcus - 1) / kChunkMcus);
    {
        const hipError_t e = reserve_workspace(ctx, cpi);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            ctx->last_error = "device memory exhausted: the workspace for this segment does not fit";
            return JPEGENC_E_NOMEM;
        }
        HIPCHK(ctx, e);
    }
    EncodeParams ep;
    ep.rgb = d_rgb;
    ep.stride = d_stride;
    ep.image_stride = 0;
    ep.mcu_w = mcu_w;
    ep.n_mcu = uint32_t(n_mcu64);
    ep.cpi = cpi;
    ep.flags = ctx->flags | (continues ? 4u : 0u);
    ep.coeffs = nullptr;
    ep.slots = ctx->slots.p;
    ep.slot_words = uint32_t(kSlotWords);
    ep.report_bits = 0xFFFFFFFFu;
    ep.stats_seq = 0;
    ep.img_flags = nullptr;
    ep.chunk_bits = ctx->chunk_bits.p;
    ep.htab = ctx->htab.p;
    ep.qmul = ctx->quant ? ctx->qmul.p : nullptr;
    // the block row above must be as aligned as the segment's own rows: the stride takes care of that
    ep.state = nullptr;
    ep.generation = 0;
    ep.lanes_per_mcu = 1;
    ep.words_out = nullptr;
    ep.out_stride = ep.cap_words = 0;
    ep.results = nullptr;
    const int seg_align = pointer_align(d_rgb, d_stride, 0, 1, rows);
    int rc_regions = choose_regions(ctx, &ep, seg_align, cpi, 1);
    if (rc_regions != JPEGENC_OK) return rc_regions;
    HIPCHK(ctx, launch_encode(ep, 1, seg_align, ctx->stream));
    rc_regions = collect_stats(ctx, ep);
    if (rc_regions != JPEGENC_OK) return rc_regions;
    ScanParams sp;
    sp.counts = ctx->chunk_bits.p;
    sp.prefix = ctx->bit_off.p;
    sp.cpi = cpi;
    sp.total_out = d_bits_tail8;
    sp.last_hi8_out = d_bits_tail8 + 1;
    HIPCHK(ctx, launch_scan(sp, 1, ctx->stream));
    ctx->seg_cpi = cpi;
    ctx->seg_bits_p = d_bits_tail8;
    ctx->seg_ready = true;
    return workspace_used(ctx);
}

int jpegenc_segment_encode(jpegenc_ctx* ctx, const uint8_t* rgb, int w, int rows, ptrdiff_t stride, uint32_t flags,
                           uint64_t* bits, uint32_t* tail8) {
    if (!ctx || !bits || !tail8) return JPEGENC_E_BAD_ARG;
    const int rc = jpegenc_segment_encode_dev(ctx, rgb, w, rows, stride, flags, ctx->seg_dev.p + SEG_BITS);
    if (rc != JPEGENC_OK) return rc;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    uint64_t v[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(v, ctx->seg_dev.p + SEG_BITS, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *bits = v[0];
    *tail8 = uint32_t(v[1]);
    return JPEGENC_OK;
}

extern "C++" {
namespace {
StuffParams segment_params(jpegenc_ctx* ctx) {
    StuffParams tp;
    tp.slots = ctx->slots.p;
    tp.slot_words = uint32_t(kSlotWords);
    tp.img_flags = nullptr;
    tp.chunk_bits = ctx->chunk_bits.p;
    tp.bit_off = ctx->bit_off.p;
    tp.chunk_ff = ctx->chunk_ff.p;
    tp.ff_off = ctx->ff_off.p;
    tp.words_out = nullptr;
    tp.out_stride = 0;
    tp.cap_words = 0;
    tp.results = nullptr;
    tp.cpi = ctx->seg_cpi;
    tp.pred_tail8 = 0;
    tp.bit_base = 0;
    tp.ff_base = 0;
    tp.out_word_base = 0;
    tp.eof = 0;
    tp.spec_depth = ctx->quant ? 1u : 5u;
    tp.inline_scan = 0;
    tp.bit_base_p = ctx->seg_bit_offset_p;
    tp.pred_tail8_p = ctx->seg_prev_tail8_p;
    return tp;
}
}  // namespace
}  // extern "C++"

int jpegenc_segment_count_dev(jpegenc_ctx* ctx, const uint64_t* d_bit_offset, const uint64_t* d_prev_tail8,
                              uint64_t* d_ff_bytes) {
    if (!ctx || !d_bit_offset || !d_prev_tail8 || !d_ff_bytes) return JPEGENC_E_BAD_ARG;
    if (!ctx->seg_ready) return JPEGENC_E_STATE;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    ctx->seg_bit_offset_p = d_bit_offset;
    ctx->seg_prev_tail8_p = d_prev_tail8;
    ctx->seg_ff_p = d_ff_bytes;
    const StuffParams tp = segment_params(ctx);
    HIPCHK(ctx, launch_stuff_count(tp, 1, ctx->stream));
    ScanParams sp;
    sp.counts = ctx->chunk_ff.p;
    sp.prefix = ctx->ff_off.p;
    sp.cpi = ctx->seg_cpi;
    sp.total_out = d_ff_bytes;
    HIPCHK(ctx, launch_scan(sp, 1, ctx->stream));
    ctx->seg_counted = true;
    return workspace_used(ctx);
}

int jpegenc_segment_count(jpegenc_ctx* ctx, uint64_t bit_offset, uint32_t prev_tail8, uint64_t* ff_bytes) {
    if (!ctx || !ff_bytes) return JPEGENC_E_BAD_ARG;
    if (!ctx->seg_ready) return JPEGENC_E_STATE;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    const uint64_t in[2] = {bit_offset, uint64_t(prev_tail8 & 0xFFu)};
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (the words may still be read by what an earlier band queued)
    HIPCHK(ctx, hipMemcpy(ctx->seg_dev.p + SEG_BIT_OFFSET, in, sizeof in, hipMemcpyHostToDevice));
    const int rc = jpegenc_segment_count_dev(ctx, ctx->seg_dev.p + SEG_BIT_OFFSET, ctx->seg_dev.p + SEG_PREV_TAIL8,
                                             ctx->seg_dev.p + SEG_FF);
    if (rc != JPEGENC_OK) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ff_bytes, ctx->seg_dev.p + SEG_FF, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return JPEGENC_OK;
}

int jpegenc_segment_write_dev(jpegenc_ctx* ctx, const uint64_t* d_ff_offset, int is_last, uint32_t* d_words,
                              size_t cap_words, uint64_t* d_span, jpegenc_result* d_result) {
    if (!ctx || !d_ff_offset || !d_words || !d_span || cap_words == 0) return JPEGENC_E_BAD_ARG;
    if (is_last && !d_result) return JPEGENC_E_BAD_ARG;
    if (!ctx->seg_ready || !ctx->seg_counted) return JPEGENC_E_STATE;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    // bytes of boundary words that belong to the neighbouring segments stay zero (the span kernel clears the two
    // words before the write step fills in the band's own bytes): the caller ORs those words
    SpanParams sp;
    sp.bit_base = ctx->seg_bit_offset_p;
    sp.bits = ctx->seg_bits_p;
    sp.ff_base = d_ff_offset;
    sp.ff = ctx->seg_ff_p;
    sp.is_last = is_last ? 1u : 0u;
    sp.words = d_words;
    sp.cap_words = cap_words;
    sp.span = d_span;
    HIPCHK(ctx, launch_segment_span(sp, ctx->stream));
    StuffParams tp = segment_params(ctx);
    tp.words_out = d_words;
    tp.cap_words = cap_words;
    tp.results = d_result;
    tp.ff_base_p = d_ff_offset;
    tp.eof = is_last ? 1u : 0u;
    HIPCHK(ctx, launch_stuff_write(tp, 1, ctx->stream));
    ctx->seg_ready = ctx->seg_counted = false;
    return workspace_used(ctx);
}

int jpegenc_segment_write(jpegenc_ctx* ctx, uint64_t ff_offset, int is_last, uint32_t flags, uint32_t* words_out,
                          size_t cap_words, uint64_t* first_word, uint64_t* n_words, jpegenc_result* result) {
    if (!ctx || !words_out || !first_word || !n_words || cap_words == 0) return JPEGENC_E_BAD_ARG;
    if (is_last && !result) return JPEGENC_E_BAD_ARG;
    if (!ctx->seg_ready || !ctx->seg_counted) return JPEGENC_E_STATE;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    uint32_t* d_words = words_out;
    if (!(flags & JPEGENC_DEVICE_PTRS)) {
        MEMCHK(ctx, ctx->out_stage.reserve(cap_words));
        d_words = ctx->out_stage.p;
    }
    MEMCHK(ctx, ctx->res_stage.reserve(1));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(ctx->seg_dev.p + SEG_FF_OFFSET, &ff_offset, 8, hipMemcpyHostToDevice));
    const int rc = jpegenc_segment_write_dev(ctx, ctx->seg_dev.p + SEG_FF_OFFSET, is_last, d_words, cap_words,
                                             ctx->seg_dev.p + SEG_FIRST_WORD, ctx->res_stage.p);
    if (rc != JPEGENC_OK) return rc;
    uint64_t span[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(span, ctx->seg_dev.p + SEG_FIRST_WORD, sizeof span, hipMemcpyDeviceToHost, ctx->stream));
    if (is_last)
        HIPCHK(ctx, hipMemcpyAsync(result, ctx->res_stage.p, sizeof(jpegenc_result), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *first_word = span[0];
    *n_words = span[1];
    const size_t n_store = size_t(std::min<uint64_t>(span[1], cap_words));
    if (!(flags & JPEGENC_DEVICE_PTRS) && n_store)
        HIPCHK(ctx, hipMemcpy(words_out, d_words, n_store * 4, hipMemcpyDeviceToHost));
    return (span[1] > cap_words) ? JPEGENC_E_TOO_SMALL : JPEGENC_OK;
}

int jpegenc_reset(jpegenc_ctx* ctx) {
    if (!ctx) return JPEGENC_E_BAD_ARG;
    ctx->port_px.clear();
    ctx->port_words.clear();
    ctx->port_done = false;
    return JPEGENC_OK;
}

int jpegenc_push_pixel(jpegenc_ctx* ctx, uint32_t d) {
    if (!ctx) return JPEGENC_E_BAD_ARG;
    if (ctx->port_done) return JPEGENC_E_STATE;  // rst is required after end_of_file
    ctx->port_px.push_back(uint8_t(d & 0xFF));          // R = data_in[7:0]
    ctx->port_px.push_back(uint8_t((d >> 8) & 0xFF));   // G = data_in[15:8]
    ctx->port_px.push_back(uint8_t((d >> 16) & 0xFF));  // B = data_in[23:16]
    return JPEGENC_OK;
}

int jpegenc_push_pixels(jpegenc_ctx* ctx, const uint32_t* d, size_t n) {
    if (!ctx || (!d && n)) return JPEGENC_E_BAD_ARG;
    if (ctx->port_done) return JPEGENC_E_STATE;
    ctx->port_px.reserve(ctx->port_px.size() + n * 3);
    for (size_t i = 0; i < n; ++i) {
        ctx->port_px.push_back(uint8_t(d[i] & 0xFF));
        ctx->port_px.push_back(uint8_t((d[i] >> 8) & 0xFF));
        ctx->port_px.push_back(uint8_t((d[i] >> 16) & 0xFF));
    }
    return JPEGENC_OK;
}

int jpegenc_end_of_file(jpegenc_ctx* ctx, jpegenc_result* result) {
    if (!ctx || !result) return JPEGENC_E_BAD_ARG;
    if (ctx->port_done) return JPEGENC_E_STATE;
    // 64 consecutive samples are one block, row-major (cb_dct.sv:478-541): the pushed stream is a
    // raster image 8 pixels wide.  A trailing partial block never completes a DCT and is dropped.
    const size_t nblk = ctx->port_px.size() / (64 * 3);
    std::memset(result, 0, sizeof *result);
    ctx->port_words.clear();
    ctx->port_done = true;
    if (nblk == 0) return JPEGENC_OK;
    if (nblk > size_t(0x7FFFFFFF) / 8) return JPEGENC_E_UNSUPPORTED;
    const int h = int(nblk * 8);
    ctx->port_words.assign(jpegenc_max_words(8, h), 0);
    int rc = jpegenc_encode_rgb8(ctx, ctx->port_px.data(), 8, h, 24, JPEGENC_HOST_PTRS, ctx->port_words.data(),
                                 ctx->port_words.size(), result);
    if (rc != JPEGENC_OK) return rc;
    ctx->port_words.resize(size_t(result->n_full_words) + (result->tail_bits ? 1 : 0));
    ctx->port_px.clear();
    return JPEGENC_OK;
}

int jpegenc_read_words(jpegenc_ctx* ctx, uint32_t* words_out, size_t cap_words, size_t* n_words) {
    if (!ctx || !n_words) return JPEGENC_E_BAD_ARG;
    if (!ctx->port_done) return JPEGENC_E_STATE;
    *n_words = ctx->port_words.size();
    if (cap_words < ctx->port_words.size() || (!words_out && !ctx->port_words.empty())) return JPEGENC_E_TOO_SMALL;
    if (!ctx->port_words.empty()) std::memcpy(words_out, ctx->port_words.data(), ctx->port_words.size() * 4);
    return JPEGENC_OK;
}

int jpegenc_words_to_bytes(const uint32_t* words, const jpegenc_result* r, uint8_t* bytes_out, size_t cap_bytes,
                           size_t* n_bytes) {
    if (!words || !r || !n_bytes) return JPEGENC_E_BAD_ARG;
    const uint64_t nbits = r->n_full_words * 32 + r->tail_bits;
    const size_t nb = size_t((nbits + 7) / 8);
    *n_bytes = nb;
    if (nb > cap_bytes || (!bytes_out && nb)) return JPEGENC_E_TOO_SMALL;
    for (size_t i = 0; i < nb; ++i) bytes_out[i] = uint8_t(words[i >> 2] >> (24 - 8 * (i & 3)));
    return JPEGENC_OK;
}

int jpegenc_set_profiling(jpegenc_ctx* ctx, int enable) {
    if (!ctx) return JPEGENC_E_BAD_ARG;
    ctx->profiling = enable != 0;
    return JPEGENC_OK;
}

int jpegenc_kernel_times(jpegenc_ctx* ctx, uint64_t launches[JPEGENC_N_KERNELS], double ms[JPEGENC_N_KERNELS],
                         int reset) {
    if (!ctx || !launches || !ms) return JPEGENC_E_BAD_ARG;
    DevGuard guard(ctx->device);
    HIPCHK(ctx, guard.err);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& ep : ctx->events) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, ep.a, ep.b) == hipSuccess) {
            ctx->k_launches[ep.kernel] += 1;
            ctx->k_ms[ep.kernel] += double(t);
        }
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    ctx->events.clear();
    for (int i = 0; i < JPEGENC_N_KERNELS; ++i) {
        launches[i] = ctx->k_launches[i];
        ms[i] = ctx->k_ms[i];
        if (reset) {
            ctx->k_launches[i] = 0;
            ctx->k_ms[i] = 0;
        }
    }
    return JPEGENC_OK;
}

const char* jpegenc_kernel_name(int index) {
    return (index >= 0 && index < JPEGENC_N_KERNELS) ? kKernelNames[index] : "";
}

int jpegenc_set_pipeline(jpegenc_ctx* ctx, int mode) {
    if (!ctx || mode < JPEGENC_PIPELINE_AUTO || mode > JPEGENC_PIPELINE_SINGLE_MCU) return JPEGENC_E_BAD_ARG;
    ctx->pipeline = mode;
    return JPEGENC_OK;
}

int jpegenc_region_stats(jpegenc_ctx* ctx, uint64_t counts[5]) {
    if (!ctx || !counts) return JPEGENC_E_BAD_ARG;
    const volatile uint32_t* h = ctx->stats_host;
    counts[0] = counts[1] = counts[3] = 0;
    for (int i = 0; i < 16; ++i) {
        counts[0] += h[16 * i];
        counts[1] += h[16 * i + 1];
        counts[3] += h[16 * i + 4];
    }
    counts[2] = h[16 * 16];
    counts[4] = ctx->regions == kRegionsLarge   ? JPEGENC_REGIONS_LARGE
                : ctx->regions == kRegionsShort ? JPEGENC_REGIONS_SHORT
                                                : JPEGENC_REGIONS_SMALL;
    return JPEGENC_OK;
}

int jpegenc_set_region_mode(jpegenc_ctx* ctx, int mode) {
    if (!ctx || mode < JPEGENC_REGIONS_AUTO || mode > JPEGENC_REGIONS_SHORT) return JPEGENC_E_BAD_ARG;
    ctx->region_mode = mode;
    return JPEGENC_OK;
}

int jpegenc_set_slot_mode(jpegenc_ctx* ctx, int mode) {
    if (!ctx || (mode != JPEGENC_SLOTS_WORST_CASE && mode != JPEGENC_SLOTS_ADAPTIVE)) return JPEGENC_E_BAD_ARG;
    ctx->slot_mode = mode;
    return JPEGENC_OK;
}

size_t jpegenc_workspace_bytes(const jpegenc_ctx* ctx) {
    if (!ctx) return 0;
    return ctx->slots.n * 4 + ctx->chunk_bits.n * 4 + ctx->chunk_ff.n * 4 + ctx->bit_off.n * 8 + ctx->ff_off.n * 8;
}

int jpegenc_set_workspace_limit(jpegenc_ctx* ctx, size_t bytes) {
    if (!ctx || bytes == 0) return JPEGENC_E_BAD_ARG;
    ctx->ws_limit = bytes;
    return JPEGENC_OK;
}

const char* jpegenc_strerror(int code) {
    switch (code) {
        case JPEGENC_OK: return "ok";
        case JPEGENC_E_BAD_ARG: return "bad argument";
        case JPEGENC_E_TOO_SMALL: return "output buffer too small";
        case JPEGENC_E_HIP: return "HIP error or no usable gfx950 device";
        case JPEGENC_E_UNSUPPORTED: return "unsupported configuration";
        case JPEGENC_E_NOMEM: return "out of memory";
        case JPEGENC_E_STATE: return "calls out of order (port protocol or segment calls)";
        case JPEGENC_E_LIMIT: return "call exceeds the static workspace of jpegenc_encode_batch_noalloc";
        case JPEGENC_E_AGAIN: return "a group of blocks outgrew its workspace slot (JPEGENC_SLOTS_ADAPTIVE): repeat the call";
        default: return "unknown error";
    }
}

const char* jpegenc_last_error(const jpegenc_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

#ifndef JPEGENC_KERNELS_SHA
#define JPEGENC_KERNELS_SHA "unknown"
#endif
const char* jpegenc_version(void) { return "jpegenc-gfx950 0.3 kernels " JPEGENC_KERNELS_SHA; }

}  // extern "C"
