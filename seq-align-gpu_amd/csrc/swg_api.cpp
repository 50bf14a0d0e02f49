// swg_api.cpp -- the C-ABI of include/swg.h over the gfx950 kernels.
//
// Replaces, for one query against a whole database, the reference's timed
// region `#pragma omp parallel for ... alignment_fill_matrices(aligners[i])`
// (src/alignment_cmdline.c:503-509) and the aligner_create/aligner_update
// bookkeeping in front of it (src/alignment.c:190-233).  There is no CPU
// fallback in this file: without a GPU swg_create fails with SWG_ERR_NODEVICE.
#include "swg_host_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;
// (query, scoring) epochs are unique in the process, not per context: a database keeps hints keyed by the epoch
// (f16_veto_epoch), and two contexts that both counted from 1 would hand each other's hints on
static std::atomic<uint64_t> g_epoch{1};

int swg_set_global_error(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int swg_set_ctx_error(swg_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return swg_set_ctx_error(ctx, e_ == hipErrorOutOfMemory ? SWG_ERR_NOMEM : SWG_ERR_HIP, \
                                     "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                                     __FILE__, __LINE__);                                       \
    } while (0)

// Wait for everything queued on the stream by polling an event from user space.  A blocking
// hipStreamSynchronize may put the thread to sleep; on a busy host the wake-up alone can cost
// milliseconds per search, several times the fill itself.
static hipError_t spin_sync(swg_ctx *ctx, hipStream_t s)
{
    hipError_t e = hipEventRecord(ctx->cur->ev_done, s);
    if (e != hipSuccess) return e;
    for (;;) {
        e = hipEventQuery(ctx->cur->ev_done);
        if (e != hipErrorNotReady) return e;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

// No C++ exception crosses the ABI (include/swg.h:9-12): the bodies below touch std containers -- the plan cache of a
// database (std::map), the fall-back key vector of swg_search_end (up to n_local keys) -- so every hot entry point
// runs its body inside this guard; an allocation failure or any other std::exception becomes SWG_ERR_NOMEM with
// the text in swg_last_error.  (The reference asserts / exits instead: src/alignment.c:63-66.)
template <class F> static int ctx_guarded(swg_ctx *ctx, const char *what, F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "%s: out of host memory", what);
    } catch (const std::exception &e) {
        return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "%s: %s", what, e.what());
    }
}

// Test hook (csrc/swg_host_internal.h, not part of the public ABI): the next time the named site is reached it
// throws std::bad_alloc as a failed allocation there would.  Sites: 1 = body of swg_search_begin, 2 = body of
// swg_search_end, 3 = the fall-back key vector of swg_search_end (reached only when the device top-K could not be
// used).  One shot: the hook clears itself when it fires.
static int g_fail_alloc_site = 0;
extern "C" void swg_debug_fail_alloc(int site) { g_fail_alloc_site = site; }
static inline void fail_alloc_here(int site)
{
    if (g_fail_alloc_site == site) {
        g_fail_alloc_site = 0;
        throw std::bad_alloc();
    }
}

extern "C" const char *swg_last_error(const swg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }
extern "C" const char *swg_global_error(void) { return g_err.c_str(); }
extern "C" int swg_abi_version(void) { return SWG_ABI_VERSION; }

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
// The long-pair kernel must run BESIDE the bulk kernel: a stream of its own priority level gets its own hardware
// queue even when other runtimes in the process (RCCL, torch) have used up the default queues, and its workgroups are
// dispatched first.  Created by the first search whose plan has two classes.
static int ensure_stream2(swg_ctx *ctx)
{
    if (ctx->stream2) return SWG_OK;
    int least = 0, greatest = 0;
    HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, greatest));
    return SWG_OK;
}

// Top-K and read-out of a finished fill run on a stream of their own, beside the NEXT search's fill: there is no next
// search before the context's second one, which is when the stream is made (the first search's read-out follows its
// fill on the fill stream).
static int ensure_stream3(swg_ctx *ctx)
{
    if (ctx->stream3) return SWG_OK;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
    return SWG_OK;
}

// Events and pinned landing buffers of one in-flight search, made when the slot is first used.
static int ensure_slot(swg_ctx *ctx, SwgSlot *sl)
{
    if (sl->ev_done) return SWG_OK;
    for (auto &ev : sl->ev)
        if (!ev) HIP_TRY(ctx, hipEventCreate(&ev));
    if (!sl->h_cand) HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&sl->h_cand), SWG_TOPK_CAND_CAP * 8, hipHostMallocDefault));
    if (!sl->h_counters) HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&sl->h_counters), 128, hipHostMallocDefault));
    HIP_TRY(ctx, hipEventCreateWithFlags(&sl->ev_done, hipEventDisableTiming)); // (last: marks the slot complete)
    return SWG_OK;
}

extern "C" int swg_create(const swg_config *cfg, swg_ctx **out)
{
    if (!out) return swg_set_global_error(SWG_ERR_ARG, "swg_create: out is NULL");
    *out = nullptr;
    const int dev = cfg ? cfg->device : 0;
    int n = 0;
    const std::chrono::steady_clock::time_point t_enter = std::chrono::steady_clock::now();
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return swg_set_global_error(SWG_ERR_NODEVICE,
                                    "swg_create: no HIP device (%s); libswg has no CPU backend",
                                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (dev < 0 || dev >= n)
        return swg_set_global_error(SWG_ERR_ARG, "swg_create: device %d out of range (have %d)", dev, n);
    swg_ctx *ctx = new (std::nothrow) swg_ctx();
    if (!ctx) return swg_set_global_error(SWG_ERR_NOMEM, "swg_create: out of memory");
    ctx->device = dev;
    ctx->epoch = g_epoch.fetch_add(1) + 1;
    memset(ctx->sub, 0, sizeof ctx->sub);
    // SWG_TIMING=1: where the wall time of this call goes (most of a one-shot tool run is here: the HIP runtime's
    // own start-up, which the first HIP call of the process pays -- hipGetDeviceCount above)
    typedef std::chrono::steady_clock clk;
    const bool timing = getenv("SWG_TIMING") != nullptr;
    clk::time_point tp = clk::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const clk::time_point t = clk::now();
        fprintf(stderr, "[swg_create] %-44s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tp).count());
        tp = t;
    };
    if (timing) fprintf(stderr, "[swg_create] %-44s %8.2f ms\n", "hipGetDeviceCount (HIP runtime start-up)", std::chrono::duration<double, std::milli>(tp - t_enter).count());
    int rc = [&]() -> int {
        HIP_TRY(ctx, hipSetDevice(dev));
        hipDeviceProp_t prop;
        HIP_TRY(ctx, hipGetDeviceProperties(&prop, dev));
        ctx->n_cu = prop.multiProcessorCount;
        lap("hipSetDevice + device properties");
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        lap("the fill stream");
        // (the long class's stream and the read-out stream: ensure_stream2 / ensure_stream3, when a search first needs
        // them -- a stream is ~18 ms of runtime work, and a one-shot tool run is start-up bound)
        // slot 0 (what swg_search uses); the other in-flight slots get their events and pinned buffers when
        // swg_search_begin first hands them out (a tool that searches once never pays for them)
        const int rs = ensure_slot(ctx, &ctx->slots[0]);
        if (rs != SWG_OK) return rs;
        lap("events + pinned buffers of one search slot");
        ctx->cur = &ctx->slots[0];
        HIP_TRY(ctx, hipMalloc(&ctx->d_sub, 32 * 32));
        lap("first device allocation");
        if (timing) {
            // the library's code object (~350 kernel instantiations) is loaded by the first launch of any of its kernels
            HIP_TRY(ctx, swg_launch_zero2(ctx->d_sub, 16, reinterpret_cast<uint8_t *>(ctx->d_sub) + 16, 16, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            lap("first kernel launch (code object load)");
        }
        return SWG_OK;
    }();
    if (rc != SWG_OK) {
        g_err = ctx->err;
        swg_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return SWG_OK;
}

extern "C" void swg_destroy(swg_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->d_sub);
    (void)hipFree(ctx->d_query);
    (void)hipFree(ctx->d_profile[0]);
    (void)hipFree(ctx->d_profile[1]);
    (void)hipFree(ctx->d_profile[2]);
    (void)hipFree(ctx->d_profile[3]);
    (void)hipFree(ctx->d_profile[4]);
    (void)hipFree(ctx->d_profile[5]);
    (void)hipFree(ctx->d_profile[6]);
    (void)hipFree(ctx->d_profile[7]);
    (void)hipFree(ctx->d_scratch);
    for (SwgSlot &sl : ctx->slots) {
        for (auto &ev : sl.ev)
            if (ev) (void)hipEventDestroy(ev);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        (void)hipHostFree(sl.h_cand);
        (void)hipHostFree(sl.h_counters);
        (void)hipHostFree(sl.h_scores);
    }
    if (ctx->b16.db) swg_db_free(ctx->b16.db);
    (void)hipHostFree(ctx->b16.h_stage);
    (void)hipHostFree(ctx->b16.h_meta);
    (void)hipFree(ctx->b16.d_stage);
    (void)hipFree(ctx->b16.d_pair_src);
    (void)hipFree(ctx->b16.d_pair_len);
    for (int i = 0; i < 4; ++i) {
        (void)hipHostFree(ctx->h_query_stage[i]);
        if (ctx->ev_query_stage[i]) (void)hipEventDestroy(ctx->ev_query_stage[i]);
    }
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int swg_set_option(swg_ctx *ctx, const char *key, long value)
{
    if (!ctx || !key) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_set_option: NULL argument");
    if (!strcmp(key, "force_bits")) {
        if (value != 0 && value != 16 && value != 32)
            return swg_set_ctx_error(ctx, SWG_ERR_ARG, "force_bits must be 0, 16 or 32");
        ctx->opt_force_bits = value;
    } else if (!strcmp(key, "cols_per_wave")) {
        ctx->opt_cols = value;
    } else if (!strcmp(key, "max_waves")) {
        if (value < 0 || value > 16) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "max_waves must be 0..16");
        ctx->opt_max_waves = value;
    } else if (!strcmp(key, "engine")) {
        if (value < 0 || value > 2) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "engine must be 0 (auto), 1 (systolic) or 2 (diagonal)");
        ctx->opt_engine = value;
    } else if (!strcmp(key, "group_lanes")) {
        if (value != 0 && value != 16 && value != 32 && value != 64)
            return swg_set_ctx_error(ctx, SWG_ERR_ARG, "group_lanes must be 0, 16, 32 or 64");
        ctx->opt_group = value;
    } else if (!strcmp(key, "long_cols")) {
        extern long g_swg_long_cols;
        g_swg_long_cols = value;
    } else if (!strcmp(key, "long_group")) {
        extern long g_swg_long_group;
        g_swg_long_group = value;
    } else if (!strcmp(key, "autotune")) {
        ctx->opt_autotune = value != 0;
    } else if (!strcmp(key, "side_readout")) {
        ctx->opt_side_readout = value != 0;
    } else if (!strcmp(key, "wide16")) {
        ctx->opt_wide = value != 0;
    } else if (!strcmp(key, "f16")) {
        if (value < 0 || value > 2) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "f16 must be 0 (off), 1 (auto) or 2 (whenever the gap scores allow)");
        ctx->opt_f16 = value;
    } else if (!strcmp(key, "last_pass")) {
        ctx->opt_last_pass = value != 0;
    } else if (!strcmp(key, "qq")) {
        ctx->opt_qq = value != 0;
    } else if (!strcmp(key, "long_helps")) {
        ctx->opt_long_helps = value != 0;
    } else if (!strcmp(key, "segment_blocks")) {
        // (tests: the multi-pass fill cuts its launches into segments of at most this many token blocks)
        if (value < 0 || value > (long)SWG_DYN_SEG_BLOCKS)
            return swg_set_ctx_error(ctx, SWG_ERR_ARG, "segment_blocks must be 0 (default) .. 2^26-64");
        ctx->opt_seg_blocks = value ? (uint32_t)value : SWG_DYN_SEG_BLOCKS;
    } else if (!strcmp(key, "q32_waves")) {
        ctx->opt_q32_waves = value;
    } else if (!strcmp(key, "wave_budget")) {
        ctx->opt_wave_budget = value;
    } else if (!strcmp(key, "batch")) {
        if (value < 0 || value > 8) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "batch must be 0..8 (pairs one queue request claims; 0 and 1: one)");
        ctx->opt_batch = value;
    } else if (!strcmp(key, "batch_blocks")) {
        if (value < 0) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "batch_blocks must be >= 0 (token blocks; 0: from the geometry)");
        ctx->opt_batch_blocks = value;
    } else if (!strcmp(key, "work_queue")) {
        ctx->opt_dynamic = value != 0;
    } else if (!strcmp(key, "prio_share")) {
        if (value < 0) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "prio_share must be >= 0 (percent)");
        ctx->opt_prio_share = value;
    } else if (!strcmp(key, "long_split")) {
        if (value < -1) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "long_split must be -1 (off), 0 (auto) or a row count");
        ctx->opt_long_split = value;
    } else if (!strcmp(key, "workgroups")) {
        if (value < 0) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "workgroups must be >= 0");
        ctx->opt_workgroups = value;
    } else {
        return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_set_option: unknown key '%s'", key);
    }
    return SWG_OK;
}

extern "C" int swg_set_scoring(swg_ctx *ctx, const int8_t sub[32][32], int gap_open, int gap_extend)
{
    if (!ctx || !sub) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_set_scoring: NULL argument");
    if (gap_open < -32768 || gap_open > 32767 || gap_extend < -32768 || gap_extend > 32767)
        return swg_set_ctx_error(ctx, SWG_ERR_ARG,
                                 "swg_set_scoring: gap scores must fit the reference's int16 score_t");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    memcpy(ctx->sub, sub, 32 * 32);
    ctx->gap_open = gap_open;
    ctx->gap_extend = gap_extend;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_sub, ctx->sub, 32 * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_scoring = true;
    ctx->epoch = g_epoch.fetch_add(1) + 1;
    return SWG_OK;
}

extern "C" int swg_set_query(swg_ctx *ctx, const int8_t *idx, size_t lq)
{
    if (!ctx || !idx || lq == 0)
        return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_set_query: NULL or empty query");
    if (lq > (1u << 24)) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_set_query: query too long");
    for (size_t i = 0; i < lq; ++i)
        if (idx[i] < 1 || idx[i] > 31)
            return swg_set_ctx_error(ctx, SWG_ERR_RESIDUE,
                                     "swg_set_query: residue index %d at %zu outside 1..31", idx[i], i);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (lq > ctx->d_query_cap) {
        (void)hipFree(ctx->d_query);
        ctx->d_query = nullptr;
        ctx->d_query_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_query, lq));
        ctx->d_query_cap = lq;
    }
    // No wait here: the copy and the profile build that consumes it are ordered on the context's
    // stream behind any search still in flight, so a caller can stream queries against a resident
    // database (set_query, search_begin, set_query, search_begin, search_end, ...).  The copy reads a
    // pinned staging buffer of its own (four in rotation, each guarded by an event): the host copy
    // ctx->query is rewritten by the next call while this one's transfer may still be queued.
    ctx->query.assign(idx, idx + lq);
    {
        const int b = ctx->query_stage_next;
        ctx->query_stage_next = (b + 1) % 4;
        if (!ctx->ev_query_stage[b]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_query_stage[b], hipEventDisableTiming));
        else HIP_TRY(ctx, hipEventSynchronize(ctx->ev_query_stage[b])); // (four transfers ago: long done)
        if (lq > ctx->h_query_stage_cap[b]) {
            (void)hipHostFree(ctx->h_query_stage[b]);
            ctx->h_query_stage[b] = nullptr;
            ctx->h_query_stage_cap[b] = 0;
            HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_query_stage[b]), lq, hipHostMallocDefault));
            ctx->h_query_stage_cap[b] = lq;
        }
        memcpy(ctx->h_query_stage[b], idx, lq);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_query, ctx->h_query_stage[b], lq, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_query_stage[b], ctx->stream));
    }
    ctx->epoch = g_epoch.fetch_add(1) + 1;
    return SWG_OK;
}

// ---------------------------------------------------------------------------
// database residency
// ---------------------------------------------------------------------------
void swg_db_release_device(swg_db *db)
{
    if (!db || db->device < 0) return;
    (void)hipSetDevice(db->device);
    (void)hipFree(db->d_codes);
    (void)hipFree(db->d_code_off);
    (void)hipFree(db->d_lens);
    (void)hipFree(db->d_packed);
    (void)hipFree(db->d_bin_off);
    (void)hipFree(db->d_bin_nblk);
    (void)hipFree(db->d_order);
    for (swg_db::Bufs &b : db->bufs) {
        (void)hipFree(b.d_scores);
        (void)hipFree(b.d_list);
        (void)hipFree(b.d_counters);
        (void)hipFree(b.d_keys);
        (void)hipFree(b.d_hist);
        b = swg_db::Bufs();
    }
    (void)hipFree(db->ptok.d_tok);
    (void)hipFree(db->ptok.d_pair_off);
    (void)hipFree(db->ptok.d_edge[0]);
    (void)hipFree(db->ptok.d_edge[1]);
    (void)hipFree(db->ptok.d_edge32[0]);
    (void)hipFree(db->ptok.d_edge32[1]);
    (void)hipFree(db->ptok.d_edge32d[0]);
    (void)hipFree(db->ptok.d_edge32d[1]);
    db->ptok = SwgPairTokens();
    for (SwgDiagLayout &L : db->diag) {
        (void)hipFree(L.d_tok);
        (void)hipFree(L.d_stream_off);
        (void)hipFree(L.d_stream_pairs);
        (void)hipFree(L.d_stream_pair_off);
        (void)hipFree(L.d_scratch);
        L = SwgDiagLayout();
    }
    db->d_codes = nullptr;
    db->d_code_off = nullptr;
    db->d_lens = nullptr;
    db->d_packed = nullptr;
    db->d_bin_off = nullptr;
    db->d_bin_nblk = nullptr;
    db->d_order = nullptr;
    db->d_scores = nullptr;
    db->d_list = nullptr;
    db->d_counters = nullptr;
    db->d_keys = nullptr;
    db->d_hist = nullptr;
    db->device = -1;
}

// Output buffers of in-flight slot `slot` (allocated on first use) become the current ones.
static int select_bufs(swg_ctx *ctx, swg_db *db, int slot)
{
    swg_db::Bufs &b = db->bufs[slot];
    const size_t ns = (size_t)db->n_bins * SWG_BIN;
    if (!b.d_scores) {
        HIP_TRY(ctx, hipMalloc(&b.d_scores, std::max<size_t>(4, ns * 4)));
        HIP_TRY(ctx, hipMalloc(&b.d_list, std::max<size_t>(4, ns * 8)));
        HIP_TRY(ctx, hipMalloc(&b.d_counters, SWG_COUNTER_BYTES));
        HIP_TRY(ctx, hipMalloc(&b.d_keys, SWG_TOPK_CAND_CAP * 8));
        HIP_TRY(ctx, hipMalloc(&b.d_hist, 4096 * 4));
    }
    db->d_scores = b.d_scores;
    db->d_list = b.d_list;
    db->d_counters = b.d_counters;
    db->d_keys = b.d_keys;
    db->d_hist = b.d_hist;
    return SWG_OK;
}

// What crosses PCIe: one byte per residue (whole dwords per sequence) and 16 bytes per slot.  The
// kernels' own layouts are built from that on the device: the pair tokens on the first search that
// uses the diagonal engine's work queue (ensure_pair_tokens), the bin image only if an engine that
// reads bins is ever used (ensure_bins).
extern "C" int swg_db_upload(swg_ctx *ctx, swg_db *db)
{
    if (!ctx || !db) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_db_upload: NULL argument");
    swg_db_release_device(db);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    db->device = ctx->device;
    const size_t nb = db->n_bins, ns = nb * SWG_BIN;
    int rc = [&]() -> int {
        std::vector<uint64_t> off_dw;
        try {
            off_dw.resize(ns + 1);
        } catch (const std::exception &) {
            return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_db_upload: out of host memory");
        }
        for (size_t i = 0; i <= ns; ++i) off_dw[i] = db->code_off[i] / 4;
        HIP_TRY(ctx, hipMalloc(&db->d_codes, std::max<size_t>(4, db->codes.size())));
        HIP_TRY(ctx, hipMalloc(&db->d_code_off, (ns + 1) * 8));
        HIP_TRY(ctx, hipMalloc(&db->d_lens, std::max<size_t>(4, ns * 4)));
        HIP_TRY(ctx, hipMalloc(&db->d_order, std::max<size_t>(4, ns * 4)));
        int rb = select_bufs(ctx, db, 0);
        if (rb != SWG_OK) return rb;
        if (!db->codes.empty())
            HIP_TRY(ctx, hipMemcpyAsync(db->d_codes, db->codes.data(), db->codes.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(db->d_code_off, off_dw.data(), (ns + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (ns) {
            HIP_TRY(ctx, hipMemcpyAsync(db->d_lens, db->lens.data(), ns * 4, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(db->d_order, db->order.data(), ns * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // off_dw goes out of scope
        db->upload_bytes = db->codes.size() + (ns + 1) * 8 + ns * 8;
        return SWG_OK;
    }();
    if (rc != SWG_OK) swg_db_release_device(db);
    return rc;
}

// The bin image: built on the device from the residue dwords the first time an engine that reads
// bins is used on this database (the default engine never does).
static int ensure_bins(swg_ctx *ctx, swg_db *db)
{
    if (db->d_packed || db->n_bins == 0) return SWG_OK;
    if (db->tokens_only)
        return swg_set_ctx_error(ctx, SWG_ERR_STATE, "this search needs the bin image, which a database built from 16-lane batches does not have");
    const size_t nb = db->n_bins;
    const uint64_t dwords = db->bin_off[nb - 1] + (uint64_t)db->bin_nblk[nb - 1] * SWG_BIN;
    HIP_TRY(ctx, hipMalloc(&db->d_bin_off, nb * 8));
    HIP_TRY(ctx, hipMalloc(&db->d_bin_nblk, nb * 4));
    HIP_TRY(ctx, hipMemcpyAsync(db->d_bin_off, db->bin_off.data(), nb * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(db->d_bin_nblk, db->bin_nblk.data(), nb * 4, hipMemcpyHostToDevice, ctx->stream));
    uint32_t *packed = nullptr;
    HIP_TRY(ctx, hipMalloc(&packed, std::max<uint64_t>(4, dwords * 4)));
    hipError_t e = swg_launch_build_bins(db->d_codes, db->d_code_off, db->d_lens, db->d_bin_off, db->d_bin_nblk,
                                         (uint32_t)nb, packed, ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(packed);
        HIP_TRY(ctx, e);
    }
    db->d_packed = packed;
    return SWG_OK;
}

// ---------------------------------------------------------------------------
// planning
// ---------------------------------------------------------------------------
struct Plan {
    int bits, variant, K, W, npass, workgroups;
    SwgKernelInfo info;
    int f16; // systolic int16 plan on the packed-f16 cells (one pass, no score of the search can reach 4096)
};

static int make_plan(swg_ctx *ctx, int bits, uint32_t n_items, Plan *pl)
{
    const int nv = swg_num_variants(bits);
    int variant = 0;
    if (ctx->opt_cols > 0) {
        variant = -1;
        for (int v = 0; v < nv; ++v)
            if (swg_variant_info(bits, v).K == (int)ctx->opt_cols) variant = v;
        if (variant < 0)
            return swg_set_ctx_error(ctx, SWG_ERR_ARG, "cols_per_wave=%ld not built for the %d-bit path",
                                     ctx->opt_cols, bits);
    }
    const SwgKernelInfo info = swg_variant_info(bits, variant);
    int maxw = info.max_waves;
    if (ctx->opt_max_waves > 0) maxw = std::min<int>(maxw, (int)ctx->opt_max_waves);
    const size_t lq = ctx->query.size();
    const size_t cols_per_pass_max = (size_t)maxw * info.K;
    const int npass = (int)((lq + cols_per_pass_max - 1) / cols_per_pass_max);
    const int W = (int)((lq + (size_t)npass * info.K - 1) / ((size_t)npass * info.K));
    // residency: info.max_waves is also the wave budget of one CU for this
    // instantiation's register allocation; LDS is the other limit
    const size_t lds = info.lds_per_wave * (size_t)W + info.lds_fixed;
    int per_cu = std::max(1, info.max_waves / W);
    per_cu = std::max(1, std::min<int>(per_cu, (int)((160 * 1024) / lds)));
    long wgs = (long)ctx->n_cu * per_cu;
    if (ctx->opt_workgroups > 0) wgs = ctx->opt_workgroups;
    wgs = std::max<long>(1, std::min<long>(wgs, (long)n_items));
    pl->bits = bits;
    pl->variant = variant;
    pl->K = info.K;
    pl->W = W;
    pl->npass = npass;
    pl->workgroups = (int)wgs;
    pl->info = info;
    return SWG_OK;
}

// tail_cols > 0: the last tail_cols layout columns (one pass) have a slice geometry of their own, tail_k_real query
// columns in tail_k_padded layout columns per lane, and begin at query column tail_qcol0.
static int ensure_profile_cols(swg_ctx *ctx, int which, uint32_t ncols, int elem_size, uint64_t geom, int k_real = 1,
                               int k_padded = 1, int chunk_cols = 4, int swizzle_lanes = 0, int f16 = 0, uint32_t tail_cols = 0,
                               int tail_k_real = 0, int tail_k_padded = 0, uint32_t tail_qcol0 = 0)
{
    const size_t bytes = (size_t)ncols * 32 * elem_size;
    // (query, scoring) epoch and geometry: the epoch is spread over all 64 bits so that no geometry field can alias it
    const uint64_t tag = (ctx->epoch * 0x9E3779B97F4A7C15ull) ^ geom ^ ((uint64_t)swizzle_lanes << 56) ^ ((uint64_t)chunk_cols << 60) ^
                         ((uint64_t)(f16 != 0) << 53) ^ (((uint64_t)tail_cols * 0xD6E8FEB86659FD93ull) ^ ((uint64_t)tail_k_real << 24));
    if (ctx->profile_tag[which] == tag && ctx->d_profile[which]) return SWG_OK;
    if (bytes > ctx->d_profile_cap[which]) {
        (void)hipFree(ctx->d_profile[which]);
        ctx->d_profile[which] = nullptr;
        ctx->d_profile_cap[which] = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_profile[which], bytes));
        ctx->d_profile_cap[which] = bytes;
    }
    HIP_TRY(ctx, swg_launch_build_profile(ctx->d_sub, ctx->d_query, (uint32_t)ctx->query.size(), ncols - tail_cols,
                                          elem_size, chunk_cols, k_real, k_padded, ctx->d_profile[which], ctx->stream,
                                          swizzle_lanes, f16));
    if (tail_cols > 0)
        HIP_TRY(ctx, swg_launch_build_profile(ctx->d_sub, ctx->d_query, (uint32_t)ctx->query.size(), tail_cols, elem_size, chunk_cols,
                                              tail_k_real, tail_k_padded, ctx->d_profile[which] + (size_t)(ncols - tail_cols) * 32 * elem_size,
                                              ctx->stream, swizzle_lanes, f16, tail_qcol0));
    ctx->profile_tag[which] = tag;
    return SWG_OK;
}

static int ensure_profile(swg_ctx *ctx, const Plan &pl)
{
    return ensure_profile_cols(ctx, pl.bits == 16 ? 0 : 1, (uint32_t)(pl.npass * pl.W * pl.K), pl.info.elem_size,
                               ((uint64_t)pl.K << 20) ^ ((uint64_t)pl.W << 12) ^ (uint64_t)pl.npass, 1, 1, 4, 0, pl.bits == 16 && pl.f16);
}

// Stream layout of the diagonal engine for this database at this stream count,
// built on the host and kept resident until the geometry changes.
static int ensure_diag_layout(swg_ctx *ctx, swg_db *db, int cls, const SwgDiagPlan &pl, uint64_t pair_begin,
                              uint64_t pair_end)
{
    SwgDiagLayout &L = db->diag[cls];
    const uint32_t spw = (uint32_t)(pl.W * (64 / pl.G));
    if (L.n_streams != pl.n_streams || L.streams_per_wg != spw || L.pair_begin != pair_begin ||
        L.pair_end != pair_end || !L.d_tok) {
        (void)hipFree(L.d_tok);
        (void)hipFree(L.d_stream_off);
        (void)hipFree(L.d_stream_pairs);
        (void)hipFree(L.d_stream_pair_off);
        (void)hipFree(L.d_scratch);
        L = SwgDiagLayout();
        try {
            swg_build_diag_layout(db, pair_begin, pair_end, pl.n_streams, spw, &L);
            L.streams_per_wg = spw;
        } catch (const std::bad_alloc &) {
            L = SwgDiagLayout();
            return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "diagonal layout: out of host memory");
        }
        const size_t S = L.n_streams;
        HIP_TRY(ctx, hipMalloc(&L.d_tok, std::max<size_t>(16, L.tok.size() * 4)));
        HIP_TRY(ctx, hipMalloc(&L.d_stream_off, (S + 1) * 8));
        HIP_TRY(ctx, hipMalloc(&L.d_stream_pairs, std::max<size_t>(4, L.stream_pairs.size() * 4)));
        HIP_TRY(ctx, hipMalloc(&L.d_stream_pair_off, (S + 1) * 4));
        if (!L.tok.empty())
            HIP_TRY(ctx, hipMemcpyAsync(L.d_tok, L.tok.data(), L.tok.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(L.d_stream_off, L.stream_off.data(), (S + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (!L.stream_pairs.empty())
            HIP_TRY(ctx, hipMemcpyAsync(L.d_stream_pairs, L.stream_pairs.data(), L.stream_pairs.size() * 4,
                                        hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(L.d_stream_pair_off, L.stream_pair_off.data(), (S + 1) * 4,
                                    hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t>().swap(L.tok); // the device copy is the one that is used
    }
    if (pl.npass > 1 && L.d_scratch_rows < L.total_blocks * 4) {
        (void)hipFree(L.d_scratch);
        L.d_scratch = nullptr;
        L.d_scratch_rows = 0;
        HIP_TRY(ctx, hipMalloc(&L.d_scratch, std::max<size_t>(8, L.total_blocks * 4 * sizeof(uint2))));
        L.d_scratch_rows = L.total_blocks * 4;
    }
    return SWG_OK;
}

// Pair-major tokens, built once per database on first use by the diagonal engine: the block
// offsets of the pairs on the host (they follow from the lengths alone, and the planner wants them
// too), the tokens themselves on the device from the resident residue dwords.
static int ensure_pair_tokens(swg_ctx *ctx, swg_db *db)
{
    SwgPairTokens &T = db->ptok;
    if (T.tried) return SWG_OK;
    T.tried = true;
    try {
        if (swg_build_pair_tokens(db, nullptr, nullptr, &T.pair_blocks_prefix) != 0) return SWG_OK; // too large: static streams
    } catch (const std::exception &) {
        return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "pair tokens: out of host memory");
    }
    T.total_blocks = T.pair_blocks_prefix.back();
    const uint32_t n_pairs = (uint32_t)(T.pair_blocks_prefix.size() - 1);
    // (one more block, all zeros, behind the last pair: what lanes that feed no pair read)
    HIP_TRY(ctx, hipMalloc(&T.d_tok, ((size_t)T.total_blocks + 1) * 16));
    HIP_TRY(ctx, hipMemsetAsync(T.d_tok + T.total_blocks, 0, 16, ctx->stream));
    HIP_TRY(ctx, hipMalloc(&T.d_pair_off, T.pair_blocks_prefix.size() * 4));
    // (the host vector lives as long as the database: no wait needed for the copy)
    HIP_TRY(ctx, hipMemcpyAsync(T.d_pair_off, T.pair_blocks_prefix.data(), T.pair_blocks_prefix.size() * 4,
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, swg_launch_build_tokens(db->d_codes, db->d_code_off, db->d_lens, T.d_pair_off, n_pairs, T.total_blocks,
                                         T.d_tok, ctx->stream));
    T.ok = true;
    return SWG_OK;
}

// Test hook (not part of the public ABI, declared in swg_host_internal.h): the pair-token image of
// a resident database as the device built it (from_host = 0) or as the host restatement of the
// same layout builds it (from_host = 1).  *n_dwords = size of the image; copied when it fits cap.
extern "C" int swg_debug_pair_tokens(swg_ctx *ctx, swg_db *db, int from_host, uint32_t *out, size_t cap_dwords,
                                     size_t *n_dwords)
{
    if (!ctx || !db || !n_dwords) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_debug_pair_tokens: NULL argument");
    *n_dwords = 0;
    if (from_host) {
        std::unique_ptr<uint32_t[]> tok;
        std::vector<uint32_t> off;
        size_t n = 0;
        try {
            if (swg_build_pair_tokens(db, &tok, &n, &off) != 0)
                return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_debug_pair_tokens: database too large for pair tokens");
        } catch (const std::exception &) {
            return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_debug_pair_tokens: out of host memory");
        }
        *n_dwords = n;
        if (out && n <= cap_dwords) memcpy(out, tok.get(), n * 4);
        return SWG_OK;
    }
    if (db->device != ctx->device || !db->d_codes)
        return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_debug_pair_tokens: database is not resident");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = ensure_pair_tokens(ctx, db);
    if (rc != SWG_OK) return rc;
    if (!db->ptok.ok) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_debug_pair_tokens: database too large for pair tokens");
    const size_t n = (size_t)db->ptok.total_blocks * 4;
    *n_dwords = n;
    if (out && n <= cap_dwords && n) {
        HIP_TRY(ctx, hipMemcpyAsync(out, db->ptok.d_tok, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SWG_OK;
}

// Classes take their pairs off a work queue (several passes: one launch per pass, the rows' edges
// go from launch to launch through memory); fixed streams only on request or when the database is
// too large for the queue's 32-bit row indices.
static bool diag_class_is_dynamic(const swg_ctx *ctx, const swg_db *db, const SwgDiagPlan &pl)
{
    if (ctx->opt_dynamic == 0 || !db->ptok.ok) return false;
    return pl.npass == 1 || db->ptok.total_blocks < (1ull << 30);
}

// The cells a class runs on (CellsDiag FORM): packed f16 and the wide form exist in the work-queue kernels
// (the wide form also in the fixed-stream one).
static int diag_class_form(const swg_ctx *ctx, const swg_db *db, const SwgDiagPlan &pl)
{
    if (pl.f16 && diag_class_is_dynamic(ctx, db, pl)) return 2;
    return pl.wide ? 1 : 0;
}

// An integer 0..2048 as an f16 bit pattern (exact), in both halves of a dword.
static uint32_t f16x2_of(int v)
{
    uint32_t b = 0;
    if (v > 0) {
        int e = 0;
        while ((v >> (e + 1)) != 0) ++e; // floor(log2 v), <= 11
        const uint32_t mant = (e <= 10 ? (uint32_t)v << (10 - e) : (uint32_t)v >> (e - 10)) & 0x3FFu;
        b = ((uint32_t)(e + 15) << 10) | mant;
    }
    return b | (b << 16);
}

// Workgroups to launch for class c.  Work-queue kernels are persistent: a workgroup that is not
// resident from the start only gets in when another one has run out of pairs, so when the long
// class runs beside the bulk the bulk leaves it its wave slots (one per SIMD per long workgroup).
static int diag_class_workgroups(const swg_ctx *ctx, const swg_db *db, const SwgDiagWork &wk, int c)
{
    const SwgDiagPlan &pl = wk.plan[c];
    if (c != 0 || !diag_class_is_dynamic(ctx, db, pl)) return pl.workgroups;
    const SwgKernelInfo info = swg_diag_variant_info(pl.variant);
    const size_t lds = swg_diag_dyn_lds_bytes(pl.K, pl.G, pl.W);
    const int per_cu = std::max(1, std::min<int>(info.max_waves / pl.W, (int)((160 * 1024) / lds)));
    if (ctx->opt_wave_budget > 0 && wk.n_classes == 1) // experiment: more resident wavefronts than the planner's 16 per CU
        return std::max(1, ctx->n_cu * std::max(1, std::min<int>((int)ctx->opt_wave_budget / pl.W, (int)((160 * 1024) / lds))));
    const int capacity = ctx->n_cu * per_cu;
    int displaced = 0;
    if (wk.n_classes == 2 && diag_class_is_dynamic(ctx, db, wk.plan[1]))
        displaced = (wk.plan[1].workgroups * wk.plan[1].W + pl.W - 1) / pl.W;
    int wgs = std::min(pl.workgroups, capacity - displaced);
    // three wavefronts per SIMD issue as fast as four and leave the scheduler more room: measured
    // (round 2 kernels, uniform database, tools/sweeps/r2_occ.sh) +1.3 % at K=16, level at K=20 and 23, +1.1 % at K=24
    if (wk.n_classes == 1 && pl.K >= 16 && pl.W == 4 && per_cu == 4) wgs = std::min(wgs, ctx->n_cu * 3);
    return std::max(1, wgs);
}

// 4-row token blocks and lane groups of class c, for the statistics
static uint64_t diag_class_blocks(const swg_ctx *ctx, const swg_db *db, const SwgDiagWork &wk, int c)
{
    if (!diag_class_is_dynamic(ctx, db, wk.plan[c])) return db->diag[c].total_blocks;
    return (uint64_t)db->ptok.pair_blocks_prefix[wk.pair_end[c]] - db->ptok.pair_blocks_prefix[wk.pair_begin[c]];
}
static uint32_t diag_class_streams(const swg_ctx *ctx, const swg_db *db, const SwgDiagWork &wk, int c)
{
    if (!diag_class_is_dynamic(ctx, db, wk.plan[c])) return db->diag[c].n_streams;
    return (uint32_t)(diag_class_workgroups(ctx, db, wk, c) * wk.plan[c].W * (64 / wk.plan[c].G));
}

static int ensure_scratch(swg_ctx *ctx, size_t dwords)
{
    if (dwords <= ctx->d_scratch_cap) return SWG_OK;
    (void)hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr;
    ctx->d_scratch_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_scratch, dwords * 4));
    ctx->d_scratch_cap = dwords;
    return SWG_OK;
}


// ---------------------------------------------------------------------------
// diagonal engine: make a work plan resident, launch it
// ---------------------------------------------------------------------------
// Which profile buffer a class reads: every class has its own ([3] the bulk, [2] the long class) -- a
// lane's slice is its K columns padded to whole chunks, its rows swizzled by the lane's position in a
// group of G: the layout depends on (K, G).  ([0] and [1] are the systolic engine's, in plain order.)
static int diag_profile_slot(const SwgDiagPlan &, int cls) { return cls == 0 ? 3 : 2; }

static int prepare_diag(swg_ctx *ctx, swg_db *db, const SwgDiagWork &wk)
{
    if (ctx->opt_dynamic) {
        int rc = ensure_pair_tokens(ctx, db);
        if (rc != SWG_OK) return rc;
    }
    for (int c = 0; c < wk.n_classes; ++c) {
        const SwgDiagPlan &pl = wk.plan[c];
        if (!diag_class_is_dynamic(ctx, db, pl)) {
            if (db->tokens_only)
                return swg_set_ctx_error(ctx, SWG_ERR_STATE, "this search needs fixed streams, which a database built from 16-lane batches does not have");
            int rc = ensure_diag_layout(ctx, db, c, pl, wk.pair_begin[c], wk.pair_end[c]);
            if (rc != SWG_OK) return rc;
        } else if (pl.npass > 1 && (!db->ptok.d_edge[0] || db->ptok.edge_blocks < db->ptok.total_blocks)) {
            (void)hipFree(db->ptok.d_edge[0]);
            (void)hipFree(db->ptok.d_edge[1]);
            db->ptok.d_edge[0] = db->ptok.d_edge[1] = nullptr;
            const size_t bytes = std::max<size_t>(8, (size_t)db->ptok.total_blocks * 4 * sizeof(uint2));
            HIP_TRY(ctx, hipMalloc(&db->ptok.d_edge[0], bytes));
            HIP_TRY(ctx, hipMalloc(&db->ptok.d_edge[1], bytes));
            db->ptok.edge_blocks = db->ptok.total_blocks;
        }
        const int kp = swg_diag_padded_cols(pl.K);
        // (a last pass with a geometry of its own: its slice follows the other passes' in the same buffer)
        const bool own_last = pl.npass > 1 && pl.last_variant >= 0 && diag_class_is_dynamic(ctx, db, pl);
        const int kp_last = own_last ? swg_diag_padded_cols(pl.last_K) : kp;
        const uint32_t tail = own_last ? (uint32_t)(pl.G * kp_last) : 0u;
        const uint32_t ncols = (uint32_t)((pl.npass - (own_last ? 1 : 0)) * pl.G * kp) + tail;
        const uint32_t qcol0 = (uint32_t)((pl.npass - 1) * pl.G * pl.K);
        int rc = ensure_profile_cols(ctx, diag_profile_slot(pl, c), ncols, 2,
                                     (1ull << 55) | ((uint64_t)pl.K << 40) | ((uint64_t)pl.G << 32) | (uint64_t)ncols, pl.K, kp, 4,
                                     SWG_LDS_SWIZZLE ? pl.G : 0, diag_class_form(ctx, db, pl) == 2, tail, pl.last_K, kp_last, qcol0);
        if (rc != SWG_OK) return rc;
        if (pl.f16_from > 0) { // both forms in this class: the f16 cells' profile of the same geometry
            rc = ensure_profile_cols(ctx, 7, ncols, 2, (1ull << 55) | ((uint64_t)pl.K << 40) | ((uint64_t)pl.G << 32) | (uint64_t)ncols,
                                     pl.K, kp, 4, SWG_LDS_SWIZZLE ? pl.G : 0, 1, tail, pl.last_K, kp_last, qcol0);
            if (rc != SWG_OK) return rc;
        }
    }
    return SWG_OK;
}

// The queue's zones for one launch over pairs [q->q_begin, q->q_end) (see the kernel's event code): the pairs of at most
// N token blocks (see below) -- lengths do not increase along the range -- are claimed opt_batch at a time, except
// the last two per lane group, which go out one by one again so that the lane groups still end together (a claim of
// B pairs is B times the granularity of the hand-out).  Whole batches only; shard c's first U1 requests are single.
static void dyn_batch_zones(const swg_ctx *ctx, const SwgPairTokens &T, SwgDiagDynParams *q, uint64_t groups, int K, int G, int form)
{
    q->batch_u1 = q->batch_u2 = q->batch_B = 0u;
    const uint32_t B = (uint32_t)ctx->opt_batch;
    if (B <= 1u || q->list || q->q_end <= q->q_begin) return;
    const std::vector<uint32_t> &pre = T.pair_blocks_prefix;
    // "Short" is a time, not a length: a request costs ~2.2 us, so batches pay where a whole pair takes a few tens of
    // microseconds -- and a claim of B pairs is B pair-times taken out of the balance at the end of the launch.  A 4-row
    // block costs 4 x (instructions per row) x 4.06 cycles of its SIMD, shared with the other wavefronts on it: 1.3 us
    // at K = 2, 3 us at K = 8, 7 us at K = 32 (three wavefronts).  Pairs of at most 40 us count as short: 30 blocks at
    // K = 2, 13 at K = 8, 5 at K = 32.  (Measured, round 4: a fixed 32 blocks cost config 3 -- K = 32 -- 6 %: 2 300
    // claims of eight 30-block pairs, 1.5 ms of work each, ended after everybody else.)  Option batch_blocks > 0 overrides.
    uint32_t N = (uint32_t)std::min<long>(ctx->opt_batch_blocks, 1l << 30);
    if (ctx->opt_batch_blocks <= 0) {
        const double waves_per_simd = std::min(4.0, std::max(1.0, (double)groups * G / 64.0 / (4.0 * ctx->n_cu)));
        const double block_us = 4.0 * ((form == 2 ? 8.5 : 10.0) * K + 30.0) * 4.06 * waves_per_simd / 2.4e3;
        N = (uint32_t)std::max(2.0, 40.0 / block_us);
    }
    uint32_t lo = q->q_begin, hi = q->q_end; // first pair with at most N blocks
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (pre[mid + 1] - pre[mid] <= N) hi = mid; else lo = mid + 1u;
    }
    const uint64_t tail = std::min<uint64_t>(q->q_end - lo, 2ull * groups);
    const uint64_t to = (uint64_t)q->q_end - tail;
    const uint64_t u1 = ((uint64_t)(lo - q->q_begin) + SWG_DYN_SHARDS - 1u) / SWG_DYN_SHARDS;
    const uint64_t p1 = (uint64_t)q->q_begin + SWG_DYN_SHARDS * u1;
    if (to <= p1) return;
    const uint64_t u2 = (to - p1) / ((uint64_t)SWG_DYN_SHARDS * B);
    if (u2 == 0u) return;
    q->batch_u1 = (uint32_t)u1;
    q->batch_u2 = (uint32_t)u2;
    q->batch_B = B;
}

// Launches the fill of one work plan.  Events: ev[1] before, ev[2] after on the main stream;
// with a long class also ev[5] (bulk end) and ev[7] (long end).
static int launch_diag(swg_ctx *ctx, const swg_db *db, const SwgDiagWork &wk, int go, int ge, bool *two_ends)
{
    hipStream_t s = ctx->stream;
    const uint32_t g = (uint32_t)(-go) & 0xFFFFu, e = (uint32_t)(-ge) & 0xFFFFu;
    *two_ends = false;
    // diagnostics: SWG_TRACE=<file> dumps one line per wavefront (class, workgroup, wave, start and
    // end in 10 ns ticks, blocks of its longest stream) for every diagonal fill
    static const char *trace_path = getenv("SWG_TRACE");
    uint64_t *d_trace[2] = {nullptr, nullptr};
    if (trace_path)
        for (int c = 0; c < wk.n_classes; ++c) {
            const size_t bytes = (size_t)wk.plan[c].workgroups * wk.plan[c].W * 4 * 8;
            HIP_TRY(ctx, hipMalloc(&d_trace[c], bytes));
            HIP_TRY(ctx, hipMemsetAsync(d_trace[c], 0, bytes, s));
        }
    HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[1], s));
    if (wk.n_classes == 2) {
        const int r2 = ensure_stream2(ctx);
        if (r2 != SWG_OK) return r2;
        // fork: the long pairs start first, on their own stream, beside the bulk
        HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[6], s));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->cur->ev[6], 0));
    }
    for (int c = wk.n_classes - 1; c >= 0; --c) {
        const SwgDiagLayout &L = db->diag[c];
        const SwgDiagPlan &pl = wk.plan[c];
        if (diag_class_is_dynamic(ctx, db, pl)) {
            const SwgPairTokens &T = db->ptok;
            SwgDiagDynParams q;
            memset(&q, 0, sizeof q);
            q.tok = T.d_tok;
            q.zero_block = (uint32_t)T.total_blocks;
            q.pair_off = T.d_pair_off;
            q.q_begin = (uint32_t)wk.pair_begin[c];
            q.q_end = (uint32_t)wk.pair_end[c];
            q.queue = db->d_counters + SWG_QUEUE_WORD(c); // zeroed with the other counters before the fill
            q.profile = ctx->d_profile[diag_profile_slot(pl, c)];
            q.scores = db->d_scores;
            q.pair_limit = (uint32_t)(((size_t)db->n_bins * SWG_BIN) / 2);
            q.G = (uint32_t)pl.G;
            const int form = diag_class_form(ctx, db, pl);
            q.go = form == 2 ? f16x2_of(-go) : g | (g << 16);
            q.ge = form == 2 ? f16x2_of(-ge) : e | (e << 16);
            // the long class always runs at raised priority; in the bulk, a pair that alone is well
            // above an average lane group's whole share
            auto bulk_prio = [&]() -> uint32_t {
                const uint64_t blocks = T.pair_blocks_prefix[wk.pair_end[0]] - T.pair_blocks_prefix[wk.pair_begin[0]];
                const uint64_t groups = (uint64_t)diag_class_workgroups(ctx, db, wk, 0) * wk.plan[0].W * (64 / wk.plan[0].G);
                return (uint32_t)std::max<uint64_t>(8, (uint64_t)(ctx->opt_prio_share * 0.01 * (double)blocks /
                                                                  (double)std::max<uint64_t>(1, groups)));
            };
            q.prio_blocks = c == 1 ? 0u : bulk_prio();
            if (c == 1 && ctx->opt_long_helps && diag_class_is_dynamic(ctx, db, wk.plan[0])) {
                // when the long pairs are done their lane groups go on with the bulk's queue
                q.q2_begin = (uint32_t)wk.pair_begin[0];
                q.q2_end = (uint32_t)wk.pair_end[0];
                q.queue2 = db->d_counters + SWG_QUEUE_WORD(0);
                q.prio_blocks2 = bulk_prio();
            }
            q.turn_levels = wk.n_classes == 2 ? 3u : 4u;
            q.simd_ranks = db->d_counters + SWG_RANK_WORD(c);
            // (f16 sums round to nearest: a computed 32768 needs a true score within a few units of it)
            q.f16_wipe = ctx->cur->score_bound >= 32000ull ? 1u : 0u;
            q.trace = d_trace[c];
            // start / end wall-clock stamps of single-pass launches: words 8..15 of the counters
            q.stamps = pl.npass == 1 ? reinterpret_cast<unsigned long long *>(db->d_counters + 8 + 4 * c) : nullptr;
            const bool edges = pl.npass > 1 || form == 1;
            const size_t slice = (size_t)pl.G * swg_diag_padded_cols(pl.K) * 64;
            hipStream_t qs = c == 1 ? ctx->stream2 : s;
            // The form with edges addresses a launch's tokens and edges by 32-bit offsets: pairs whose
            // token blocks span more than SWG_DYN_SEG_BLOCKS go in several launches per pass, each over a
            // run of consecutive pairs (a segment).  Normally there is one, the whole token buffer.
            std::vector<std::pair<uint32_t, uint32_t>> segs; // pair ranges
            if (!edges || T.total_blocks <= ctx->opt_seg_blocks) {
                segs.push_back(std::make_pair(q.q_begin, q.q_end));
                q.seg_origin = 0;
                q.seg_blocks = (uint32_t)std::min<uint64_t>(T.total_blocks, ctx->opt_seg_blocks);
            } else {
                const std::vector<uint32_t> &pre = T.pair_blocks_prefix;
                for (uint32_t b = q.q_begin; b < q.q_end;) {
                    const uint64_t limit = (uint64_t)pre[b] + ctx->opt_seg_blocks;
                    const uint32_t e = (uint32_t)(std::upper_bound(pre.begin() + b, pre.begin() + q.q_end + 1, limit,
                                                                   [](uint64_t v, uint32_t x) { return v < (uint64_t)x; }) -
                                                  pre.begin()) - 1u;
                    if (e <= b) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "a pair of sequences too long for the multi-pass fill");
                    segs.push_back(std::make_pair(b, e));
                    b = e;
                }
                q.q2_begin = q.q2_end = 0; // (the other class's pairs lie outside a segment)
                q.queue2 = nullptr;
            }
            // Both 16-bit forms in one class (pl.f16_from, see swg_search_begin): the pairs before it -- the longest --
            // take all their passes on the wide form, then the rest theirs on the f16 cells, the same geometry
            // throughout; ev[5] between the two parts tells their times apart.
            const uint32_t class_begin = q.q_begin, class_end = q.q_end;
            const bool split = c == 0 && wk.n_classes == 1 && form != 2 && pl.f16_from > class_begin && pl.f16_from < class_end;
            bool first_launch = true;
            int launches = 0, f16_launches = 0;
            for (int part = 0; part < (split ? 2 : 1); ++part) {
                const int pform = split && part == 1 ? 2 : form;
                const uint32_t part_begin = split && part == 1 ? pl.f16_from : class_begin;
                const uint32_t part_end = split && part == 0 ? pl.f16_from : class_end;
                const uint8_t *prof = ctx->d_profile[split && part == 1 ? 7 : diag_profile_slot(pl, c)];
                q.go = pform == 2 ? f16x2_of(-go) : g | (g << 16);
                q.ge = pform == 2 ? f16x2_of(-ge) : e | (e << 16);
                if (split && part == 1) HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[5], qs));
                for (int pass = 0; pass < pl.npass; ++pass) {
                    // one launch per pass: the kernel boundary is what lets any lane group take any pair
                    q.profile = prof + (size_t)pass * slice;
                    q.edge_in = pass > 0 ? T.d_edge[(pass - 1) & 1] : nullptr;
                    q.edge_out = pass + 1 < pl.npass ? T.d_edge[pass & 1] : nullptr;
                    for (const std::pair<uint32_t, uint32_t> &sg : segs) {
                        const uint32_t b = std::max(sg.first, part_begin), en = std::min(sg.second, part_end);
                        if (b >= en) continue;
                        if (!first_launch)
                            HIP_TRY(ctx, hipMemsetAsync(q.queue, 0, (size_t)SWG_DYN_SHARDS * SWG_DYN_SHARD_STRIDE * 4, qs));
                        first_launch = false;
                        q.q_begin = b;
                        q.q_end = en;
                        if (segs.size() > 1) {
                            q.seg_origin = T.pair_blocks_prefix[sg.first];
                            q.seg_blocks = T.pair_blocks_prefix[sg.second] - q.seg_origin;
                        }
                        const int variant = pass + 1 == pl.npass && pl.npass > 1 && pl.last_variant >= 0 ? pl.last_variant : pl.variant;
                        const int wgs_c = diag_class_workgroups(ctx, db, wk, c);
                        // (with long_helps the long class's kernel reads the bulk's counters too, as single pairs: no zones then)
                        if (!(ctx->opt_long_helps && wk.n_classes == 2)) dyn_batch_zones(ctx, T, &q, (uint64_t)wgs_c * pl.W * (64 / pl.G), swg_diag_variant_info(variant).K, pl.G, pform);
                        HIP_TRY(ctx, swg_launch_diag_dyn(variant, edges, pform, pl.W, wgs_c, q, qs));
                        ++launches;
                        if (split && part == 1) ++f16_launches;
                    }
                }
            }
            if (c == 0) ctx->cur->fill_launches = launches, ctx->cur->fill_f16_launches = f16_launches;
            continue;
        }
        SwgDiagParams d;
        memset(&d, 0, sizeof d);
        d.tok = L.d_tok;
        d.stream_off = L.d_stream_off;
        d.stream_pairs = L.d_stream_pairs;
        d.stream_pair_off = L.d_stream_pair_off;
        d.n_streams = L.n_streams;
        d.profile = ctx->d_profile[diag_profile_slot(pl, c)];
        d.scores = db->d_scores;
        d.scratch = L.d_scratch;
        d.npass = (uint32_t)pl.npass;
        d.G = (uint32_t)pl.G;
        d.go = g | (g << 16);
        d.ge = e | (e << 16);
        // wavefronts on the critical path get issue priority over the ones they share a SIMD
        // with: all of the long class; in the bulk, streams that hold little more than one very
        // long pair
        const double mean_blocks = (double)L.total_blocks / std::max<uint32_t>(1, L.n_streams);
        d.prio_blocks = c == 1 ? 0u
                        : (double)L.max_stream_blocks > 1.1 * mean_blocks ? (uint32_t)(0.75 * (double)L.max_stream_blocks)
                                                                          : 0xFFFFFFFFu;
        d.trace = d_trace[c];
        HIP_TRY(ctx, swg_launch_diag(pl.variant, pl.npass > 1, pl.wide != 0, pl.W, pl.workgroups, pl.lds_bytes, d,
                                     c == 1 ? ctx->stream2 : s));
    }
    if (wk.n_classes == 2) {
        // join; the end of the fill is the later of the two kernels' ends
        HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[5], s));
        HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[7], ctx->stream2));
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->cur->ev[7], 0));
        *two_ends = true;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[2], s));
    if (trace_path) {
        HIP_TRY(ctx, hipStreamSynchronize(s));
        FILE *f = fopen(trace_path, "a");
        if (f) fprintf(f, "# fill K=%d G=%d W=%d wgs=%d classes=%d\n", wk.plan[0].K, wk.plan[0].G, wk.plan[0].W,
                       wk.plan[0].workgroups, wk.n_classes);
        for (int c = 0; c < wk.n_classes; ++c) {
            const size_t n = (size_t)wk.plan[c].workgroups * wk.plan[c].W;
            std::vector<uint64_t> h(n * 4);
            HIP_TRY(ctx, hipMemcpy(h.data(), d_trace[c], n * 32, hipMemcpyDeviceToHost));
            (void)hipFree(d_trace[c]);
            for (size_t i = 0; f && i < n; ++i)
                fprintf(f, "%d %zu %zu %llu %llu %llu %llu\n", c, i / wk.plan[c].W, i % wk.plan[c].W,
                        (unsigned long long)h[4 * i], (unsigned long long)h[4 * i + 1], (unsigned long long)h[4 * i + 2],
                        (unsigned long long)h[4 * i + 3]);
        }
        if (f) fclose(f);
    }
    return SWG_OK;
}

static int diag_fill_ms(swg_ctx *ctx, bool two_ends, double *out)
{
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->cur->ev[1], ctx->cur->ev[2]));
    *out = ms;
    if (two_ends) {
        float a = 0.f, b = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->cur->ev[1], ctx->cur->ev[5]));
        HIP_TRY(ctx, hipEventElapsedTime(&b, ctx->cur->ev[1], ctx->cur->ev[7]));
        *out = std::max(*out, (double)std::max(a, b));
    }
    return SWG_OK;
}

// Launches the systolic int16/int32 fill of one plan over the whole database (events ev[1], ev[2]).
static int launch_systolic(swg_ctx *ctx, const swg_db *db, const Plan &pl, int go, int ge)
{
    hipStream_t s = ctx->stream;
    SwgFillParams p;
    memset(&p, 0, sizeof p);
    p.residues = db->d_packed;
    p.bin_off = db->d_bin_off;
    p.bin_nblk = db->d_bin_nblk;
    p.n_bins = db->n_bins;
    p.scores = db->d_scores;
    p.scratch = ctx->d_scratch;
    p.profile = ctx->d_profile[pl.bits == 16 ? 0 : 1];
    p.queue = db->d_counters + 0;
    p.n_items = pl.bits == 16 ? db->n_bins : db->n_bins * 2;
    p.npass = (uint32_t)pl.npass;
    if (pl.bits == 16 && pl.f16) {
        p.go = (int32_t)f16x2_of(-go);
        p.ge = (int32_t)f16x2_of(-ge);
    } else if (pl.bits == 16) {
        const uint32_t g = (uint32_t)(-go) & 0xFFFFu, e = (uint32_t)(-ge) & 0xFFFFu;
        p.go = (int32_t)(g | (g << 16));
        p.ge = (int32_t)(e | (e << 16));
    } else {
        p.go = go;
        p.ge = ge;
    }
    p.scratch_wg_dwords = (uint64_t)db->max_nblk * SWG_ROWS_PER_BLK * 64 * pl.info.nb;
    HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[1], s));
    HIP_TRY(ctx, swg_launch_fill(pl.bits, pl.variant, pl.W, pl.workgroups, p, s, pl.bits == 16 && pl.f16));
    HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[2], s));
    return SWG_OK;
}

static int prepare_systolic(swg_ctx *ctx, const swg_db *db, const Plan &pl)
{
    int rc = ensure_bins(ctx, const_cast<swg_db *>(db));
    if (rc != SWG_OK) return rc;
    if ((rc = ensure_profile(ctx, pl)) != SWG_OK) return rc;
    const size_t need = pl.npass > 1 ? (size_t)pl.workgroups * db->max_nblk * SWG_ROWS_PER_BLK * 64 * pl.info.nb : 0;
    return ensure_scratch(ctx, need);
}

// ---------------------------------------------------------------------------
// int32 with a work queue (swg_diag32q_kernel): non-positive gap scores, one pass
// ---------------------------------------------------------------------------
// Occupancy of an int32 work plan.  The int32 profile is twice the int16 one, so LDS, not registers,
// decides how many workgroups a CU holds: the long class keeps one workgroup of four wavefronts per
// CU, and the bulk takes the smallest workgroup size W (4, 8, 12 wavefronts sharing one profile) with
// which the LDS that is left still holds three wavefronts per SIMD, or as many as it can.
static void q32_occupancy(const swg_ctx *ctx, const SwgDiagWork &wk, int *bulk_W, int *bulk_per_cu)
{
    const SwgDiagPlan &pl = wk.plan[0];
    const SwgKernelInfo info = swg_diag_variant_info(pl.variant);
    size_t room = 160 * 1024;
    int cap_waves = std::min(12, info.max_waves);
    if (wk.n_classes == 2) {
        room -= std::min(room, swg_diag32q_lds_bytes(wk.plan[1].K, wk.plan[1].G, 4));
        cap_waves = std::min(cap_waves, info.max_waves - 4);
    }
    int best_W = 4, best_n = 1, best_waves = 0;
    for (int W = 4; W <= info.max_waves; W += 4) {
        const size_t lds = swg_diag32q_lds_bytes(pl.K, pl.G, W);
        const int n = std::min<int>((int)(room / lds), cap_waves / W);
        if (n >= 1 && n * W > best_waves) {
            best_waves = n * W;
            best_W = W;
            best_n = n;
        }
    }
    if (ctx->opt_q32_waves > 0 && ctx->opt_q32_waves <= info.max_waves && wk.n_classes == 1 &&
        swg_diag32q_lds_bytes(pl.K, pl.G, (int)ctx->opt_q32_waves) <= room) { // experiment: the workgroup size of an int32 launch
        best_W = (int)ctx->opt_q32_waves;
        best_n = std::max(1, std::min<int>((int)(room / swg_diag32q_lds_bytes(pl.K, pl.G, best_W)), cap_waves / best_W));
    }
    *bulk_W = best_W;
    *bulk_per_cu = best_n;
}

static int q32_class_workgroups(const swg_ctx *ctx, const SwgDiagWork &wk, int c, uint64_t items)
{
    const SwgDiagPlan &pl = wk.plan[c];
    int W = 4, per_cu = 1;
    if (c == 0) q32_occupancy(ctx, wk, &W, &per_cu);
    const uint64_t per_wg = (uint64_t)W * (64 / pl.G);
    return (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->n_cu * per_cu, (items + per_wg - 1) / per_wg));
}

static int q32_class_waves(const swg_ctx *ctx, const SwgDiagWork &wk, int c)
{
    int W = 4, per_cu = 1;
    if (c == 0) q32_occupancy(ctx, wk, &W, &per_cu);
    return W;
}

static bool q32_plan_fits(const SwgDiagWork &wk, size_t lq)
{
    if (wk.n_classes < 1) return false;
    for (int c = 0; c < wk.n_classes; ++c) {
        const SwgDiagPlan &pl = wk.plan[c];
        if (pl.npass != 1 || (size_t)pl.G * pl.K < lq || swg_diag32q_lds_bytes(pl.K, pl.G, 4) > 160 * 1024) return false;
    }
    // both classes run side by side on every CU
    if (wk.n_classes == 2 && swg_diag32q_lds_bytes(wk.plan[0].K, wk.plan[0].G, 4) + swg_diag32q_lds_bytes(wk.plan[1].K, wk.plan[1].G, 4) >
                                 160 * 1024)
        return false;
    return true;
}

// Geometry for a list of `n_items` flagged sequences (or, with a plan the int16 planner's choice does not
// fit, the whole database): few items get 64 lanes each (the shortest chain per row), many the narrowest
// lane group that covers the query in one pass.  A query no single pass holds (LDS: G*K int32 columns of
// 128 bytes in 160 KB, about 1150) takes several passes of the widest geometry for the item count.
static bool q32_list_plan(const swg_ctx *ctx, size_t lq, uint32_t n_items, SwgDiagWork *wk)
{
    const bool few = n_items <= 8u * (uint32_t)ctx->n_cu;
    const int order[3] = {few ? 64 : 16, 32, few ? 16 : 64};
    auto set = [&](int v, int K, int G, int npass) {
        *wk = SwgDiagWork();
        wk->n_classes = 1;
        wk->plan[0].variant = v;
        wk->plan[0].K = K;
        wk->plan[0].G = G;
        wk->plan[0].W = 4;
        wk->plan[0].npass = npass;
    };
    for (int gi = 0; gi < 3; ++gi) {
        const int G = order[gi];
        int best = -1, bestK = 1 << 30;
        for (int v = 0; v < swg_num_diag_variants(); ++v) {
            const int K = swg_diag_variant_info(v).K;
            if ((size_t)G * K >= lq && K < bestK && swg_diag32q_lds_bytes(K, G, 4) <= 160 * 1024) {
                best = v;
                bestK = K;
            }
        }
        if (best >= 0) {
            set(best, bestK, G, 1);
            return true;
        }
    }
    // several passes: the most columns per pass that fit LDS (few items: 64 lanes; many: 32)
    const int G = few ? 64 : 32;
    int best = -1, bestK = 0;
    for (int v = 0; v < swg_num_diag_variants(); ++v) {
        const int K = swg_diag_variant_info(v).K;
        if (K > bestK && swg_diag32q_lds_bytes(K, G, 4) <= 160 * 1024) {
            best = v;
            bestK = K;
        }
    }
    if (best < 0) return false;
    const size_t cols = (size_t)G * bestK;
    const size_t npass = (lq + cols - 1) / cols;
    if (npass > 64) return false;
    // the fewest columns per lane that still need no more passes (less padding in the last one)
    for (int v = 0; v < swg_num_diag_variants(); ++v) {
        const int K = swg_diag_variant_info(v).K;
        if (K < bestK && (size_t)G * K * npass >= lq) {
            best = v;
            bestK = K;
        }
    }
    set(best, bestK, G, (int)npass);
    return true;
}

// Geometry of the exact int32 cells (gap scores of any sign) for a whole database: one class; the narrowest lane
// group that covers the query in one pass with at most SWG_X32_MAX_K columns per lane (forced cols_per_wave /
// group_lanes are honoured where they fit), else several passes of 64 lanes.
static bool x32_plan(const swg_ctx *ctx, const swg_db *db, size_t lq, SwgDiagWork *wk)
{
    auto set = [&](int v, int K, int G, size_t npass) {
        *wk = SwgDiagWork();
        wk->n_classes = 1;
        wk->plan[0].variant = v;
        wk->plan[0].K = K;
        wk->plan[0].G = G;
        wk->plan[0].W = 4;
        wk->plan[0].npass = (int)npass;
        wk->pair_begin[0] = 0;
        wk->pair_end[0] = swg_db_pair_count(db);
    };
    auto fits = [&](int K, int G) { return K <= SWG_X32_MAX_K && swg_diag32q_lds_bytes(K, G, 4) <= 160 * 1024; };
    if (ctx->opt_cols > 0 && ctx->opt_group > 0) {
        for (int v = 0; v < swg_num_diag_variants(); ++v) {
            const int K = swg_diag_variant_info(v).K, G = (int)ctx->opt_group;
            if (K != (int)ctx->opt_cols || !fits(K, G)) continue;
            const size_t np = (lq + (size_t)G * K - 1) / ((size_t)G * K);
            if (np > 64) continue;
            set(v, K, G, np);
            return true;
        }
    }
    const int groups[3] = {16, 32, 64};
    for (int gi = 0; gi < 3; ++gi) {
        const int G = groups[gi];
        int best = -1, bestK = 1 << 30;
        for (int v = 0; v < swg_num_diag_variants(); ++v) {
            const int K = swg_diag_variant_info(v).K;
            if ((size_t)G * K >= lq && K < bestK && fits(K, G)) best = v, bestK = K;
        }
        if (best >= 0) {
            set(best, bestK, G, 1);
            return true;
        }
    }
    const int G = 64;
    int best = -1, bestK = 0;
    for (int v = 0; v < swg_num_diag_variants(); ++v) {
        const int K = swg_diag_variant_info(v).K;
        if (K > bestK && fits(K, G)) best = v, bestK = K;
    }
    if (best < 0) return false;
    const size_t npass = (lq + (size_t)G * bestK - 1) / ((size_t)G * bestK);
    if (npass > 64) return false;
    for (int v = 0; v < swg_num_diag_variants(); ++v) {
        const int K = swg_diag_variant_info(v).K;
        if (K < bestK && (size_t)G * K * npass >= lq) best = v, bestK = K;
    }
    set(best, bestK, G, npass);
    return true;
}

// Launches the int32 work-queue fill: every sequence of the plan's classes (list == NULL), or the
// device-side list of ranks with one class.  A plan of several passes (one class) is one launch per
// pass, the rows' edges going from launch to launch through memory.  Events as launch_diag.
static int launch_q32(swg_ctx *ctx, const swg_db *db, const SwgDiagWork &wk, int go, int ge, const uint32_t *d_list,
                      const uint32_t *d_list_count, uint32_t list_items, uint32_t *queue_words, bool *two_ends,
                      bool timing_events = true, bool exact = false)
{
    hipStream_t s = ctx->stream;
    SwgPairTokens &T = const_cast<swg_db *>(db)->ptok;
    const size_t n_slots = (size_t)db->n_bins * SWG_BIN;
    *two_ends = false;
    const int npass = wk.plan[0].npass;
    if (npass > 1) {
        if (wk.n_classes != 1 || T.total_blocks >= (1ull << 28))
            return swg_set_ctx_error(ctx, SWG_ERR_ARG, "int32 multi-pass fill: plan not supported");
        if (!T.d_edge32[0] || T.edge32_blocks < T.total_blocks) {
            (void)hipFree(T.d_edge32[0]);
            (void)hipFree(T.d_edge32[1]);
            T.d_edge32[0] = T.d_edge32[1] = nullptr;
            const size_t bytes = std::max<size_t>(8, (size_t)T.total_blocks * 4 * 2 * sizeof(int2));
            HIP_TRY(ctx, hipMalloc(&T.d_edge32[0], bytes));
            HIP_TRY(ctx, hipMalloc(&T.d_edge32[1], bytes));
            T.edge32_blocks = T.total_blocks;
        }
        if (exact && (!T.d_edge32d[0] || T.edge32d_blocks < T.total_blocks)) { // the exact cells' third edge value
            (void)hipFree(T.d_edge32d[0]);
            (void)hipFree(T.d_edge32d[1]);
            T.d_edge32d[0] = T.d_edge32d[1] = nullptr;
            const size_t bytes = std::max<size_t>(8, (size_t)T.total_blocks * 4 * 2 * sizeof(int32_t));
            HIP_TRY(ctx, hipMalloc(&T.d_edge32d[0], bytes));
            HIP_TRY(ctx, hipMalloc(&T.d_edge32d[1], bytes));
            T.edge32d_blocks = T.total_blocks;
        }
    }
    for (int c = 0; c < wk.n_classes; ++c) {
        const SwgDiagPlan &pl = wk.plan[c];
        const int kp = swg_q32_padded_cols(pl.K);
        const uint32_t ncols = (uint32_t)(pl.npass * pl.G * kp);
        int rc = ensure_profile_cols(ctx, 4 + c, ncols, 4, (1ull << 54) | ((uint64_t)pl.K << 40) | ((uint64_t)pl.G << 32) | (uint64_t)ncols,
                                     pl.K, kp, 2, SWG_LDS_SWIZZLE ? pl.G : 0);
        if (rc != SWG_OK) return rc;
    }
    if (timing_events) HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[1], s));
    if (wk.n_classes == 2) {
        const int r2 = ensure_stream2(ctx);
        if (r2 != SWG_OK) return r2;
        HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[6], s));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->cur->ev[6], 0));
    }
    for (int c = wk.n_classes - 1; c >= 0; --c) {
        const SwgDiagPlan &pl = wk.plan[c];
        SwgDiagQ32Params q;
        memset(&q, 0, sizeof q);
        q.tok = T.d_tok;
        q.zero_block = (uint32_t)T.total_blocks;
        q.pair_off = T.d_pair_off;
        uint64_t items;
        if (d_list) {
            q.list = d_list;
            q.list_count = d_list_count;
            items = list_items;
        } else {
            q.q_begin = (uint32_t)std::min<uint64_t>(2 * wk.pair_begin[c], n_slots);
            q.q_end = (uint32_t)std::min<uint64_t>(2 * wk.pair_end[c], n_slots);
            items = q.q_end - q.q_begin;
        }
        q.queue = queue_words + (size_t)c * SWG_DYN_SHARDS * SWG_DYN_SHARD_STRIDE;
        q.scores = db->d_scores;
        q.seq_limit = (uint32_t)n_slots;
        q.G = (uint32_t)pl.G;
        q.go = exact ? go : -go; // (the exact cells add the signed scores, the reduced ones subtract magnitudes)
        q.ge = exact ? ge : -ge;
        q.turn_levels = wk.n_classes == 2 ? 3u : 4u;
        q.simd_ranks = db->d_counters + SWG_RANK_WORD(c);
        const int wgs = q32_class_workgroups(ctx, wk, c, items);
        const int W = q32_class_waves(ctx, wk, c);
        if (c == 0 && !d_list) {
            const uint64_t blocks = T.pair_blocks_prefix[wk.pair_end[0]] - T.pair_blocks_prefix[wk.pair_begin[0]];
            const uint64_t groups = (uint64_t)wgs * W * (64 / pl.G);
            q.prio_blocks = (uint32_t)std::max<uint64_t>(8, (uint64_t)(ctx->opt_prio_share * 0.01 * 2.0 * (double)blocks /
                                                                       (double)std::max<uint64_t>(1, groups)));
        } else {
            q.prio_blocks = c == 1 ? 0u : 0xFFFFFFFFu;
        }
        const size_t slice = (size_t)pl.G * swg_q32_padded_cols(pl.K) * 128;
        hipStream_t qs = c == 1 ? ctx->stream2 : s;
        for (int pass = 0; pass < pl.npass; ++pass) {
            if (pass > 0) HIP_TRY(ctx, hipMemsetAsync(q.queue, 0, (size_t)SWG_DYN_SHARDS * SWG_DYN_SHARD_STRIDE * 4, qs));
            q.profile = ctx->d_profile[4 + c] + (size_t)pass * slice;
            q.edge_in = pass > 0 ? T.d_edge32[(pass - 1) & 1] : nullptr;
            q.edge_out = pass + 1 < pl.npass ? T.d_edge32[pass & 1] : nullptr;
            q.edge_d_in = exact && pass > 0 ? T.d_edge32d[(pass - 1) & 1] : nullptr;
            q.edge_d_out = exact && pass + 1 < pl.npass ? T.d_edge32d[pass & 1] : nullptr;
            HIP_TRY(ctx, swg_launch_diag32q(pl.variant, pl.npass > 1, exact, W, wgs, q, qs));
        }
    }
    if (wk.n_classes == 2) {
        HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[5], s));
        HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[7], ctx->stream2));
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->cur->ev[7], 0));
        *two_ends = true;
    }
    if (timing_events) HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[2], s));
    return SWG_OK;
}

// ---------------------------------------------------------------------------
// the pairs the f16 cells flagged, again on int16 cells
// ---------------------------------------------------------------------------
// A pair whose f16 score reached 4096 is run again by the same work-queue kernel on the packed int16 cells (or
// the wide form when the query can score beyond 32767), which are exact where the f16 cells are not: 10
// instead of 8.5 instructions per column pair, against 16 for the int32 kernel, and -- unlike the int32
// kernel's 32-bit edge indices -- on databases of any size.  The launch reads the list's length on the device
// and leaves at once when it is empty.  Geometry: few pairs get 64 lanes each (the shortest chain per row),
// many the main fill's own.
static bool i16_list_plan(int n_cu, size_t lq, uint32_t n_pairs_guess, const SwgDiagPlan &main_plan, SwgDiagPlan *out)
{
    // 64 lanes per pair: one pair per wavefront, so a list shorter than the launch has lane groups still fills every
    // wavefront it occupies (with 16-lane groups a list of 3 100 pairs -- config 4's share with 0.5 % relatives --
    // lands one pair in every fourth group and every wavefront issues its rows for one group in four: 26 ms against
    // 15), and at 512 columns or more its rows cost the same instructions per pair as the main fill's narrower groups
    // in more passes (lq 3000: 2 x 269 per pair-row against 6 x 349 / 4).  Only a short query's long list -- where a
    // 64-lane group would hold two or three columns per lane -- keeps the main fill's geometry.
    if (n_pairs_guess > 4u * (uint32_t)n_cu && lq < 512) {
        *out = main_plan;
        out->f16 = 0;
        out->f16_from = 0;
        out->last_variant = -1;
        out->last_K = 0;
        return true;
    }
    const int G = 64;
    int best = -1, bestK = 0;
    for (int v = 0; v < swg_num_diag_variants(); ++v) { // the most columns per pass that fit LDS
        const int K = swg_diag_variant_info(v).K;
        if (K > bestK && swg_diag_dyn_lds_bytes(K, G, 4) <= 160 * 1024) best = v, bestK = K;
    }
    if (best < 0) return false;
    const size_t npass = (lq + (size_t)G * bestK - 1) / ((size_t)G * bestK);
    for (int v = 0; v < swg_num_diag_variants(); ++v) { // the fewest columns per lane that need no more passes
        const int K = swg_diag_variant_info(v).K;
        if (K < bestK && (size_t)G * K * npass >= lq) best = v, bestK = K;
    }
    *out = SwgDiagPlan();
    out->variant = best;
    out->K = bestK;
    out->G = G;
    // One workgroup per CU is all that fits beside a 64-lane profile of 512 columns or more (98 KB at K = 24), so the
    // workgroup is as large as the kernel was compiled for, and the kernel sends home the wavefronts a shorter list
    // does not need (it knows the list's length; the host does not).  Until round 4's last day the workgroup had 4
    // wavefronts -- ONE per SIMD, 5.2 cycles per instruction instead of 4.07 -- whatever the list, and config 4's
    // 3 100 pairs took three rounds and 28 left-overs on 1 024 wavefronts (7.7 ms per pass, now 5.3; the text of
    // DESIGN 9 counted 3 072).
    int W = swg_diag_variant_info(best).max_waves / 4 * 4;
    while (W > 4 && swg_diag_dyn_lds_bytes(bestK, G, W) > 160 * 1024) W -= 4;
    out->W = std::max(4, W);
    out->npass = (int)npass;
    return true;
}

// test hook (not in the public headers): the list re-run's geometry for a query of lq columns, on the host.
// main_kgw: the main fill's (K, G, W), which a short query's long list keeps; out: {K, G, W, passes}
extern "C" int swg_debug_list_plan(size_t lq, uint32_t n_pairs_guess, int n_cu, const int32_t *main_kgw, int32_t *out)
{
    if (!main_kgw || !out || n_cu < 1) return SWG_ERR_ARG;
    SwgDiagPlan mp = SwgDiagPlan(), lp = SwgDiagPlan();
    mp.K = main_kgw[0];
    mp.G = main_kgw[1];
    mp.W = main_kgw[2];
    mp.npass = 1;
    if (!i16_list_plan(n_cu, lq, n_pairs_guess, mp, &lp)) return SWG_ERR_ARG;
    out[0] = lp.K;
    out[1] = lp.G;
    out[2] = lp.W;
    out[3] = lp.npass;
    return SWG_OK;
}

static int launch_dyn_list(swg_ctx *ctx, swg_db *db, const SwgDiagPlan &pl, bool wide, int go, int ge, const uint32_t *d_list,
                           const uint32_t *d_count, hipStream_t s)
{
    SwgPairTokens &T = db->ptok;
    const int form = wide ? 1 : 0;
    const bool edges = pl.npass > 1 || form == 1;
    if (pl.npass > 1 && (!T.d_edge[0] || T.edge_blocks < T.total_blocks)) {
        (void)hipFree(T.d_edge[0]);
        (void)hipFree(T.d_edge[1]);
        T.d_edge[0] = T.d_edge[1] = nullptr;
        const size_t bytes = std::max<size_t>(8, (size_t)T.total_blocks * 4 * sizeof(uint2));
        HIP_TRY(ctx, hipMalloc(&T.d_edge[0], bytes));
        HIP_TRY(ctx, hipMalloc(&T.d_edge[1], bytes));
        T.edge_blocks = T.total_blocks;
    }
    const int kp = swg_diag_padded_cols(pl.K);
    const uint32_t ncols = (uint32_t)(pl.npass * pl.G * kp);
    int rc = ensure_profile_cols(ctx, 6, ncols, 2, (1ull << 52) | ((uint64_t)pl.K << 40) | ((uint64_t)pl.G << 32) | (uint64_t)ncols, pl.K, kp,
                                 4, SWG_LDS_SWIZZLE ? pl.G : 0, 0);
    if (rc != SWG_OK) return rc;
    const uint32_t g = (uint32_t)(-go) & 0xFFFFu, e = (uint32_t)(-ge) & 0xFFFFu;
    SwgDiagDynParams q;
    memset(&q, 0, sizeof q);
    q.tok = T.d_tok;
    q.zero_block = (uint32_t)T.total_blocks;
    q.pair_off = T.d_pair_off;
    q.list = d_list;
    q.list_count = d_count;
    q.queue = db->d_counters + SWG_QUEUE_WORD(0);
    q.scores = db->d_scores;
    q.pair_limit = (uint32_t)(((size_t)db->n_bins * SWG_BIN) / 2);
    q.G = (uint32_t)pl.G;
    q.go = g | (g << 16);
    q.ge = e | (e << 16);
    q.prio_blocks = 0xFFFFFFFFu;
    q.prio_blocks2 = 0xFFFFFFFFu;
    q.turn_levels = 4u;
    q.simd_ranks = db->d_counters + SWG_RANK_WORD(0);
    const SwgKernelInfo info = swg_diag_variant_info(pl.variant);
    const size_t lds = swg_diag_dyn_lds_bytes(pl.K, pl.G, pl.W);
    const int per_cu = std::max(1, std::min<int>(info.max_waves / pl.W, (int)((160 * 1024) / lds)));
    const int wgs = ctx->n_cu * std::min(per_cu, 3);
    const uint32_t n_pairs = (uint32_t)swg_db_pair_count(db);
    std::vector<std::pair<uint32_t, uint32_t>> segs;
    if (!edges || T.total_blocks <= ctx->opt_seg_blocks) {
        segs.push_back(std::make_pair(0u, n_pairs));
        q.seg_origin = 0;
        q.seg_blocks = (uint32_t)std::min<uint64_t>(T.total_blocks, ctx->opt_seg_blocks);
    } else {
        const std::vector<uint32_t> &pre = T.pair_blocks_prefix;
        for (uint32_t b = 0; b < n_pairs;) {
            const uint64_t limit = (uint64_t)pre[b] + ctx->opt_seg_blocks;
            const uint32_t en = (uint32_t)(std::upper_bound(pre.begin() + b, pre.begin() + n_pairs + 1, limit,
                                                            [](uint64_t v, uint32_t x) { return v < (uint64_t)x; }) - pre.begin()) - 1u;
            if (en <= b) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "a pair of sequences too long for the multi-pass fill");
            segs.push_back(std::make_pair(b, en));
            b = en;
        }
    }
    const size_t slice = (size_t)pl.G * kp * 64;
    bool first_launch = true;
    for (int pass = 0; pass < pl.npass; ++pass) {
        q.profile = ctx->d_profile[6] + (size_t)pass * slice;
        q.edge_in = pass > 0 ? T.d_edge[(pass - 1) & 1] : nullptr;
        q.edge_out = pass + 1 < pl.npass ? T.d_edge[pass & 1] : nullptr;
        for (const std::pair<uint32_t, uint32_t> &sg : segs) {
            if (!first_launch) HIP_TRY(ctx, hipMemsetAsync(q.queue, 0, (size_t)SWG_DYN_SHARDS * SWG_DYN_SHARD_STRIDE * 4, s));
            first_launch = false;
            q.q_begin = sg.first;
            q.q_end = sg.second;
            if (segs.size() > 1) {
                q.seg_origin = T.pair_blocks_prefix[sg.first];
                q.seg_blocks = T.pair_blocks_prefix[sg.second] - q.seg_origin;
            }
            HIP_TRY(ctx, swg_launch_diag_dyn(pl.variant, edges, form, pl.W, wgs, q, s));
        }
    }
    return SWG_OK;
}

// First search of a query length on a database: the cost model ranks the geometries, the few
// best are timed once on this device (each is a complete, valid fill) and the fastest is kept.
static int autotune_diag(swg_ctx *ctx, swg_db *db, size_t lq, int go, int ge, SwgTuned *tuned, int form)
{
    SwgDiagWork *best = &tuned->wk;
    std::vector<SwgDiagWork> cands;
    const bool work_queue = ctx->opt_dynamic != 0 && db->ptok.ok;
    if (swg_plan_diag_candidates(db, lq, ctx->n_cu, 0, 0, 0, 0, true, work_queue, &cands, 1.0, form) <= 0) return SWG_ERR_ARG;
    for (SwgDiagWork &c : cands) // (the trials run on the cells the search will use)
        for (int k = 0; k < c.n_classes; ++k) c.plan[k].f16 = form == 2;
    // distinct (K, G, W, split) among the best-ranked
    std::vector<SwgDiagWork> pick;
    for (const SwgDiagWork &c : cands) {
        bool dup = false;
        for (const SwgDiagWork &p : pick)
            dup |= p.plan[0].K == c.plan[0].K && p.plan[0].G == c.plan[0].G && p.plan[0].W == c.plan[0].W &&
                   p.n_classes == c.n_classes &&
                   (c.n_classes < 2 || (p.plan[1].K == c.plan[1].K && p.plan[1].G == c.plan[1].G &&
                                        p.pair_end[1] == c.pair_end[1]));
        if (!dup) pick.push_back(c);
        // long fills: fewer trials (a trial is a warm-up and a few complete fills)
        if (pick.size() >= (cands[0].plan[0].est_ms > 50.0 ? 4u : 8u)) break;
    }
    const size_t n_slots = (size_t)db->n_bins * SWG_BIN;
    double best_ms = 1e300;
    // One trial = a warm-up fill, then four fills queued back to back the way consecutive searches
    // are, timed as a whole: what happens where one fill ends and the next begins (workgroups of
    // two classes competing for the freed slots) is part of what is being chosen.
    auto time_one = [&](const SwgDiagWork &c, double *ms_out) -> int {
        int rc = prepare_diag(ctx, db, c);
        if (rc != SWG_OK) return rc;
        // (enough repetitions for about 20 ms of fills: four 2 ms fills differ by more from trial to trial than the
        // geometries being compared do -- round 3: the tuner picked K=24 over the model's K=23 on config 2 and lost 3 %)
        const int reps = c.plan[0].est_ms > 20.0 ? 2 : std::max(4, std::min(16, (int)(20.0 / std::max(0.5, c.plan[0].est_ms))));
        for (int rep = -1; rep < reps; ++rep) {
            bool two = false;
            if (rep == 0) HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[0], ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(db->d_scores, 0, n_slots * 4, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(db->d_counters, 0, SWG_COUNTER_BYTES, ctx->stream));
            rc = launch_diag(ctx, db, c, go, ge, &two);
            if (rc != SWG_OK) return rc;
        }
        HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[4], ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->cur->ev[0], ctx->cur->ev[4]));
        *ms_out = (double)ms / reps;
        return SWG_OK;
    };
    if (!pick.empty()) {
        // (one untimed trial first: the first candidate -- the model's choice -- otherwise pays for the clocks ramping
        // up and the code objects' first touch, and loses to candidates that are in fact slower)
        double warm = 0;
        int rc = time_one(pick[0], &warm);
        if (rc != SWG_OK) return rc;
    }
    for (const SwgDiagWork &c : pick) {
        double ms = 0;
        int rc = time_one(c, &ms);
        if (rc != SWG_OK) return rc;
        // the model's first choice stays unless another geometry is clearly (2 %) faster: two
        // timings of the same fill differ by about a percent
        if (ms < best_ms * (best_ms < 1e299 ? 0.98 : 1.0)) {
            best_ms = ms;
            *best = c;
            best->plan[0].est_ms = ms;
        }
    }
    // second stage: with the winning geometry, where to cut the long class off.  Fixed streams only:
    // with the work queue the model's cut (the longest pair a fair-share wavefront still finishes
    // within the search) is within a percent of the best measured one, less than two trials differ.
    if (best->n_classes == 2 && !diag_class_is_dynamic(ctx, db, best->plan[0])) {
        const SwgDiagWork base = *best;
        const uint64_t n_pairs = swg_db_pair_count(db);
        const bool dyn = diag_class_is_dynamic(ctx, db, base.plan[0]);
        // static streams: multiples of a stream's mean share; work queue: around the model's cut
        const double unit = dyn ? (double)(2ull + db->lens[2 * base.pair_end[1]])
                                : (double)swg_db_pair_rows(db, 0, n_pairs, nullptr) / (double)base.plan[0].n_streams;
        const double fr_static[] = {0.25, 0.45, 0.8, 1.0, 1.3, 1.7};
        const double fr_dyn[] = {0.6, 0.75, 0.88, 1.15, 1.35, 1.7};
        for (int i = 0; i < 6; ++i) {
            const long thr = (long)std::max(64.0, (dyn ? fr_dyn[i] : fr_static[i]) * unit);
            std::vector<SwgDiagWork> alt;
            if (swg_plan_diag_candidates(db, lq, ctx->n_cu, base.plan[0].K, base.plan[0].G, base.plan[0].W, thr, true,
                                         work_queue, &alt, 1.0, form) <= 0)
                continue;
            for (SwgDiagWork &c : alt)
                for (int k = 0; k < c.n_classes; ++k) c.plan[k].f16 = form == 2;
            const SwgDiagWork *same = nullptr;
            for (const SwgDiagWork &c : alt)
                if (c.n_classes == 2 && c.plan[1].K == base.plan[1].K && c.plan[1].G == base.plan[1].G) {
                    same = &c;
                    break;
                }
            if (!same || same->pair_end[1] == base.pair_end[1]) continue;
            double ms = 0;
            int rc = time_one(*same, &ms);
            if (rc != SWG_OK) return rc;
            if (ms < best_ms * 0.985) {
                best_ms = ms;
                *best = *same;
                best->plan[0].est_ms = ms;
            }
        }
    }
    tuned->engine = 2;
    tuned->ms = best_ms;
    // third stage: the systolic engine (less bookkeeping per row, coarse work units); it has not won
    // a measured case since the diagonal engine got its work queue and is only tried where a trial
    // is cheap
    const long keep_cols = ctx->opt_cols;
    for (int v = 0; v < swg_num_variants(16) && tuned->ms < 100.0; ++v) {
        Plan pl;
        memset(&pl, 0, sizeof pl);
        ctx->opt_cols = swg_variant_info(16, v).K;
        int rc = make_plan(ctx, 16, db->n_bins, &pl);
        ctx->opt_cols = keep_cols;
        if (rc != SWG_OK) continue;
        if ((rc = prepare_systolic(ctx, db, pl)) != SWG_OK) return rc;
        double ms_min = 1e300;
        for (int rep = 0; rep < 2; ++rep) {
            HIP_TRY(ctx, hipMemsetAsync(db->d_scores, 0, n_slots * 4, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(db->d_counters, 0, SWG_COUNTER_BYTES, ctx->stream));
            if ((rc = launch_systolic(ctx, db, pl, go, ge)) != SWG_OK) return rc;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            double ms = 0;
            if ((rc = diag_fill_ms(ctx, false, &ms)) != SWG_OK) return rc;
            ms_min = std::min(ms_min, ms);
        }
        if (ms_min < tuned->ms) {
            tuned->ms = ms_min;
            tuned->engine = 1;
            tuned->systolic_K = pl.K;
        }
    }
    return SWG_OK;
}

// ---------------------------------------------------------------------------
// the hot path
// ---------------------------------------------------------------------------
extern "C" uint64_t swg_hit_key(int32_t score, uint32_t index)
{
    return ((uint64_t)(uint32_t)(score < 0 ? 0 : score) << 32) | (uint64_t)(0xFFFFFFFFu - index);
}
extern "C" void swg_key_hit(uint64_t key, swg_hit *out)
{
    if (!out) return;
    out->score = (int32_t)(key >> 32);
    out->index = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
}
extern "C" size_t swg_topk_merge_keys(const uint64_t *keys, size_t n, size_t k, swg_hit *out)
{
    if (!keys || !out || k == 0) return 0;
    std::vector<uint64_t> v;
    v.reserve(n);
    for (size_t i = 0; i < n; ++i)
        if (keys[i] != 0) v.push_back(keys[i]);
    const size_t m = std::min(k, v.size());
    std::partial_sort(v.begin(), v.begin() + m, v.end(), std::greater<uint64_t>());
    for (size_t i = 0; i < m; ++i) swg_key_hit(v[i], &out[i]);
    return m;
}

// Pinned landing buffer of a slot's score read-out (a pageable destination would make the "async"
// copy block the host behind everything queued before it).
static int slot_scores(swg_ctx *ctx, SwgSlot *S, size_t n)
{
    if (n <= S->h_scores_cap) return SWG_OK;
    (void)hipHostFree(S->h_scores);
    S->h_scores = nullptr;
    S->h_scores_cap = 0;
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&S->h_scores), std::max<size_t>(n, 64) * sizeof(int32_t), hipHostMallocDefault));
    S->h_scores_cap = std::max<size_t>(n, 64);
    return SWG_OK;
}

// Queues one whole search on the context's stream and returns without waiting (except on the
// first search of a query length, which tunes the geometry, and when int16 scores may
// saturate, where the number of flagged sequences is read back to size the re-score).
static int search_begin(swg_ctx *ctx, const swg_db *db, bool want_scores, size_t k, SwgSlot *S)
{
    if (!ctx || !db) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_search: NULL argument");
    if (!ctx->have_scoring) return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_search: no scoring set");
    if (ctx->query.empty()) return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_search: no query set");
    if (db->device != ctx->device || !db->d_codes)
        return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_search: database is not resident on device %d",
                                 ctx->device);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        const int rs = ensure_slot(ctx, S);
        if (rs != SWG_OK) return rs;
    }
    ctx->cur = S;
    {
        const int rb = select_bufs(ctx, const_cast<swg_db *>(db), (int)(S - ctx->slots));
        if (rb != SWG_OK) return rb;
    }
    S->bufs = db->bufs[S - ctx->slots];
    S->db = db;
    S->k = k;
    S->want_scores = want_scores;
    swg_stats &st = S->st;
    memset(&st, 0, sizeof st);
    const size_t lq = ctx->query.size();
    const uint32_t n_bins = db->n_bins;
    const size_t n_slots = (size_t)n_bins * SWG_BIN;
    st.cells = (uint64_t)lq * db->residues;
    st.bytes_alg = db->residues + 8ull * db->n_local + 32ull * lq + 1024ull;
    S->bits = 0; // marks "nothing queued" for an empty database
    if (n_bins == 0) return SWG_OK;

    // which arithmetic: the packed int16 form needs non-positive gap scores
    const int go = ctx->gap_open + ctx->gap_extend, ge = ctx->gap_extend;
    const bool fast_ok = ctx->gap_open <= 0 && ctx->gap_extend <= 0 && -go <= 32767;
    int bits = fast_ok ? 16 : 32;
    if (ctx->opt_force_bits == 32) bits = 32;
    if (ctx->opt_force_bits == 16 && !fast_ok)
        return swg_set_ctx_error(ctx, SWG_ERR_ARG,
                                 "force_bits=16 needs gap_open <= 0 and gap_extend <= 0");

    // How high can a score get?  Not above the query's best possible total (every column paired
    // with its best-scoring residue) nor above the longest sequence times the largest table entry.
    // Below 32767 nothing can saturate; below 65535 the wide form of the diagonal engine (values
    // biased by -32768, same instruction count) is exact and nothing needs the int32 re-score.
    int smax = 0;
    uint64_t qbound = 0;
    for (int a = 0; a < 32; ++a)
        for (int b = 0; b < 32; ++b) smax = std::max<int>(smax, ctx->sub[a][b]);
    for (size_t i = 0; i < lq; ++i) {
        int best = 0;
        for (int b = 1; b < 32; ++b) best = std::max<int>(best, ctx->sub[(uint8_t)ctx->query[i] & 31][b]);
        qbound += (uint64_t)best;
    }
    const uint64_t longest = (uint64_t)db->max_nblk * SWG_ROWS_PER_BLK;
    const uint64_t score_bound = std::min<uint64_t>(qbound, std::min<uint64_t>(lq, longest) * (uint64_t)smax);
    S->score_bound = score_bound;
    bool wide = bits == 16 && score_bound >= 32767ull && ctx->opt_engine != 1 && ctx->opt_wide != 0;
    // The packed-f16 cells (three-operand maxima, 8.5 instead of 10 instructions per column pair) are exact
    // while scores stay below 4096; a sequence that reaches it is flagged and re-scored in int32.  They are the
    // first step whenever the gap magnitudes are f16 integers and the query is not so long that scores are
    // expected far beyond (where the wide form is exact on its own) -- unless this database has shown, for this
    // query, that a good part of its rows gets flagged ("f16" option: 0 never, 2 regardless of both).
    const bool want_f16 = bits == 16 && ctx->opt_engine != 1 && ctx->opt_dynamic != 0 && ctx->opt_f16 != 0 && -go <= 2048 &&
                          -ge <= 2048 && (ctx->opt_f16 == 2 || (score_bound < 32767ull && db->f16_veto_epoch != ctx->epoch));
    if (want_f16) wide = false; // (only with f16 = 2: the f16 cells first, whatever the score bound)
    const int plan_form = want_f16 ? 2 : 0;

    Plan main_pl, re_pl;
    memset(&main_pl, 0, sizeof main_pl);
    memset(&re_pl, 0, sizeof re_pl);
    // (the systolic plan: only the systolic engine needs it to exist -- a cols_per_wave meant for the
    // diagonal engine has no systolic instantiation, and with engine != 1 both widths run on lane groups)
    int rc = make_plan(ctx, bits, bits == 16 ? n_bins : n_bins * 2, &main_pl);
    if (rc != SWG_OK && ctx->opt_engine == 1) return rc;
    // int16: the diagonal engine unless the systolic one is asked for
    SwgDiagWork wk;
    bool use_diag = false, tuned_systolic = false;
    if (bits == 16 && ctx->opt_engine != 1) {
        if (ctx->opt_dynamic && (rc = ensure_pair_tokens(ctx, const_cast<swg_db *>(db))) != SWG_OK) return rc;
        const bool free_geometry = ctx->opt_cols == 0 && ctx->opt_group == 0 && ctx->opt_max_waves == 0 &&
                                   ctx->opt_long_split == 0 && ctx->opt_workgroups == 0;
        swg_db *mdb = const_cast<swg_db *>(db);
        const uint64_t tuned_key = (uint64_t)lq | ((uint64_t)plan_form << 40); // (a geometry is tuned for the cells it ran on)
        auto it = free_geometry ? mdb->tuned.find(tuned_key) : mdb->tuned.end();
        if (it == mdb->tuned.end() && free_geometry && ctx->opt_autotune && ctx->opt_engine == 0 &&
            db->n_local >= 4096 && db->n_local <= (4u << 20)) {
            SwgTuned tn;
            if (autotune_diag(ctx, mdb, lq, go, ge, &tn, plan_form) == SWG_OK && tn.wk.n_classes > 0)
                it = mdb->tuned.insert(std::make_pair(tuned_key, tn)).first;
        }
        if (it != mdb->tuned.end() && !(wide && it->second.engine == 1)) {
            if (it->second.engine == 1 && ctx->opt_engine == 0) {
                // the systolic engine measured faster for this database and query length
                const long keep = ctx->opt_cols;
                ctx->opt_cols = it->second.systolic_K;
                rc = make_plan(ctx, 16, n_bins, &main_pl);
                ctx->opt_cols = keep;
                if (rc != SWG_OK) return rc;
                tuned_systolic = true;
            } else {
                wk = it->second.wk;
                use_diag = true;
            }
        }
        if (!use_diag && !tuned_systolic)
            use_diag = swg_plan_diag_work(db, lq, ctx->n_cu, ctx->opt_cols, ctx->opt_group, ctx->opt_max_waves,
                                          ctx->opt_long_split, ctx->opt_workgroups == 0,
                                          ctx->opt_dynamic != 0 && db->ptok.ok, &wk, 1.0, plan_form) > 0;
        // The cost model's word on the ENGINE (round 4; until then only the autotuner could pick the systolic one, and it
        // is off for databases beyond 4 M sequences and wherever the caller turned it off): a database of short sequences
        // of near-equal length -- peptides -- is what the systolic engine is good at (no reset rows, no flags, nothing per
        // pair: 2 M peptides of 20-40 residues, lq 128: 6 570 GCUPS against the lane groups' 4 840, lq 30: 4 820 against
        // 1 930), and the two estimates tell: the lane groups' from the planner, the systolic one from the bin table.
        // It has to win by 15 % (both models are good to about 10 %).
        if (use_diag && !tuned_systolic && free_geometry && ctx->opt_engine == 0 && ctx->opt_f16 != 2 && !wide && !db->tokens_only && it == mdb->tuned.end() &&
            wk.plan[0].est_ms > 0.0) {
            int sys_K = 0;
            const bool sys_f16 = ctx->opt_f16 != 0 && -go <= 2048 && -ge <= 2048 && score_bound < 4096ull;
            const double sys_ms = swg_systolic_estimate_ms(db, lq, ctx->n_cu, &sys_K, sys_f16);
            if (sys_K > 0 && sys_ms < SWG_SYSTOLIC_MARGIN * wk.plan[0].est_ms * swg_diag_short_pair_factor(db, wk.plan[0], plan_form)) {
                const long keep = ctx->opt_cols;
                ctx->opt_cols = sys_K;
                const int rs = make_plan(ctx, 16, n_bins, &main_pl);
                ctx->opt_cols = keep;
                if (rs == SWG_OK) {
                    use_diag = false;
                    tuned_systolic = true;
                }
            }
        }
        if (!use_diag && !tuned_systolic && ctx->opt_engine == 2)
            return swg_set_ctx_error(ctx, SWG_ERR_ARG, "no diagonal-engine geometry for these options");
        if (use_diag && ctx->opt_workgroups > 0) {
            SwgDiagPlan &d0 = wk.plan[0];
            const uint64_t per_wg = (uint64_t)d0.W * (64 / d0.G);
            d0.workgroups = (int)std::min<long>(ctx->opt_workgroups, d0.workgroups);
            d0.n_streams = (uint32_t)((uint64_t)d0.workgroups * per_wg);
        }
        if (!use_diag && !tuned_systolic && rc != SWG_OK) return rc;
        // the wide form exists in the diagonal engine only
        if (wide && use_diag)
            for (int c = 0; c < wk.n_classes; ++c) wk.plan[c].wide = 1;
        else
            wide = false;
    }
    // The systolic engine on packed-f16 cells (round 4: 8.5 instead of 10 instructions per column pair) where NO score of
    // this search can reach their ceiling -- the engine has no flag-and-re-run route, and the databases the cost model
    // gives it are the ones whose longest sequence is short (372 residues x BLOSUM62's 11 < 4096).
    if (bits == 16 && !use_diag && main_pl.npass == 1 && ctx->opt_f16 != 0 && -go <= 2048 && -ge <= 2048 && score_bound < 4096ull)
        main_pl.f16 = 1;
    // int32 work (forced / unusual gap scores / re-score of saturated sequences) also runs on
    // the diagonal engine unless the systolic one is asked for
    const bool use_diag32 = ctx->opt_engine != 1;
    // Non-positive gap scores: the int32 work-queue kernel (8 instructions per cell, any lane-group
    // geometry, sequences off the queue) instead of the bin-based one (12 per cell, 64 lanes x 16 columns
    // whatever the query length).
    bool q32_ok = fast_ok && use_diag32 && ctx->opt_dynamic != 0;
    {
        SwgDiagWork probe; // (one pass up to about 1150 columns, else several)
        q32_ok = q32_ok && q32_list_plan(ctx, lq, 1, &probe);
    }
    // the f16 cells: every class on the work queue, and a re-score path for what they flag
    bool use_f16 = want_f16 && use_diag;
    for (int c = 0; use_f16 && c < wk.n_classes; ++c) use_f16 = diag_class_is_dynamic(ctx, db, wk.plan[c]);
    if (q32_ok && (bits == 32 || score_bound >= (wide ? 65535ull : 32767ull))) {
        if ((rc = ensure_pair_tokens(ctx, const_cast<swg_db *>(db))) != SWG_OK) return rc;
        SwgDiagWork probe;
        q32_ok = db->ptok.ok && q32_list_plan(ctx, lq, 1, &probe) &&
                 (probe.plan[0].npass == 1 || db->ptok.total_blocks < (1ull << 28)); // (32-bit edge indices)
    }
    // A query long enough to score beyond 32767 gets the wide form (wide16 = 0: plain int16 cells and the int32
    // re-score from 32767) -- but only a sequence that is long itself can get
    // anywhere near: an exact copy of a stretch of the query scores qbound / lq per row on average, so one of fewer
    // than 4096 * lq / qbound rows stays below the f16 cells' ceiling even then.  Those (most of a protein database:
    // 730 rows with BLOSUM62) take the f16 cells, 8.5 instructions per column pair instead of 10, in launches of
    // their own after the long ones' (launch_diag).  The threshold is an expectation, not a bound: whatever the f16
    // cells flag all the same is run again on the wide form like any other flagged pair, so results do not depend on it.
    uint32_t split_at = 0, split_rows = 0;
    uint64_t split_residues = 0;
    if (!use_f16 && bits == 16 && score_bound >= 32767ull && use_diag && ctx->opt_f16 == 1 && db->f16_veto_epoch != ctx->epoch && -go <= 2048 && -ge <= 2048 &&
        wk.n_classes == 1 && diag_class_is_dynamic(ctx, db, wk.plan[0]) && !db->tokens_only && qbound > 0) {
        const uint32_t rows = swg_split_rows(lq, qbound);
        swg_db_split_at(const_cast<swg_db *>(db), rows); // (per database and length: a binary search and one pass over the lengths)
        const uint32_t cut = std::max<uint32_t>(db->split_pair, (uint32_t)wk.pair_begin[0]);
        if (cut <= wk.pair_begin[0] && (score_bound < 65535ull || q32_ok)) {
            // nothing long in this database: the f16 cells for all of it (what they flag: the wide form, as below)
            use_f16 = true;
            wide = false;
            wk.plan[0].wide = 0;
        } else if (cut > wk.pair_begin[0] && cut < wk.pair_end[0]) {
            split_at = cut;
            split_rows = rows;
            split_residues = db->split_residues;
        }
    }
    wk.plan[0].f16_from = split_at;
    // Several passes of G*K columns each leave the last one partly empty (3000 columns in 6 passes of 512: 72 of them,
    // 2.3 % of the work): it runs the instantiation with the fewest columns per lane that cover what is left.
    for (int c = 0; c < wk.n_classes; ++c) {
        SwgDiagPlan &pl = wk.plan[c];
        pl.last_variant = -1;
        pl.last_K = 0;
        if (!use_diag || wk.n_classes != 1 || pl.npass < 2 || !diag_class_is_dynamic(ctx, db, pl) || ctx->opt_last_pass == 0) continue;
        if (!swg_plan_last_pass(pl, lq, &pl.last_variant, &pl.last_K)) pl.last_variant = -1, pl.last_K = 0;
    }
    // what the f16 cells flag is run again on int16 cells (the wide form if scores may pass 32767); only what
    // saturates those too needs the int32 kernel
    const bool rerun_wide = (use_f16 && score_bound >= 32767ull && ctx->opt_wide != 0) || (split_at != 0u && wide);
    if (use_f16 && score_bound >= (rerun_wide ? 65535ull : 32767ull) && !q32_ok) use_f16 = false;
    for (int c = 0; c < wk.n_classes; ++c) wk.plan[c].f16 = use_f16 ? 1 : 0;
    const bool some_f16 = use_f16 || split_at != 0u; // some pairs run on the f16 cells: their flags are collected and re-run
    const int32_t ceiling = use_f16 ? 4096 : wide ? 65535 : 32767;
    const bool may_saturate = bits == 16 && (score_bound >= (uint64_t)ceiling || split_at != 0u);
    if (may_saturate) {
        const long keep_cols = ctx->opt_cols;
        ctx->opt_cols = 0; // the int32 re-score uses its default geometry
        rc = make_plan(ctx, 32, (uint32_t)std::min<size_t>(n_slots / 64, 1u << 30), &re_pl);
        ctx->opt_cols = keep_cols;
        if (rc != SWG_OK) return rc;
    }
    const int npass32 = (int)((lq + 64 * SWG_DIAG32_K - 1) / (64 * SWG_DIAG32_K));
    SwgDiagWork wk32;
    bool use_q32 = false, exact32 = false;
    if (bits == 32 && !fast_ok && use_diag32 && ctx->opt_dynamic != 0) {
        // gap scores the reduced algebra cannot express (a positive one): the same work-queue kernel on the
        // exact cells, unless the token array is beyond its 32-bit edge indices
        if ((rc = ensure_pair_tokens(ctx, const_cast<swg_db *>(db))) != SWG_OK) return rc;
        if (db->ptok.ok) {
            // the planner's split into a bulk and a long class where it fits the int32 profile and the exact cells'
            // register budget (config 2's shape: 2 580 GCUPS as one class, the longest pairs' chains last), else one class
            bool two = swg_plan_diag_work(db, lq, ctx->n_cu, ctx->opt_cols, ctx->opt_group, ctx->opt_max_waves, ctx->opt_long_split,
                                          ctx->opt_workgroups == 0, true, &wk32) > 0 && q32_plan_fits(wk32, lq);
            for (int c = 0; two && c < wk32.n_classes; ++c) two = wk32.plan[c].K <= SWG_X32_MAX_K;
            if (two || (x32_plan(ctx, db, lq, &wk32) && (wk32.plan[0].npass == 1 || db->ptok.total_blocks < (1ull << 28))))
                use_q32 = exact32 = true;
        }
    }
    if (bits == 32 && q32_ok) {
        use_q32 = swg_plan_diag_work(db, lq, ctx->n_cu, ctx->opt_cols, ctx->opt_group, ctx->opt_max_waves, ctx->opt_long_split,
                                     ctx->opt_workgroups == 0, true, &wk32) > 0 && q32_plan_fits(wk32, lq);
        if (!use_q32 && ctx->opt_cols > 0 && ctx->opt_group > 0) {
            // a forced geometry that needs several passes (or whose long class did not fit): one class of exactly
            // that geometry, a launch per pass -- if its int32 profile fits LDS; otherwise the library's own pick
            // below, which swg_stats reports
            for (int v = 0; v < swg_num_diag_variants() && !use_q32; ++v) {
                const int K = swg_diag_variant_info(v).K, G = (int)ctx->opt_group;
                if (K != (int)ctx->opt_cols || swg_diag32q_lds_bytes(K, G, 4) > 160 * 1024) continue;
                const size_t np = (lq + (size_t)G * K - 1) / ((size_t)G * K);
                if (np > 64 || (np > 1 && db->ptok.total_blocks >= (1ull << 28))) continue;
                wk32 = SwgDiagWork();
                wk32.n_classes = 1;
                wk32.plan[0].variant = v;
                wk32.plan[0].K = K;
                wk32.plan[0].G = G;
                wk32.plan[0].W = 4;
                wk32.plan[0].npass = (int)np;
                wk32.pair_begin[0] = 0;
                wk32.pair_end[0] = swg_db_pair_count(db);
                use_q32 = true;
            }
        }
        if (!use_q32) {
            // the int16 planner's choice does not fit (LDS holds half as many int32 columns): fewest lanes that do
            use_q32 = q32_list_plan(ctx, lq, (uint32_t)std::min<size_t>(n_slots, 1u << 30), &wk32);
            if (use_q32) {
                wk32.pair_begin[0] = 0;
                wk32.pair_end[0] = swg_db_pair_count(db);
            }
        }
    }
    // (the int32 level follows the last 16-bit one: the fill's own cells, or the re-run's when f16 cells came first)
    const bool int32_level_planned = bits == 16 && score_bound >= (use_f16 ? (rerun_wide ? 65535ull : 32767ull) : (uint64_t)ceiling);
    const bool bin32 = use_diag32 && ((bits == 32 && !use_q32) || (int32_level_planned && !q32_ok)); // the bin-based int32 kernel is needed
    if (bin32) {
        rc = ensure_profile_cols(ctx, 1, (uint32_t)(npass32 * 64 * SWG_DIAG32_K), 4, (1ull << 30) ^ (uint64_t)npass32);
        if (rc != SWG_OK) return rc;
        const size_t per_wave = ((size_t)db->max_nblk * 4 + 4) * 4; // dwords: one uint4 per stream row
        if ((rc = ensure_scratch(ctx, npass32 > 1 ? per_wave * 16 * (size_t)ctx->n_cu : 0)) != SWG_OK) return rc;
    }
    if (use_diag) {
        rc = prepare_diag(ctx, const_cast<swg_db *>(db), wk);
    } else if (use_q32) {
        rc = SWG_OK; // profiles are built at the launch
    } else if (!(bits == 32 && use_diag32)) {
        rc = ensure_profile(ctx, main_pl);
    }
    if (rc != SWG_OK) return rc;
    if (may_saturate && !use_diag32 && (rc = ensure_profile(ctx, re_pl)) != SWG_OK) return rc;
    {
        size_t need = 0;
        if (bin32 && npass32 > 1) need = ((size_t)db->max_nblk * 4 + 4) * 4 * 16 * (size_t)ctx->n_cu;
        if (!use_diag && !(bits == 32 && use_diag32) && main_pl.npass > 1)
            need = std::max(need, (size_t)main_pl.workgroups * db->max_nblk * SWG_ROWS_PER_BLK * 64 * main_pl.info.nb);
        if (may_saturate && !use_diag32 && re_pl.npass > 1)
            need = std::max(need, (size_t)re_pl.workgroups * db->max_nblk * SWG_ROWS_PER_BLK * 64 * re_pl.info.nb);
        if ((rc = ensure_scratch(ctx, need)) != SWG_OK) return rc;
    }

    // the systolic engine and the bin-based int32 kernel read the bin image (built on the device on first use)
    if (((!use_diag && !use_q32) || (int32_level_planned && !q32_ok)) && (rc = ensure_bins(ctx, const_cast<swg_db *>(db))) != SWG_OK)
        return rc;

    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[0], s));
    HIP_TRY(ctx, swg_launch_zero2(db->d_scores, n_slots * 4, db->d_counters, SWG_COUNTER_BYTES, s)); // (one launch, not two memsets)

    SwgFillParams p;
    memset(&p, 0, sizeof p);
    p.residues = db->d_packed;
    p.bin_off = db->d_bin_off;
    p.bin_nblk = db->d_bin_nblk;
    p.n_bins = n_bins;
    p.scores = db->d_scores;
    p.scratch = ctx->d_scratch;

    bool two_ends = false;
    ctx->cur->fill_launches = 0;
    if (!use_diag && !use_q32) HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[1], s));
    if (use_diag) {
        if ((rc = launch_diag(ctx, db, wk, go, ge, &two_ends)) != SWG_OK) return rc;
    } else if (use_q32) {
        if ((rc = launch_q32(ctx, db, wk32, go, ge, nullptr, nullptr, 0, db->d_counters + SWG_QUEUE_WORD(0), &two_ends, true, exact32)) != SWG_OK)
            return rc;
    } else if (bits == 32 && use_diag32) {
        p.profile = ctx->d_profile[1];
        p.queue = db->d_counters + 0;
        p.list = nullptr;
        p.list_count = nullptr;
        p.n_items = (uint32_t)n_slots;
        p.npass = (uint32_t)npass32;
        p.go = go;
        p.ge = ge;
        p.scratch_wg_dwords = ((uint64_t)db->max_nblk * 4 + 4) * 4;
        HIP_TRY(ctx, swg_launch_diag32(16, (int)std::min<size_t>((size_t)ctx->n_cu, (n_slots + 15) / 16), p, s));
    } else {
        p.profile = ctx->d_profile[bits == 16 ? 0 : 1];
        p.queue = db->d_counters + 0;
        p.list = nullptr;
        p.list_count = nullptr;
        p.n_items = bits == 16 ? n_bins : n_bins * 2;
        p.npass = (uint32_t)main_pl.npass;
        if (bits == 16 && main_pl.f16) {
            p.go = (int32_t)f16x2_of(-go);
            p.ge = (int32_t)f16x2_of(-ge);
        } else if (bits == 16) {
            const uint32_t g = (uint32_t)(-go) & 0xFFFFu, e = (uint32_t)(-ge) & 0xFFFFu;
            p.go = (int32_t)(g | (g << 16));
            p.ge = (int32_t)(e | (e << 16));
        } else {
            p.go = go;
            p.ge = ge;
        }
        p.scratch_wg_dwords = (uint64_t)db->max_nblk * SWG_ROWS_PER_BLK * 64 * main_pl.info.nb;
        HIP_TRY(ctx, swg_launch_fill(bits, main_pl.variant, main_pl.W, main_pl.workgroups, p, s, bits == 16 && main_pl.f16 != 0));
    }
    if (!use_diag && !use_q32) HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[2], s));
    int32_t level_ceiling = ceiling; // what the fill before the int32 level saturates at
    bool int32_level = may_saturate;
    uint32_t *seq_list = db->d_list;
    if (may_saturate && some_f16) {
        // counters [17] = flagged pairs (the list's length), [16] = flagged sequences, [6] = their rows / 16 (the veto's input)
        HIP_TRY(ctx, swg_launch_collect_flagged_pairs(db->d_scores, split_at, (uint32_t)(n_slots / 2), 4096, db->d_list, db->d_counters + 17,
                                                      db->d_counters + 16, db->d_lens, db->d_counters + 6, s));
        SwgDiagPlan lp;
        const uint32_t guess = db->sat_hint > 0 ? (uint32_t)std::min<long long>(db->sat_hint, 1ll << 30) : 1u;
        if (!i16_list_plan(ctx->n_cu, lq, guess, wk.plan[0], &lp)) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "no geometry for the int16 re-run");
        HIP_TRY(ctx, hipMemsetAsync(db->d_counters + SWG_QUEUE_WORD(0), 0, (size_t)(SWG_COUNTER_BYTES - SWG_QUEUE_WORD(0) * 4u), s));
        if ((rc = launch_dyn_list(ctx, const_cast<swg_db *>(db), lp, rerun_wide, go, ge, db->d_list, db->d_counters + 17, s)) != SWG_OK) return rc;
        level_ceiling = rerun_wide ? 65535 : 32767;
        int32_level = score_bound >= (uint64_t)level_ceiling;
        seq_list = db->d_list + n_slots; // (the pair list keeps the first half)
    }
    if (int32_level) {
        // counters [1] = saturated sequences (the list's length), [6] = their rows in units of 16
        HIP_TRY(ctx, swg_launch_collect_saturated(db->d_scores, (uint32_t)n_slots, level_ceiling, seq_list, db->d_counters + 1,
                                                  some_f16 ? nullptr : db->d_lens, db->d_counters + 6, s));
        p.profile = ctx->d_profile[1];
        p.queue = db->d_counters + 2;
        p.list = seq_list;
        p.list_count = db->d_counters + 1;
        p.n_items = 0;
        p.go = go;
        p.ge = ge;
        SwgDiagWork wkl;
        // The work-queue re-score reads the count on the device and leaves at once when it is zero, so it is queued
        // behind every fill that may flag something and the host never waits for the count in the middle of a
        // search (it did until round 3: a round trip per search, and the end of the two-deep pipeline of
        // swg_search_begin).  Only its lane-group width is a guess -- few flagged sequences get 64 lanes each,
        // many the narrowest group that covers the query -- made from what the last search of this database saw.
        const uint32_t guess = !some_f16 && db->sat_hint > 0 ? (uint32_t)std::min<long long>(db->sat_hint, 1ll << 30) : 1u;
        if (use_diag32 && q32_ok && q32_list_plan(ctx, lq, guess, &wkl)) {
            // (fresh queue counters and rank table: the fill's are spent; no events of its own: the
            // re-score is timed as ev[2] .. ev[3] like the other re-score forms)
            HIP_TRY(ctx, hipMemsetAsync(db->d_counters + SWG_QUEUE_WORD(0), 0,
                                        (size_t)(SWG_COUNTER_BYTES - SWG_QUEUE_WORD(0) * 4u), s));
            bool two = false;
            rc = launch_q32(ctx, db, wkl, go, ge, seq_list, db->d_counters + 1, std::max<uint32_t>(2u * guess, 4096u),
                            db->d_counters + SWG_QUEUE_WORD(0), &two, false);
            if (rc != SWG_OK) return rc;
        } else if (use_diag32) {
            // the bin-based kernel (positive gap scores never get here; a database beyond the queue's
            // indices, work_queue = 0): its shape comes from the count, read back over PCIe
            uint32_t n_sat = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&n_sat, db->d_counters + 1, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, spin_sync(ctx, s));
            if (n_sat > 0) {
                int W = (int)((n_sat + (uint32_t)ctx->n_cu - 1) / (uint32_t)ctx->n_cu);
                W = std::min(16, std::max(4, (W + 3) / 4 * 4));
                const int wgs = (int)std::min<uint32_t>((uint32_t)ctx->n_cu, (n_sat + W - 1) / W);
                p.npass = (uint32_t)npass32;
                p.scratch_wg_dwords = ((uint64_t)db->max_nblk * 4 + 4) * 4;
                HIP_TRY(ctx, swg_launch_diag32(W, wgs, p, s));
            }
        } else {
            p.npass = (uint32_t)re_pl.npass;
            p.scratch_wg_dwords = (uint64_t)db->max_nblk * SWG_ROWS_PER_BLK * 64 * re_pl.info.nb;
            HIP_TRY(ctx, swg_launch_fill(32, re_pl.variant, re_pl.W, re_pl.workgroups, p, s));
        }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[3], s));

    // Top-K and read-out go to their own stream: the next search's fill is queued right behind this
    // one's on the main stream and these small kernels run beside its start instead of holding it
    // up (the output buffers belong to this in-flight slot until swg_search_end).
    // (the read-out stream exists from the context's second search on: ensure_stream3)
    S->side = false;
    if (ctx->opt_side_readout && ctx->n_begun > 0) {
        const int r3 = ensure_stream3(ctx);
        if (r3 != SWG_OK) return r3;
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream3, ctx->cur->ev[3], 0));
        s = ctx->stream3;
        S->side = true;
    }
    ++ctx->n_begun;
    // top-K on the device unless every score goes to the host anyway
    const bool dev_topk = k > 0 && !want_scores && k <= SWG_TOPK_CAND_CAP / 2;
    if (dev_topk)
        HIP_TRY(ctx, swg_launch_topk(db->d_scores, db->d_order, (uint32_t)n_slots, (uint32_t)k, db->d_hist,
                                     db->d_counters + 4, db->d_keys, SWG_TOPK_CAND_CAP, db->d_counters + 3, s));
    HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[4], s));

    // read-out, queued behind the kernels; swg_search_end waits for it
    S->bits = bits;
    S->dev_topk = dev_topk;
    S->need_scores = want_scores || (k > 0 && !dev_topk);
    S->first_chunk = SWG_TOPK_CAND_CAP;
    S->two_ends = two_ends;
    S->may_saturate = may_saturate;
    S->use_diag = use_diag;
    S->use_diag32 = use_diag32;
    S->use_q32 = use_q32;
    S->used_f16 = some_f16;
    S->split_rows = split_rows;
    S->split_residues = split_residues;
    S->epoch = ctx->epoch;
    S->wk32 = wk32;
    S->npass32 = npass32;
    S->wk = wk;
    S->main_K = main_pl.K;
    S->main_f16 = main_pl.f16 != 0;
    S->main_W = main_pl.W;
    S->main_npass = main_pl.npass;
    S->main_wgs = main_pl.workgroups;
    if (dev_topk)
        HIP_TRY(ctx, hipMemcpyAsync(S->h_cand, db->d_keys, S->first_chunk * 8, hipMemcpyDeviceToHost, s));
    if (S->need_scores) {
        if ((rc = slot_scores(ctx, S, n_slots)) != SWG_OK) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(S->h_scores, db->d_scores, n_slots * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipMemcpyAsync(S->h_counters, db->d_counters, 128, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipEventRecord(S->ev_done, s));
    return SWG_OK;
}

static int search_end(swg_ctx *ctx, SwgSlot *S, int32_t *scores_out, swg_hit *topk_out, size_t *n_hits,
                      swg_stats *stats)
{
    const swg_db *db = S->db;
    const size_t k = S->k;
    if (n_hits) *n_hits = 0;
    if (k > 0 && !topk_out) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_search: k > 0 but topk_out NULL");
    if (scores_out && !S->want_scores)
        return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_search_end: scores were not requested at swg_search_begin");
    swg_stats &st = S->st;
    if (S->bits == 0) { // empty database
        if (stats) *stats = st;
        return SWG_OK;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->cur = S;
    hipStream_t s = S->side ? ctx->stream3 : ctx->stream; // (the stream this search's read-out was queued on: not behind queued fills)
    for (;;) { // poll: a blocking wait can cost milliseconds of wake-up latency on a busy host
        const hipError_t q = hipEventQuery(S->ev_done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIP_TRY(ctx, q);
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    const size_t n_slots = (size_t)db->n_bins * SWG_BIN;
    const bool dev_topk = S->dev_topk, two_ends = S->two_ends, may_saturate = S->may_saturate;
    const bool use_diag = S->use_diag, use_diag32 = S->use_diag32;
    const int bits = S->bits, npass32 = S->npass32;
    const SwgDiagWork &wk = S->wk;
    const SwgDiagPlan &dpl = wk.plan[0];
    uint32_t *h_counters = S->h_counters;
    uint64_t *h_cand = S->h_cand;
    int32_t *&h_scores = S->h_scores;
    int rc = SWG_OK;
    bool cand_ok = dev_topk && h_counters[5] == 0 && h_counters[3] <= SWG_TOPK_CAND_CAP;
    if (dev_topk && !cand_ok) { // threshold beyond the histogram or too many ties: select on the host
        if ((rc = slot_scores(ctx, S, n_slots)) != SWG_OK) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(h_scores, S->bufs.d_scores, n_slots * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, spin_sync(ctx, s));
    }

    float ms = 0.f;
    if ((rc = diag_fill_ms(ctx, two_ends, &st.fill_ms)) != SWG_OK) return rc;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->cur->ev[2], ctx->cur->ev[3]));
    st.rescore_ms = may_saturate ? ms : 0.0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->cur->ev[0], ctx->cur->ev[4]));
    st.total_ms = ms;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->cur->ev[3], ctx->cur->ev[4]));
    const double topk_dev_ms = ms;
    st.n_rescored = S->used_f16 ? h_counters[16] : h_counters[1];
    st.path_bits = bits;
    st.cell_form = use_diag ? diag_class_form(ctx, db, dpl) : S->main_f16 ? 2 : 0;
    if (use_diag && dpl.f16_from != 0u) {
        st.cell_form = dpl.wide ? 4 : 5;
        st.n_rescored = (uint64_t)h_counters[16] + h_counters[1]; // flagged by the f16 cells + saturated on the wide form
        st.split_rows = (int32_t)S->split_rows;
        st.fill_f16_launches = S->fill_f16_launches;
        st.cells_f16 = st.cells / std::max<uint64_t>(1, db->residues) * S->split_residues; // (cells = lq * residues)
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->cur->ev[5], ctx->cur->ev[2]));
        st.fill_f16_ms = ms;
    }
    if (may_saturate) {
        // what the next search's plan may assume (never its results)
        swg_db *mdb = const_cast<swg_db *>(db);
        mdb->sat_hint = (long long)(S->used_f16 ? h_counters[17] : h_counters[1]); // (f16: pairs; else sequences)
        // f16 cells whose flagged pairs hold more than 1/16 of the pair rows (counter 6: rows of the flagged pairs / 16;
        // a pair row is two residues) cost more in re-runs than they save: the cells save 15 % of the fill, the list
        // re-run of a share f of the rows costs f x 10 / 8.5 at about half the fill's efficiency.  Measured on config
        // 4's share with 4 % of the pair rows flagged: f16 + re-run 191 ms, int16 cells alone 200.
        if (S->used_f16 && (uint64_t)h_counters[6] * 16ull * 2ull * 16ull > db->residues + 2ull * db->n_local) mdb->f16_veto_epoch = S->epoch;
    }
    st.classes_overlapped = -1;
    if (use_diag && wk.n_classes == 2 && wk.plan[0].npass == 1 && wk.plan[1].npass == 1) {
        // did the two classes run side by side?  (stamps: complement of the earliest start, latest end)
        unsigned long long t[4];
        memcpy(t, h_counters + 8, sizeof t);
        if (t[0] && t[1] && t[2] && t[3]) {
            const unsigned long long bulk_start = ~t[0], bulk_end = t[1], long_start = ~t[2];
            st.classes_overlapped = (bulk_end > bulk_start && long_start < bulk_start + (bulk_end - bulk_start) / 10) ? 1 : 0;
        }
    }
    if (use_diag) {
        st.engine = 2;
        st.cols_per_wave = dpl.K;
        st.group_lanes = dpl.G;
        st.waves = dpl.W;
        st.passes = dpl.npass;
        st.fill_launches = S->fill_launches > 0 ? S->fill_launches : dpl.npass;
        st.workgroups = diag_class_workgroups(ctx, db, wk, 0);
        st.work_queue = diag_class_is_dynamic(ctx, db, dpl) ? 1 : 0;
        st.streams = (int32_t)diag_class_streams(ctx, db, wk, 0);
        st.cells_padded = 2ull * ((uint64_t)(dpl.npass - (dpl.last_variant >= 0 ? 1 : 0)) * dpl.K + (dpl.last_variant >= 0 ? dpl.last_K : 0)) *
                          dpl.G * diag_class_blocks(ctx, db, wk, 0) * 4ull;
        st.last_pass_cols = dpl.last_variant >= 0 ? dpl.last_K : 0;
        if (wk.n_classes == 2) {
            const SwgDiagPlan &lp = wk.plan[1];
            st.long_pairs = (int32_t)(wk.pair_end[1] - wk.pair_begin[1]);
            st.long_cols_per_lane = lp.K;
            st.long_streams = (int32_t)diag_class_streams(ctx, db, wk, 1);
            st.cells_padded += 2ull * lp.npass * lp.G * lp.K * diag_class_blocks(ctx, db, wk, 1) * 4ull;
        }
    } else if (S->use_q32) {
        const SwgDiagWork &w32 = S->wk32;
        st.engine = 2;
        st.work_queue = 1;
        st.cols_per_wave = w32.plan[0].K;
        st.group_lanes = w32.plan[0].G;
        st.waves = q32_class_waves(ctx, w32, 0);
        st.passes = w32.plan[0].npass;
        const uint64_t items0 = 2 * (w32.pair_end[0] - w32.pair_begin[0]);
        st.workgroups = q32_class_workgroups(ctx, w32, 0, items0);
        st.streams = st.workgroups * st.waves * (64 / w32.plan[0].G);
        for (int c = 0; c < w32.n_classes; ++c)
            st.cells_padded += 2ull * w32.plan[c].npass * w32.plan[c].G * w32.plan[c].K *
                               (uint64_t)(db->ptok.pair_blocks_prefix[w32.pair_end[c]] - db->ptok.pair_blocks_prefix[w32.pair_begin[c]]) * 4ull;
        if (w32.n_classes == 2) {
            st.long_pairs = (int32_t)(w32.pair_end[1] - w32.pair_begin[1]);
            st.long_cols_per_lane = w32.plan[1].K;
        }
    } else if (bits == 32 && use_diag32) {
        st.engine = 2;
        st.cols_per_wave = SWG_DIAG32_K;
        st.group_lanes = 64;
        st.waves = 16;
        st.passes = npass32;
        st.workgroups = (int)std::min<size_t>((size_t)ctx->n_cu, (n_slots + 15) / 16);
        st.cells_padded = (uint64_t)npass32 * 64 * SWG_DIAG32_K * ((uint64_t)db->rows_padded);
    } else {
        st.engine = 1;
        st.cols_per_wave = S->main_K;
        st.waves = S->main_W;
        st.passes = S->main_npass;
        st.workgroups = S->main_wgs;
        st.cells_padded = (uint64_t)S->main_npass * S->main_W * S->main_K * db->rows_padded;
    }
    if (st.fill_launches <= 0) st.fill_launches = std::max(1, st.passes);

    const auto t0 = std::chrono::steady_clock::now();
    if (scores_out) {
        for (size_t i = 0; i < n_slots; ++i) {
            const uint32_t oi = db->order[i];
            if (oi != 0xFFFFFFFFu) scores_out[oi] = h_scores[i];
        }
    }
    if (k > 0) {
        std::vector<uint64_t> keys;
        if (cand_ok) {
            keys.assign(h_cand, h_cand + h_counters[3]);
        } else {
            fail_alloc_here(3);
            keys.reserve(db->n_local);
            for (size_t i = 0; i < n_slots; ++i) {
                const uint32_t oi = db->order[i];
                if (oi != 0xFFFFFFFFu) keys.push_back(swg_hit_key(h_scores[i], oi));
            }
        }
        const size_t m = std::min(k, keys.size());
        std::partial_sort(keys.begin(), keys.begin() + m, keys.end(), std::greater<uint64_t>());
        for (size_t i = 0; i < m; ++i) swg_key_hit(keys[i], &topk_out[i]);
        if (n_hits) *n_hits = m;
    }
    st.topk_ms = topk_dev_ms + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    return SWG_OK;
}

extern "C" int swg_search_begin(swg_ctx *ctx, const swg_db *db, int want_scores, size_t k, int *ticket)
{
    return ctx_guarded(ctx, "swg_search_begin", [&]() -> int {
        fail_alloc_here(1);
        if (!ctx || !db || !ticket) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_search_begin: NULL argument");
        int slot = -1;
        for (int i = 0; i < SWG_MAX_INFLIGHT; ++i) {
            const int c = (ctx->next_slot + i) % SWG_MAX_INFLIGHT;
            if (!ctx->slots[c].busy) {
                slot = c;
                break;
            }
        }
        if (slot < 0)
            return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_search_begin: %d searches already in flight", SWG_MAX_INFLIGHT);
        const int rc = search_begin(ctx, db, want_scores != 0, k, &ctx->slots[slot]);
        if (rc != SWG_OK) return rc;
        ctx->slots[slot].busy = true;
        ctx->next_slot = (slot + 1) % SWG_MAX_INFLIGHT;
        *ticket = slot;
        return SWG_OK;
    });
}

extern "C" int swg_search_end(swg_ctx *ctx, int ticket, int32_t *scores_out, swg_hit *topk_out, size_t *n_hits,
                              swg_stats *stats)
{
    bool valid = false; // the ticket named a search in flight: it is spent whatever happens next (a failed read-out is not retried)
    const int rc = ctx_guarded(ctx, "swg_search_end", [&]() -> int {
        fail_alloc_here(2);
        if (!ctx || ticket < 0 || ticket >= SWG_MAX_INFLIGHT || !ctx->slots[ticket].busy)
            return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_search_end: no search in flight under that ticket");
        valid = true;
        return search_end(ctx, &ctx->slots[ticket], scores_out, topk_out, n_hits, stats);
    });
    if (valid) ctx->slots[ticket].busy = false;
    return rc;
}

extern "C" int swg_search(swg_ctx *ctx, const swg_db *db, int32_t *scores_out, swg_hit *topk_out, size_t k,
                          size_t *n_hits, swg_stats *stats)
{
    if (k > 0 && !topk_out) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_search: k > 0 but topk_out NULL");
    int ticket = -1;
    const int rc = swg_search_begin(ctx, db, scores_out != nullptr, k, &ticket);
    if (rc != SWG_OK) return rc;
    return swg_search_end(ctx, ticket, scores_out, topk_out, n_hits, stats);
}

// ---------------------------------------------------------------------------
// many queries against one resident database in one pass
// ---------------------------------------------------------------------------
// The reference runs one query per process (src/alignment_cmdline.c:381-396 reads a single query
// record); its report says the design "extends naturally" to many-to-many (Final Report p.7).  Here
// that is one launch per class for a whole batch of queries: row y of the grid works for query y (its
// own profile, its own pair queue, its own score array), so a database too small to fill the chip
// with one query -- where a search lasts as long as its longest pair's chain of rows -- fills it with
// many.  Queries that cannot take this path (several passes, scores that may saturate int16, gap
// scores outside the packed form, options that ask for another engine) are searched one after
// another with swg_search: same results, no batching.
namespace {
struct MultiBufs {
    int8_t *d_q = nullptr;
    uint32_t *d_qoff = nullptr;
    uint32_t *d_order = nullptr;
    uint8_t *d_prof[2] = {nullptr, nullptr};
    int32_t *d_scores = nullptr;
    uint32_t *d_cnt = nullptr;
    uint32_t *d_hist = nullptr, *d_meta = nullptr; // device top-K of the batch (no score array asked for)
    uint64_t *d_cand = nullptr;
    // How many rows each per-query buffer holds, written where the buffer is allocated and checked against what a
    // launch will index before every launch (multi_rows_ok): with two queries per lane the grid's row y stands for
    // queries 2y and 2y+1, so a batch of odd size indexes ONE ROW MORE than it has queries.  Round 3 sized these
    // buffers by the queries and a wavefront of the last pair ran off the end ("the fault of round 3").
    size_t rows_scores = 0, rows_order = 0, rows_prof[2] = {0, 0}, rows_cnt = 0, rows_topk = 0;
    ~MultiBufs()
    {
        (void)hipFree(d_hist);
        (void)hipFree(d_meta);
        (void)hipFree(d_cand);
        (void)hipFree(d_q);
        (void)hipFree(d_qoff);
        (void)hipFree(d_order);
        (void)hipFree(d_prof[0]);
        (void)hipFree(d_prof[1]);
        (void)hipFree(d_scores);
        (void)hipFree(d_cnt);
    }
};
} // namespace

// Every per-query buffer of a batch launch against the rows the launch indexes: Qb queries, Qrows grid rows (query
// pairs when two queries share a lane: the kernels then address score rows 2y and 2y+1 for y < Qrows, i.e. Qb + 1
// rows for an odd batch).  A mismatch is a bug of this file; it is reported, not launched.
static int multi_rows_ok(swg_ctx *ctx, const MultiBufs &B, bool qq, size_t Qb, size_t Qrows, int n_classes, bool dev_topk)
{
    const size_t score_rows = qq ? 2 * Qrows : Qb; // rows of d_scores a launch may write
    const size_t order_rows = qq ? 2 * Qrows : 0;  // entries of d_order the profile builder may read
    struct { const char *name; size_t have, need; } chk[] = {
        {"d_scores", B.rows_scores, score_rows},       {"d_order", B.rows_order, order_rows},
        {"d_prof[0]", B.rows_prof[0], Qrows},          {"d_prof[1]", n_classes == 2 ? B.rows_prof[1] : Qrows, Qrows},
        {"d_cnt", B.rows_cnt, Qrows},                  {"d_hist/d_meta/d_cand", dev_topk ? B.rows_topk : Qb, Qb},
    };
    if ((qq && Qrows != (Qb + 1) / 2) || (!qq && Qrows != Qb))
        return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_search_multi: %zu grid rows for %zu queries (qq %d)", Qrows, Qb, (int)qq);
    for (const auto &c : chk)
        if (c.have < c.need)
            return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_search_multi: %s holds %zu rows, the launch indexes %zu (batch of %zu, qq %d)",
                                     c.name, c.have, c.need, Qb, (int)qq);
    return SWG_OK;
}

static void multi_deliver(const swg_db *db, const int32_t *h_scores, size_t n_slots, int32_t *scores_out, swg_hit *topk_out,
                          size_t k, size_t *n_hits)
{
    if (scores_out)
        for (size_t i = 0; i < n_slots; ++i) {
            const uint32_t oi = db->order[i];
            if (oi != 0xFFFFFFFFu) scores_out[oi] = h_scores[i];
        }
    if (k > 0 && topk_out) {
        std::vector<uint64_t> keys;
        keys.reserve(db->n_local);
        for (size_t i = 0; i < n_slots; ++i) {
            const uint32_t oi = db->order[i];
            if (oi != 0xFFFFFFFFu) keys.push_back(swg_hit_key(h_scores[i], oi));
        }
        const size_t m = std::min(k, keys.size());
        std::partial_sort(keys.begin(), keys.begin() + m, keys.end(), std::greater<uint64_t>());
        for (size_t i = 0; i < m; ++i) swg_key_hit(keys[i], &topk_out[i]);
        if (n_hits) *n_hits = m;
    } else if (n_hits) {
        *n_hits = 0;
    }
}

static int search_multi_impl(swg_ctx *ctx, const swg_db *db, const int8_t *queries, const uint64_t *q_offsets,
                             size_t n_queries, int32_t *scores_out, swg_hit *topk_out, size_t k, size_t *n_hits,
                             swg_stats *stats);

extern "C" int swg_search_multi(swg_ctx *ctx, const swg_db *db, const int8_t *queries, const uint64_t *q_offsets,
                                size_t n_queries, int32_t *scores_out, swg_hit *topk_out, size_t k, size_t *n_hits,
                                swg_stats *stats)
{
    try { // no C++ exception crosses the ABI (the body sizes host vectors by the batch)
        return search_multi_impl(ctx, db, queries, q_offsets, n_queries, scores_out, topk_out, k, n_hits, stats);
    } catch (const std::bad_alloc &) {
        return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_search_multi: out of host memory");
    } catch (const std::exception &e) {
        return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_search_multi: %s", e.what());
    }
}

static int search_multi_impl(swg_ctx *ctx, const swg_db *db, const int8_t *queries, const uint64_t *q_offsets,
                             size_t n_queries, int32_t *scores_out, swg_hit *topk_out, size_t k, size_t *n_hits,
                             swg_stats *stats)
{
    if (!ctx || !db || (n_queries && (!queries || !q_offsets)))
        return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_search_multi: NULL argument");
    if (k > 0 && !topk_out) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_search_multi: k > 0 but topk_out NULL");
    if (!ctx->have_scoring) return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_search_multi: no scoring set");
    if (db->device != ctx->device || !db->d_codes)
        return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_search_multi: database is not resident on device %d", ctx->device);
    for (const SwgSlot &sl : ctx->slots)
        if (sl.busy) return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_search_multi: searches are in flight on this context");
    swg_stats st;
    memset(&st, 0, sizeof st);
    size_t lq_max = 0;
    for (size_t i = 0; i < n_queries; ++i) {
        if (q_offsets[i + 1] <= q_offsets[i])
            return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_search_multi: query %zu is empty or the offsets are not increasing", i);
        const size_t lq = (size_t)(q_offsets[i + 1] - q_offsets[i]);
        if (lq > (1u << 24)) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_search_multi: query %zu too long", i);
        lq_max = std::max(lq_max, lq);
        for (uint64_t j = q_offsets[i]; j < q_offsets[i + 1]; ++j)
            if (queries[j] < 1 || queries[j] > 31)
                return swg_set_ctx_error(ctx, SWG_ERR_RESIDUE, "swg_search_multi: residue index %d in query %zu outside 1..31",
                                         queries[j], i);
        st.cells += (uint64_t)lq * db->residues;
        st.bytes_alg += db->residues + 8ull * db->n_local + 32ull * lq + 1024ull;
    }
    if (stats) *stats = st;
    if (n_queries == 0) return SWG_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_slots = (size_t)db->n_bins * SWG_BIN;
    const size_t n_total = db->n_total;

    // ---- can the batch go through one launch? ------------------------------------------------
    const int go = ctx->gap_open + ctx->gap_extend, ge = ctx->gap_extend;
    bool fast = ctx->gap_open <= 0 && ctx->gap_extend <= 0 && -go <= 32767 && ctx->opt_force_bits != 32 &&
                ctx->opt_engine != 1 && ctx->opt_dynamic != 0 && db->n_bins > 0 && n_queries > 1 && ctx->opt_cols == 0 &&
                ctx->opt_group == 0 && ctx->opt_max_waves == 0 && ctx->opt_workgroups == 0;
    SwgDiagWork wk;
    const size_t Qb_max = 256; // queries per launch
    uint64_t bound_max = 0;    // the largest score any query of the batch can reach
    if (fast) {
        // no score of any query may reach the int16 ceiling (the batch path has no re-score)
        int smax = 0;
        for (int a = 0; a < 32; ++a)
            for (int b = 0; b < 32; ++b) smax = std::max<int>(smax, ctx->sub[a][b]);
        const uint64_t longest = (uint64_t)db->max_nblk * SWG_ROWS_PER_BLK;
        for (size_t i = 0; i < n_queries && fast; ++i) {
            uint64_t qbound = 0;
            const uint64_t lq = q_offsets[i + 1] - q_offsets[i];
            for (uint64_t j = q_offsets[i]; j < q_offsets[i + 1]; ++j) {
                int best = 0;
                for (int b = 1; b < 32; ++b) best = std::max<int>(best, ctx->sub[(uint8_t)queries[j] & 31][b]);
                qbound += (uint64_t)best;
            }
            const uint64_t bound = std::min<uint64_t>(qbound, std::min<uint64_t>(lq, longest) * (uint64_t)smax);
            bound_max = std::max(bound_max, bound);
            if (bound >= 32767ull) fast = false;
        }
    }
    // the packed-f16 cells (8.5 instead of 10 instructions per column pair) where no query of the batch can reach
    // their ceiling: the batch path has no re-run either
    const int form = fast && ctx->opt_f16 != 0 && bound_max < 4096ull && -go <= 2048 && -ge <= 2048 ? 2 : 0;
    if (fast) {
        int rc = ensure_pair_tokens(ctx, const_cast<swg_db *>(db));
        if (rc != SWG_OK) return rc;
        fast = db->ptok.ok &&
               swg_plan_diag_work(db, lq_max, ctx->n_cu, 0, 0, 0, ctx->opt_long_split, true, true, &wk,
                                  (double)std::min(n_queries, Qb_max), form) > 0;
        for (int c = 0; fast && c < wk.n_classes; ++c)
            fast = wk.plan[c].npass == 1 && diag_class_is_dynamic(ctx, db, wk.plan[c]) && (size_t)wk.plan[c].G * wk.plan[c].K >= lq_max;
        // A database of short sequences is faster on the systolic engine, one query after another (no batch form of
        // that engine exists), than as a batch on the lane groups: 16 queries against 500 000 peptides 3 720 GCUPS as a
        // batch, ~7 000 one by one.  Same comparison as a single search makes (swg_search_begin), per query.
        if (fast && ctx->opt_engine == 0 && ctx->opt_f16 != 2 && !db->tokens_only && wk.plan[0].est_ms > 0.0) {
            int sys_K = 0;
            const bool sys_f16 = ctx->opt_f16 != 0 && -go <= 2048 && -ge <= 2048 && bound_max < 4096ull;
            const double sys_ms = swg_systolic_estimate_ms(db, lq_max, ctx->n_cu, &sys_K, sys_f16) * (double)std::min(n_queries, Qb_max);
            if (sys_K > 0 && sys_ms < SWG_SYSTOLIC_MARGIN * wk.plan[0].est_ms * swg_diag_short_pair_factor(db, wk.plan[0], form)) fast = false;
        }
    }
    if (!fast) {
        // one after another; the context's own query is put back afterwards
        const std::vector<int8_t> keep = ctx->query;
        int rc = SWG_OK;
        swg_stats one;
        for (size_t i = 0; i < n_queries && rc == SWG_OK; ++i) {
            rc = swg_set_query(ctx, queries + q_offsets[i], (size_t)(q_offsets[i + 1] - q_offsets[i]));
            if (rc == SWG_OK)
                rc = swg_search(ctx, db, scores_out ? scores_out + i * n_total : nullptr, topk_out ? topk_out + i * k : nullptr, k,
                                n_hits ? n_hits + i : nullptr, &one);
            if (rc == SWG_OK) {
                st.fill_ms += one.fill_ms;
                st.rescore_ms += one.rescore_ms;
                st.topk_ms += one.topk_ms;
                st.total_ms += one.total_ms;
                st.n_rescored += one.n_rescored;
                st.cells_padded += one.cells_padded;
                st.path_bits = one.path_bits;
                st.cell_form = one.cell_form;
                st.engine = one.engine;
                st.cols_per_wave = one.cols_per_wave;
                st.group_lanes = one.group_lanes;
                st.waves = one.waves;
                st.passes = one.passes;
                st.workgroups = one.workgroups;
                st.work_queue = one.work_queue;
            }
        }
        if (!keep.empty()) {
            const int rq = swg_set_query(ctx, keep.data(), keep.size());
            if (rc == SWG_OK) rc = rq;
        } else {
            ctx->query.clear();
        }
        if (stats) *stats = st;
        return rc;
    }

    // ---- one launch per class for up to Qb_max queries ----------------------------------------
    // Two queries per lane (swg_diag_qq_kernel, 7.5 instead of 8.5 instructions per column pair: the two halves of a
    // register hold two QUERIES against one sequence, so no v_perm pairs two sequences' profile words) where the batch
    // runs on the f16 cells and the pairs' 4-byte profile fits LDS beside the other class's; option "qq" = 0 turns it off.
    bool qq = form == 2 && ctx->opt_qq != 0;
    int qq_per_cu = 1;
    if (qq) {
        // The pairs' profile is twice the size, so LDS decides the occupancy (as for the int32 kernel): the long class
        // keeps one workgroup of four wavefronts per CU, the bulk takes the workgroup size -- 4, 8, 12 or 16 wavefronts
        // sharing one profile -- that leaves the most wavefronts resident (four 57 KB workgroups of four do not fit a
        // CU; two of eight do: the first version of this path ran at two wavefronts per SIMD and lost to the perm).
        size_t room = 160 * 1024;
        const SwgKernelInfo info = swg_diag_variant_info(wk.plan[0].variant);
        int cap_waves = info.max_waves;
        if (wk.n_classes == 2) {
            wk.plan[1].W = 4;
            room -= std::min(room, swg_diag32q_lds_bytes(wk.plan[1].K, wk.plan[1].G, 4));
            cap_waves -= 4;
        }
        int best_W = 0, best_waves = 0;
        for (int W = 4; W <= info.max_waves; W += 4) {
            const int n = std::min<int>((int)(room / swg_diag32q_lds_bytes(wk.plan[0].K, wk.plan[0].G, W)), cap_waves / W);
            if (n >= 1 && n * W > best_waves) best_waves = n * W, best_W = W, qq_per_cu = n;
        }
        if (best_W == 0) qq = false; // (no room beside the long class: two sequences per lane, as before)
        else wk.plan[0].W = best_W;
    }
    const SwgPairTokens &T = db->ptok;
    const uint32_t gm = (uint32_t)(-go) & 0xFFFFu, em = (uint32_t)(-ge) & 0xFFFFu;
    const uint32_t cnt_class = SWG_DYN_SHARDS * SWG_DYN_SHARD_STRIDE; // queue dwords of one class of one query
    MultiBufs B;
    std::vector<int32_t> h_scores;
    std::vector<uint32_t> qoff32, order, h_meta;
    std::vector<uint64_t> h_cand;
    // Top-K only (no score array asked for): selected on the device for the whole batch in three launches, and a few
    // hundred keys per query come back instead of every score (round 3: with 32 queries against 100 000 sequences the
    // copy and the host's selection took longer than the fill: 82 ms of wall time for 49 ms of device time).
    const bool dev_topk = scores_out == nullptr && k > 0 && k <= SWG_TOPK_MULTI_CAP / 2;
    hipStream_t s = ctx->stream;
    ctx->cur = &ctx->slots[0];
    for (size_t q0 = 0; q0 < n_queries; q0 += Qb_max) {
        const size_t Qb = std::min(Qb_max, n_queries - q0);
        const size_t Qrows = qq ? (Qb + 1) / 2 : Qb; // rows of the grid: query pairs, or queries
        const uint64_t qbytes = q_offsets[q0 + Qb] - q_offsets[q0];
        try {
            qoff32.resize(Qb + 1);
            if (!dev_topk) h_scores.resize(Qb * n_slots);
            if (dev_topk) {
                h_meta.resize(Qb * 4);
                h_cand.resize(Qb * (size_t)SWG_TOPK_MULTI_CAP);
            }
        } catch (const std::exception &) {
            return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_search_multi: out of host memory");
        }
        for (size_t i = 0; i <= Qb; ++i) qoff32[i] = (uint32_t)(q_offsets[q0 + i] - q_offsets[q0]);
        // qq: queries of similar length share a lane (row r of the score buffer belongs to query order[r] of this chunk)
        order.resize(Qb);
        for (size_t i = 0; i < Qb; ++i) order[i] = (uint32_t)i;
        if (qq)
            std::stable_sort(order.begin(), order.end(),
                             [&](uint32_t a, uint32_t b) { return qoff32[a + 1] - qoff32[a] > qoff32[b + 1] - qoff32[b]; });
        if (q0 == 0) {
            // rows are counted in QUERIES for scores / order / top-K, in grid rows (query pairs with qq) for the
            // profiles and the queues; a qq batch gets an even number of query rows, so that the absent partner of an
            // odd batch's last query has a row of its own everywhere (nothing reads it back)
            const size_t qm = std::min(Qb_max, n_queries);
            const size_t q_rows = qq ? (qm + 1) / 2 * 2 : qm, g_rows = qq ? (qm + 1) / 2 : qm;
            const size_t cnt_dwords = Qb_max * 2 * cnt_class + 2 * SWG_DYN_SIMD_SLOTS;
            HIP_TRY(ctx, hipMalloc(&B.d_cnt, cnt_dwords * 4));
            B.rows_cnt = Qb_max;
            HIP_TRY(ctx, hipMalloc(&B.d_scores, q_rows * n_slots * 4));
            B.rows_scores = q_rows;
            HIP_TRY(ctx, hipMalloc(&B.d_qoff, (Qb_max + 1) * 4));
            HIP_TRY(ctx, hipMalloc(&B.d_order, (Qb_max + 1) * 4));
            B.rows_order = Qb_max + 1;
            if (dev_topk) {
                HIP_TRY(ctx, hipMalloc(&B.d_hist, qm * 4096 * 4));
                HIP_TRY(ctx, hipMalloc(&B.d_meta, qm * 16));
                HIP_TRY(ctx, hipMalloc(&B.d_cand, qm * (size_t)SWG_TOPK_MULTI_CAP * 8));
                B.rows_topk = qm;
            }
            for (int c = 0; c < wk.n_classes; ++c) {
                // (qq: one profile of 128 bytes per column per query PAIR, and an odd batch's last pair is a whole pair)
                const size_t per_row = (size_t)wk.plan[c].G * (qq ? (size_t)swg_q32_padded_cols(wk.plan[c].K) * 128 : (size_t)swg_diag_padded_cols(wk.plan[c].K) * 64);
                HIP_TRY(ctx, hipMalloc(&B.d_prof[c], g_rows * per_row));
                B.rows_prof[c] = g_rows;
            }
        }
        {
            // the fence: what this chunk's launches will index, against what the buffers hold
            const int rf = multi_rows_ok(ctx, B, qq, Qb, Qrows, wk.n_classes, dev_topk);
            if (rf != SWG_OK) return rf;
        }
        (void)hipFree(B.d_q);
        B.d_q = nullptr;
        HIP_TRY(ctx, hipMalloc(&B.d_q, std::max<uint64_t>(4, qbytes)));
        HIP_TRY(ctx, hipMemcpyAsync(B.d_q, queries + q_offsets[q0], qbytes, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(B.d_qoff, qoff32.data(), (Qb + 1) * 4, hipMemcpyHostToDevice, s));
        order.push_back(order.back()); // (qq, odd batch: the last pair's absent partner is its query once more)
        HIP_TRY(ctx, hipMemcpyAsync(B.d_order, order.data(), (Qb + 1) * 4, hipMemcpyHostToDevice, s));
        order.pop_back();
        HIP_TRY(ctx, hipMemsetAsync(B.d_scores, 0, (qq ? 2 * Qrows : Qb) * n_slots * 4, s));
        HIP_TRY(ctx, hipMemsetAsync(B.d_cnt, 0, (Qb_max * 2 * cnt_class + 2 * SWG_DYN_SIMD_SLOTS) * 4, s));
        for (int c = 0; c < wk.n_classes; ++c) {
            const SwgDiagPlan &pl = wk.plan[c];
            if (qq)
                HIP_TRY(ctx, swg_launch_build_profiles_qq(ctx->d_sub, B.d_q, B.d_qoff, B.d_order, (uint32_t)Qb,
                                                          (uint32_t)(pl.G * swg_q32_padded_cols(pl.K)), pl.K, swg_q32_padded_cols(pl.K),
                                                          B.d_prof[c], s, SWG_LDS_SWIZZLE ? pl.G : 0));
            else
                HIP_TRY(ctx, swg_launch_build_profiles_multi(ctx->d_sub, B.d_q, B.d_qoff, (uint32_t)Qb,
                                                             (uint32_t)(pl.G * swg_diag_padded_cols(pl.K)), pl.K,
                                                             swg_diag_padded_cols(pl.K), B.d_prof[c], s, SWG_LDS_SWIZZLE ? pl.G : 0, form == 2));
        }
        // workgroups per query: the chip's resident workgroups shared out over the batch
        int wgs[2] = {1, 1};
        uint64_t groups0 = 1;
        for (int c = 0; c < wk.n_classes; ++c) {
            const SwgDiagPlan &pl = wk.plan[c];
            const SwgKernelInfo info = swg_diag_variant_info(pl.variant);
            const size_t lds = qq ? swg_diag32q_lds_bytes(pl.K, pl.G, pl.W) : swg_diag_dyn_lds_bytes(pl.K, pl.G, pl.W);
            const int per_cu = qq ? (c == 0 ? qq_per_cu : 1) : std::max(1, std::min<int>(info.max_waves / pl.W, (int)((160 * 1024) / lds)));
            int total = ctx->n_cu * per_cu;
            if (wk.n_classes == 2 && !qq) total = c == 1 ? ctx->n_cu : std::max(ctx->n_cu, total - ctx->n_cu); // one wavefront per SIMD for the long class
            const uint64_t items = (wk.pair_end[c] - wk.pair_begin[c]) * (qq ? 2u : 1u); // pairs, or (qq) single sequences
            const uint64_t per_wg = (uint64_t)pl.W * (64 / pl.G);
            wgs[c] = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)total / Qrows, (items + per_wg - 1) / per_wg));
            if (c == 0) groups0 = (uint64_t)wgs[0] * Qrows * per_wg;
        }
        HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[1], s));
        if (wk.n_classes == 2) {
            const int r2 = ensure_stream2(ctx);
            if (r2 != SWG_OK) return r2;
            HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[6], s));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->cur->ev[6], 0));
        }
        for (int c = wk.n_classes - 1; c >= 0; --c) {
            const SwgDiagPlan &pl = wk.plan[c];
            if (qq) {
                SwgDiagQQParams q;
                memset(&q, 0, sizeof q);
                q.tok = T.d_tok;
                q.zero_block = (uint32_t)T.total_blocks;
                q.pair_off = T.d_pair_off;
                q.q_begin = (uint32_t)std::min<uint64_t>(2 * wk.pair_begin[c], n_slots);
                q.q_end = (uint32_t)std::min<uint64_t>(2 * wk.pair_end[c], n_slots);
                q.queue = B.d_cnt + (size_t)c * cnt_class;
                q.queue_stride = 2 * cnt_class;
                q.profile = B.d_prof[c];
                q.profile_stride = (uint64_t)pl.G * swg_q32_padded_cols(pl.K) * 128;
                q.scores = B.d_scores;
                q.score_stride = n_slots;
                q.n_queries = (uint32_t)Qb;
                q.seq_limit = (uint32_t)n_slots;
                q.G = (uint32_t)pl.G;
                q.go = f16x2_of(-go);
                q.ge = f16x2_of(-ge);
                q.prio_blocks = 0xFFFFFFFFu;
                if (c == 0) {
                    const uint64_t blocks = 2ull * (uint64_t)(T.pair_blocks_prefix[wk.pair_end[0]] - T.pair_blocks_prefix[wk.pair_begin[0]]) * Qrows;
                    q.prio_blocks = (uint32_t)std::max<uint64_t>(8, (uint64_t)(ctx->opt_prio_share * 0.01 * (double)blocks / (double)groups0));
                } else {
                    q.prio_blocks = 0u;
                }
                q.turn_levels = wk.n_classes == 2 ? 3u : 4u;
                q.simd_ranks = B.d_cnt + Qb_max * 2 * cnt_class + (size_t)c * SWG_DYN_SIMD_SLOTS;
                HIP_TRY(ctx, swg_launch_diag_qq(pl.variant, pl.W, wgs[c], (int)Qrows, q, c == 1 ? ctx->stream2 : s));
                continue;
            }
            SwgDiagDynParams q;
            memset(&q, 0, sizeof q);
            q.tok = T.d_tok;
            q.zero_block = (uint32_t)T.total_blocks;
            q.pair_off = T.d_pair_off;
            q.q_begin = (uint32_t)wk.pair_begin[c];
            q.q_end = (uint32_t)wk.pair_end[c];
            q.queue = B.d_cnt + (size_t)c * cnt_class;
            q.queue_stride = 2 * cnt_class;
            q.profile = B.d_prof[c];
            q.profile_stride = (uint64_t)pl.G * swg_diag_padded_cols(pl.K) * 64;
            q.scores = B.d_scores;
            q.score_stride = n_slots;
            q.pair_limit = (uint32_t)(n_slots / 2);
            q.G = (uint32_t)pl.G;
            q.go = form == 2 ? f16x2_of(-go) : gm | (gm << 16);
            q.ge = form == 2 ? f16x2_of(-ge) : em | (em << 16);
            if (c == 0) {
                const uint64_t blocks = (uint64_t)(T.pair_blocks_prefix[wk.pair_end[0]] - T.pair_blocks_prefix[wk.pair_begin[0]]) * Qb;
                q.prio_blocks = (uint32_t)std::max<uint64_t>(8, (uint64_t)(ctx->opt_prio_share * 0.01 * (double)blocks / (double)groups0));
            }
            q.turn_levels = wk.n_classes == 2 ? 3u : 4u;
            q.simd_ranks = B.d_cnt + Qb_max * 2 * cnt_class + (size_t)c * SWG_DYN_SIMD_SLOTS;
            dyn_batch_zones(ctx, T, &q, (uint64_t)wgs[c] * pl.W * (64 / pl.G), pl.K, pl.G, form);
            HIP_TRY(ctx, swg_launch_diag_dyn(pl.variant, false, form, pl.W, wgs[c], q, c == 1 ? ctx->stream2 : s, (int)Qb));
        }
        if (wk.n_classes == 2) {
            HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[7], ctx->stream2));
            HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->cur->ev[7], 0));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->cur->ev[2], s));
        if (dev_topk) {
            HIP_TRY(ctx, swg_launch_topk_multi(B.d_scores, n_slots, db->d_order, (uint32_t)n_slots, (uint32_t)Qb, (uint32_t)k, B.d_hist,
                                               B.d_meta, B.d_cand, SWG_TOPK_MULTI_CAP, s));
            HIP_TRY(ctx, hipMemcpyAsync(h_meta.data(), B.d_meta, Qb * 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipMemcpyAsync(h_cand.data(), B.d_cand, Qb * (size_t)SWG_TOPK_MULTI_CAP * 8, hipMemcpyDeviceToHost, s));
        } else {
            HIP_TRY(ctx, hipMemcpyAsync(h_scores.data(), B.d_scores, Qb * n_slots * 4, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(ctx, spin_sync(ctx, s));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->cur->ev[1], ctx->cur->ev[2]));
        st.fill_ms += ms;
        st.total_ms += ms;
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t r = 0; r < Qb; ++r) { // (row r of the score buffer = query order[r] of this chunk)
            const size_t i = order[r];
            if (dev_topk && h_meta[4 * r + 1] == 0 && h_meta[4 * r + 2] <= SWG_TOPK_MULTI_CAP) {
                // every hit with a score >= the k-th best one: sort those few keys
                uint64_t *c = h_cand.data() + r * (size_t)SWG_TOPK_MULTI_CAP;
                const size_t nc = h_meta[4 * r + 2], m = std::min(k, nc);
                std::partial_sort(c, c + m, c + nc, std::greater<uint64_t>());
                for (size_t j = 0; j < m; ++j) swg_key_hit(c[j], &topk_out[(q0 + i) * k + j]);
                if (n_hits) n_hits[q0 + i] = m;
                continue;
            }
            if (dev_topk) { // threshold beyond the histogram, or too many ties: this query's scores to the host after all
                if (h_scores.size() < n_slots) h_scores.resize(n_slots);
                HIP_TRY(ctx, hipMemcpyAsync(h_scores.data(), B.d_scores + r * n_slots, n_slots * 4, hipMemcpyDeviceToHost, s));
                HIP_TRY(ctx, spin_sync(ctx, s));
                multi_deliver(db, h_scores.data(), n_slots, nullptr, topk_out + (q0 + i) * k, k, n_hits ? n_hits + q0 + i : nullptr);
                continue;
            }
            multi_deliver(db, h_scores.data() + r * n_slots, n_slots, scores_out ? scores_out + (q0 + i) * n_total : nullptr,
                          topk_out ? topk_out + (q0 + i) * k : nullptr, k, n_hits ? n_hits + q0 + i : nullptr);
        }
        st.topk_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        st.workgroups = wgs[0] * (int)Qrows;
        st.streams = (int32_t)groups0;
    }
    st.path_bits = 16;
    st.engine = 2;
    st.work_queue = 1;
    st.classes_overlapped = -1; // (not measured for a batch)
    st.cell_form = qq ? 3 : form;
    st.cols_per_wave = wk.plan[0].K;
    st.group_lanes = wk.plan[0].G;
    st.waves = wk.plan[0].W;
    st.passes = 1;
    st.fill_launches = 1;
    if (wk.n_classes == 2) {
        st.long_pairs = (int32_t)(wk.pair_end[1] - wk.pair_begin[1]);
        st.long_cols_per_lane = wk.plan[1].K;
    }
    for (int c = 0; c < wk.n_classes; ++c)
        st.cells_padded += 2ull * wk.plan[c].G * wk.plan[c].K * diag_class_blocks(ctx, db, wk, c) * 4ull * n_queries;
    if (stats) *stats = st;
    return SWG_OK;
}

// ---------------------------------------------------------------------------
// reference-shaped replay (16-lane batches as alignment_fill_matrices gets them)
// ---------------------------------------------------------------------------
// The device route.  The batches go to the GPU as they are -- [max_len][16] table indices, copied end to end into
// pinned staging by all cores and from there in one transfer -- and the pair tokens are built from that image on the
// device: the two sequences of a pair are two adjacent lanes of one batch, so a row's two residues are two adjacent
// bytes (swg_build_tokens16_kernel).  Nothing is un-transposed or re-coded on the host, and nothing is allocated or
// freed per call once the buffers have grown to the size of the caller's macro-batches (round 2 rebuilt a whole
// database per call: 26-34 ms of host work and hipFree around 2.5 ms of device time).  Returns 1 when the search
// cannot take this route (options that ask for an engine that reads the bin image, a query the work-queue kernels
// cannot hold): the caller falls back to the host route.
static int fill_batches16_device(swg_ctx *ctx, const swg_batch16 *batches, size_t n_batches, size_t n_records,
                                 const size_t *first_rec, double *fill_seconds, double *t_ms)
{
    typedef std::chrono::steady_clock clk;
    const clk::time_point t0 = clk::now();
    if (ctx->opt_engine == 1 || ctx->opt_dynamic == 0 || !ctx->have_scoring || ctx->query.empty()) return 1;
    for (const SwgSlot &sl : ctx->slots)
        if (sl.busy) return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_fill_batches16: searches are in flight on this context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    SwgBatch16Cache &C = ctx->b16;
    // batches longest first (the reference's caller passes them sorted: then this is the identity)
    std::vector<uint32_t> bo(n_batches);
    for (size_t b = 0; b < n_batches; ++b) bo[b] = (uint32_t)b;
    bool sorted = true;
    for (size_t b = 1; b < n_batches && sorted; ++b) sorted = batches[b].max_len <= batches[b - 1].max_len;
    if (!sorted)
        std::stable_sort(bo.begin(), bo.end(), [&](uint32_t x, uint32_t y) { return batches[x].max_len > batches[y].max_len; });
    // sizes
    size_t n_pairs = 0;
    uint64_t stage_bytes = 0, total_blocks = 0, residues = 0, longest = 0;
    for (size_t b = 0; b < n_batches; ++b) {
        const swg_batch16 &bt = batches[b];
        if (bt.max_len > 0x3FFFFFFFull) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_fill_batches16: batch %zu too long", b);
        const size_t np = (bt.vector_size + 1) / 2;
        n_pairs += np;
        stage_bytes += (uint64_t)bt.max_len * 16u;
        total_blocks += (uint64_t)np * ((2ull + bt.max_len + 3) / 4);
        residues += (uint64_t)bt.vector_size * bt.max_len;
        longest = std::max<uint64_t>(longest, bt.max_len);
    }
    if (n_pairs >= (1ull << 31) || total_blocks + 1 >= (1ull << 32)) return 1;
    const size_t n_slots = (2 * n_pairs + SWG_BIN - 1) / SWG_BIN * SWG_BIN;
    // ---- buffers: grown, never shrunk ---------------------------------------------------------
    if (!C.db || n_slots > C.slots_cap || n_pairs > C.pairs_cap || total_blocks > C.blocks_cap || stage_bytes > C.stage_cap) {
        // The new capacities live in locals until EVERY allocation has succeeded: a failure half way leaves the cache
        // empty (capacities 0, no database), so the next call grows again instead of writing through NULL buffers.
        auto release = [&C]() {
            if (C.db) swg_db_free(C.db);
            C.db = nullptr;
            (void)hipHostFree(C.h_stage);
            (void)hipHostFree(C.h_meta);
            (void)hipFree(C.d_stage);
            (void)hipFree(C.d_pair_src);
            (void)hipFree(C.d_pair_len);
            C.h_stage = C.h_meta = nullptr;
            C.d_stage = nullptr;
            C.d_pair_src = nullptr;
            C.d_pair_len = nullptr;
        };
        auto grow = [](uint64_t need, uint64_t have) { return std::max<uint64_t>(need + need / 4 + 64, have); };
        const size_t slots_cap = (size_t)((grow(n_slots, C.slots_cap) + SWG_BIN - 1) / SWG_BIN * SWG_BIN);
        const size_t pairs_cap = (size_t)grow(n_pairs, C.pairs_cap);
        const uint64_t blocks_cap = grow(total_blocks, C.blocks_cap);
        const size_t stage_cap = (size_t)grow(stage_bytes, C.stage_cap);
        release();
        C.slots_cap = C.pairs_cap = C.stage_cap = 0;
        C.blocks_cap = 0;
        const int ra = [&]() -> int {
            swg_db *db = new (std::nothrow) swg_db();
            if (!db) return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_fill_batches16: out of memory");
            C.db = db;
            db->tokens_only = true;
            db->device = ctx->device;
            db->n_bins = (uint32_t)(slots_cap / SWG_BIN); // (sizes the per-search output buffers)
            HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&C.h_stage), stage_cap, hipHostMallocDefault));
            HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&C.h_meta), pairs_cap * 16 + 8 + slots_cap * 8, hipHostMallocDefault));
            HIP_TRY(ctx, hipMalloc(&C.d_stage, stage_cap));
            HIP_TRY(ctx, hipMalloc(&C.d_pair_src, pairs_cap * 8));
            HIP_TRY(ctx, hipMalloc(&C.d_pair_len, pairs_cap * 4));
            HIP_TRY(ctx, hipMalloc(&db->d_codes, 16)); // (marks the database resident; there are no residue bytes by rank)
            HIP_TRY(ctx, hipMalloc(&db->d_lens, slots_cap * 4));
            HIP_TRY(ctx, hipMalloc(&db->d_order, slots_cap * 4));
            HIP_TRY(ctx, hipMalloc(&db->ptok.d_tok, (size_t)(blocks_cap + 1) * 16));
            HIP_TRY(ctx, hipMalloc(&db->ptok.d_pair_off, (pairs_cap + 1) * 4));
            return select_bufs(ctx, db, 0);
        }();
        if (ra != SWG_OK) {
            release();
            return ra;
        }
        C.slots_cap = slots_cap;
        C.pairs_cap = pairs_cap;
        C.blocks_cap = blocks_cap;
        C.stage_cap = stage_cap;
    }
    swg_db *db = C.db;
    // ---- this call's database: ranks 2p, 2p+1 = lanes 2i, 2i+1 of a batch --------------------------------
    uint64_t *pair_src = reinterpret_cast<uint64_t *>(C.h_meta);
    uint32_t *pair_len = reinterpret_cast<uint32_t *>(C.h_meta + C.pairs_cap * 8);
    uint32_t *h_lens = reinterpret_cast<uint32_t *>(C.h_meta + C.pairs_cap * 12);
    uint32_t *h_order = h_lens + C.slots_cap;
    try {
        db->lens.assign(n_slots, 0u);
        db->order.assign(n_slots, 0xFFFFFFFFu);
        db->ptok.pair_blocks_prefix.assign(n_pairs + 1, 0u);
    } catch (const std::exception &) {
        return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_fill_batches16: out of host memory");
    }
    std::vector<uint64_t> stage_off(n_batches);
    {
        uint64_t so = 0;
        size_t p = 0;
        uint32_t blocks = 0;
        for (size_t k = 0; k < n_batches; ++k) {
            const size_t b = bo[k];
            const swg_batch16 &bt = batches[b];
            stage_off[b] = so;
            for (size_t i = 0; 2 * i < bt.vector_size; ++i, ++p) {
                const bool has_y = 2 * i + 1 < bt.vector_size;
                pair_src[p] = (so + 2 * i) | (has_y ? 0ull : 1ull << 63);
                pair_len[p] = (uint32_t)bt.max_len;
                db->lens[2 * p] = (uint32_t)bt.max_len;
                db->order[2 * p] = (uint32_t)(first_rec[b] + 2 * i);
                if (has_y) {
                    db->lens[2 * p + 1] = (uint32_t)bt.max_len;
                    db->order[2 * p + 1] = (uint32_t)(first_rec[b] + 2 * i + 1);
                }
                blocks += (uint32_t)((2ull + bt.max_len + 3) / 4);
                db->ptok.pair_blocks_prefix[p + 1] = blocks;
            }
            so += (uint64_t)bt.max_len * 16u;
        }
    }
    db->n_total = n_records;
    db->n_local = 2 * n_pairs; // (an odd batch's missing lane is an empty slot in the middle: pairs stay batch-aligned)
    db->n_bins = (uint32_t)(n_slots / SWG_BIN);
    db->residues = residues;
    db->max_nblk = (uint32_t)((longest + 3) / 4);
    db->rows_padded = 0;
    // one macro-batch has nothing to tell the next: plans and hints start afresh with the cache's database
    db->tuned.clear();
    db->pair_rows_prefix.clear(); // (the lengths changed under the same pair count)
    db->sat_hint = -1;
    db->f16_veto_epoch = 0;
    SwgPairTokens &T = db->ptok;
    T.tried = T.ok = true;
    T.total_blocks = total_blocks;
    memcpy(h_lens, db->lens.data(), n_slots * 4);
    memcpy(h_order, db->order.data(), n_slots * 4);
    t_ms[0] = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    const clk::time_point t1 = clk::now();
    hipStream_t s = ctx->stream;
    {
        // the batches into pinned staging by all cores, in runs of about 4 MB (in staging order): a run's transfer is
        // queued as soon as it is copied, so the upload of one run overlaps the copy of the next
        size_t k0 = 0;
        while (k0 < n_batches) {
            size_t k1 = k0;
            uint64_t bytes = 0;
            while (k1 < n_batches && bytes < (4u << 20)) bytes += (uint64_t)batches[bo[k1++]].max_len * 16u;
            const long long lo = (long long)k0, hi = (long long)k1;
            swg_stage_batches16(batches, bo.data(), stage_off.data(), (size_t)lo, (size_t)hi, C.h_stage); // all cores (swg_pack.cpp)
            const uint64_t from = stage_off[bo[k0]];
            HIP_TRY(ctx, hipMemcpyAsync(C.d_stage + from, C.h_stage + from, bytes, hipMemcpyHostToDevice, s));
            k0 = k1;
        }
    }
    t_ms[1] = std::chrono::duration<double, std::milli>(clk::now() - t1).count();
    const clk::time_point t2 = clk::now();
    HIP_TRY(ctx, hipMemcpyAsync(C.d_pair_src, pair_src, n_pairs * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(C.d_pair_len, pair_len, n_pairs * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(T.d_pair_off, T.pair_blocks_prefix.data(), (n_pairs + 1) * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(db->d_lens, h_lens, n_slots * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(db->d_order, h_order, n_slots * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(T.d_tok + total_blocks, 0, 16, s)); // the block of zeros behind the last pair
    uint32_t *d_bad = reinterpret_cast<uint32_t *>(db->d_codes);
    HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, 4, s));
    HIP_TRY(ctx, swg_launch_build_tokens16(C.d_stage, C.d_pair_src, C.d_pair_len, T.d_pair_off, (uint32_t)n_pairs, total_blocks,
                                           T.d_tok, d_bad, s));
    // ---- the search itself (the cost model's geometry: this database is searched once) ----------------------
    swg_stats st;
    const auto keep_autotune = ctx->opt_autotune;
    ctx->opt_autotune = 0;
    int rc = search_begin(ctx, db, true, 0, &ctx->slots[0]);
    if (rc == SWG_OK) {
        ctx->slots[0].busy = true;
        // (scores by record index: straight from the slot's pinned landing buffer, no int32 array in between)
        rc = search_end(ctx, &ctx->slots[0], nullptr, nullptr, nullptr, &st);
        ctx->slots[0].busy = false;
    }
    ctx->opt_autotune = keep_autotune;
    if (rc != SWG_OK) return rc == SWG_ERR_STATE && strstr(ctx->err.c_str(), "built from 16-lane batches") ? 1 : rc;
    uint32_t bad = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, spin_sync(ctx, s));
    if (bad) return swg_set_ctx_error(ctx, SWG_ERR_RESIDUE, "swg_fill_batches16: residue index outside 1..31 in a batch");
    t_ms[2] = std::chrono::duration<double, std::milli>(clk::now() - t2).count();
    const clk::time_point t3 = clk::now();
    const int32_t *hs = ctx->slots[0].h_scores;
    {
        size_t p = 0;
        for (size_t k = 0; k < n_batches; ++k) {
            const swg_batch16 &bt = batches[bo[k]];
            for (size_t i = 0; 2 * i < bt.vector_size; ++i, ++p) {
                bt.max_scores[2 * i] = (int16_t)std::min<int32_t>(hs[2 * p], 32767);
                if (2 * i + 1 < bt.vector_size) bt.max_scores[2 * i + 1] = (int16_t)std::min<int32_t>(hs[2 * p + 1], 32767);
            }
        }
    }
    t_ms[3] = std::chrono::duration<double, std::milli>(clk::now() - t3).count();
    t_ms[4] = st.total_ms;
    if (fill_seconds) *fill_seconds = st.total_ms * 1e-3;
    return SWG_OK;
}

static int fill_batches16_impl(swg_ctx *ctx, const swg_batch16 *batches, size_t n_batches, double *fill_seconds);

extern "C" int swg_fill_batches16(swg_ctx *ctx, const swg_batch16 *batches, size_t n_batches,
                                  double *fill_seconds)
{
    try { // no C++ exception crosses the ABI
        return fill_batches16_impl(ctx, batches, n_batches, fill_seconds);
    } catch (const std::bad_alloc &) {
        return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_fill_batches16: out of host memory");
    } catch (const std::exception &e) {
        return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_fill_batches16: %s", e.what());
    }
}

static int fill_batches16_impl(swg_ctx *ctx, const swg_batch16 *batches, size_t n_batches, double *fill_seconds)
{
    if (!ctx || (!batches && n_batches))
        return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_fill_batches16: NULL argument");
    if (fill_seconds) *fill_seconds = 0.0;
    // diagnostics: SWG_TIMING=1 prints where the wall time of this call went
    static const bool timing = getenv("SWG_TIMING") != nullptr;
    typedef std::chrono::steady_clock clk;
    clk::time_point tp[8];
    tp[0] = clk::now();
    std::vector<uint64_t> offsets(1, 0);
    std::vector<size_t> first_rec(n_batches);
    try {
        for (size_t b = 0; b < n_batches; ++b) {
            if (!batches[b].db_idx_t || !batches[b].max_scores || batches[b].vector_size > 16 ||
                batches[b].max_len == 0)
                return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_fill_batches16: bad batch %zu", b);
            first_rec[b] = offsets.size() - 1;
            // lane l, padded rows included: the reference computes them as real
            // rows (src/alignment_cmdline.c:448-450)
            for (size_t l = 0; l < batches[b].vector_size; ++l) offsets.push_back(offsets.back() + batches[b].max_len);
        }
    } catch (const std::exception &) {
        return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_fill_batches16: out of host memory");
    }
    const size_t n_all = offsets.size() - 1;
    if (n_all == 0) return SWG_OK;
    {
        double t_ms[5] = {0, 0, 0, 0, 0};
        int rd = SWG_OK;
        try {
            rd = fill_batches16_device(ctx, batches, n_batches, n_all, first_rec.data(), fill_seconds, t_ms);
        } catch (const std::exception &) {
            return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_fill_batches16: out of host memory");
        }
        if (rd != 1) {
            if (rd == SWG_OK && timing)
                fprintf(stderr, "[swg_fill_batches16] %zu records, device route: tables %.2f ms, staging copy %.2f, upload + tokens + search + "
                                "read-out %.2f (device, first kernel to last: %.2f), scores to the batches %.2f; wall %.2f\n",
                        n_all, t_ms[0], t_ms[1], t_ms[2], t_ms[4], t_ms[3],
                        std::chrono::duration<double, std::milli>(clk::now() - tp[0]).count());
            return rd;
        }
    }
    // the host route (searches the device route cannot take: engine = 1, work_queue = 0, what needs the bin image)
    std::unique_ptr<int8_t[]> flat(new (std::nothrow) int8_t[std::max<uint64_t>(1, offsets.back())]);
    if (!flat) return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_fill_batches16: out of host memory");
    swg_untranspose_batches16(batches, n_batches, first_rec.data(), offsets.data(), flat.get());
    tp[1] = clk::now();
    const size_t n = offsets.size() - 1;
    if (n == 0) return SWG_OK;
    swg_db *db = nullptr;
    int rc = swg_db_pack(flat.get(), offsets.data(), n, 0, 1, &db);
    if (rc != SWG_OK) {
        ctx->err = swg_global_error();
        return rc;
    }
    tp[2] = clk::now();
    rc = swg_db_upload(ctx, db);
    tp[3] = clk::now();
    std::vector<int32_t> scores;
    try {
        scores.assign(n, 0);
    } catch (const std::exception &) {
        swg_db_free(db);
        return swg_set_ctx_error(ctx, SWG_ERR_NOMEM, "swg_fill_batches16: out of host memory");
    }
    swg_stats st;
    // (this database is searched exactly once: the cost model's geometry, no timed trials)
    const auto keep_autotune = ctx->opt_autotune;
    ctx->opt_autotune = 0;
    if (rc == SWG_OK) rc = swg_search(ctx, db, scores.data(), nullptr, 0, nullptr, &st);
    ctx->opt_autotune = keep_autotune;
    tp[4] = clk::now();
    swg_db_free(db);
    tp[5] = clk::now();
    if (rc != SWG_OK) return rc;
    size_t i = 0;
    for (size_t b = 0; b < n_batches; ++b)
        for (size_t l = 0; l < batches[b].vector_size; ++l, ++i)
            batches[b].max_scores[l] = (int16_t)std::min<int32_t>(scores[i], 32767);
    if (fill_seconds) *fill_seconds = st.total_ms * 1e-3;
    if (timing) {
        auto ms = [&](int i) { return std::chrono::duration<double, std::milli>(tp[i + 1] - tp[i]).count(); };
        fprintf(stderr, "[swg_fill_batches16] %zu records: un-transpose %.2f ms, pack %.2f, upload %.2f, search %.2f (device %.2f), free %.2f\n",
                n, ms(0), ms(1), ms(2), ms(3), st.total_ms, ms(4));
    }
    return SWG_OK;
}
