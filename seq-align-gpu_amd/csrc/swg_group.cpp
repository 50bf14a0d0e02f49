// swg_group.cpp -- one process, several GPUs: the database is dealt round-robin (by bin) over
// the devices, every device searches its shard with its own swg_ctx, and the shards' top-K
// lists are merged by ONE RCCL max-all-reduce of n_gpus*K 64-bit keys over xGMI.
//
// This is the in-process counterpart of what bench.py does with one process per GPU and
// torch.distributed; the C tool (`smith_waterman --gpus N`) uses it.  RCCL is loaded lazily
// (dlopen) so that nothing else in libswg depends on it; with one device no collective is
// issued.  The reference has no multi-device code at all: this is a build-side
// addition named by the north star.
#include "swg_host_internal.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
// the few RCCL entry points used, resolved at run time
typedef void *ncclComm_t;
typedef int ncclResult_t;
enum { kNcclUint64 = 5, kNcclMax = 2 };
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        if (lib) return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce;
    }
};
Rccl g_rccl;
} // namespace

struct swg_group {
    int n = 0;
    bool force_collective = false; // run the all-reduce even with one device (rehearsal)
    // several contexts on ONE device (a device named more than once: the rehearsal of n > 1 on a one-GPU box): RCCL refuses
    // duplicate devices, so the shards' keys are merged on the host -- which is what the max-all-reduce of disjoint
    // segments computes.  Everything else of the group path (one sort, shards built and uploaded side by side, all
    // searches queued before any is awaited, alignments routed to the shard that holds the sequence) runs as with n devices.
    bool host_merge = false;
    std::vector<int> devices;
    std::vector<swg_ctx *> ctx;
    std::vector<swg_db *> db;
    std::vector<ncclComm_t> comm;
    std::vector<uint64_t *> d_keys; // [n] device buffers of n*kcap keys
    size_t kcap = 0;
    size_t n_total = 0;
    std::string err;
};

static int gerr(swg_group *g, int code, const std::string &msg)
{
    if (g) g->err = msg;
    swg_set_global_error(code, "%s", msg.c_str());
    return code;
}

extern "C" const char *swg_group_last_error(const swg_group *g) { return g ? g->err.c_str() : swg_global_error(); }

extern "C" void swg_group_destroy(swg_group *g)
{
    if (!g) return;
    for (size_t i = 0; i < g->db.size(); ++i) swg_db_free(g->db[i]);
    for (size_t i = 0; i < g->d_keys.size(); ++i)
        if (g->d_keys[i]) {
            (void)hipSetDevice(g->devices[i]);
            (void)hipFree(g->d_keys[i]);
        }
    for (ncclComm_t c : g->comm)
        if (c && g_rccl.CommDestroy) g_rccl.CommDestroy(c);
    for (swg_ctx *c : g->ctx) swg_destroy(c);
    delete g;
}

extern "C" int swg_group_create(const int *devices, int n, int force_collective, swg_group **out)
{
    if (!out || n < 1 || n > 64) return swg_set_global_error(SWG_ERR_ARG, "swg_group_create: bad arguments");
    *out = nullptr;
    swg_group *g = new (std::nothrow) swg_group();
    if (!g) return swg_set_global_error(SWG_ERR_NOMEM, "swg_group_create: out of memory");
    g->n = n;
    g->force_collective = force_collective != 0;
    for (int i = 0; i < n; ++i) g->devices.push_back(devices ? devices[i] : i);
    for (int i = 0; i < n; ++i) {
        swg_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.device = g->devices[i];
        swg_ctx *c = nullptr;
        const int rc = swg_create(&cfg, &c);
        if (rc != SWG_OK) {
            swg_group_destroy(g);
            return rc; // message already in swg_global_error
        }
        g->ctx.push_back(c);
    }
    g->db.assign(n, nullptr);
    g->d_keys.assign(n, nullptr);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (g->devices[i] == g->devices[j]) g->host_merge = true;
    if ((n > 1 || g->force_collective) && !g->host_merge) {
        if (!g_rccl.load()) {
            swg_group_destroy(g);
            return swg_set_global_error(SWG_ERR_HIP, "swg_group_create: cannot load librccl (%s)", dlerror());
        }
        g->comm.assign(n, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(g->comm.data(), n, g->devices.data());
        if (r != 0) {
            const char *m = g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?";
            swg_group_destroy(g);
            return swg_set_global_error(SWG_ERR_HIP, "ncclCommInitAll failed: %s", m);
        }
    }
    *out = g;
    return SWG_OK;
}

extern "C" int swg_group_size(const swg_group *g) { return g ? g->n : 0; }

extern "C" int swg_group_set_option(swg_group *g, const char *key, long value)
{
    if (!g) return swg_set_global_error(SWG_ERR_ARG, "swg_group_set_option: NULL group");
    for (swg_ctx *c : g->ctx) {
        const int rc = swg_set_option(c, key, value);
        if (rc != SWG_OK) return gerr(g, rc, swg_last_error(c));
    }
    return SWG_OK;
}

extern "C" int swg_group_set_scoring(swg_group *g, const int8_t sub[32][32], int gap_open, int gap_extend)
{
    if (!g) return swg_set_global_error(SWG_ERR_ARG, "swg_group_set_scoring: NULL group");
    for (swg_ctx *c : g->ctx) {
        const int rc = swg_set_scoring(c, sub, gap_open, gap_extend);
        if (rc != SWG_OK) return gerr(g, rc, swg_last_error(c));
    }
    return SWG_OK;
}

extern "C" int swg_group_set_query(swg_group *g, const int8_t *idx, size_t lq)
{
    if (!g) return swg_set_global_error(SWG_ERR_ARG, "swg_group_set_query: NULL group");
    for (swg_ctx *c : g->ctx) {
        const int rc = swg_set_query(c, idx, lq);
        if (rc != SWG_OK) return gerr(g, rc, swg_last_error(c));
    }
    return SWG_OK;
}

extern "C" int swg_group_load(swg_group *g, const int8_t *flat, const uint64_t *offsets, size_t n)
{
    if (!g) return swg_set_global_error(SWG_ERR_ARG, "swg_group_load: NULL group");
    for (int i = 0; i < g->n; ++i) {
        swg_db_free(g->db[i]);
        g->db[i] = nullptr;
    }
    // ONE global sort for all the devices (round 3 packed -- i.e. sorted -- the whole database once per device, one
    // after another); the shards are cut from it side by side, one host thread per device, and each thread uploads
    // its shard as soon as it is built, so a device's transfer runs beside the other shards' re-coding.
    std::vector<std::string> up_err((size_t)g->n);
    try {
        const int rc = swg_pack_shards(flat, offsets, n, g->n, g->db.data(), [&](int r, swg_db *db) -> int {
            const int ru = swg_db_upload(g->ctx[r], db);
            if (ru != SWG_OK) swg_set_global_error(ru, "upload to device %d: %s", g->devices[r], swg_last_error(g->ctx[r]));
            return ru;
        });
        if (rc != SWG_OK) return gerr(g, rc, swg_global_error());
    } catch (const std::exception &e) {
        for (int i = 0; i < g->n; ++i) {
            swg_db_free(g->db[i]);
            g->db[i] = nullptr;
        }
        return gerr(g, SWG_ERR_NOMEM, std::string("swg_group_load: ") + e.what());
    }
    g->n_total = n;
    return SWG_OK;
}

extern "C" int swg_group_search(swg_group *g, int32_t *scores_out, swg_hit *topk_out, size_t k, size_t *n_hits,
                                swg_stats *stats /* [n] or NULL */)
{
    if (!g) return swg_set_global_error(SWG_ERR_ARG, "swg_group_search: NULL group");
    if (n_hits) *n_hits = 0;
    if (k > 0 && !topk_out) return gerr(g, SWG_ERR_ARG, "swg_group_search: k > 0 but topk_out NULL");
    for (int i = 0; i < g->n; ++i)
        if (!g->db[i]) return gerr(g, SWG_ERR_STATE, "swg_group_search: no database loaded");
    // every GPU starts its shard; nothing waits until all are queued
    std::vector<int> ticket(g->n, -1);
    for (int i = 0; i < g->n; ++i) {
        const int rc = swg_search_begin(g->ctx[i], g->db[i], scores_out != nullptr, k, &ticket[i]);
        if (rc != SWG_OK) {
            for (int j = 0; j < i; ++j) (void)swg_search_end(g->ctx[j], ticket[j], nullptr, nullptr, nullptr, nullptr);
            return gerr(g, rc, swg_last_error(g->ctx[i]));
        }
    }
    std::vector<swg_hit> local(std::max<size_t>(1, k));
    std::vector<uint64_t> keys((size_t)g->n * std::max<size_t>(1, k), 0);
    int first_err = SWG_OK;
    for (int i = 0; i < g->n; ++i) {
        size_t nh = 0;
        swg_stats st;
        // shards write disjoint entries of the caller's score vector (original indices)
        const int rc = swg_search_end(g->ctx[i], ticket[i], scores_out, k ? local.data() : nullptr, &nh, &st);
        if (rc != SWG_OK && first_err == SWG_OK) {
            first_err = rc;
            g->err = swg_last_error(g->ctx[i]);
        }
        if (stats) stats[i] = st;
        for (size_t j = 0; j < nh; ++j) keys[(size_t)i * k + j] = swg_hit_key(local[j].score, local[j].index);
    }
    if (first_err != SWG_OK) return first_err;
    if (k == 0) return SWG_OK;

    if ((g->n > 1 || g->force_collective) && !g->host_merge) {
        // each device holds the n*k buffer with only its own segment filled; one max-all-reduce
        // leaves the union everywhere (n*k*8 bytes: latency-bound, link bandwidth irrelevant)
        const size_t count = (size_t)g->n * k;
        if (g->kcap < count) {
            for (int i = 0; i < g->n; ++i) {
                if (hipSetDevice(g->devices[i]) != hipSuccess) return gerr(g, SWG_ERR_HIP, "hipSetDevice failed");
                (void)hipFree(g->d_keys[i]);
                g->d_keys[i] = nullptr;
                if (hipMalloc(&g->d_keys[i], count * 8) != hipSuccess) return gerr(g, SWG_ERR_NOMEM, "hipMalloc failed");
            }
            g->kcap = count;
        }
        std::vector<uint64_t> seg(count);
        for (int i = 0; i < g->n; ++i) {
            std::fill(seg.begin(), seg.end(), 0);
            std::copy(keys.begin() + (size_t)i * k, keys.begin() + (size_t)(i + 1) * k, seg.begin() + (size_t)i * k);
            if (hipSetDevice(g->devices[i]) != hipSuccess ||
                hipMemcpy(g->d_keys[i], seg.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess)
                return gerr(g, SWG_ERR_HIP, "upload of top-K keys failed");
        }
        if (g_rccl.GroupStart() != 0) return gerr(g, SWG_ERR_HIP, "ncclGroupStart failed");
        for (int i = 0; i < g->n; ++i) {
            const ncclResult_t r = g_rccl.AllReduce(g->d_keys[i], g->d_keys[i], count, kNcclUint64, kNcclMax,
                                                    g->comm[i], g->ctx[i]->stream);
            if (r != 0) {
                (void)g_rccl.GroupEnd();
                return gerr(g, SWG_ERR_HIP, std::string("ncclAllReduce failed: ") +
                                                (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
            }
        }
        if (g_rccl.GroupEnd() != 0) return gerr(g, SWG_ERR_HIP, "ncclGroupEnd failed");
        for (int i = 0; i < g->n; ++i)
            if (hipSetDevice(g->devices[i]) != hipSuccess || hipStreamSynchronize(g->ctx[i]->stream) != hipSuccess)
                return gerr(g, SWG_ERR_HIP, "synchronising the all-reduce failed");
        if (hipSetDevice(g->devices[0]) != hipSuccess ||
            hipMemcpy(keys.data(), g->d_keys[0], count * 8, hipMemcpyDeviceToHost) != hipSuccess)
            return gerr(g, SWG_ERR_HIP, "download of merged keys failed");
    }
    const size_t m = swg_topk_merge_keys(keys.data(), keys.size(), k, topk_out);
    if (n_hits) *n_hits = m;
    return SWG_OK;
}

// Alignments of hits of a group search: every hit is re-run on the GPU whose shard holds the
// sequence (swg_align_hits, swg_trace.hip), shard after shard -- a cold path.
extern "C" size_t swg_group_align_ops_bound(const swg_group *g)
{
    size_t b = 0;
    if (g)
        for (int i = 0; i < g->n; ++i)
            if (g->db[i]) b = std::max(b, swg_align_ops_bound(g->ctx[i], g->db[i]));
    return b;
}

extern "C" int swg_group_align_hits(swg_group *g, const swg_hit *hits, size_t n_hits, swg_alignment *out, char *ops,
                                    size_t ops_stride)
{
    if (!g) return swg_set_global_error(SWG_ERR_ARG, "swg_group_align_hits: NULL group");
    if (n_hits && (!hits || !out)) return gerr(g, SWG_ERR_ARG, "swg_group_align_hits: NULL argument");
    if (ops && ops_stride == 0) return gerr(g, SWG_ERR_ARG, "swg_group_align_hits: ops_stride is 0");
    for (int i = 0; i < g->n; ++i)
        if (!g->db[i]) return gerr(g, SWG_ERR_STATE, "swg_group_align_hits: no database loaded");
    // shard of every wanted sequence: bins are dealt round-robin, so ask the shards' own index lists
    std::vector<int> shard(n_hits, -1);
    {
        std::unordered_map<uint32_t, int> owner;
        owner.reserve(n_hits * 2);
        for (size_t h = 0; h < n_hits; ++h) owner[hits[h].index] = -1;
        for (int i = 0; i < g->n; ++i) {
            for (const uint32_t oi : g->db[i]->order) { // ~0u marks an empty slot of the last bin
                auto it = oi == 0xFFFFFFFFu ? owner.end() : owner.find(oi);
                if (it != owner.end()) it->second = i;
            }
        }
        for (size_t h = 0; h < n_hits; ++h) {
            shard[h] = owner[hits[h].index];
            if (shard[h] < 0)
                return gerr(g, SWG_ERR_ARG, "swg_group_align_hits: sequence " + std::to_string(hits[h].index) +
                                                " is not in the loaded database");
        }
    }
    for (int i = 0; i < g->n; ++i) {
        std::vector<size_t> mine;
        for (size_t h = 0; h < n_hits; ++h)
            if (shard[h] == i) mine.push_back(h);
        if (mine.empty()) continue;
        std::vector<swg_hit> sub(mine.size());
        std::vector<swg_alignment> al(mine.size());
        std::vector<char> sub_ops(ops ? mine.size() * ops_stride : 0);
        for (size_t m = 0; m < mine.size(); ++m) sub[m] = hits[mine[m]];
        const int rc = swg_align_hits(g->ctx[i], g->db[i], sub.data(), sub.size(), al.data(), ops ? sub_ops.data() : nullptr,
                                      ops_stride);
        if (rc != SWG_OK) return gerr(g, rc, swg_last_error(g->ctx[i]));
        for (size_t m = 0; m < mine.size(); ++m) {
            out[mine[m]] = al[m];
            if (ops) memcpy(ops + mine[m] * ops_stride, sub_ops.data() + m * ops_stride, (size_t)al[m].n_ops + 1);
        }
    }
    return SWG_OK;
}
