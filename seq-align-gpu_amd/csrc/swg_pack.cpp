// swg_pack.cpp -- host-side database packer (no GPU needed).
//
// Replaces what the reference does per 16 records at
// src/alignment_cmdline.c:429-452 (letters -> indices, transpose to [j][lane],
// pad with '*'): here the whole database is sorted by length once (the
// reference instead REQUIRES a pre-sorted input, src/alignment_cmdline.c:431-439)
// and stored as residue bytes (index << 3) by sorted rank, every sequence a run
// of whole dwords (filled up with the padding residue 0).  That array, the
// lengths and the original indices are all that is copied to the GPU; the
// kernels' own layouts (pair tokens, bins) are built from them on the device.
#include "swg_host_internal.h"
#include "../../include/swg_host.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <new>
#include <string>
#include <thread>
#include <vector>

// Bins, slot tables and offsets from the lengths by sorted slot (db->lens, db->order filled in).
static void derive_tables(swg_db *db)
{
    const size_t nb = db->n_bins, ns = nb * SWG_BIN;
    db->bin_off.resize(nb);
    db->bin_nblk.resize(nb);
    uint64_t dwords = 0, residues = 0, rows_padded = 0;
    size_t n_local = 0;
    for (size_t lb = 0; lb < nb; ++lb) {
        const uint64_t len0 = db->lens[lb * SWG_BIN]; // sorted: the first of a bin is its longest
        const uint32_t nblk = (uint32_t)std::max<uint64_t>(1, (len0 + SWG_ROWS_PER_BLK - 1) / SWG_ROWS_PER_BLK);
        db->bin_off[lb] = dwords;
        db->bin_nblk[lb] = nblk;
        dwords += (uint64_t)nblk * SWG_BIN;
        rows_padded += (uint64_t)nblk * SWG_ROWS_PER_BLK * SWG_BIN;
    }
    db->code_off.assign(ns + 1, 0);
    for (size_t i = 0; i < ns; ++i) {
        db->code_off[i + 1] = db->code_off[i] + ((uint64_t)db->lens[i] + 3) / 4 * 4;
        residues += db->lens[i];
        n_local += db->order[i] != 0xFFFFFFFFu;
    }
    db->n_local = n_local;
    db->residues = residues;
    db->rows_padded = rows_padded;
    // rows of the pairs before pair p (two reset rows + the longer sequence): what the planner sums on every search
    // (swg_db_pair_rows); built here, once, so that searches never write to the database's host image
    {
        const size_t n_pairs = (n_local + 1) / 2;
        db->pair_rows_prefix.assign(n_pairs + 1, 0);
        for (size_t p = 0; p < n_pairs; ++p) db->pair_rows_prefix[p + 1] = db->pair_rows_prefix[p] + 2ull + db->lens[2 * p];
    }
    db->max_nblk = nb ? *std::max_element(db->bin_nblk.begin(), db->bin_nblk.end()) : 0;
}

// Step 1 of packing, once per database whatever the number of shards: validation (the reference exits in
// letters_to_index on anything outside A-Z/a-z/'*'; here an index outside 1..31 is an error code) and the global
// order -- a stable counting sort by length, descending.
static std::atomic<unsigned long> g_sorts{0}; // how many global sorts this process has run (swg_debug_sort_count)
extern "C" unsigned long swg_debug_sort_count(void) { return g_sorts.load(); }

static int sort_database(const char *what, const int8_t *flat, const uint64_t *offsets, size_t n, std::vector<uint32_t> *sorted)
{
    if (!offsets && n > 0) return swg_set_global_error(SWG_ERR_ARG, "%s: NULL input", what);
    if (n >= 0xFFFFFFF0ull) return swg_set_global_error(SWG_ERR_ARG, "%s: too many sequences", what);
    sorted->clear();
    if (n == 0) return SWG_OK; // nothing to read: offsets may be NULL
    if (!flat && offsets[n] > offsets[0]) return swg_set_global_error(SWG_ERR_ARG, "%s: NULL input", what);
    uint64_t max_len = 0;
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return swg_set_global_error(SWG_ERR_ARG, "%s: offsets not monotone at %zu", what, i);
        max_len = std::max<uint64_t>(max_len, offsets[i + 1] - offsets[i]);
    }
    if (max_len > 0x3FFFFFFFull) return swg_set_global_error(SWG_ERR_ARG, "%s: sequence too long", what);
    const uint64_t total = offsets[n] - offsets[0];
    {
        int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad) num_threads(swg_host_threads())
        for (long long i = (long long)offsets[0]; i < (long long)(offsets[0] + total); ++i) {
            const int v = flat[i];
            bad |= (v < 1 || v > 31);
        }
        if (bad) return swg_set_global_error(SWG_ERR_RESIDUE, "%s: residue index outside 1..31 in database", what);
    }
    sorted->resize(n);
    std::vector<uint64_t> cnt(max_len + 2, 0);
    for (size_t i = 0; i < n; ++i) cnt[max_len - (offsets[i + 1] - offsets[i]) + 1]++;
    for (size_t l = 1; l < cnt.size(); ++l) cnt[l] += cnt[l - 1];
    for (size_t i = 0; i < n; ++i) (*sorted)[cnt[max_len - (offsets[i + 1] - offsets[i])]++] = (uint32_t)i;
    g_sorts.fetch_add(1);
    return SWG_OK;
}

// Step 2, once per shard: the bins b % shard_count == shard_rank of the global order, re-coded.
// global_index: NULL (the sequences given ARE the database, index = position), or the original
// index of each sequence given (a shard that was cut elsewhere: swg_db_pack_shard); n_total: size
// of the whole database the indices refer to.  threads: cores the re-coding loop may take.
static int build_shard(const int8_t *flat, const uint64_t *offsets, size_t n, const std::vector<uint32_t> &sorted,
                       const uint32_t *global_index, size_t n_total, int shard_rank, int shard_count, int threads, swg_db **out)
{
    swg_db *db = new (std::nothrow) swg_db();
    if (!db) return swg_set_global_error(SWG_ERR_NOMEM, "swg_db_pack: out of memory");
    std::unique_ptr<swg_db> holder(db); // freed on every error path, exceptions included
    db->n_total = n_total;
    if (n == 0) {
        derive_tables(db);
        *out = holder.release();
        return SWG_OK;
    }
    // bins of the global order that belong to this shard
    const size_t n_bins_global = (n + SWG_BIN - 1) / SWG_BIN;
    std::vector<size_t> my_bins;
    for (size_t b = (size_t)shard_rank; b < n_bins_global; b += (size_t)shard_count)
        my_bins.push_back(b);
    const size_t nb = my_bins.size();
    db->n_bins = (uint32_t)nb;
    db->order.assign(nb * SWG_BIN, 0xFFFFFFFFu);
    db->lens.assign(nb * SWG_BIN, 0u);
    std::vector<uint32_t> src_of(nb * SWG_BIN, 0u); // position of a slot's sequence in the caller's arrays
    for (size_t lb = 0; lb < nb; ++lb) {
        const size_t first = my_bins[lb] * SWG_BIN;
        for (size_t s = 0; s < SWG_BIN && first + s < n; ++s) {
            const uint32_t oi = sorted[first + s];
            src_of[lb * SWG_BIN + s] = oi;
            db->order[lb * SWG_BIN + s] = global_index ? global_index[oi] : oi;
            db->lens[lb * SWG_BIN + s] = (uint32_t)(offsets[oi + 1] - offsets[oi]);
        }
    }
    derive_tables(db);
    db->codes.resize(db->code_off.back()); // written completely, sequence by sequence, below
    const long long ns = (long long)(nb * SWG_BIN);
#pragma omp parallel for schedule(dynamic, 512) num_threads(threads)
    for (long long s = 0; s < ns; ++s) {
        if (db->order[s] == 0xFFFFFFFFu) continue;
        const int8_t *src = flat + offsets[src_of[s]];
        const uint32_t len = db->lens[s];
        uint8_t *cd = db->codes.data() + db->code_off[s];
        for (uint32_t j = 0; j < len; ++j) cd[j] = (uint8_t)((uint8_t)src[j] << 3);
        for (uint32_t j = len; j < (len + 3) / 4 * 4; ++j) cd[j] = 0; // padding residue up to the dword
    }
    *out = holder.release();
    return SWG_OK;
}

static int pack_impl(const int8_t *flat, const uint64_t *offsets, size_t n, const uint32_t *global_index,
                     size_t n_total, int shard_rank, int shard_count, swg_db **out)
{
    if (!out) return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack: out is NULL");
    *out = nullptr;
    if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count)
        return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack: bad shard %d/%d", shard_rank,
                                    shard_count);
    std::vector<uint32_t> sorted;
    const int rc = sort_database("swg_db_pack", flat, offsets, n, &sorted);
    if (rc != SWG_OK) return rc;
    return build_shard(flat, offsets, n, sorted, global_index, n_total, shard_rank, shard_count, swg_host_threads(), out);
}

// Every shard of one database from ONE global sort (swg_group_load: one process, several GPUs).  ready(r, db), when
// given, is called on the thread that built shard r as soon as it is built (the group uploads it from there, so a
// device's transfer overlaps the other shards' re-coding); its non-zero return stops that shard.  The shards are
// built side by side, one host thread each, sharing the cores.
int swg_pack_shards(const int8_t *flat, const uint64_t *offsets, size_t n, int shard_count, swg_db **out,
                    const std::function<int(int, swg_db *)> &ready)
{
    if (!out || shard_count < 1) return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack_shards: bad argument");
    for (int r = 0; r < shard_count; ++r) out[r] = nullptr;
    std::vector<uint32_t> sorted;
    int rc = sort_database("swg_db_pack_shards", flat, offsets, n, &sorted);
    if (rc != SWG_OK) return rc;
    const int threads = std::max(1, swg_host_threads() / shard_count);
    std::vector<int> rcs((size_t)shard_count, SWG_OK);
    std::vector<std::string> errs((size_t)shard_count);
    auto work = [&](int r) {
        try {
            rcs[r] = build_shard(flat, offsets, n, sorted, nullptr, n, r, shard_count, threads, &out[r]);
            if (rcs[r] == SWG_OK && ready) rcs[r] = ready(r, out[r]);
        } catch (const std::exception &e) {
            rcs[r] = SWG_ERR_NOMEM;
            errs[r] = e.what();
        }
        if (rcs[r] != SWG_OK && errs[r].empty()) errs[r] = swg_global_error(); // (thread-local: carried to the caller's thread below)
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)shard_count);
    int started = 1; // shards 0 .. started-1 have a worker (shard 0: this thread)
    try {
        for (int r = 1; r < shard_count; ++r) {
            pool.emplace_back(work, r);
            started = r + 1;
        }
    } catch (const std::exception &) {
        // (no more threads to be had: the shards without one are built here, after the others -- a joinable std::thread
        // must not be destroyed, so nothing may leave this function before the joins below)
    }
    work(0);
    for (int r = started; r < shard_count; ++r) work(r);
    for (std::thread &t : pool) t.join();
    for (int r = 0; r < shard_count; ++r)
        if (rcs[r] != SWG_OK) {
            rc = rcs[r];
            const std::string msg = errs[r];
            for (int q = 0; q < shard_count; ++q) {
                swg_db_free(out[q]);
                out[q] = nullptr;
            }
            return swg_set_global_error(rc, "swg_db_pack_shards: shard %d: %s", r, msg.c_str());
        }
    return SWG_OK;
}

// No C++ exception crosses the ABI: allocation failures and length errors of the containers
// become SWG_ERR_NOMEM.
template <class F> static int guarded(const char *what, F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return swg_set_global_error(SWG_ERR_NOMEM, "%s: out of memory", what);
    } catch (const std::exception &e) {
        return swg_set_global_error(SWG_ERR_NOMEM, "%s: %s", what, e.what());
    }
}

extern "C" int swg_db_pack(const int8_t *flat, const uint64_t *offsets, size_t n, int shard_rank,
                           int shard_count, swg_db **out)
{
    return guarded("swg_db_pack", [&] { return pack_impl(flat, offsets, n, nullptr, n, shard_rank, shard_count, out); });
}

extern "C" int swg_db_pack_shard(const int8_t *flat, const uint64_t *offsets, size_t n_local,
                                 const uint32_t *global_index, size_t n_total, swg_db **out)
{
    if (n_local > 0 && !global_index)
        return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack_shard: global_index is NULL");
    if (n_local > n_total || n_total >= 0xFFFFFFF0ull)
        return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack_shard: %zu sequences of a database of %zu", n_local, n_total);
    for (size_t i = 0; i < n_local; ++i)
        if (global_index[i] >= n_total)
            return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack_shard: index %u at %zu outside the database", global_index[i], i);
    return guarded("swg_db_pack_shard", [&] { return pack_impl(flat, offsets, n_local, global_index, n_total, 0, 1, out); });
}

extern "C" int swg_db_pack_shards(const int8_t *flat, const uint64_t *offsets, size_t n, int shard_count, swg_db **out)
{
    return guarded("swg_db_pack_shards", [&] { return swg_pack_shards(flat, offsets, n, shard_count, out, nullptr); });
}

// ---------------------------------------------------------------------------
// packed database file: the host image of a swg_db (sorting and re-coding are the expensive
// part of ingest), loaded with plain reads afterwards.  Layout: header, order[n_slots],
// lens[n_slots], codes.
// ---------------------------------------------------------------------------
namespace {
const char kMagic[8] = {'S', 'W', 'G', 'D', 'B', '0', '2', '\0'};
struct FileHeader {
    char magic[8];
    uint64_t n_total, n_local, n_bins, residues, n_codes, reserved[3];
};
template <class V> bool put(FILE *f, const V &v)
{
    return v.empty() || fwrite(v.data(), sizeof(typename V::value_type), v.size(), f) == v.size();
}
template <class V> bool get(FILE *f, V &v, size_t n)
{
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(typename V::value_type), n, f) == n;
}
} // namespace

// Reference-shaped batches (alignment_fill_matrices' input: db_idx_t[max_len][16], the 16 lanes of a row
// side by side, src/alignment_cmdline.c:434,445) back into records laid end to end, padded rows included
// (the reference computes them as real rows).  first_rec[b] = index of batch b's lane 0 among the records,
// rec_off = the records' offsets in flat.  All cores: this is most of what the compatibility route costs.
// The same batches as they are, end to end, into pinned staging (swg_fill_batches16's device route): batches
// order[lo..hi) to stage + stage_off[batch].  Lives here because this file is compiled by g++ with OpenMP (hipcc
// compiles swg_api.cpp without it: a pragma there is ignored).
void swg_stage_batches16(const swg_batch16 *batches, const uint32_t *order, const uint64_t *stage_off, size_t lo, size_t hi,
                         uint8_t *stage)
{
#pragma omp parallel for schedule(dynamic, 16) num_threads(swg_host_threads())
    for (long long k = (long long)lo; k < (long long)hi; ++k) {
        const swg_batch16 &bt = batches[order[k]];
        memcpy(stage + stage_off[order[k]], bt.db_idx_t, (size_t)bt.max_len * 16u);
    }
}

void swg_untranspose_batches16(const swg_batch16 *batches, size_t n_batches, const size_t *first_rec,
                               const uint64_t *rec_off, int8_t *flat)
{
#pragma omp parallel for schedule(dynamic, 16) num_threads(swg_host_threads())
    for (long long b = 0; b < (long long)n_batches; ++b) {
        const swg_batch16 &bt = batches[b];
        const int8_t *src = bt.db_idx_t;
        int8_t *dst[16];
        for (size_t l = 0; l < bt.vector_size; ++l) dst[l] = flat + rec_off[first_rec[b] + l];
        if (bt.vector_size == 16) {
            for (size_t j = 0; j < bt.max_len; ++j, src += 16)
                for (int l = 0; l < 16; ++l) dst[l][j] = src[l];
        } else {
            for (size_t j = 0; j < bt.max_len; ++j, src += 16)
                for (size_t l = 0; l < bt.vector_size; ++l) dst[l][j] = src[l];
        }
    }
}

extern "C" int swg_db_save(const swg_db *db, const char *path)
{
    if (!db || !path) return swg_set_global_error(SWG_ERR_ARG, "swg_db_save: NULL argument");
    FILE *f = fopen(path, "wb");
    if (!f) return swg_set_global_error(SWG_ERR_IO, "swg_db_save: cannot write %s", path);
    FileHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, kMagic, 8);
    h.n_total = db->n_total;
    h.n_local = db->n_local;
    h.n_bins = db->n_bins;
    h.residues = db->residues;
    h.n_codes = db->codes.size();
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && put(f, db->order) && put(f, db->lens) && put(f, db->codes);
    ok = (fclose(f) == 0) && ok;
    if (!ok) return swg_set_global_error(SWG_ERR_IO, "swg_db_save: short write to %s", path);
    return SWG_OK;
}

static int load_impl(const char *path, swg_db **out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return swg_set_global_error(SWG_ERR_IO, "swg_db_load: cannot read %s", path);
    FileHeader h;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, kMagic, 8) != 0) {
        fclose(f);
        return swg_set_global_error(SWG_ERR_IO, "swg_db_load: %s is not a packed database", path);
    }
    // every count is checked against the size of the file before anything is sized by it
    uint64_t file_bytes = 0;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long end = ftell(f);
        if (end >= 0) file_bytes = (uint64_t)end;
    }
    const bool seek_ok = fseek(f, (long)sizeof h, SEEK_SET) == 0;
    const uint64_t ns = h.n_bins <= 0xFFFFFFFFull / SWG_BIN ? h.n_bins * SWG_BIN : 0;
    const bool counts_ok = seek_ok && h.n_bins <= 0xFFFFFFFFull / SWG_BIN && h.n_local <= ns && h.n_local <= h.n_total &&
                           h.n_total < 0xFFFFFFF0ull && h.n_codes <= file_bytes && h.n_codes % 4 == 0 &&
                           sizeof h + ns * 8 + h.n_codes == file_bytes;
    if (!counts_ok) {
        fclose(f);
        return swg_set_global_error(SWG_ERR_IO, "swg_db_load: inconsistent header in %s", path);
    }
    swg_db *db = new (std::nothrow) swg_db();
    if (!db) {
        fclose(f);
        return swg_set_global_error(SWG_ERR_NOMEM, "swg_db_load: out of memory");
    }
    std::unique_ptr<swg_db> holder(db);
    struct Closer {
        FILE *f;
        ~Closer() { fclose(f); }
    } closer{f};
    db->n_total = h.n_total;
    db->n_bins = (uint32_t)h.n_bins;
    bool ok = get(f, db->order, ns) && get(f, db->lens, ns);
    // structural checks before the tables are derived: nothing read from the file is trusted as an
    // index or a size unchecked, and the invariants the kernels rely on are verified, not assumed
    // (lengths sorted longest first, so that a pair's second sequence is the shorter one; empty
    // slots only at the end)
    for (size_t s = 0; s < ns && ok; ++s) {
        const bool empty = db->order[s] == 0xFFFFFFFFu;
        ok = (empty || db->order[s] < db->n_total) && (!empty || db->lens[s] == 0) && db->lens[s] <= 0x3FFFFFFFu &&
             (s == 0 || db->lens[s] <= db->lens[s - 1]) && (s == 0 || empty || db->order[s - 1] != 0xFFFFFFFFu);
    }
    if (ok) {
        derive_tables(db);
        ok = db->n_local == h.n_local && db->residues == h.residues && db->code_off.back() == h.n_codes &&
             get(f, db->codes, h.n_codes);
    }
    if (ok) { // residue bytes: index << 3 of an index in 1..31, zeros up to the dword boundary
        int bad = 0;
        const long long nsl = (long long)ns;
#pragma omp parallel for schedule(dynamic, 512) reduction(| : bad) num_threads(swg_host_threads())
        for (long long s = 0; s < nsl; ++s) {
            const uint8_t *cd = db->codes.data() + db->code_off[s];
            const uint32_t len = db->lens[s];
            for (uint32_t j = 0; j < len; ++j) bad |= (cd[j] & 7u) != 0 || cd[j] == 0;
            for (uint32_t j = len; j < (len + 3) / 4 * 4; ++j) bad |= cd[j] != 0;
        }
        ok = bad == 0;
    }
    if (!ok) return swg_set_global_error(SWG_ERR_IO, "swg_db_load: %s is truncated or corrupt", path);
    *out = holder.release();
    return SWG_OK;
}

extern "C" int swg_db_load(const char *path, swg_db **out)
{
    if (!path || !out) return swg_set_global_error(SWG_ERR_ARG, "swg_db_load: NULL argument");
    *out = nullptr;
    return guarded("swg_db_load", [&] { return load_impl(path, out); });
}

extern "C" void swg_db_free(swg_db *db)
{
    if (!db) return;
    swg_db_release_device(db);
    delete db;
}

extern "C" size_t swg_db_count(const swg_db *db) { return db ? db->n_local : 0; }
extern "C" size_t swg_db_total_count(const swg_db *db) { return db ? db->n_total : 0; }
extern "C" uint64_t swg_db_residues(const swg_db *db) { return db ? db->residues : 0; }
// what swg_db_upload copies to the GPU: the residue bytes and 16 bytes per slot (length, original
// index, offset), plus one block offset per pair of sequences
extern "C" uint64_t swg_db_packed_bytes(const swg_db *db)
{
    if (!db) return 0;
    const uint64_t ns = (uint64_t)db->n_bins * SWG_BIN;
    return (uint64_t)db->codes.size() + ns * 16u + 8u + (ns / 2 + 1) * 4u;
}
extern "C" const uint32_t *swg_db_order(const swg_db *db) { return db ? db->order.data() : nullptr; }
