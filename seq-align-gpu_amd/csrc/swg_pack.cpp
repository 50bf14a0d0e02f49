// swg_pack.cpp -- host-side database packer (no GPU needed).
//
// Replaces what the reference does per 16 records at
// src/alignment_cmdline.c:429-452 (letters -> indices, transpose to [j][lane],
// pad with '*'): here the whole database is sorted by length once (the
// reference instead REQUIRES a pre-sorted input, src/alignment_cmdline.c:431-439),
// cut into bins of 128 sequences and stored row-block-major, one dword per four
// residues per sequence, padding = residue 0.
#include "swg_host_internal.h"
#include "../../include/swg_host.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

extern "C" int swg_db_pack(const int8_t *flat, const uint64_t *offsets, size_t n, int shard_rank,
                           int shard_count, swg_db **out)
{
    if (!out) return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack: out is NULL");
    *out = nullptr;
    if ((!flat && n > 0 && offsets && offsets[n] > 0) || (!offsets && n > 0))
        return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack: NULL input");
    if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count)
        return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack: bad shard %d/%d", shard_rank,
                                    shard_count);
    if (n >= 0xFFFFFFF0ull)
        return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack: too many sequences");

    // lengths + validation (the reference exits in letters_to_index on anything
    // outside A-Z/a-z/'*'; here an index outside 1..31 is an error code)
    uint64_t max_len = 0;
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i])
            return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack: offsets not monotone at %zu", i);
        max_len = std::max<uint64_t>(max_len, offsets[i + 1] - offsets[i]);
    }
    if (max_len > 0x3FFFFFFFull)
        return swg_set_global_error(SWG_ERR_ARG, "swg_db_pack: sequence too long");
    const uint64_t total = n ? offsets[n] - offsets[0] : 0;
    {
        int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad) num_threads(swg_host_threads())
        for (long long i = (long long)offsets[0]; i < (long long)(offsets[0] + total); ++i) {
            const int v = flat[i];
            bad |= (v < 1 || v > 31);
        }
        if (bad)
            return swg_set_global_error(SWG_ERR_RESIDUE,
                                        "swg_db_pack: residue index outside 1..31 in database");
    }

    swg_db *db = new (std::nothrow) swg_db();
    if (!db) return swg_set_global_error(SWG_ERR_NOMEM, "swg_db_pack: out of memory");
    db->n_total = n;

    // stable counting sort by length, descending
    std::vector<uint32_t> sorted(n);
    {
        std::vector<uint64_t> cnt(max_len + 2, 0);
        for (size_t i = 0; i < n; ++i) cnt[max_len - (offsets[i + 1] - offsets[i]) + 1]++;
        for (size_t l = 1; l < cnt.size(); ++l) cnt[l] += cnt[l - 1];
        for (size_t i = 0; i < n; ++i) sorted[cnt[max_len - (offsets[i + 1] - offsets[i])]++] = (uint32_t)i;
    }

    // bins of the global order that belong to this shard
    const size_t n_bins_global = (n + SWG_BIN - 1) / SWG_BIN;
    std::vector<size_t> my_bins;
    for (size_t b = (size_t)shard_rank; b < n_bins_global; b += (size_t)shard_count)
        my_bins.push_back(b);
    const size_t nb = my_bins.size();
    db->n_bins = (uint32_t)nb;
    db->bin_off.resize(nb);
    db->bin_nblk.resize(nb);
    db->order.assign(nb * SWG_BIN, 0xFFFFFFFFu);
    db->lens.assign(nb * SWG_BIN, 0u);
    uint64_t dwords = 0;
    uint64_t residues = 0, rows_padded = 0;
    size_t n_local = 0;
    for (size_t lb = 0; lb < nb; ++lb) {
        const size_t first = my_bins[lb] * SWG_BIN;
        const uint64_t len0 = offsets[sorted[first] + 1] - offsets[sorted[first]];
        const uint32_t nblk = (uint32_t)std::max<uint64_t>(1, (len0 + SWG_ROWS_PER_BLK - 1) / SWG_ROWS_PER_BLK);
        db->bin_off[lb] = dwords;
        db->bin_nblk[lb] = nblk;
        dwords += (uint64_t)nblk * SWG_BIN;
        rows_padded += (uint64_t)nblk * SWG_ROWS_PER_BLK * SWG_BIN;
        for (size_t s = 0; s < SWG_BIN && first + s < n; ++s) {
            const uint32_t oi = sorted[first + s];
            db->order[lb * SWG_BIN + s] = oi;
            const uint32_t len = (uint32_t)(offsets[oi + 1] - offsets[oi]);
            db->lens[lb * SWG_BIN + s] = len;
            residues += len;
            ++n_local;
        }
    }
    db->n_local = n_local;
    db->residues = residues;
    db->rows_padded = rows_padded;
    db->max_nblk = nb ? *std::max_element(db->bin_nblk.begin(), db->bin_nblk.end()) : 0;
    try {
        db->packed.resize(dwords); // every bin zeroes its own blocks in the parallel loop below
    } catch (const std::bad_alloc &) {
        delete db;
        return swg_set_global_error(SWG_ERR_NOMEM, "swg_db_pack: out of memory (%llu dwords)",
                                    (unsigned long long)dwords);
    }

    try {
        db->code_off.assign(nb * SWG_BIN + 1, 0);
        for (size_t i = 0; i < nb * SWG_BIN; ++i) db->code_off[i + 1] = db->code_off[i] + db->lens[i];
        db->codes.resize(residues); // written completely, sequence by sequence, below
    } catch (const std::bad_alloc &) {
        delete db;
        return swg_set_global_error(SWG_ERR_NOMEM, "swg_db_pack: out of memory");
    }

    uint32_t *pk = db->packed.data();
#pragma omp parallel for schedule(dynamic, 4) num_threads(swg_host_threads())
    for (long long lb = 0; lb < (long long)nb; ++lb) {
        uint32_t *base = pk + db->bin_off[lb];
        memset(base, 0, (size_t)db->bin_nblk[lb] * SWG_BIN * sizeof(uint32_t)); // padding rows and empty slots
        for (size_t s = 0; s < SWG_BIN; ++s) {
            const uint32_t oi = db->order[lb * SWG_BIN + s];
            if (oi == 0xFFFFFFFFu) continue;
            const int8_t *src = flat + offsets[oi];
            const uint32_t len = db->lens[lb * SWG_BIN + s];
            uint8_t *cd = db->codes.data() + db->code_off[lb * SWG_BIN + s];
            for (uint32_t j = 0; j < len; ++j) cd[j] = (uint8_t)((uint8_t)src[j] << 3);
            for (uint32_t j = 0; j < len; j += 4) {
                uint32_t wd = 0;
                const uint32_t m = std::min<uint32_t>(4, len - j);
                for (uint32_t r = 0; r < m; ++r) wd |= ((uint32_t)(uint8_t)src[j + r] << 3) << (8 * r);
                base[(size_t)(j / 4) * SWG_BIN + SWG_BIN_COLUMN((uint32_t)s)] = wd;
            }
        }
    }
    *out = db;
    return SWG_OK;
}

// ---------------------------------------------------------------------------
// packed database file: the host image of a swg_db, written once (sorting, binning and
// dword-packing are the expensive part of ingest), loaded with plain reads afterwards
// ---------------------------------------------------------------------------
namespace {
const char kMagic[8] = {'S', 'W', 'G', 'D', 'B', '0', '1', '\0'};
struct FileHeader {
    char magic[8];
    uint64_t n_total, n_local, n_bins, max_nblk, residues, rows_padded;
    uint64_t n_packed, n_codes;
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
    h.max_nblk = db->max_nblk;
    h.residues = db->residues;
    h.rows_padded = db->rows_padded;
    h.n_packed = db->packed.size();
    h.n_codes = db->codes.size();
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && put(f, db->bin_off) && put(f, db->bin_nblk) && put(f, db->order) &&
              put(f, db->lens) && put(f, db->code_off) && put(f, db->codes) && put(f, db->packed);
    ok = (fclose(f) == 0) && ok;
    if (!ok) return swg_set_global_error(SWG_ERR_IO, "swg_db_save: short write to %s", path);
    return SWG_OK;
}

extern "C" int swg_db_load(const char *path, swg_db **out)
{
    if (!path || !out) return swg_set_global_error(SWG_ERR_ARG, "swg_db_load: NULL argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return swg_set_global_error(SWG_ERR_IO, "swg_db_load: cannot read %s", path);
    FileHeader h;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, kMagic, 8) != 0) {
        fclose(f);
        return swg_set_global_error(SWG_ERR_IO, "swg_db_load: %s is not a packed database", path);
    }
    const uint64_t ns = h.n_bins * SWG_BIN;
    if (h.n_local > ns || h.n_bins > 0xFFFFFFFFull / SWG_BIN || h.n_local > h.n_total) {
        fclose(f);
        return swg_set_global_error(SWG_ERR_IO, "swg_db_load: inconsistent header in %s", path);
    }
    swg_db *db = new (std::nothrow) swg_db();
    if (!db) {
        fclose(f);
        return swg_set_global_error(SWG_ERR_NOMEM, "swg_db_load: out of memory");
    }
    bool ok = false;
    try {
        db->n_total = h.n_total;
        db->n_local = h.n_local;
        db->n_bins = (uint32_t)h.n_bins;
        db->max_nblk = (uint32_t)h.max_nblk;
        db->residues = h.residues;
        db->rows_padded = h.rows_padded;
        ok = get(f, db->bin_off, h.n_bins) && get(f, db->bin_nblk, h.n_bins) && get(f, db->order, ns) &&
             get(f, db->lens, ns) && get(f, db->code_off, ns + 1) && get(f, db->codes, h.n_codes) &&
             get(f, db->packed, h.n_packed);
    } catch (const std::bad_alloc &) {
        ok = false;
    }
    fclose(f);
    // cheap structural checks: nothing read from the file is trusted as an index unchecked
    if (ok) {
        uint64_t dwords = 0;
        for (size_t b = 0; b < db->n_bins && ok; ++b) {
            ok = db->bin_off[b] == dwords && db->bin_nblk[b] >= 1 && db->bin_nblk[b] <= db->max_nblk;
            dwords += (uint64_t)db->bin_nblk[b] * SWG_BIN;
        }
        ok = ok && dwords == db->packed.size() && db->code_off[ns] == db->codes.size() &&
             db->residues == db->codes.size() && db->rows_padded == dwords * SWG_ROWS_PER_BLK;
        for (size_t s = 0; s < ns && ok; ++s) {
            ok = db->code_off[s + 1] - db->code_off[s] == db->lens[s] &&
                 (db->order[s] == 0xFFFFFFFFu || db->order[s] < db->n_total) &&
                 db->lens[s] <= (uint64_t)db->bin_nblk[s / SWG_BIN] * SWG_ROWS_PER_BLK;
        }
    }
    if (!ok) {
        delete db;
        return swg_set_global_error(SWG_ERR_IO, "swg_db_load: %s is truncated or corrupt", path);
    }
    *out = db;
    return SWG_OK;
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
extern "C" uint64_t swg_db_packed_bytes(const swg_db *db)
{
    return db ? (uint64_t)db->packed.size() * 4u + (uint64_t)db->n_bins * 12u : 0;
}
extern "C" const uint32_t *swg_db_order(const swg_db *db) { return db ? db->order.data() : nullptr; }
