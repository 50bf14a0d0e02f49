// swg_trace.hip -- alignments of reported hits ("traceback for the top-K only").
//
// The reference prints scores only: its fork removed the traceback of upstream seq-align (Final
// Report p.7, p.10; what is left is the comment at src/alignment.c:46).  A search here returns the K
// best pairs; this unit re-runs exactly those K pairs with the recurrence of src/alignment.c:124-161
// kept whole and walks back from the best match cell.  It is a cold path (K pairs, not the
// database): one workgroup per pair sweeps the anti-diagonals of the pair's matrix, 256 cells at a
// time, in int32, three rotating diagonals per state (in LDS for queries up to 1700 columns, else
// in HBM/L2), one predecessor byte per cell stored diagonal-major (coalesced), then one lane follows
// the bytes back.  Everything runs on
// the GPU; like the rest of the library there is no CPU path.
#include "swg_host_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <unordered_map>

struct SwgTraceJob {
    uint64_t res_off; // into the gathered residue indices
    uint64_t dir_off; // into the predecessor bytes
    uint32_t len;     // database sequence length
    uint32_t pad;
};

struct SwgTraceOut {
    int32_t score;
    uint32_t q_begin, q_end, d_begin, d_end, n_ops, pad[2];
};

struct SwgTraceParams {
    const int8_t *query; // [lq] table indices
    const int8_t *sub;   // [32][32], row = query residue
    const int8_t *res;   // database residue indices of the jobs, back to back
    const SwgTraceJob *jobs;
    int32_t *diag;       // per job 9 * (lq + 1): three rotating anti-diagonals of H, A and B
    uint8_t *dir;        // per job (lq + len - 1) * lq predecessor bytes, diagonal-major
    char *ops;           // per job ops_stride bytes
    SwgTraceOut *out;
    uint32_t lq, ops_stride;
    int go, ge;
};

#define SWG_TRACE_THREADS 256
#define SWG_TRACE_LDS_COLS 1700 /* 9 * (cols + 1) * 4 bytes <= 60 KB */

// predecessor code of one state: 0 = the alignment starts here, 1/2/3 = came from H/A/B
__device__ __forceinline__ uint32_t trace_pick(int32_t m, int32_t x, int32_t y)
{
    return m == 0 ? 0u : x == m ? 1u : y == m ? 2u : 3u;
}

// IN_LDS: the nine rotating anti-diagonals live in the workgroup's LDS (queries up to
// SWG_TRACE_LDS_COLS columns; one dependent sweep then waits for LDS, not for L2).
template <bool IN_LDS>
__global__ __launch_bounds__(SWG_TRACE_THREADS) void swg_trace_kernel(SwgTraceParams p)
{
    extern __shared__ int32_t s_diag[];
    __shared__ int8_t s_sub[1024];
    __shared__ int s_best;
    __shared__ unsigned long long s_pos;
    __shared__ uint32_t s_n;
    const SwgTraceJob job = p.jobs[blockIdx.x];
    const uint32_t lq = p.lq, len = job.len, tid = threadIdx.x, w = lq + 1;
    for (uint32_t k = tid; k < 1024; k += SWG_TRACE_THREADS) s_sub[k] = p.sub[k];
    if (tid == 0) {
        s_best = 0;
        s_pos = ~0ull;
        s_n = 0;
    }
    int32_t *X;
    if (IN_LDS) X = s_diag;
    else X = p.diag + (size_t)blockIdx.x * 9 * w;
    const int8_t *d = p.res + job.res_off;
    uint8_t *dir = p.dir + job.dir_off;
    const int go = p.go, ge = p.ge;
    __syncthreads();

    // cell (j, i): database row j, query column i, both from 1; anti-diagonal dg = i + j.  The
    // buffers are indexed by i: (j-1, i) and (j, i-1) lie on dg-1 at i and i-1, (j-1, i-1) on dg-2 at i-1.
    int32_t best = 0;
    uint32_t bj = 0, bi = 0;
    for (uint32_t dg = 2; dg <= lq + len; ++dg) {
        const uint32_t c0 = dg % 3, c1 = (dg + 2) % 3, c2 = (dg + 1) % 3;
        int32_t *H0 = X + c0 * w, *A0 = X + (3 + c0) * w, *B0 = X + (6 + c0) * w;
        const int32_t *H1 = X + c1 * w, *A1 = X + (3 + c1) * w, *B1 = X + (6 + c1) * w;
        const int32_t *H2 = X + c2 * w, *A2 = X + (3 + c2) * w, *B2 = X + (6 + c2) * w;
        const uint32_t ilo = dg > len ? dg - len : 1u, ihi = min(lq, dg - 1);
        uint8_t *drow = dir + (size_t)(dg - 2) * lq;
        for (uint32_t i = ilo + tid; i <= ihi; i += SWG_TRACE_THREADS) {
            const uint32_t j = dg - i;
            int32_t hd = 0, ad = 0, bd = 0, hu = 0, au = 0, bu = 0, hl = 0, al = 0, bl = 0;
            if (j > 1) {
                hu = H1[i], au = A1[i], bu = B1[i];
                if (i > 1) hd = H2[i - 1], ad = A2[i - 1], bd = B2[i - 1];
            }
            if (i > 1) hl = H1[i - 1], al = A1[i - 1], bl = B1[i - 1];
            const int32_t s = s_sub[(int)p.query[i - 1] * 32 + (int)d[j - 1]];
            const int32_t mh = max(max(hd, ad), max(bd, 0));
            const int32_t xa = hu + go, ya = au + ge, za = bu + go;
            const int32_t ma = max(max(xa, ya), max(za, 0));
            const int32_t xb = hl + go, yb = al + go, zb = bl + ge;
            const int32_t mb = max(max(xb, yb), max(zb, 0));
            drow[i - 1] = (uint8_t)(trace_pick(mh, hd, ad) | trace_pick(ma, xa, ya) << 2 | trace_pick(mb, xb, yb) << 4);
            const int32_t h = mh + s;
            H0[i] = h, A0[i] = ma, B0[i] = mb;
            if (h > best || (h == best && h > 0 && (j < bj || (j == bj && i < bi)))) best = h, bj = j, bi = i;
        }
        __syncthreads();
    }

    // best match cell: highest score, then smallest database position, then smallest query position
    atomicMax(&s_best, best);
    __syncthreads();
    if (best == s_best && best > 0) atomicMin(&s_pos, (unsigned long long)bj << 32 | bi);
    __syncthreads();

    char *ops = p.ops + (size_t)blockIdx.x * p.ops_stride;
    if (tid == 0) {
        SwgTraceOut o = {};
        uint32_t n = 0;
        if (s_best > 0) {
            uint32_t j = (uint32_t)(s_pos >> 32), i = (uint32_t)s_pos;
            o.score = s_best, o.q_end = i, o.d_end = j;
            uint32_t state = 1;
            while (j > 0 && i > 0 && n + 1 < p.ops_stride) {
                const uint32_t c = dir[(size_t)(i + j - 2) * lq + (i - 1)];
                uint32_t from;
                if (state == 1) ops[n++] = 'M', from = c & 3, --j, --i;
                else if (state == 2) ops[n++] = 'I', from = (c >> 2) & 3, --j;
                else ops[n++] = 'D', from = (c >> 4) & 3, --i;
                if (from == 0) break;
                state = from;
            }
            o.q_begin = i, o.d_begin = j;
        }
        o.n_ops = n;
        ops[n] = 0;
        p.out[blockIdx.x] = o;
        s_n = n;
    }
    __syncthreads();
    const uint32_t n = s_n; // written last to first: turn it round
    for (uint32_t k = tid; k < n / 2; k += SWG_TRACE_THREADS) {
        const char a = ops[k], b = ops[n - 1 - k];
        ops[k] = b, ops[n - 1 - k] = a;
    }
}

#define TRACE_TRY(ctx, expr)                                                                            \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            rc = swg_set_ctx_error(ctx, e_ == hipErrorOutOfMemory ? SWG_ERR_NOMEM : SWG_ERR_HIP,        \
                                   "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            goto done;                                                                                  \
        }                                                                                               \
    } while (0)

extern "C" size_t swg_align_ops_bound(const swg_ctx *ctx, const swg_db *db)
{
    if (!ctx || !db) return 0;
    uint32_t longest = 0;
    for (uint32_t l : db->lens) longest = std::max(longest, l);
    return ctx->query.size() + longest + 1;
}

extern "C" int swg_align_hits(swg_ctx *ctx, const swg_db *db, const swg_hit *hits, size_t n_hits,
                              swg_alignment *out, char *ops, size_t ops_stride)
{
    if (!ctx) return swg_set_global_error(SWG_ERR_ARG, "swg_align_hits: NULL context");
    if (!db || (n_hits && (!hits || !out)))
        return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_align_hits: NULL argument");
    if (!ctx->have_scoring || ctx->query.empty())
        return swg_set_ctx_error(ctx, SWG_ERR_STATE, "swg_align_hits: scoring and query must be set first");
    if (ops && ops_stride == 0) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_align_hits: ops_stride is 0");
    if (n_hits == 0) return SWG_OK;
    if (n_hits > (1u << 20)) return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_align_hits: more than 2^20 hits");
    const size_t lq = ctx->query.size();

    // original index -> slot of the sorted order, for the wanted sequences only
    std::unordered_map<uint32_t, size_t> slot_of;
    slot_of.reserve(n_hits * 2);
    for (size_t h = 0; h < n_hits; ++h) slot_of[hits[h].index] = SIZE_MAX;
    for (size_t s = 0; s < db->order.size(); ++s) {
        if (db->order[s] == ~0u) continue;
        auto it = slot_of.find(db->order[s]);
        if (it != slot_of.end()) it->second = s;
    }
    std::vector<SwgTraceJob> jobs(n_hits);
    std::vector<int8_t> res;
    size_t longest = 0;
    for (size_t h = 0; h < n_hits; ++h) {
        const size_t s = slot_of[hits[h].index];
        if (s == SIZE_MAX)
            return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_align_hits: sequence %u is not in this database shard",
                                     hits[h].index);
        const size_t len = db->lens[s];
        const uint64_t cells = (uint64_t)(lq + len) * lq;
        if (len == 0 || cells > (16ull << 30))
            return swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_align_hits: pair %u (%zu x %zu) is outside what a traceback holds",
                                     hits[h].index, lq, len);
        jobs[h].res_off = res.size();
        jobs[h].len = (uint32_t)len;
        jobs[h].pad = 0;
        const uint8_t *c = db->codes.data() + db->code_off[s];
        for (size_t r = 0; r < len; ++r) res.push_back((int8_t)(c[r] >> 3)); // codes are index << 3
        longest = std::max(longest, len);
    }
    const size_t dev_stride = lq + longest + 1; // a path has at most lq + len steps

    int rc = SWG_OK;
    int8_t *d_query = nullptr, *d_sub = nullptr, *d_res = nullptr;
    SwgTraceJob *d_jobs = nullptr;
    int32_t *d_diag = nullptr;
    uint8_t *d_dir = nullptr;
    char *d_ops = nullptr;
    SwgTraceOut *d_out = nullptr;
    std::vector<SwgTraceOut> h_out(n_hits);
    std::vector<char> h_ops;
    size_t max_chunk = 0, max_dir = 0;
    const bool in_lds = lq <= SWG_TRACE_LDS_COLS;
    {
        // chunks of consecutive hits whose predecessor bytes fit 2 GiB together (a single larger pair goes alone)
        const uint64_t budget = 2ull << 30;
        for (size_t b = 0; b < n_hits;) {
            uint64_t bytes = 0;
            size_t e = b;
            while (e < n_hits) {
                const uint64_t need = (uint64_t)(lq + jobs[e].len - 1) * lq;
                if (e > b && bytes + need > budget) break;
                bytes += need, ++e;
            }
            max_chunk = std::max(max_chunk, e - b), max_dir = std::max<size_t>(max_dir, bytes);
            b = e;
        }
    }
    TRACE_TRY(ctx, hipSetDevice(ctx->device));
    TRACE_TRY(ctx, hipMalloc(&d_query, lq));
    TRACE_TRY(ctx, hipMalloc(&d_sub, 1024));
    TRACE_TRY(ctx, hipMalloc(&d_res, std::max<size_t>(res.size(), 4)));
    TRACE_TRY(ctx, hipMalloc(&d_jobs, n_hits * sizeof(SwgTraceJob)));
    TRACE_TRY(ctx, hipMalloc(&d_diag, in_lds ? 16 : max_chunk * 9 * (lq + 1) * sizeof(int32_t)));
    TRACE_TRY(ctx, hipMalloc(&d_dir, std::max<size_t>(max_dir, 4)));
    TRACE_TRY(ctx, hipMalloc(&d_ops, max_chunk * dev_stride));
    TRACE_TRY(ctx, hipMalloc(&d_out, n_hits * sizeof(SwgTraceOut)));
    TRACE_TRY(ctx, hipMemcpyAsync(d_query, ctx->query.data(), lq, hipMemcpyHostToDevice, ctx->stream));
    TRACE_TRY(ctx, hipMemcpyAsync(d_sub, &ctx->sub[0][0], 1024, hipMemcpyHostToDevice, ctx->stream));
    TRACE_TRY(ctx, hipMemcpyAsync(d_res, res.data(), res.size(), hipMemcpyHostToDevice, ctx->stream));
    if (ops) h_ops.resize(max_chunk * dev_stride);
    for (size_t b = 0; b < n_hits;) {
        const uint64_t budget = 2ull << 30;
        uint64_t bytes = 0;
        size_t e = b;
        while (e < n_hits) {
            const uint64_t need = (uint64_t)(lq + jobs[e].len - 1) * lq;
            if (e > b && bytes + need > budget) break;
            jobs[e].dir_off = bytes;
            bytes += need, ++e;
        }
        const size_t nb = e - b;
        TRACE_TRY(ctx, hipMemcpyAsync(d_jobs + b, jobs.data() + b, nb * sizeof(SwgTraceJob), hipMemcpyHostToDevice,
                                      ctx->stream));
        SwgTraceParams p;
        p.query = d_query, p.sub = d_sub, p.res = d_res, p.jobs = d_jobs + b, p.diag = d_diag, p.dir = d_dir;
        p.ops = d_ops, p.out = d_out + b, p.lq = (uint32_t)lq, p.ops_stride = (uint32_t)dev_stride;
        p.go = ctx->gap_open + ctx->gap_extend, p.ge = ctx->gap_extend; // src/alignment.c:58-59
        if (in_lds)
            hipLaunchKernelGGL(swg_trace_kernel<true>, dim3((unsigned)nb), dim3(SWG_TRACE_THREADS),
                               9 * (lq + 1) * sizeof(int32_t), ctx->stream, p);
        else
            hipLaunchKernelGGL(swg_trace_kernel<false>, dim3((unsigned)nb), dim3(SWG_TRACE_THREADS), 0, ctx->stream, p);
        TRACE_TRY(ctx, hipGetLastError());
        TRACE_TRY(ctx, hipMemcpyAsync(h_out.data() + b, d_out + b, nb * sizeof(SwgTraceOut), hipMemcpyDeviceToHost,
                                      ctx->stream));
        if (ops)
            TRACE_TRY(ctx, hipMemcpyAsync(h_ops.data(), d_ops, nb * dev_stride, hipMemcpyDeviceToHost, ctx->stream));
        TRACE_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t h = b; h < e; ++h) {
            const SwgTraceOut &o = h_out[h];
            swg_alignment &a = out[h];
            a.score = o.score, a.index = hits[h].index;
            a.q_begin = o.q_begin, a.q_end = o.q_end, a.d_begin = o.d_begin, a.d_end = o.d_end;
            a.n_ops = o.n_ops, a.reserved = 0;
            if (!ops) continue;
            if ((size_t)o.n_ops + 1 > ops_stride) {
                rc = swg_set_ctx_error(ctx, SWG_ERR_ARG, "swg_align_hits: ops_stride %zu too small for a path of %u steps "
                                       "(swg_align_ops_bound() is always enough)", ops_stride, o.n_ops);
                goto done;
            }
            memcpy(ops + h * ops_stride, h_ops.data() + (h - b) * dev_stride, (size_t)o.n_ops + 1);
        }
        b = e;
    }
done:
    (void)hipFree(d_query), (void)hipFree(d_sub), (void)hipFree(d_res), (void)hipFree(d_jobs);
    (void)hipFree(d_diag), (void)hipFree(d_dir), (void)hipFree(d_ops), (void)hipFree(d_out);
    return rc;
}
