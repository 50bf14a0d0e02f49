// swg_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the Smith-Waterman fill.
//
// What is computed: the three-state affine-gap local alignment recurrence of the
// reference, src/alignment.c:124-161 (H "match", A "gap_a" from the row above,
// B "gap_b" from the left, all floored at 0, score = max H), for ONE query
// against every sequence of a packed database -- i.e. the reference's whole
// timed region, src/alignment_cmdline.c:503-509, not one 16-lane call.
//
// How (nothing here follows the reference's AVX2 code).  Three families of fill kernels live in this
// file; the host planner (swg_diag_host.cpp, swg_api.cpp) picks one per search:
//
//  * The lane-group fill with a work queue (swg_diag_dyn_kernel, the default): G = 16, 32 or 64
//    lanes of a wavefront share ONE pair of database sequences, one in each int16 half of their
//    VGPRs (v_pk_* packed math).  Lane l of the group owns K query columns (3 VGPRs per column:
//    M = max(H,A,B), G = M - |go| floored, A) and runs one database row behind lane l-1, so the
//    group sweeps the DP matrix as an anti-diagonal wavefront; the right edge (M, B) of a lane's
//    strip reaches its neighbour by one DPP row_shr/wave_shr move per row, never through memory.
//    Groups take pairs (longest first) from sharded atomic counters and stream them back to back:
//    the row tokens of the next pair follow the last row of the previous one, so the diagonal
//    pipeline never drains.  Queries longer than G*K columns take several passes with the last
//    lane's edge kept in a per-pair HBM buffer (8 B per row per pair per pass).
//    swg_diag_kernel is the same engine with a fixed stream of pairs per group (option dynamic = 0).
//
//  * The same engine in 32-bit cells (swg_diag32q_kernel): one sequence per group, used when scores
//    may pass 16 bits, for re-scoring flagged sequences and when forced; with a second cell type, the exact
//    three-state recurrence, for gap scores of any sign (swg_diag32_kernel, the older bin-based form of that,
//    is left for work_queue = 0 and databases beyond 32-bit edge indices).
//
//  * The systolic fill (swg_fill_kernel): a wavefront owns 128 whole sequences of a bin and W waves
//    of a workgroup split the query's columns, edges crossing through an LDS ring.  It needs no
//    per-pair tokens and serves swg_fill_batches16 (the reference-shaped 16-lane batches) and
//    option engine = 1.
//
//  * The query profile (substitution scores of every query column against all 32 residue indices)
//    lives in LDS in [4-column chunk][32 residues][4] int16 order: a chunk is exactly one 256-byte
//    LDS bank row and a lane's read address is chunk*256 + residue*8 (rows XOR-swizzled by lane in
//    the lane-group kernels, so lanes that read the same residue still spread over the banks).
//    The per-lane gather that dominates the reference (scoring_lookup, src/alignment.c:31-44)
//    costs one ds_read_b64 per 8 cells and its address one SDWA xor per row.
//
//  * No MFMA: the recurrence is integer max/add with a loop-carried dependency.  The bound is VALU
//    issue: 10 packed instructions per 2 cells on the int16 cells, 8.5 on the packed-f16 cells that gfx950's
//    three-operand maximum allows (CellsDiag FORM 2: exact below a score of 4096, anything above flagged and run
//    again), 7.5 with two queries of a batch per lane (swg_diag_qq_kernel).
//
// int16 fast path (gap_open <= 0 and gap_extend <= 0, the normal case): with
// M = max(H,A,B) the recurrence collapses to
//     A' = max(M_up - |go|, A_up - |ge|)   B' = max(M_left - |go|, B_left - |ge|)
//     M' = max(M_diag + s, A', B')         all floored at 0
// which is value-identical to the reference's (max distributes over +const and
// go <= ge makes the extra transitions redundant; DESIGN.md gives the proof).
// Floors come for free from unsigned saturating subtracts (v_pk_sub_u16 clamp); the
// diagonal add is a signed saturating add (v_pk_add_i16 clamp), so a score that reaches
// 32767 sticks there and the sequence is flagged for a wider path (the biased "wide"
// variant to 65535, then 32-bit cells).  The reference wraps silently instead.
#include "swg_internal.h"
#include <type_traits>

// This file is compiled several times, once per PART (build.py, in parallel: one translation unit with all of its
// ~350 kernel instantiations took four minutes): every part sees all the templates and instantiates one family.
//   0  the non-template kernels, the systolic fill, the fixed-stream diagonal fill, the bin-based int32 fill and the
//      launchers that only choose among other parts' kernels
//   1  swg_diag_dyn_kernel on the int16 and wide cells     2  swg_diag_dyn_kernel on the f16 cells
//   3  swg_diag32q_kernel (reduced and exact cells)        4  swg_diag_qq_kernel
// Undefined: everything in one unit (tools/probe_isa.sh with SWG_PROBE_VARIANT).
#ifdef SWG_PART
#define SWG_HAS_PART(n) (SWG_PART == (n))
#else
#define SWG_HAS_PART(n) 1
#endif
// The (K, wave budget) instantiations of the lane-group kernels; the position in this list is the variant index.
#ifdef SWG_PROBE_VARIANT // (ISA experiments on one instantiation: tools/probe_isa.sh)
#define SWG_APPLY(X, ...) X(__VA_ARGS__)
#define SWG_DIAG_VARIANTS(X) SWG_APPLY(X, SWG_PROBE_VARIANT)
#else
#define SWG_DIAG_VARIANTS(X)                                                                                          \
    X(24, 16) X(12, 16) X(8, 16) X(16, 16) X(32, 12) X(6, 16) X(10, 16) X(20, 16) X(28, 12) X(4, 16) X(14, 16) X(18, 16) \
    X(22, 16) X(2, 16) X(23, 16) X(21, 16) X(19, 16) X(17, 16) X(15, 16) X(13, 16) X(11, 16) X(9, 16) X(7, 16) X(31, 12)  \
    X(29, 12) X(27, 12) X(25, 12) X(30, 12) X(26, 12) X(5, 16) X(3, 16)
#endif

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define DEVINL __device__ __forceinline__

DEVINL uint32_t pk_add_i16_sat(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a),
                                                                      __builtin_bit_cast(s16x2, b)));
}
DEVINL uint32_t pk_sub_u16_sat(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a),
                                                                      __builtin_bit_cast(u16x2, b)));
}
DEVINL uint32_t pk_sub_i16_sat(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a),
                                                                      __builtin_bit_cast(s16x2, b)));
}
DEVINL uint32_t pk_max_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a),
                                                                  __builtin_bit_cast(s16x2, b)));
}
// packed f16 (the three-operand-maximum cells, CellsDiag FORM 2): values are integers, exact up to 2048
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
DEVINL uint32_t pk_add_f16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b));
}
DEVINL uint32_t pk_sub_f16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) - __builtin_bit_cast(f16x2, b));
}
// v_pk_maximum3_f16 (gfx950): the compiler forms it from two nested IEEE-754-2019 maxima
DEVINL uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a),
                                                                                                    __builtin_bit_cast(f16x2, b)),
                                                                      __builtin_bit_cast(f16x2, c)));
}
// LDS access by 32-bit LDS address: the lane-group kernels form their profile addresses with one SDWA
// instruction from a per-lane base that already holds the array's LDS address (no add of the array
// symbol per read).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
DEVINL uint2 lds_read_u2(uint32_t addr)
{
    const u32x2 v = *reinterpret_cast<__attribute__((address_space(3))) const u32x2 *>((uintptr_t)addr);
    return make_uint2(v.x, v.y);
}
DEVINL int2 lds_read_i2(uint32_t addr)
{
    const i32x2 v = *reinterpret_cast<__attribute__((address_space(3))) const i32x2 *>((uintptr_t)addr);
    return make_int2(v.x, v.y);
}
// buffer resources (raw, no stride): loads beyond num_records return zero, stores beyond it are dropped
typedef __amdgpu_buffer_rsrc_t rsrc_t;
DEVINL rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
// 32-bit LDS address of a __shared__ array
#define SWG_LDS_ADDRESS(arr) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(arr))
DEVINL int imax(int a, int b) { return a > b ? a : b; }
DEVINL int imax3(int a, int b, int c) { return imax(imax(a, b), c); }

// ---------------------------------------------------------------------------
// int16 cells: two sequences per lane
// ---------------------------------------------------------------------------
template <int K> struct CellsI16 {
    typedef uint2 edge_t;         // (M, B) of a strip's last column in one row
    static constexpr int SPL = 2; // sequences per lane
    static constexpr int ESZ = 2;
    static constexpr int CHUNK = 32 * 4 * ESZ; // 256 B = one LDS bank row
    static constexpr int SLICE = (K / 4) * CHUNK;
    static constexpr int OFF_SHIFT = 0; // residue byte (index<<3) is the LDS offset

    uint32_t M[K], G[K], A[K];
    uint32_t best, mdl;

    DEVINL static edge_t zero_edge() { return make_uint2(0u, 0u); }                       // the column left of the query
    DEVINL static int score(uint32_t b, int s) { return (int)((b >> (16 * s)) & 0xFFFFu); } // sequence s of the lane

    DEVINL void reset()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) M[k] = G[k] = A[k] = 0u;
        best = 0u;
        mdl = 0u;
    }

    // One database row.  prof: this wave's LDS profile slice; off[s]: byte
    // offset of sequence s's residue row; ein = (M,B) of the column left of the
    // strip in this row; returns (M,B) of the strip's last column.
    DEVINL edge_t row(const uint8_t *prof, const uint32_t (&off)[SPL], const edge_t ein,
                      uint32_t go, uint32_t ge)
    {
        uint32_t md = mdl;                      // M[j-1][first-1]
        uint32_t gl = pk_sub_u16_sat(ein.x, go); // max(M_left - |go|, 0)
        uint32_t bl = ein.y;
#pragma unroll
        for (int c = 0; c < K / 4; ++c) {
            const uint2 wx = *reinterpret_cast<const uint2 *>(prof + off[0] + c * CHUNK);
            const uint2 wy = *reinterpret_cast<const uint2 *>(prof + off[1] + c * CHUNK);
            uint32_t s[4];
            s[0] = __builtin_amdgcn_perm(wy.x, wx.x, 0x05040100u);
            s[1] = __builtin_amdgcn_perm(wy.x, wx.x, 0x07060302u);
            s[2] = __builtin_amdgcn_perm(wy.y, wx.y, 0x05040100u);
            s[3] = __builtin_amdgcn_perm(wy.y, wx.y, 0x07060302u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * c + u;
                const uint32_t t = pk_add_i16_sat(md, s[u]); // M_diag + s, sticks at 32767
                md = M[k];
                const uint32_t a = pk_max_i16(G[k], pk_sub_u16_sat(A[k], ge));
                const uint32_t b = pk_max_i16(gl, pk_sub_u16_sat(bl, ge));
                const uint32_t m = pk_max_i16(pk_max_i16(t, a), b);
                M[k] = m;
                A[k] = a;
                gl = G[k] = pk_sub_u16_sat(m, go);
                bl = b;
                best = pk_max_i16(best, m);
            }
        }
        mdl = ein.x;
        return make_uint2(M[K - 1], bl);
    }
};

// ---------------------------------------------------------------------------
// packed-f16 cells for the systolic fill (round 4): the arithmetic of CellsDiag<K, 2> -- gfx950's three-operand maxima,
// 8.5 instead of 10 instructions per column pair, a score v held as v - 2048, exact below 4096 -- for the engine that
// short sequences run on.  Used only where no score of the search can reach 4096 (the host knows the bound: longest
// sequence x largest table entry; 372 residues under BLOSUM62), so nothing is flagged and nothing re-run.
// ---------------------------------------------------------------------------
template <int K> struct CellsSF16 {
    typedef uint2 edge_t;
    static constexpr int SPL = 2;
    static constexpr int ESZ = 2;
    static constexpr int CHUNK = 32 * 4 * ESZ;
    static constexpr int SLICE = (K / 4) * CHUNK;
    static constexpr int OFF_SHIFT = 0;
    static constexpr uint32_t Z = 0xE800E800u; // -2048.0 in both halves: score 0 (SWG_F16_ZERO)

    uint32_t M[K], G[K], A[K];
    uint32_t best, mdl, zero;

    DEVINL static edge_t zero_edge() { return make_uint2(Z, Z); }
    DEVINL static int score(uint32_t b, int s)
    {
        return (int)(float)__builtin_bit_cast(_Float16, (unsigned short)((b >> (16 * s)) & 0xFFFFu)) + 2048;
    }

    DEVINL void reset()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) M[k] = G[k] = A[k] = Z;
        best = Z;
        mdl = Z;
        zero = Z; // (the floor operand of the maxima lives in a register: a literal would be re-materialised per use)
        asm volatile("" : "+v"(zero));
    }

    DEVINL edge_t row(const uint8_t *prof, const uint32_t (&off)[SPL], const edge_t ein, uint32_t go, uint32_t ge)
    {
        uint32_t md = mdl;
        uint32_t gl = pk_sub_f16(ein.x, go); // unfloored: the maximum that uses it floors it
        uint32_t bl = ein.y;
#pragma unroll
        for (int c = 0; c < K / 4; ++c) {
            const uint2 wx = *reinterpret_cast<const uint2 *>(prof + off[0] + c * CHUNK);
            const uint2 wy = *reinterpret_cast<const uint2 *>(prof + off[1] + c * CHUNK);
            uint32_t s[4];
            s[0] = __builtin_amdgcn_perm(wy.x, wx.x, 0x05040100u);
            s[1] = __builtin_amdgcn_perm(wy.x, wx.x, 0x07060302u);
            s[2] = __builtin_amdgcn_perm(wy.y, wx.y, 0x05040100u);
            s[3] = __builtin_amdgcn_perm(wy.y, wx.y, 0x07060302u);
            uint32_t mprev = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * c + u;
                const uint32_t t = pk_add_f16(md, s[u]);
                md = M[k];
                const uint32_t a = pk_max3_f16(G[k], pk_sub_f16(A[k], ge), zero);
                const uint32_t b = pk_max3_f16(gl, pk_sub_f16(bl, ge), zero);
                const uint32_t m = pk_max3_f16(t, a, b);
                M[k] = m;
                A[k] = a;
                gl = G[k] = pk_sub_f16(m, go);
                bl = b;
                if (u & 1) best = pk_max3_f16(best, mprev, m); // the running best takes two columns at a time
                mprev = m;
            }
        }
        mdl = ein.x;
        return make_uint2(M[K - 1], bl);
    }
};

// ---------------------------------------------------------------------------
// int32 cells: one sequence per lane, the reference's recurrence term by term
// ---------------------------------------------------------------------------
template <int K> struct CellsI32 {
    typedef uint4 edge_t;         // (L = max(H,A), B, D = max(H,A,B), unused)
    static constexpr int SPL = 1;
    static constexpr int ESZ = 4;
    static constexpr int CHUNK = 32 * 4 * ESZ; // 512 B
    static constexpr int SLICE = (K / 4) * CHUNK;
    static constexpr int OFF_SHIFT = 1; // residue byte (index<<3) -> index*16

    int U[K], A[K], D[K]; // previous row: U = max(H,B), A, D = max(H,A,B)
    int best, ddl;

    DEVINL static edge_t zero_edge() { return make_uint4(0u, 0u, 0u, 0u); }

    DEVINL void reset()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) U[k] = A[k] = D[k] = 0;
        best = 0;
        ddl = 0;
    }

    DEVINL edge_t row(const uint8_t *prof, const uint32_t (&off)[SPL], const edge_t ein, int go,
                      int ge)
    {
        int dd = ddl;
        int ll = (int)ein.x, bl = (int)ein.y;
#pragma unroll
        for (int c = 0; c < K / 4; ++c) {
            const int4 sv = *reinterpret_cast<const int4 *>(prof + off[0] + c * CHUNK);
            const int s[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * c + u;
                const int h = imax(dd + s[u], 0);             // src/alignment.c:124-129
                const int a = imax3(U[k] + go, A[k] + ge, 0); // src/alignment.c:142-147
                const int b = imax3(ll + go, bl + ge, 0);     // src/alignment.c:156-161
                dd = D[k];
                U[k] = imax(h, b);
                ll = imax(h, a);
                D[k] = imax(ll, b);
                A[k] = a;
                bl = b;
                best = imax(best, h);                         // src/alignment.c:133
            }
        }
        ddl = (int)ein.z;
        return make_uint4((uint32_t)ll, (uint32_t)bl, (uint32_t)D[K - 1], 0u);
    }
};

DEVINL uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = v > o ? v : o;
    }
    return v;
}

// L1-bypassing loads of a spilled edge: it was written by another wavefront of
// this workgroup; its stores are in L2 once that wave has passed the barrier,
// but this CU's vector L1 may still hold an older copy of the line.
DEVINL uint2 load_edge_l2(const uint2 *p)
{
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
DEVINL uint4 load_edge_l2(const uint4 *p)
{
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}

// ---------------------------------------------------------------------------
// The systolic fill
// ---------------------------------------------------------------------------
#define SWG_ITEM_RING 32

// LDS map of a workgroup of W waves:
//   [W] profile slices | [(W+1)][2 parities][4 rows][64 lanes] edges | work ids
// Edge slot w is the INPUT of wave w (slot 0: zeros or the re-read spill) and
// the output of wave w-1; parity = phase & 1.
template <class Cells, int K, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void swg_fill_kernel(const SwgFillParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef typename Cells::edge_t edge_t;
    constexpr int SPL = Cells::SPL;
    constexpr int EDGE = SWG_ROWS_PER_BLK * 64; // edges of one row-block
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);

    uint8_t *prof = smem + (size_t)w * Cells::SLICE;
    edge_t *ring = reinterpret_cast<edge_t *>(smem + (size_t)W * Cells::SLICE);
    uint32_t *items = reinterpret_cast<uint32_t *>(ring + (size_t)(W + 1) * 2 * EDGE);
    if (threadIdx.x < SWG_ITEM_RING + 2) items[threadIdx.x] = 0u;
    __syncthreads();

    const uint32_t n_items = p.list_count ? (*p.list_count + 63u) / 64u : p.n_items;
    const int npass = (int)p.npass;
    edge_t *scratch =
        reinterpret_cast<edge_t *>(p.scratch + (size_t)blockIdx.x * p.scratch_wg_dwords);

    Cells cells;
    cells.reset();

    // stream position of THIS wave (every wave walks the same stream of
    // row-blocks, one phase apart)
    uint32_t seq = 0;      // work items started so far
    int pass = 0, blk = 0; // position inside the current item
    int nb = 0, nblk_real = 0;
    bool finished = false;
    int prof_pass = -1;     // pass whose profile slice is in LDS
    uint32_t next_item = 0; // wave 0: prefetched work id

    const uint32_t *rptr[SPL];
    uint32_t sid[SPL];
    uint32_t cur[SPL];
    int my_nblk = 0;

    if (w == 0) {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(p.queue, 1u);
        next_item = __builtin_amdgcn_readfirstlane(v);
    }

    for (int phase = 0;; ++phase) {
        if (phase >= w && !finished) {
            if (blk == 0 && pass == 0) {
                // ---- next work item -------------------------------------
                uint32_t item;
                if (w == 0) {
                    item = next_item;
                    if (lane == 0) items[seq % SWG_ITEM_RING] = item;
                    uint32_t v = 0;
                    if (lane == 0 && item < n_items) v = atomicAdd(p.queue, 1u);
                    next_item = __builtin_amdgcn_readfirstlane(v);
                } else {
                    item = items[seq % SWG_ITEM_RING];
                }
                item = __builtin_amdgcn_readfirstlane(item);
                if (item >= n_items) {
                    finished = true;
                    // the last wave is the last to run dry; the flag is double
                    // buffered by phase parity so a wave that is already in the
                    // next phase cannot make a slower one leave early
                    if (w == W - 1 && lane == 0) items[SWG_ITEM_RING + (phase & 1)] = 1u;
                } else {
                    if constexpr (SPL == 2) {
                        nblk_real = (int)p.bin_nblk[item];
                        my_nblk = nblk_real;
                        rptr[0] = p.residues + p.bin_off[item] + lane;
                        rptr[1] = rptr[0] + 64;
                        sid[0] = item * SWG_BIN + 2u * lane; // lane l: sorted ranks 2l, 2l+1 of the bin
                        sid[1] = sid[0] + 1u;
                    } else {
                        const uint32_t gi = item * 64u + lane;
                        uint32_t s;
                        bool valid;
                        if (p.list) {
                            valid = gi < *p.list_count;
                            s = valid ? p.list[gi] : 0u;
                        } else {
                            s = gi;
                            valid = s < p.n_bins * SWG_BIN;
                        }
                        const uint32_t b = s / SWG_BIN;
                        my_nblk = valid ? (int)p.bin_nblk[b] : 0;
                        rptr[0] = p.residues + (valid ? p.bin_off[b] : 0) + SWG_BIN_COLUMN(s % SWG_BIN);
                        sid[0] = valid ? s : 0xFFFFFFFFu;
                        nblk_real = __builtin_amdgcn_readfirstlane(
                            (int)wave_max_u32((uint32_t)my_nblk));
                    }
                    // a spilled edge is re-read W-1 phases after it was
                    // written: keep passes at least W blocks apart
                    nb = (npass > 1 && nblk_real < W) ? W : nblk_real;
                }
            }
            if (!finished) {
                if (blk == 0) {
                    cells.reset();
                    if (prof_pass != pass) {
                        // this wave's slice of the query profile -> LDS
                        const uint8_t *src = p.profile + ((size_t)pass * W + w) * Cells::SLICE;
                        for (int o = lane * 16; o < Cells::SLICE; o += 64 * 16)
                            *reinterpret_cast<uint4 *>(prof + o) =
                                *reinterpret_cast<const uint4 *>(src + o);
                        prof_pass = pass;
                    }
#pragma unroll
                    for (int s = 0; s < SPL; ++s) cur[s] = (0 < my_nblk) ? rptr[s][0] : 0u;
                }
                if (blk < nblk_real) {
                    // ---- one row-block: 4 database rows ------------------
                    uint32_t res[SPL];
#pragma unroll
                    for (int s = 0; s < SPL; ++s) {
                        res[s] = cur[s];
                        cur[s] = (blk + 1 < my_nblk) ? rptr[s][(size_t)(blk + 1) * SWG_BIN] : 0u;
                    }
                    // this lane's 4 input edges and 4 output edges
                    edge_t *ein_p = ring + ((size_t)w * 2 + ((phase + 1) & 1)) * EDGE + lane;
                    edge_t *eout_p = ring + ((size_t)(w + 1) * 2 + (phase & 1)) * EDGE + lane;
                    edge_t *sc = scratch + (size_t)blk * EDGE + lane;
                    if (w == 0) {
                        // the column left of the query: zeros in the first pass,
                        // otherwise what the last wave spilled in the previous
                        // pass.  Same lane writes and reads: no barrier needed.
#pragma unroll
                        for (int r = 0; r < SWG_ROWS_PER_BLK; ++r)
                            ein_p[r * 64] = (pass == 0) ? Cells::zero_edge() : load_edge_l2(sc + r * 64);
                    }
#pragma unroll
                    for (int r = 0; r < SWG_ROWS_PER_BLK; ++r) {
                        uint32_t off[SPL];
#pragma unroll
                        for (int s = 0; s < SPL; ++s)
                            off[s] = ((res[s] >> (8 * r)) & 0xFFu) << Cells::OFF_SHIFT;
                        eout_p[r * 64] = cells.row(prof, off, ein_p[r * 64], p.go, p.ge);
                    }
                    if (w == W - 1 && pass + 1 < npass) {
                        // spill the query-side edge for the next pass; the stores
                        // must have reached L2 before this wave signals the
                        // barrier (hipcc's __syncthreads only drains lgkmcnt here)
#pragma unroll
                        for (int r = 0; r < SWG_ROWS_PER_BLK; ++r) sc[r * 64] = eout_p[r * 64];
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                ++blk;
                if (blk == nb) {
                    // ---- end of this (item, pass): publish the maxima ------
                    if constexpr (SPL == 2) {
                        atomicMax(p.scores + sid[0], Cells::score(cells.best, 0));
                        atomicMax(p.scores + sid[1], Cells::score(cells.best, 1));
                    } else {
                        if (sid[0] != 0xFFFFFFFFu) atomicMax(p.scores + sid[0], cells.best);
                    }
                    blk = 0;
                    if (++pass == npass) {
                        pass = 0;
                        ++seq;
                    }
                }
            }
        }
        __syncthreads();
        if (items[SWG_ITEM_RING + (phase & 1)] != 0u) break;
    }
}

// ---------------------------------------------------------------------------
// The diagonal fill: G lanes share one pair of sequences
// ---------------------------------------------------------------------------
// Same packed recurrence, finer decomposition.  A group of G lanes (16, 32 or 64)
// owns ONE pair of database sequences; lane g of the group holds K query columns
// [g*K, (g+1)*K) and works on database row (t - g) at step t, so the group sweeps
// the DP matrix as an anti-diagonal wavefront.  What crosses a lane boundary per
// step -- the strip's right edge (M, B) and the row's token -- moves with DPP lane
// shifts (row_shr:1 inside a 16-lane row, wave_shr:1 across the wavefront);
// nothing goes through LDS or memory.
//
// Why it exists next to the systolic kernel: there a lane walks K=32 columns of
// 128 lock-stepped sequences, so a bin of long sequences is a serial chain of
// rows*K*11 instructions on one CU while the rest of the chip drains; here the
// chain is rows*(K/G-th of the query), the unit of scheduling is a pair of
// sequences instead of 128, and sequences stream through a lane group back to
// back (two all-padding "reset" rows between pairs clear the carried state), so
// there is no pipeline fill per sequence and no padding to a bin's longest
// member.
//
// Token per database row, 32 bits: byte 0 = X residue (index<<3), byte 1 = Y residue
// (index<<3), bit 16 = reset row, bit 17 = last row of the pair.  The residue bytes are the
// LDS byte offsets of their profile rows as they stand (one SDWA add each forms the address),
// a row with a flag is any token above 0xFFFF (one compare), and a 4-row block is one uint4
// whose components are the rows (nothing to extract).  Costs 4 bytes per pair-row of HBM
// instead of 2 -- irrelevant at 0.2 % of the HBM roofline -- and saves five VALU instructions
// per lane-row.
#define DPP_ROW_SHR1 0x111
#define DPP_WAVE_SHR1 0x138

// The cells come in three forms of the same recurrence (FORM):
//  0  packed int16, unsigned saturating subtracts do the floors: scores to 32767 (a score that reaches it sticks
//     there and the sequence is flagged), 10 VALU instructions per 2 cells.
//  1  "wide": the same on values biased by -32768 (a score v is held as v - 32768), which doubles the range to
//     65535 at the same instruction count: the signed saturating subtract floors at -32768 = score 0, the signed
//     saturating add sticks at 32767 = score 65535, signed max keeps the order.  Used when the query is long
//     enough for a score to pass 32767 (the reference's int16 lanes wrap there).  Reset rows cannot
//     use the all-ones gap trick in this form and wipe the state explicitly.
//  2  packed f16 with gfx950's three-operand maximum (v_pk_maximum3_f16): the floors are the third operand of
//     the maxima that are needed anyway, M = max3(t, a, b) is one instruction instead of two and the running
//     best takes two columns per instruction: 8.5 instructions per 2 cells.  Every value is an integer, and f16
//     holds the integers of [-2048, 2048] exactly: a score v is held as v - 2048 (as the wide form biases by
//     32768), so the cells are exact for scores 0 .. 4096.  Sums round monotonically beyond that range: a sum
//     whose true value is above 2048 comes out >= 2048, one below -2048 comes out <= -2048 and is floored to
//     -2048 (= score 0) like its true value.  So a pair whose computed best stays below 2048 (score 4096) has
//     had every cell computed exactly; a pair that reaches it is flagged and re-scored.  G is kept
//     unfloored (the max3 floors it where it is used).  Reset rows: gap magnitudes of 65504, the largest finite
//     f16, clear every state below 32768 in two rows; a lane whose best got beyond that (a score far above the
//     flag level in the making) is wiped explicitly.  The profile holds f16 bit patterns, -65504 for padding; no
//     NaN can arise (-inf exists only as a transient diagonal sum that the floor replaces, +inf meets only finite
//     values).
#define SWG_F16_BIG 0x7BFF7BFFu  // 65504 in both halves
#define SWG_F16_ZERO 0xE800E800u // -2048.0 in both halves: score 0
#define SWG_F16_FLAG 0x6800u     // +2048.0 = score 4096: a pair that reaches it is flagged
#define SWG_F16_CEILING 4096
template <int K, int FORM = 0> struct CellsDiag {
    static constexpr bool WIDE = FORM == 1, F16 = FORM == 2;
    static constexpr uint32_t ZERO = WIDE ? 0x80008000u : F16 ? SWG_F16_ZERO : 0u;
    DEVINL static uint32_t sub(uint32_t a, uint32_t b) { return F16 ? pk_sub_f16(a, b) : WIDE ? pk_sub_i16_sat(a, b) : pk_sub_u16_sat(a, b); }
    // A profile chunk is [32 residues][4 columns] int16 = 256 bytes; a lane's slice of the profile
    // is KP = K rounded up to whole chunks, the columns it works on are the first K (any K: G*K
    // lands within G/2 columns of the query length on average instead of 2*G).
    static constexpr int CH = 4;
    static constexpr int KP = (K + CH - 1) / CH * CH;
    static constexpr int CHUNK = 32 * CH * 2;
    uint32_t M[K], G[K], A[K];
    uint32_t best, mdl;

    DEVINL void reset()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) M[k] = G[k] = A[k] = ZERO;
        best = ZERO;
        mdl = ZERO;
    }

    // lanes with fm = all ones forget everything (WIDE form of a reset row; F16: a lane that reached +inf)
    DEVINL void wipe(uint32_t fm)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            M[k] = (M[k] & ~fm) | (ZERO & fm);
            G[k] = (G[k] & ~fm) | (ZERO & fm);
            A[k] = (A[k] & ~fm) | (ZERO & fm);
        }
        best = (best & ~fm) | (ZERO & fm);
        mdl = (mdl & ~fm) | (ZERO & fm);
    }

    // F16: all ones in lanes whose running best is 32768 or more (+inf included) in either half.  Two reset rows
    // clear what a pair leaves behind only up to 65504 - 2048 (a row subtracts 65504, the floor is -2048): a lane
    // with more than that in it is wiped by hand, long before.
    DEVINL uint32_t best_is_huge() const
    {
        return 0u - (uint32_t)((((best & 0x78007800u) + 0x08000800u) & 0x80008000u) != 0u);
    }

    // as CellsI16::row, with per-lane gap magnitudes (all ones on reset rows).  ax / ay: 32-bit LDS
    // address of the row's residue in this lane's first chunk, for sequence X / Y.  FENCED: software
    // pipeline with a depth of one chunk -- chunk c+1's profile reads are issued before chunk c's
    // arithmetic and nothing moves across the chunk boundary, so at most two chunks' words (8
    // registers) are in flight.  Left alone the scheduler hoists eight reads and K=24 spills.
    // zero: F16 only: score 0 (SWG_F16_ZERO) in a register the caller keeps (a literal would be re-materialised per use)
    template <bool FENCED = false>
    DEVINL uint2 row(uint32_t ax, uint32_t ay, uint32_t em, uint32_t eb, uint32_t go, uint32_t ge, uint32_t zero = 0u)
    {
        constexpr int NCH = KP / CH;
        uint32_t md = mdl;
        uint32_t gl = sub(em, go);
        uint32_t bl = eb;
        uint2 nx = lds_read_u2(ax), ny = lds_read_u2(ay);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint2 wx = nx, wy = ny;
            if (c + 1 < NCH) {
                nx = lds_read_u2(ax + (c + 1) * CHUNK);
                ny = lds_read_u2(ay + (c + 1) * CHUNK);
            }
            uint32_t s[CH];
            s[0] = __builtin_amdgcn_perm(wy.x, wx.x, 0x05040100u);
            s[1] = __builtin_amdgcn_perm(wy.x, wx.x, 0x07060302u);
            s[2] = __builtin_amdgcn_perm(wy.y, wx.y, 0x05040100u);
            s[3] = __builtin_amdgcn_perm(wy.y, wx.y, 0x07060302u);
            uint32_t mprev = 0u;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int k = CH * c + u;
                if (k >= K) break; // unused tail of the last chunk
                if (F16) {
                    const uint32_t t = pk_add_f16(md, s[u]);
                    md = M[k];
                    const uint32_t a = pk_max3_f16(G[k], pk_sub_f16(A[k], ge), zero);
                    const uint32_t b = pk_max3_f16(gl, pk_sub_f16(bl, ge), zero);
                    const uint32_t m = pk_max3_f16(t, a, b);
                    M[k] = m;
                    A[k] = a;
                    gl = G[k] = pk_sub_f16(m, go);
                    bl = b;
                    // the running best takes two columns at a time (the last one of an odd K alone)
                    if (u & 1) best = pk_max3_f16(best, mprev, m);
                    else if (k == K - 1) best = pk_max3_f16(best, m, m);
                    mprev = m;
                } else {
                    const uint32_t t = pk_add_i16_sat(md, s[u]);
                    md = M[k];
                    const uint32_t a = pk_max_i16(G[k], sub(A[k], ge));
                    const uint32_t b = pk_max_i16(gl, sub(bl, ge));
                    const uint32_t m = pk_max_i16(pk_max_i16(t, a), b);
                    M[k] = m;
                    A[k] = a;
                    gl = G[k] = sub(m, go);
                    bl = b;
                    best = pk_max_i16(best, m);
                }
            }
            if (FENCED && c + 1 < NCH) __builtin_amdgcn_sched_barrier(0);
        }
        mdl = em;
        return make_uint2(M[K - 1], bl);
    }
};

// The LDS maximum of the f16 bests is taken on order-preserving 16-bit keys: a non-negative value's bits with the
// sign bit set, a negative value's bits complemented (so 0 = below everything: an untouched slot).
DEVINL uint32_t f16_key(uint32_t bits) { return (bits & 0x8000u) ? (~bits & 0xFFFFu) : (bits | 0x8000u); }
// integer score of such a key: value + 2048, flagged (SWG_F16_CEILING) from +2048.0 up, +inf included
DEVINL int f16_score(uint32_t key)
{
    if (key >= (0x8000u | SWG_F16_FLAG)) return SWG_F16_CEILING;
    const uint32_t bits = (key & 0x8000u) ? (key & 0x7FFFu) : (~key & 0xFFFFu);
    return (int)(float)__builtin_bit_cast(_Float16, (unsigned short)bits) + 2048;
}

template <int CTRL> DEVINL uint32_t dpp_zero(uint32_t src)
{
    // lanes without a source lane get 0 (bound_ctrl): no v_mov to set up an old value
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, 0xf, 0xf, true);
}

template <int CTRL> DEVINL uint32_t dpp_keep(uint32_t keep, uint32_t src)
{
    // lanes without a source lane keep `keep`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)keep, (int)src, CTRL, 0xf, 0xf, false);
}

// row r of a 4-row token block (r is a constant after unrolling)
DEVINL uint32_t block_row(const uint4 &b, int r) { return r == 0 ? b.x : r == 1 ? b.y : r == 2 ? b.z : b.w; }

// A wave-uniform integer the optimizer cannot see through: a test on it stays one scalar compare
// per use.  (A uniform bool that lives across the row loop is kept as a lane mask and turned back
// into a condition with v_cndmask + v_cmp on every row.)
DEVINL int opaque_uniform(int v)
{
    asm volatile("" : "+s"(v));
    return v;
}

// LDS address of a residue's row in a lane's first profile chunk.  base: the lane's slice (multiple of 256)
// with, when the profile is swizzled, (g & 31) << 3 in its low byte; off: the token's residue byte.
// BYTE: which byte of the token (0: sequence X, 1: sequence Y).  One SDWA instruction either way; the XOR is
// written out because the compiler's SDWA peephole only finds the add (it makes v_and + v_bfe + 2 v_xad of the XOR).
template <int BYTE> DEVINL uint32_t prof_addr(uint32_t base, uint32_t tok)
{
#if SWG_LDS_SWIZZLE
    uint32_t a;
    if (BYTE == 0)
        asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(base), "v"(tok));
    else
        asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(base), "v"(tok));
    return a;
#else
    return base + ((tok >> (8 * BYTE)) & 0xFFu);
#endif
}
// slice_base: LDS address of the lane's slice, a multiple of 256 (the arrays are declared aligned(256))
DEVINL uint32_t prof_base(uint32_t slice_base, int g) { return SWG_LDS_SWIZZLE ? (slice_base | (((uint32_t)g & 31u) << 3)) : slice_base; }

template <int K, int MAXW, bool MULTIPASS, bool WIDE = false>
__global__ __launch_bounds__(MAXW * 64) void swg_diag_kernel(const SwgDiagParams p)
{
    extern __shared__ __attribute__((aligned(256))) uint8_t smem[]; // query profile of this pass
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);
    const int G = (int)p.G;
    const int g = lane & (G - 1);
    const bool leader = g == 0, tail = g == G - 1;
    const uint32_t stream = (uint32_t)((blockIdx.x * W + w) * (64 / G) + lane / G);
    uint64_t boff = 0;
    uint32_t nblk = 0, pair0 = 0;
    if (stream < p.n_streams) {
        boff = p.stream_off[stream];
        nblk = (uint32_t)(p.stream_off[stream + 1] - boff);
        pair0 = p.stream_pair_off[stream];
    }
    const uint32_t wave_nblk = __builtin_amdgcn_readfirstlane(wave_max_u32(nblk));
    const uint32_t nsteps = wave_nblk * 4u + (uint32_t)G; // >= rows + G - 1, multiple of 4
    const uint64_t t_start = p.trace ? wall_clock64() : 0ull;
    if (wave_nblk >= p.prio_blocks) __builtin_amdgcn_s_setprio(3);
    const uint32_t rows = nblk * 4u;
    const uint4 *tp = p.tok + boff;
    uint2 *sp = p.scratch + boff * 4u;
    const uint32_t base = prof_base(SWG_LDS_ADDRESS(smem) + (uint32_t)g * (CellsDiag<K>::KP / 4) * CellsDiag<K>::CHUNK, g);
    const uint32_t slice = (uint32_t)G * CellsDiag<K>::KP * 64u;
    const int npass = MULTIPASS ? (int)p.npass : 1;

    for (int pass = 0; pass < npass; ++pass) {
        if (MULTIPASS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // spills of the previous pass
            __syncthreads();
        }
        {
            const uint8_t *src = p.profile + (size_t)pass * slice;
            for (uint32_t o = threadIdx.x * 16u; o < slice; o += blockDim.x * 16u)
                *reinterpret_cast<uint4 *>(smem + o) = *reinterpret_cast<const uint4 *>(src + o);
        }
        __syncthreads();

        constexpr uint32_t Z = CellsDiag<K, WIDE>::ZERO; // score 0 in the cells' representation
        const uint4 none = make_uint4(0u, 0u, 0u, 0u);
        CellsDiag<K, WIDE> cells;
        cells.reset();
        uint32_t tok = 0u, m_out = Z, b_out = Z, c_out = Z, done = 0u;
        uint4 cur = (leader && nblk > 0u) ? tp[0] : none;
        uint4 nxt = (leader && nblk > 1u) ? tp[1] : none;
        uint2 spc[4], spn[4];
        if (MULTIPASS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                spc[r] = (leader && pass > 0 && (uint32_t)r < rows) ? load_edge_l2(sp + r) : make_uint2(Z, Z);
                spn[r] = (leader && pass > 0 && 4u + r < rows) ? load_edge_l2(sp + 4 + r) : make_uint2(Z, Z);
            }
        }

        for (uint32_t s4 = 0; s4 < nsteps; s4 += 4u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t fresh = block_row(cur, r);
                const uint32_t lm = MULTIPASS ? spc[r].x : Z, lb = MULTIPASS ? spc[r].y : Z;
                uint32_t em, eb, cin;
                if (G == 16) {
                    tok = dpp_keep<DPP_ROW_SHR1>(fresh, tok);
                    em = MULTIPASS ? dpp_keep<DPP_ROW_SHR1>(lm, m_out) : dpp_zero<DPP_ROW_SHR1>(m_out);
                    eb = MULTIPASS ? dpp_keep<DPP_ROW_SHR1>(lb, b_out) : dpp_zero<DPP_ROW_SHR1>(b_out);
                    cin = WIDE ? dpp_keep<DPP_ROW_SHR1>(Z, c_out) : dpp_zero<DPP_ROW_SHR1>(c_out);
                } else {
                    const uint32_t t0 = dpp_keep<DPP_WAVE_SHR1>(fresh, tok);
                    const uint32_t t1 = MULTIPASS ? dpp_keep<DPP_WAVE_SHR1>(lm, m_out) : dpp_zero<DPP_WAVE_SHR1>(m_out);
                    const uint32_t t2 = MULTIPASS ? dpp_keep<DPP_WAVE_SHR1>(lb, b_out) : dpp_zero<DPP_WAVE_SHR1>(b_out);
                    const uint32_t t3 = WIDE ? dpp_keep<DPP_WAVE_SHR1>(Z, c_out) : dpp_zero<DPP_WAVE_SHR1>(c_out);
                    if (G == 32) { // lane 32 starts a group too
                        tok = leader ? fresh : t0;
                        em = leader ? lm : t1;
                        eb = leader ? lb : t2;
                        cin = leader ? Z : t3;
                    } else {
                        tok = t0;
                        em = t1;
                        eb = t2;
                        cin = t3;
                    }
                }
                // reset / last rows are rare: one wave-uniform test keeps their bookkeeping out of
                // the common step (the recurrence itself is issued once, with per-lane gap operands)
                const bool special = __builtin_amdgcn_ballot_w64(tok > 0xFFFFu) != 0ull;
                uint32_t go_t = p.go, ge_t = p.ge;
                if (special) {
                    // reset rows: gap magnitudes of all ones wipe A/G/B, two such rows wipe M
                    const uint32_t fm = 0u - ((tok >> 16) & 1u);
                    if (WIDE) {
                        cells.wipe(fm);
                    } else {
                        cells.best &= ~fm;
                        go_t |= fm;
                        ge_t |= fm;
                    }
                }
                const uint2 e = cells.template row<(K > 16)>(prof_addr<0>(base, tok), prof_addr<1>(base, tok), em,
                                                              eb, go_t, ge_t);
                c_out = pk_max_i16(cin, cells.best);
                if (special && tail && (tok & SWG_TOK_LAST)) {
                    const uint32_t pr = p.stream_pairs[pair0 + done];
                    atomicMax(p.scores + 2u * pr, (int)((c_out ^ Z) & 0xFFFFu));
                    atomicMax(p.scores + 2u * pr + 1u, (int)((c_out ^ Z) >> 16));
                    ++done;
                }
                m_out = e.x;
                b_out = e.y;
                if (MULTIPASS && tail) {
                    const uint32_t row = s4 + (uint32_t)r - (uint32_t)(G - 1);
                    if (pass + 1 < npass && row < rows) sp[row] = e; // row wraps negative -> huge
                }
            }
            const uint32_t bi = s4 / 4u + 2u;
            cur = nxt;
            nxt = (leader && bi < nblk) ? tp[bi] : none;
            if (MULTIPASS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    spc[r] = spn[r];
                    const uint32_t row = (bi)*4u + (uint32_t)r;
                    spn[r] = (leader && pass > 0 && row < rows) ? load_edge_l2(sp + row) : make_uint2(Z, Z);
                }
            }
        }
    }
    if (p.trace && lane == 0) {
        uint64_t *t = p.trace + (size_t)(blockIdx.x * W + w) * 4u;
        t[0] = t_start;
        t[1] = wall_clock64();
        t[2] = wave_nblk;
        t[3] = 0ull;
    }
}

// ---------------------------------------------------------------------------
// The diagonal fill with a work queue (one launch per pass of the query)
// ---------------------------------------------------------------------------
// Same lanes, same step, but a lane group is not handed a fixed stream of pairs:
// its leader lane takes the next pair off a device-wide counter when the one it
// is feeding runs out (pairs are stored longest first, so the queue is the
// longest-processing-time-first order).  Why: the wavefronts of a SIMD do not
// advance at the same rate (the issue arbiter favours the oldest), so equal
// static shares finish at very different times and the SIMDs spend the last
// fifth of a search with one or two wavefronts -- too few to saturate the
// issue port.  With the queue every resident wavefront stays busy until the
// pairs are gone, whatever rate it ran at.
//
// A lane group's bookkeeping lives in LDS (128 dwords per group, after the
// profile) and is only touched when one of the wavefront's pairs begins or ends; between
// such events a block costs a scalar compare and the leader's token load.  The
// register file holds the DP state and one token index.  When a pair runs out its
// leader takes the next one there and then (counter, then the pair's token range: two
// round trips, at raised priority so that a wavefront whose turn it is to yield does not
// crawl through the bookkeeping); that stalls one wavefront for a few microseconds
// about once per hundred blocks while the others keep the issue port busy.
//
// A pair's two scores: every lane keeps the maximum of its own columns over the pair's rows and,
// on the pair's last row, joins it to the pair's slot of the record with two LDS atomic maxima;
// the tail lane -- the last of the group to see that row -- then reads the slot, writes the two
// scores and clears it.  (Carrying a running maximum from lane to lane instead costs a DPP move
// and a packed max on EVERY row.)  Pair ids reach the tail lane through a ring in the same record;
// a lane finds both by counting the last rows it has seen.  A launch may name a second range of
// pairs to go on with when its own is empty: the long class ends early and then helps with the bulk.
//
// LDS record of a lane group: [0] value of the wavefront's block counter at which
// the current pair has no more tokens to load (NONE: no pair and none to come),
// [1] flags, [2] ids pushed, [3], [4], [6], [8..16] the group's batch of short pairs (SWG_DYN_MAX_BATCH),
// [32..63] ring of pair ids ([64..127] held a ring of score pairs until round 4: the scores travel in registers now).
#ifndef SWG_DYN_FENCE_ABOVE
#define SWG_DYN_FENCE_ABOVE 16 // fence the profile prefetch for K above this
#endif
#define SWG_DYN_STATE 128u
// Pairs between the leader (which is up to two token blocks ahead) and the tail lane: at most
// (2*4 + 63) rows / 4 rows per shortest pair = 18 with 64 lanes per pair.
#define SWG_DYN_RING 32u
#define SWG_DYN_MAXES 64u
// Pairs one queue request may claim (swg_diag_dyn_kernel, batch_B): the group's record keeps their block offsets
// in [8 .. 8 + SWG_DYN_MAX_BATCH], the pairs left in [3], the next id in [4], the batch's first id in [6].
#ifndef SWG_DYN_MAX_BATCH
#define SWG_DYN_MAX_BATCH 8u
#endif
#ifndef SWG_DYN_TURN_SHIFT
#define SWG_DYN_TURN_SHIFT 14 // a turn lasts 2^14 ticks of the 100 MHz clock (164 us): long against a block even for the wavefront whose turn it is to yield
#endif
#ifndef SWG_DYN_TURN_EVERY
#define SWG_DYN_TURN_EVERY 1u // the clock is looked at every block (a stale priority ties with a fresh one, and ties go to the oldest)
#endif
#define SWG_DYN_HOT 4u     // current pair is long: raised priority
#define SWG_DYN_SECOND 8u  // on the second range
#define SWG_DYN_IDLING 16u // EDGES: out of pairs, the block with the idle row is on its way
#define SWG_DYN_KEEP 0xFFFFFFFEu
// EDGES: byte offsets of lanes with nothing to load or store.  With at most SWG_DYN_SEG_BLOCKS < 2^26
// blocks per launch the token resource is < 2^30 bytes and the edge resources < 2^31; a parked offset
// moves like the others (16 bytes per block, the tail's 32) for at most that many blocks, so it stays
// in [2^30, 2^31) -- twice that in [2^31, 2^32) -- and the tail's in [2^31, 2^32): out of bounds
// wherever they are used.
#define SWG_OFF_PARKED 0x40000000u
#define SWG_OFF_PARKED_TAIL 0x80000000u
#define SWG_DYN_NONE 0xFFFFFFFFu
// flags bits 8..: shards of the current range found empty so far

// quad_perm DPP controls: every lane of a quad reads the quad's lane Q
#define DPP_QUAD(Q) ((Q) | ((Q) << 2) | ((Q) << 4) | ((Q) << 6))
// value of lane r of the caller's quad (r is a constant after unrolling)
DEVINL uint32_t quad_bcast(uint32_t x, int r)
{
    switch (r) {
    case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, DPP_QUAD(0), 0xf, 0xf, true);
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, DPP_QUAD(1), 0xf, 0xf, true);
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, DPP_QUAD(2), 0xf, 0xf, true);
    default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, DPP_QUAD(3), 0xf, 0xf, true);
    }
}

// EDGES: the launch is one pass of a query longer than G*K columns.  The left edge of every row
// comes from the previous pass's launch (edge_in, null in the first pass) and the right edge goes
// to the next one (edge_out, null in the last), both indexed by the row's position in the pair-major
// token order, so any lane group can take any pair in any pass; the kernel boundary is the
// synchronisation.  Scores are the maximum over the passes.  Nothing of it is paid per row beyond two
// quad broadcasts and the store itself, and it takes few registers (at K=32 there are none to spare):
// tokens and edges are addressed through buffer resources by ONE 32-bit byte offset per lane.  In
// lanes 0..3 of a group it is the token offset of row g of the block being loaded -- the leader's own
// tokens; twice that is the offset of the row's left edge, which each of the four lanes fetches for
// the leader to pick up with a quad broadcast --; in the tail lane it is where the right edges of its
// current block go.  Offsets outside a resource read as zero and store nothing, which is what lanes
// with nothing to do want: they sit in a range that is out of bounds for all three resources and stays
// so for as many blocks as a launch can have (hence SWG_DYN_SEG_BLOCKS).  The tail lane sets its offset
// when a pair's first reset row reaches it (always at unrolled row 3: the tail is G-1 = 3 mod 4 rows
// behind the leader) from the pair ring, and parks it when the row flagged IDLE -- sent once by a
// leader that finds the queue empty -- reaches it.
// FORM: the cells (see CellsDiag): 0 packed int16, 1 wide (scores to 65535; needs EDGES: a query that can pass
// 32767 is long), 2 packed f16 with three-operand maxima (scores below 4096, anything above is flagged).
template <int K, int MAXW, bool EDGES = false, int FORM = 0>
__global__ __launch_bounds__(MAXW * 64) void swg_diag_dyn_kernel(const SwgDiagDynParams p)
{
    constexpr bool WIDE = FORM == 1, F16 = FORM == 2;
    static_assert(EDGES || !WIDE, "the wide form is instantiated with edges only");
    extern __shared__ __attribute__((aligned(256))) uint8_t smem[]; // query profile, then the group records
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);
    const int G = (int)p.G;
    const int gshift = G == 16 ? 4 : G == 32 ? 5 : 6;
    const int g = lane & (G - 1);
    const bool leader = g == 0, tail = g == G - 1;
    constexpr uint32_t Z = CellsDiag<K, FORM>::ZERO; // score 0 in the cells' representation
    const uint32_t base = prof_base(SWG_LDS_ADDRESS(smem) + (uint32_t)g * (CellsDiag<K>::KP / 4) * CellsDiag<K>::CHUNK, g);
    const uint32_t slice = (uint32_t)G * CellsDiag<K>::KP * 64u;
    // recomputed where it is needed (rarely) instead of living in a register
    auto record = [&]() -> uint32_t * {
        const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        return reinterpret_cast<uint32_t *>(smem + slice) + (((uint32_t)w << (6 - gshift)) + (l >> gshift)) * SWG_DYN_STATE;
    };
    // List mode (the re-run of the pairs the f16 cells flagged): the queue hands out positions of a device-side
    // list of pair ids instead of the ids themselves; [q_begin, q_end) then only says which of them belong to this
    // launch (a segment of a very large database).  An empty list -- the usual case -- ends the launch here.
    const uint32_t n_list = p.list ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*p.list_count) : 0u;
    if (p.list && n_list == 0u) return;
    const uint64_t t_start = p.trace ? wall_clock64() : 0ull;
    if (p.stamps && lane == 0) atomicMax(p.stamps, ~(unsigned long long)wall_clock64()); // earliest start, as the maximum of the complements (0 = not run)
    // several queries in one launch: row y of the grid works for query y (its profile, its queue, its scores)
    const uint8_t *profile = p.profile + (size_t)blockIdx.y * p.profile_stride;
    uint32_t *const queue = p.queue + (size_t)blockIdx.y * p.queue_stride;
    int32_t *const scores = p.scores + (size_t)blockIdx.y * p.score_stride;
    for (uint32_t o = threadIdx.x * 16u; o < slice; o += blockDim.x * 16u)
        *reinterpret_cast<uint4 *>(smem + o) = *reinterpret_cast<const uint4 *>(profile + o);
    {
        uint32_t *st = record();
        if (g < 4) st[g] = 0u; // every group is due at block 0
    }
    __syncthreads();
    // List mode: the host sizes the workgroup for a long list (the count is on the device); of a shorter one's only
    // the wavefronts the list can keep busy stay, in fours -- one more per SIMD.  The queue gives a pair to whoever
    // asks first, and the sixteen wavefronts of the first workgroups would each run theirs at a quarter of the speed
    // that four have.  (The one barrier of the kernel is behind them.)
    if (p.list) {
        const uint32_t per_wave = 64u >> gshift;
        const uint32_t want = (n_list + gridDim.x * per_wave - 1u) / (gridDim.x * per_wave);
        if ((uint32_t)w >= max(4u, (want + 3u) & ~3u)) return;
    }

    CellsDiag<K, FORM> cells;
    cells.reset();
    uint32_t tok = 0u, m_out = Z, b_out = Z;
    // F16 without edges: score 0 is not a zero bit pattern, so a lane without a DPP source (the leader) cannot be
    // zero-filled.  The incoming edge lives in registers that are only ever written by the DPP move itself: the
    // leader keeps the score-0 pattern they start with, without a v_mov per row (two of each: a row's M edge is
    // still wanted as the next row's diagonal when the next edge arrives).
    uint32_t em_rot[2] = {Z, Z}, eb_rot[2] = {Z, Z};
    uint32_t zero_v = Z; // (F16: the floor operand of the maxima, kept in a register)
    asm volatile("" : "+v"(zero_v));
    uint32_t go_v = p.go, ge_v = p.ge; // per-lane gap magnitudes: all ones while the lane is on a reset row
    uint32_t nlast = 0u;               // tail lane: last rows it has seen = position of its pair in the group's ring of ids
    uint32_t bc = Z;                   // the pair's best on its way along the last row (see the flagged-row branch)
    // Tokens: T0..T3 are the rows of the block being worked on.  Each is re-loaded with the same row of
    // the NEXT block right after its use (one block of time for the load to land), from the per-lane
    // pointer tp: the leader's runs through its pair's blocks, everybody else's -- and an idle leader's --
    // stays on a block of zeros (padding rows).  No register copies between blocks, nothing to clear.
    uint32_t T0 = 0u, T1 = 0u, T2 = 0u, T3 = 0u;
    const uint32_t *const zero_blk = reinterpret_cast<const uint32_t *>(p.tok + p.zero_block);
    const uint32_t *tp = zero_blk;
    uint32_t tstep = 0u;         // dwords tp advances per block: 4 while the leader feeds a pair
    // EDGES: left edges of the current / next block (lanes 0..3 of a group, one row each), and the lane's
    // byte offset into the segment's tokens (lanes 0..3; times two: into the incoming edges) or into
    // the outgoing edges (tail lane)
    uint2 ec = make_uint2(Z, Z), en = make_uint2(Z, Z);
    uint32_t off = tail ? SWG_OFF_PARKED_TAIL : SWG_OFF_PARKED;
    const uint64_t tails = __builtin_amdgcn_ballot_w64(tail);
    const rsrc_t tokR = make_rsrc(p.tok + p.seg_origin, EDGES ? p.seg_blocks * 16u : 0u);
    const rsrc_t einR = make_rsrc(p.edge_in + (size_t)p.seg_origin * 4u, EDGES && p.edge_in ? p.seg_blocks * 32u : 0u);
    const rsrc_t eoutR = make_rsrc(p.edge_out + (size_t)p.seg_origin * 4u, EDGES && p.edge_out ? p.seg_blocks * 32u : 0u);
    // wave-uniform: block counter, the count at which the next pair runs out (none: all leaders idle)
    uint32_t blocks = 0u, next_event = 0u, drain = 0u, events = 0u;
    uint64_t event_ticks = 0ull; // diagnostics
    bool hot = false;
    // rank of this wavefront among the launch's wavefronts on its SIMD (0, 1, 2 ..): one counter per
    // physical SIMD, found through HW_REG_HW_ID (id 4: simd 5:4, cu 11:8, sh 12, se 15:13) and
    // HW_REG_XCC_ID (id 20, bits 3:0).  The wave slot (HW_ID 3:0) would not do: another launch's
    // wavefronts sit between ours, so our slots are not consecutive.
    uint32_t rank;
    {
        const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((11 << 11) | (4 << 6) | 4);  // bits 15:4
        const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20); // bits 3:0
        const uint32_t simd = (((hw >> 4) << 2) | (hw & 3u) | (xcc << 10)) & (SWG_DYN_SIMD_SLOTS - 1u);
        uint32_t r = 0u;
        if (lane == 0) r = atomicAdd(p.simd_ranks + simd, 1u);
        rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    }
    // The arbiter serves the highest priority first and among equals the oldest wavefront, and
    // these wavefronts live as long as the kernel: left alone, the youngest of a SIMD would crawl
    // and the pair it holds would end long after the queue is empty.  Everyone not on the critical
    // path therefore takes turns at the lower priorities.  The turn comes from the wall clock (the
    // wavefronts of a SIMD see the same phase) plus the rank, so that at any time they hold
    // different priorities: rotating on a private counter leaves ties, and ties go to the oldest.
    auto take_turn = [&]() {
        // (3 or 4 levels: a remainder by a run-time value would be a float-reciprocal sequence on the
        // VALU every block; by the constant 3 it is one scalar multiply-high)
        const uint32_t x = ((uint32_t)(wall_clock64() >> SWG_DYN_TURN_SHIFT) + rank) & 0xFFFFu;
        const uint32_t turn = p.turn_levels == 4u ? (x & 3u) : x - 3u * ((x * 0xAAABu) >> 17);
        if (turn == 0u) __builtin_amdgcn_s_setprio(0);
        else if (turn == 1u) __builtin_amdgcn_s_setprio(1);
        else if (turn == 2u) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    };

    // The main loop exists three times, for groups of 16 lanes (edges handed over by row_shr), of 32 (wave_shr and a
    // select at lane 32) and of 64 (wave_shr alone): a run-time test of G inside the row makes the register allocator's
    // job harder than it is, and a 64-lane group -- the long class, config 5's passes -- paid for the 32-lane select
    // (three v_cndmask per row) without needing it.
    auto main_loop = [&](auto gw_tag) {
    constexpr int GW = decltype(gw_tag)::value; // lanes per group, a constant inside the loop
    constexpr bool G16 = GW == 16;
    for (;;) {
        if (blocks == next_event) {
            // some pair has run out (or this is the start): its leader takes the next one
            __builtin_amdgcn_s_setprio(3);
            const uint64_t ev0 = p.trace ? wall_clock64() : 0ull;
            ++events;
            uint32_t *st = record();
            uint32_t end_at = st[0];
            uint32_t fl = st[1];
            uint32_t efirst = SWG_DYN_KEEP; // EDGES: first block of the pair taken in this event (none: gone idle)
            if (leader && end_at == blocks) {
                // The queue is SWG_DYN_SHARDS counters, shard c handing out pairs begin+c, begin+c+S, ..
                // (one counter for everybody saturates the atomic unit of its memory channel at ~16
                // requests/us and every request then takes 16 us).  A leader starts at its workgroup's
                // home shard and moves on when a shard is empty.
                //
                // Short pairs by the batch (round 4): a request is an atomic and two dependent loads, 2.2 us during
                // which the wavefront issues nothing -- and with pairs of a few token blocks (peptides) more than the
                // pair itself takes, while the counters see hundreds of requests per microsecond.  Where the host says
                // so (batch_B > 1: the pairs of the range's middle zone are short) one request claims batch_B
                // consecutive pairs: the leader loads their batch_B + 1 block offsets at once into the group's record
                // and the next batch_B - 1 hand-outs are two LDS reads.  A counter value u of shard c then means:
                //   u < U1            pair q_begin + c + 8 u                          (the long pairs, one by one)
                //   u < U1 + U2       pairs P1 + (8 (u - U1) + c) B .. + B - 1       (P1 = q_begin + 8 U1)
                //   else              pair P2 + c + 8 (u - U1 - U2)                   (P2 = P1 + 8 U2 B: the last
                //                     pairs of the range one by one again, so that the lane groups end together)
                bool second = (fl & SWG_DYN_SECOND) != 0u;
                uint32_t tried = fl >> 8;
                uint32_t nq = SWG_DYN_NONE, first = 0u, len = 0u;
                const uint32_t left = p.batch_B > 1u ? st[3] : 0u; // pairs of the group's own batch not handed out yet
                if (left != 0u) {
                    nq = st[4];
                    const uint32_t j = nq - st[6];
                    first = st[8u + j];
                    len = st[9u + j] - first;
                    st[3] = left - 1u;
                    st[4] = nq + 1u;
                } else {
                if (EDGES && (fl & SWG_DYN_IDLING)) tried = SWG_DYN_SHARDS, second = true; // (the queues were empty a block ago)
                uint32_t claimed = 1u;
                for (;;) {
                    if (tried >= SWG_DYN_SHARDS) {
                        if (second || p.q2_end <= p.q2_begin) break;
                        second = true; // own range empty: go on with the other launch's
                        tried = 0u;
                    }
                    const uint32_t shard = (blockIdx.x + tried) & (SWG_DYN_SHARDS - 1u);
                    uint32_t *ctr = (second ? p.queue2 : queue) + shard * SWG_DYN_SHARD_STRIDE;
                    if (p.list) {
                        const uint32_t at = shard + SWG_DYN_SHARDS * atomicAdd(ctr, 1u);
                        if (at < n_list) {
                            const uint32_t cand = p.list[at];
                            if (cand >= p.q_begin && cand < p.q_end) {
                                nq = cand;
                                break;
                            }
                            continue; // (another segment's pair: the same shard again)
                        }
                        ++tried;
                        continue;
                    }
                    const uint32_t u = atomicAdd(ctr, 1u);
                    uint32_t cand;
                    if (second) {
                        cand = p.q2_begin + shard + SWG_DYN_SHARDS * u;
                    } else if (u < p.batch_u1 || p.batch_B <= 1u) {
                        cand = p.q_begin + shard + SWG_DYN_SHARDS * u;
                    } else if (u - p.batch_u1 < p.batch_u2) {
                        cand = p.q_begin + SWG_DYN_SHARDS * p.batch_u1 + (SWG_DYN_SHARDS * (u - p.batch_u1) + shard) * p.batch_B;
                        claimed = p.batch_B; // (whole batches only: the host's U2 ends before the range does)
                    } else {
                        cand = p.q_begin + SWG_DYN_SHARDS * (p.batch_u1 + p.batch_u2 * p.batch_B) + shard + SWG_DYN_SHARDS * (u - p.batch_u1 - p.batch_u2);
                    }
                    if (cand < (second ? p.q2_end : p.q_end)) {
                        nq = cand;
                        break;
                    }
                    ++tried;
                }
                if (nq != SWG_DYN_NONE) {
                    if (claimed > 1u) {
                        // the batch's block offsets: independent loads, one wait
                        uint32_t o[SWG_DYN_MAX_BATCH + 1u];
#pragma unroll
                        for (uint32_t i = 0u; i <= SWG_DYN_MAX_BATCH; ++i) o[i] = i <= claimed ? p.pair_off[nq + i] : 0u;
#pragma unroll
                        for (uint32_t i = 0u; i <= SWG_DYN_MAX_BATCH; ++i) st[8u + i] = o[i];
                        st[3] = claimed - 1u;
                        st[4] = nq + 1u;
                        st[6] = nq;
                        first = o[0];
                        len = o[1] - o[0];
                    } else {
                        first = p.pair_off[nq];
                        len = p.pair_off[nq + 1u] - first;
                    }
                }
                }
                const bool idling = (fl & SWG_DYN_IDLING) != 0u;
                fl = (second ? SWG_DYN_SECOND : 0u) | (tried << 8);
                if (nq != SWG_DYN_NONE) {
                    if (EDGES) {
                        efirst = first;
                    } else {
                        tp = reinterpret_cast<const uint32_t *>(p.tok + first);
                        tstep = 4u;
                    }
                    end_at = blocks + len;
                    const uint32_t pushed = st[2];
                    st[SWG_DYN_RING + (pushed & (SWG_DYN_RING - 1u))] = nq;
                    st[2] = pushed + 1u;
                    if (len >= (second ? p.prio_blocks2 : p.prio_blocks)) fl |= SWG_DYN_HOT;
                } else if (EDGES && !idling) {
                    // out of pairs: this block's loads bring zeros; the next event flags their first row
                    end_at = blocks + 1u;
                    efirst = SWG_DYN_NONE;
                    fl |= SWG_DYN_IDLING;
                } else if (EDGES) {
                    end_at = SWG_DYN_NONE;
                    T0 = SWG_TOK_IDLE; // (loaded as zero during the last block)
                    fl |= SWG_DYN_IDLING;
                } else {
                    end_at = SWG_DYN_NONE;
                    tp = zero_blk;
                    tstep = 0u;
                }
                st[0] = end_at;
                st[1] = fl;
            }
            if (EDGES) {
                // the first four lanes follow their leader: row g of the new pair's first block, or parked
                const uint32_t ef = quad_bcast(efirst, 0);
                if (g < 4 && ef != SWG_DYN_KEEP)
                    off = ef != SWG_DYN_NONE ? (ef - p.seg_origin) * 16u + (uint32_t)g * 4u : SWG_OFF_PARKED;
            }
            next_event = SWG_DYN_NONE;
            for (int i = 0; i < 64; i += G)
                next_event = min(next_event, (uint32_t)__builtin_amdgcn_readlane((int)end_at, i));
            // a wavefront feeding a long pair is on the critical path: it keeps the raised priority
            hot = __builtin_amdgcn_ballot_w64(leader && end_at != SWG_DYN_NONE && (fl & SWG_DYN_HOT) != 0u) != 0ull;
            if (!hot) take_turn();
            if (p.trace) event_ticks += wall_clock64() - ev0;
        }
        if (next_event == SWG_DYN_NONE) {
            // every leader out of pairs: let the rows in flight reach the tail lanes, then leave
            if (drain >= (uint32_t)G + 12u) break;
            drain += 4u;
        }
        if (EDGES) {
            // lanes 0..3 of a group fetch the next block's left edges, one row each
            ec = en;
            if (p.edge_in) { // (the first pass has none: the edges stay zero)
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(einR, off * 2u, 0, 0);
                en = make_uint2(v.x, v.y);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t fresh = r == 0 ? T0 : r == 1 ? T1 : r == 2 ? T2 : T3;
            uint32_t lm = Z, lb = Z;
            if (EDGES) {
                lm = quad_bcast(ec.x, r);
                lb = quad_bcast(ec.y, r);
            }
            uint32_t em, eb;
            constexpr int Gs = GW;
            constexpr bool KEEP = F16 && !EDGES; // (see em_rot)
            if (G16) {
                tok = dpp_keep<DPP_ROW_SHR1>(fresh, tok);
                if (KEEP) {
                    em = em_rot[r & 1] = dpp_keep<DPP_ROW_SHR1>(em_rot[r & 1], m_out);
                    eb = eb_rot[r & 1] = dpp_keep<DPP_ROW_SHR1>(eb_rot[r & 1], b_out);
                } else {
                    em = EDGES ? dpp_keep<DPP_ROW_SHR1>(lm, m_out) : dpp_zero<DPP_ROW_SHR1>(m_out);
                    eb = EDGES ? dpp_keep<DPP_ROW_SHR1>(lb, b_out) : dpp_zero<DPP_ROW_SHR1>(b_out);
                }
            } else {
                const uint32_t u0 = dpp_keep<DPP_WAVE_SHR1>(fresh, tok);
                uint32_t u1, u2;
                if (KEEP) {
                    u1 = em_rot[r & 1] = dpp_keep<DPP_WAVE_SHR1>(em_rot[r & 1], m_out);
                    u2 = eb_rot[r & 1] = dpp_keep<DPP_WAVE_SHR1>(eb_rot[r & 1], b_out);
                } else {
                    u1 = EDGES ? dpp_keep<DPP_WAVE_SHR1>(lm, m_out) : dpp_zero<DPP_WAVE_SHR1>(m_out);
                    u2 = EDGES ? dpp_keep<DPP_WAVE_SHR1>(lb, b_out) : dpp_zero<DPP_WAVE_SHR1>(b_out);
                }
                if (Gs == 32) { // lane 32 starts a group too
                    tok = leader ? fresh : u0;
                    em = leader ? lm : u1;
                    eb = leader ? lb : u2;
                } else {
                    tok = u0;
                    em = u1;
                    eb = u2;
                }
            }
            // this row of the next block (the register is free: its value went into the DPP move above)
            if (EDGES) {
                const uint32_t t = __builtin_amdgcn_raw_buffer_load_b32(tokR, off + (uint32_t)r * 4u, 0, 0);
                if (r == 0) T0 = t;
                else if (r == 1) T1 = t;
                else if (r == 2) T2 = t;
                else T3 = t;
                if (r == 3) off += tail ? 32u : 16u; // (the tail lane is 3 (mod 4) rows behind: its blocks begin here)
            } else {
                if (r == 0) T0 = tp[0];
                else if (r == 1) T1 = tp[1];
                else if (r == 2) T2 = tp[2];
                else T3 = tp[3];
            }
            // Rows with a flag are rare: ONE wave-uniform test (any token above 0xFFFF) keeps their
            // bookkeeping out of the common step; the recurrence itself is issued once, with per-lane
            // gap operands that hold the real gap scores except while the lane is on a reset row.
            const bool special = __builtin_amdgcn_ballot_w64(tok > 0xFFFFu) != 0ull;
            if (special) {
                // reset rows: gap magnitudes of all ones wipe A/G/B, two such rows wipe M
                uint32_t fm = 0u - ((tok >> 16) & 1u);
                asm volatile("" : "+v"(fm)); // (keeps this a branch: flattened into selects it costs six instructions on every row)
                if (WIDE) {
                    // (a wiped lane fed zero edges stays wiped: the first of the two reset rows is enough, and
                    // rows that are special for another flag skip the 3K+2 selects)
                    const uint32_t fw = fm & (((tok >> 19) & 1u) - 1u);
                    if (__builtin_amdgcn_ballot_w64(fw != 0u) != 0ull) cells.wipe(fw);
                } else if (F16) {
                    // (two rows at 65504 clear any state below 32768; a lane that got beyond is wiped by hand: its best
                    // -- still the ended pair's on the first reset row, nothing has grown it since that pair's last
                    // row -- is 32768 or more; on the second reset row the best is what the first left: small)
                    // -- and only a launch whose query can score that much looks at all (p.f16_wipe, a scalar test).
                    if (p.f16_wipe) {
                        const uint32_t fi = fm & cells.best_is_huge();
                        if (__builtin_amdgcn_ballot_w64(fi != 0u) != 0ull) cells.wipe(fi);
                    }
                    cells.best = (cells.best & ~fm) | (Z & fm);
                    go_v = (go_v & ~fm) | (SWG_F16_BIG & fm);
                    ge_v = (ge_v & ~fm) | (SWG_F16_BIG & fm);
                } else {
                    cells.best &= ~fm;
                    go_v |= fm;
                    ge_v |= fm;
                }
                if (EDGES && r == 3 && p.edge_out) {
                    // a pair's first row at the tail lane: its edges go to the pair's rows from here on
                    // (the pair is the next one of the ring: every earlier one has had its last row);
                    // the idle row: nowhere
                    if (tail && (tok & SWG_TOK_RESET) != 0u) {
                        const uint32_t pr = record()[SWG_DYN_RING + (nlast & (SWG_DYN_RING - 1u))];
                        off = (p.pair_off[pr] - p.seg_origin) * 32u;
                    }
                    if (tail && (tok & SWG_TOK_IDLE) != 0u) off = SWG_OFF_PARKED_TAIL;
                }
            }
            const uint2 e = cells.template row<(SWG_DYN_FENCE_ABOVE < K)>(prof_addr<0>(base, tok), prof_addr<1>(base, tok),
                                                                           em, eb, go_v, ge_v, zero_v);
            if (special) {
                if (!WIDE) {
                    go_v = p.go;
                    ge_v = p.ge;
                }
                // The pair's score.  Every lane holds the maximum of its own columns over the pair (cells.best); on the
                // pair's LAST row that maximum travels with the row from lane to lane -- bc: what came in from the lane
                // before, joined with this lane's best -- so the tail lane, the last of the group to see the row, holds
                // the pair's maximum and writes the two scores.  Only the chain along the last row matters (a lane's bc
                // is read by its neighbour one step after it was written, on that neighbour's turn at the same row), so
                // the move and the maximum are made on flagged rows only: two instructions on those, nothing on the
                // others.  (Round 2-3 joined the bests through two LDS atomics per lane and pair, in a divergent branch
                // that short pairs -- peptides: 35 rows -- entered on every step: more than half of such a fill.)
                {
                    uint32_t cin;
                    if (G16) {
                        cin = dpp_keep<DPP_ROW_SHR1>(Z, bc);
                    } else {
                        const uint32_t u3 = dpp_keep<DPP_WAVE_SHR1>(Z, bc);
                        cin = (GW == 32 && leader) ? Z : u3; // lane 32 starts a group too
                    }
                    bc = F16 ? pk_max3_f16(cin, cells.best, cells.best) : pk_max_i16(cin, cells.best);
                }
                if (tail && (tok & SWG_TOK_LAST) != 0u) {
                    // the pair's last row at the tail lane: the pair is the next one of the ring
                    const uint32_t pr = record()[SWG_DYN_RING + (nlast & (SWG_DYN_RING - 1u))];
                    uint32_t sx, sy;
                    if (F16) {
                        sx = (uint32_t)f16_score(f16_key(bc & 0xFFFFu));
                        sy = (uint32_t)f16_score(f16_key(bc >> 16));
                    } else {
                        sx = (bc ^ Z) & 0xFFFFu;
                        sy = (bc ^ Z) >> 16;
                    }
                    if (pr >= p.pair_limit) {
                        // cannot happen with a well-formed token stream; a stray write must not, either
                    } else if (EDGES) { // one pass of several: the score is the maximum over the passes
                        atomicMax(scores + 2u * pr, (int)sx);
                        atomicMax(scores + 2u * pr + 1u, (int)sy);
                    } else {
                        scores[2u * pr] = (int)sx;
                        scores[2u * pr + 1u] = (int)sy;
                    }
                    ++nlast;
                }
            }
            m_out = e.x;
            b_out = e.y;
            if (EDGES) {
                // the tail lane's row is the pass's right edge (L2 gathers a pair's consecutive rows)
                if (__builtin_expect(__builtin_amdgcn_inverse_ballot_w64(tails), 1))
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{e.x, e.y}, eoutR, off + (uint32_t)((r + 1) & 3) * 8u, 0, 0);
            }
        }
        if (!EDGES) tp += tstep;
        ++blocks;
        if (!hot && (blocks & (SWG_DYN_TURN_EVERY - 1u)) == 0u) take_turn();
    }
    };
    if (G == 16) main_loop(std::integral_constant<int, 16>());
    else if (G == 32) main_loop(std::integral_constant<int, 32>());
    else main_loop(std::integral_constant<int, 64>());
    if (p.stamps && lane == 0) atomicMax(p.stamps + 1, (unsigned long long)wall_clock64()); // latest end
    if (p.trace && lane == 0) {
        uint64_t *t = p.trace + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) * W + w) * 4u;
        t[0] = t_start;
        t[1] = wall_clock64();
        t[2] = (uint64_t)blocks | ((uint64_t)events << 32) | (event_ticks << 44);
        t[3] = (uint64_t)(uint32_t)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) | ((uint64_t)rank << 32) |
               ((uint64_t)(uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 40);
    }
}

// ---------------------------------------------------------------------------
// The int32 fill with a work queue: G lanes share ONE sequence, scores beyond 16 bits
// ---------------------------------------------------------------------------
// For everything the packed int16 form cannot hold -- sequences whose 16-bit score saturated,
// force_bits = 32 -- when the gap scores are non-positive and the query fits one pass (G*K columns).
// Same skeleton as swg_diag_dyn_kernel: persistent wavefronts, lane groups of 16 / 32 / 64 lanes
// sweeping an anti-diagonal, work items off sharded counters, the token stream of the PAIR the
// sequence belongs to (the leader lane picks the pair's X or Y residue byte), scores collected by an
// LDS atomic maximum on the last row.  An item is one sequence (sorted rank): either every rank of a
// range, or the entries of a device-side list (the saturated ones).
//
// Cells: the same algebra as the packed form, in int32, with three-operand maxima doing the floors:
//     a = max3(G[k], A[k] - e, 0)    b = max3(gl, bl - e, 0)    m = max3(md + s, a, b)    G' = m - g
// (G = M - |go| unfloored: max3's zero does it) = 8 instructions per cell against 12 for the
// reference's recurrence term by term (which stays the path for positive gap scores:
// swg_diag32_kernel).  Profile: [col/2][32 residues][2] int32 -- a residue's two columns are 8 bytes,
// so the token's residue byte is the LDS offset as it stands, like the int16 kernels.
template <int K> struct CellsQ32 {
    static constexpr int KP = (K + 1) / 2 * 2;
    static constexpr int CHUNK = 32 * 2 * 4; // 256 bytes: 32 residues x 2 columns x int32
    int M[K], G[K], A[K];
    int best, mdl;

    DEVINL void reset()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) M[k] = G[k] = A[k] = 0;
        best = 0;
        mdl = 0;
    }

    // em / eb: M and B of the column left of this lane's strip in this row; go / ge: gap magnitudes
    template <bool FENCED = false> DEVINL int2 row(uint32_t ax, int em, int eb, int go, int ge)
    {
        constexpr int NCH = KP / 2;
        int md = mdl;
        int gl = em - go;
        int bl = eb;
        int2 nx = lds_read_i2(ax);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int2 w = nx;
            if (c + 1 < NCH) nx = lds_read_i2(ax + (c + 1) * CHUNK);
            const int s[2] = {w.x, w.y};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = 2 * c + u;
                if (k >= K) break; // unused tail of an odd K's last chunk
                const int t = md + s[u];
                md = M[k];
                const int a = imax3(G[k], A[k] - ge, 0);
                const int b = imax3(gl, bl - ge, 0);
                const int m = imax3(t, a, b);
                M[k] = m;
                A[k] = a;
                gl = G[k] = m - go;
                bl = b;
                best = imax(best, m);
            }
            if (FENCED && (c & 1) && c + 1 < NCH) __builtin_amdgcn_sched_barrier(0);
        }
        mdl = em;
        return make_int2(M[K - 1], bl);
    }
};

// The reference's recurrence term by term (src/alignment.c:124-161), for gap scores of any sign (the reduced algebra
// above needs gap_open <= 0 and gap_extend <= 0; the reference's CLI accepts positive ones,
// src/alignment_cmdline.c:255-267): per column U = max(H, B), A, D = max(H, A, B) of the previous row; a lane's
// right edge is (L = max(H, A), B, D).  12 instructions per cell.  go / ge are the signed scores; on a reset row
// they are -2^29, which zeroes every state in one row.
template <int K> struct CellsX32 {
    static constexpr int KP = (K + 1) / 2 * 2;
    static constexpr int CHUNK = 32 * 2 * 4;
    int U[K], A[K], D[K];
    int best, ddl;

    DEVINL void reset()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) U[k] = A[k] = D[k] = 0;
        best = 0;
        ddl = 0;
    }

    // el / eb / ed: L, B and D of the column left of this lane's strip in this row (ed becomes the next row's diagonal)
    template <bool FENCED = false> DEVINL int2 row(uint32_t ax, int el, int eb, int ed, int go, int ge, int &od)
    {
        constexpr int NCH = KP / 2;
        int dd = ddl;
        int ll = el, bl = eb;
        int2 nx = lds_read_i2(ax);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int2 w = nx;
            if (c + 1 < NCH) nx = lds_read_i2(ax + (c + 1) * CHUNK);
            const int s[2] = {w.x, w.y};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = 2 * c + u;
                if (k >= K) break;
                const int h = imax(dd + s[u], 0);             // src/alignment.c:124-129
                const int a = imax3(U[k] + go, A[k] + ge, 0); // src/alignment.c:142-147
                const int b = imax3(ll + go, bl + ge, 0);     // src/alignment.c:156-161
                dd = D[k];
                U[k] = imax(h, b);
                ll = imax(h, a);
                D[k] = imax(ll, b);
                A[k] = a;
                bl = b;
                best = imax(best, h);                         // src/alignment.c:133
            }
            if (FENCED && (c & 1) && c + 1 < NCH) __builtin_amdgcn_sched_barrier(0);
        }
        ddl = ed;
        od = D[K - 1];
        return make_int2(ll, bl);
    }
};

#define SWG_Q32_RESET_GAP (1 << 29) // gap magnitude on reset rows: one such row zeroes A, G and B, two zero M

// EDGES: one pass of a query longer than G*K columns, as in swg_diag_dyn_kernel: the left edge (M, B) of every
// row comes from the previous pass's launch and the right edge goes to the next one, indexed by
// 2 * (the row's position in the pair-major token order) + (which sequence of the pair), so that both
// sequences of a pair can be items of the same launch; scores are the maximum over the passes.
// EXACT: the cells of CellsX32 (gap scores of any sign; p.go / p.ge are then the signed scores) instead of the
// reduced algebra; a third edge value (D) travels with the other two, through edge_d_in / edge_d_out between passes.
template <int K, int MAXW, bool EDGES, bool EXACT = false>
__global__ __launch_bounds__(MAXW * 64) void swg_diag32q_kernel(const SwgDiagQ32Params p)
{
    extern __shared__ __attribute__((aligned(256))) uint8_t smem[]; // int32 query profile, then the group records
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int G = (int)p.G;
    const int gshift = G == 16 ? 4 : G == 32 ? 5 : 6;
    const int g = lane & (G - 1);
    const bool leader = g == 0, tail = g == G - 1;
    const uint32_t base = prof_base(SWG_LDS_ADDRESS(smem) + (uint32_t)g * (CellsQ32<K>::KP / 2) * CellsQ32<K>::CHUNK, g);
    const uint32_t slice = (uint32_t)G * CellsQ32<K>::KP * 128u;
    auto record = [&]() -> uint32_t * {
        const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        return reinterpret_cast<uint32_t *>(smem + slice) + (((uint32_t)w << (6 - gshift)) + (l >> gshift)) * SWG_DYN_STATE;
    };
    // (a re-score launch is queued behind every fill that MAY flag something: with an empty list -- the usual
    // case -- it leaves before it has read a byte of the profile)
    const uint32_t n_items = p.list_count ? *p.list_count : p.q_end - p.q_begin;
    if (n_items == 0u) return;
    for (uint32_t o = threadIdx.x * 16u; o < slice; o += blockDim.x * 16u)
        *reinterpret_cast<uint4 *>(smem + o) = *reinterpret_cast<const uint4 *>(p.profile + o);
    {
        uint32_t *st = record();
        if (g < 4) st[g] = 0u; // every group is due at block 0
    }
    __syncthreads();

    typename std::conditional<EXACT, CellsX32<K>, CellsQ32<K>>::type cells;
    cells.reset();
    uint32_t tok = 0u;
    int m_out = 0, b_out = 0, d_out = 0; // right edge of the lane's strip: (M, B), or (L, B, D) in the exact form
    int go_v = (int)p.go, ge_v = (int)p.ge;
    int bc = 0; // the current sequence's best on its way along its last row (see swg_diag_dyn_kernel)
    uint32_t nlast = 0u;
    // v_perm selector of the leader: residue byte of X (.. 00) or Y (.. 01), zero, flags byte, zero.  It
    // belongs to the sequence whose tokens are being LOADED; the blocks in flight keep the selector they
    // were loaded under (pick_nxt, pick_cur), or the last rows of a sequence would be read with its
    // successor's.
    uint32_t pick = 0x0C020C00u, pick_nxt = 0x0C020C00u, pick_cur = 0x0C020C00u;
    // tokens as in swg_diag_dyn_kernel: the rows of the current block, each re-loaded for the next block right after its use
    uint32_t T0 = 0u, T1 = 0u, T2 = 0u, T3 = 0u;
    const uint32_t *const zero_blk = reinterpret_cast<const uint32_t *>(p.tok + p.zero_block);
    const uint32_t *tp = zero_blk;
    uint32_t tstep = 0u;
    // EDGES: edge index of row 0 of the block tp points at / the block loaded during this iteration / the
    // block being worked on (leader; none: idle), the index travelling with the token, left edges of the
    // current / next block (lanes 0..3 of a group, one row each)
    uint32_t rb_load = SWG_DYN_NONE, rb_nxt = SWG_DYN_NONE, rb_cur = SWG_DYN_NONE, ridx = SWG_DYN_NONE;
    int2 ec = make_int2(0, 0), en = make_int2(0, 0);
    int ecd = 0, end_ = 0; // EXACT with EDGES: the third edge value of the current / next block
    uint32_t blocks = 0u, next_event = 0u, drain = 0u;
    bool hot = false;
    uint32_t rank;
    {
        const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((11 << 11) | (4 << 6) | 4);
        const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        const uint32_t simd = (((hw >> 4) << 2) | (hw & 3u) | (xcc << 10)) & (SWG_DYN_SIMD_SLOTS - 1u);
        uint32_t r = 0u;
        if (lane == 0) r = atomicAdd(p.simd_ranks + simd, 1u);
        rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    }
    auto take_turn = [&]() {
        const uint32_t x = ((uint32_t)(wall_clock64() >> SWG_DYN_TURN_SHIFT) + rank) & 0xFFFFu;
        const uint32_t turn = p.turn_levels == 4u ? (x & 3u) : x - 3u * ((x * 0xAAABu) >> 17);
        if (turn == 0u) __builtin_amdgcn_s_setprio(0);
        else if (turn == 1u) __builtin_amdgcn_s_setprio(1);
        else if (turn == 2u) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    };

    // (two loops, for 16-lane groups and for wider ones: see swg_diag_qq_kernel)
    auto main_loop = [&](auto gw_tag) {
    constexpr int GW = decltype(gw_tag)::value; // lanes per group, a constant inside the loop
    constexpr bool G16 = GW == 16;
    for (;;) {
        if (blocks == next_event) {
            // some sequence has run out (or this is the start): its leader takes the next one
            __builtin_amdgcn_s_setprio(3);
            uint32_t *st = record();
            uint32_t end_at = st[0];
            uint32_t fl = st[1];
            if (leader && end_at == blocks) {
                uint32_t tried = fl >> 8;
                uint32_t item = SWG_DYN_NONE;
                while (tried < SWG_DYN_SHARDS) {
                    const uint32_t shard = (blockIdx.x + tried) & (SWG_DYN_SHARDS - 1u);
                    const uint32_t cand = shard + SWG_DYN_SHARDS * atomicAdd(p.queue + shard * SWG_DYN_SHARD_STRIDE, 1u);
                    if (cand < n_items) {
                        item = cand;
                        break;
                    }
                    ++tried;
                }
                fl = tried << 8;
                uint32_t seq = SWG_DYN_NONE;
                if (item != SWG_DYN_NONE) seq = p.list ? p.list[item] : p.q_begin + item;
                if (seq != SWG_DYN_NONE && seq < p.seq_limit) {
                    const uint32_t pr = seq >> 1;
                    const uint32_t first = p.pair_off[pr];
                    const uint32_t len = p.pair_off[pr + 1u] - first;
                    tp = reinterpret_cast<const uint32_t *>(p.tok + first);
                    tstep = 4u;
                    if (EDGES) rb_load = first * 8u + (seq & 1u);
                    end_at = blocks + len;
                    pick = 0x0C020C00u | (seq & 1u);
                    const uint32_t pushed = st[2];
                    st[SWG_DYN_RING + (pushed & (SWG_DYN_RING - 1u))] = seq;
                    st[2] = pushed + 1u;
                    if (len >= p.prio_blocks) fl |= SWG_DYN_HOT;
                } else {
                    end_at = SWG_DYN_NONE;
                    tp = zero_blk;
                    tstep = 0u;
                    rb_load = SWG_DYN_NONE;
                }
                st[0] = end_at;
                st[1] = fl;
            }
            next_event = SWG_DYN_NONE;
            for (int i = 0; i < 64; i += G)
                next_event = min(next_event, (uint32_t)__builtin_amdgcn_readlane((int)end_at, i));
            hot = __builtin_amdgcn_ballot_w64(leader && end_at != SWG_DYN_NONE && (fl & SWG_DYN_HOT) != 0u) != 0ull;
            if (!hot) take_turn();
        }
        if (next_event == SWG_DYN_NONE) {
            if (drain >= (uint32_t)G + 12u) break;
            drain += 4u;
        }
        pick_cur = pick_nxt; // the block worked on now was loaded during the previous iteration ...
        pick_nxt = pick;     // ... and the one loaded during this iteration belongs to the leader's current sequence
        if (EDGES) {
            rb_cur = rb_nxt;
            rb_nxt = rb_load;
            if (rb_load != SWG_DYN_NONE) rb_load += 8u;
            ec = en;
            ecd = end_;
            // lanes 0..3 of a group fetch the next block's left edges, one row each
            const uint32_t bq = quad_bcast(rb_nxt, 0);
            en = make_int2(0, 0);
            end_ = 0;
            if (g < 4 && bq != SWG_DYN_NONE && p.edge_in) {
                en = p.edge_in[(size_t)bq + 2u * (uint32_t)g];
                if (EXACT) end_ = p.edge_d_in[(size_t)bq + 2u * (uint32_t)g];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // the leader's token of this row: the residue byte of ITS sequence of the pair, and the flags
            const uint32_t raw = r == 0 ? T0 : r == 1 ? T1 : r == 2 ? T2 : T3;
            const uint32_t fresh = __builtin_amdgcn_perm(raw, raw, pick_cur);
            uint32_t lm = 0u, lb = 0u, ld = 0u, fresh_ridx = SWG_DYN_NONE;
            if (EDGES) {
                lm = quad_bcast((uint32_t)ec.x, r);
                lb = quad_bcast((uint32_t)ec.y, r);
                if (EXACT) ld = quad_bcast((uint32_t)ecd, r);
                fresh_ridx = rb_cur != SWG_DYN_NONE ? rb_cur + 2u * (uint32_t)r : SWG_DYN_NONE;
            }
            int em, eb, ed = 0;
            constexpr int Gs = GW;
            if (G16) {
                tok = dpp_keep<DPP_ROW_SHR1>(fresh, tok);
                em = (int)(EDGES ? dpp_keep<DPP_ROW_SHR1>(lm, (uint32_t)m_out) : dpp_zero<DPP_ROW_SHR1>((uint32_t)m_out));
                eb = (int)(EDGES ? dpp_keep<DPP_ROW_SHR1>(lb, (uint32_t)b_out) : dpp_zero<DPP_ROW_SHR1>((uint32_t)b_out));
                if (EXACT) ed = (int)(EDGES ? dpp_keep<DPP_ROW_SHR1>(ld, (uint32_t)d_out) : dpp_zero<DPP_ROW_SHR1>((uint32_t)d_out));
                if (EDGES) ridx = dpp_keep<DPP_ROW_SHR1>(fresh_ridx, ridx);
            } else {
                const uint32_t u0 = dpp_keep<DPP_WAVE_SHR1>(fresh, tok);
                const uint32_t u1 = EDGES ? dpp_keep<DPP_WAVE_SHR1>(lm, (uint32_t)m_out) : dpp_zero<DPP_WAVE_SHR1>((uint32_t)m_out);
                const uint32_t u2 = EDGES ? dpp_keep<DPP_WAVE_SHR1>(lb, (uint32_t)b_out) : dpp_zero<DPP_WAVE_SHR1>((uint32_t)b_out);
                const uint32_t u3 = !EXACT ? 0u : EDGES ? dpp_keep<DPP_WAVE_SHR1>(ld, (uint32_t)d_out) : dpp_zero<DPP_WAVE_SHR1>((uint32_t)d_out);
                const uint32_t u4 = EDGES ? dpp_keep<DPP_WAVE_SHR1>(fresh_ridx, ridx) : 0u;
                if (Gs == 32) { // lane 32 starts a group too
                    tok = leader ? fresh : u0;
                    em = leader ? (int)lm : (int)u1;
                    eb = leader ? (int)lb : (int)u2;
                    if (EXACT) ed = leader ? (int)ld : (int)u3;
                    if (EDGES) ridx = leader ? fresh_ridx : u4;
                } else {
                    tok = u0;
                    em = (int)u1;
                    eb = (int)u2;
                    if (EXACT) ed = (int)u3;
                    if (EDGES) ridx = u4;
                }
            }
            if (r == 0) T0 = tp[0];
            else if (r == 1) T1 = tp[1];
            else if (r == 2) T2 = tp[2];
            else T3 = tp[3];
            const bool special = __builtin_amdgcn_ballot_w64(tok > 0xFFFFu) != 0ull;
            if (special) {
                uint32_t fm = 0u - ((tok >> 16) & 1u);
                asm volatile("" : "+v"(fm)); // (keeps this a branch)
                cells.best &= (int)~fm;
                // (the reduced form subtracts gap magnitudes, the exact form adds signed gap scores)
                const uint32_t rg = EXACT ? (uint32_t)-SWG_Q32_RESET_GAP : (uint32_t)SWG_Q32_RESET_GAP;
                go_v = (int)(((uint32_t)go_v & ~fm) | (rg & fm));
                ge_v = (int)(((uint32_t)ge_v & ~fm) | (rg & fm));
            }
            int2 e;
            if constexpr (EXACT) e = cells.template row<(K > 12)>(prof_addr<0>(base, tok), em, eb, ed, go_v, ge_v, d_out);
            else e = cells.template row<(K > 12)>(prof_addr<0>(base, tok), em, eb, go_v, ge_v);
            if (special) {
                go_v = (int)p.go;
                ge_v = (int)p.ge;
                // the sequence's score travels along its last row from lane to lane (see swg_diag_dyn_kernel)
                {
                    uint32_t cin;
                    if (G16) {
                        cin = dpp_zero<DPP_ROW_SHR1>((uint32_t)bc);
                    } else {
                        const uint32_t u5 = dpp_zero<DPP_WAVE_SHR1>((uint32_t)bc);
                        cin = (GW == 32 && leader) ? 0u : u5; // lane 32 starts a group too
                    }
                    bc = imax((int)cin, cells.best);
                }
                if (tail && (tok & SWG_TOK_LAST) != 0u) {
                    const uint32_t seq = record()[SWG_DYN_RING + (nlast & (SWG_DYN_RING - 1u))];
                    if (seq < p.seq_limit) {
                        if (EDGES) atomicMax(p.scores + seq, bc); // one pass of several
                        else p.scores[seq] = bc;
                    }
                    ++nlast;
                }
            }
            m_out = e.x;
            b_out = e.y;
            if (EDGES) {
                if (tail && ridx != SWG_DYN_NONE && p.edge_out) {
                    p.edge_out[ridx] = e;
                    if (EXACT) p.edge_d_out[ridx] = d_out;
                }
            }
        }
        tp += tstep;
        ++blocks;
        if (!hot) take_turn();
    }
    };
    if (G == 16) main_loop(std::integral_constant<int, 16>());
    else if (G == 32) main_loop(std::integral_constant<int, 32>());
    else main_loop(std::integral_constant<int, 64>());
}

// ---------------------------------------------------------------------------
// Two QUERIES per lane against one sequence (batches of queries: swg_search_multi)
// ---------------------------------------------------------------------------
// Of the f16 cells' 8.5 instructions per column pair one is the v_perm_b32 that pairs the profile words of the two
// SEQUENCES that share a lane: two residues, two LDS reads, one combine.  With a batch of queries the two halves of a
// register can instead hold two queries against the SAME sequence: one residue, and a profile that keeps the two
// queries' scores for a column side by side -- one ds_read_b64 is two columns ready to add.  7.5 instructions per two
// cells.  Items are sequences (as in swg_diag32q_kernel: the token stream of the pair a sequence belongs to, the
// leader lane picking its byte); row y of the grid works for query pair y.  Single pass, f16 cells (biased by -2048:
// see CellsDiag FORM 2 for the arithmetic, the reset rows and the leader's edge registers); the host takes this path
// only when no query of the batch can score 4096.
template <int K> struct CellsQQ {
    static constexpr int KP = (K + 1) / 2 * 2;
    static constexpr int CHUNK = 32 * 2 * 4; // 256 bytes: 32 residues x 2 columns x (2 queries x f16)
    uint32_t M[K], G[K], A[K];
    uint32_t best, mdl;

    DEVINL void reset()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) M[k] = G[k] = A[k] = SWG_F16_ZERO;
        best = SWG_F16_ZERO;
        mdl = SWG_F16_ZERO;
    }
    DEVINL void wipe(uint32_t fm)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            M[k] = (M[k] & ~fm) | (SWG_F16_ZERO & fm);
            G[k] = (G[k] & ~fm) | (SWG_F16_ZERO & fm);
            A[k] = (A[k] & ~fm) | (SWG_F16_ZERO & fm);
        }
        best = (best & ~fm) | (SWG_F16_ZERO & fm);
        mdl = (mdl & ~fm) | (SWG_F16_ZERO & fm);
    }
    DEVINL uint32_t best_is_huge() const { return 0u - (uint32_t)((((best & 0x78007800u) + 0x08000800u) & 0x80008000u) != 0u); }

    template <bool FENCED = false> DEVINL uint2 row(uint32_t ax, uint32_t em, uint32_t eb, uint32_t go, uint32_t ge, uint32_t zero)
    {
        constexpr int NCH = KP / 2;
        uint32_t md = mdl;
        uint32_t gl = pk_sub_f16(em, go);
        uint32_t bl = eb;
        // (a chunk is two columns = 60 cycles of arithmetic, less than an LDS read takes: the reads run TWO chunks ahead)
        uint2 n0 = lds_read_u2(ax), n1 = NCH > 1 ? lds_read_u2(ax + CHUNK) : make_uint2(0u, 0u);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint2 w = n0;
            n0 = n1;
            if (c + 2 < NCH) n1 = lds_read_u2(ax + (c + 2) * CHUNK);
            const uint32_t s[2] = {w.x, w.y};
            uint32_t mprev = 0u;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = 2 * c + u;
                if (k >= K) break; // unused tail of an odd K's last chunk
                const uint32_t t = pk_add_f16(md, s[u]);
                md = M[k];
                const uint32_t a = pk_max3_f16(G[k], pk_sub_f16(A[k], ge), zero);
                const uint32_t b = pk_max3_f16(gl, pk_sub_f16(bl, ge), zero);
                const uint32_t m = pk_max3_f16(t, a, b);
                M[k] = m;
                A[k] = a;
                gl = G[k] = pk_sub_f16(m, go);
                bl = b;
                if (u & 1) best = pk_max3_f16(best, mprev, m);
                else if (k == K - 1) best = pk_max3_f16(best, m, m);
                mprev = m;
            }
            if (FENCED && (c & 1) && c + 1 < NCH) __builtin_amdgcn_sched_barrier(0);
        }
        mdl = em;
        return make_uint2(M[K - 1], bl);
    }
};

template <int K, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void swg_diag_qq_kernel(const SwgDiagQQParams p)
{
    extern __shared__ __attribute__((aligned(256))) uint8_t smem[]; // the query pair's profile, then the group records
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int G = (int)p.G;
    const int gshift = G == 16 ? 4 : G == 32 ? 5 : 6;
    const int g = lane & (G - 1);
    const bool leader = g == 0, tail = g == G - 1;
    constexpr uint32_t Z = SWG_F16_ZERO;
    const uint32_t base = prof_base(SWG_LDS_ADDRESS(smem) + (uint32_t)g * (CellsQQ<K>::KP / 2) * CellsQQ<K>::CHUNK, g);
    const uint32_t slice = (uint32_t)G * CellsQQ<K>::KP * 128u;
    auto record = [&]() -> uint32_t * {
        const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        return reinterpret_cast<uint32_t *>(smem + slice) + (((uint32_t)w << (6 - gshift)) + (l >> gshift)) * SWG_DYN_STATE;
    };
    const uint8_t *profile = p.profile + (size_t)blockIdx.y * p.profile_stride;
    uint32_t *const queue = p.queue + (size_t)blockIdx.y * p.queue_stride;
    int32_t *const scores_a = p.scores + (size_t)(2u * blockIdx.y) * p.score_stride;      // first query of the pair
    int32_t *const scores_b = p.scores + (size_t)(2u * blockIdx.y + 1u) * p.score_stride; // second (a copy of the first for an odd batch's last pair)
    const bool have_b = 2u * blockIdx.y + 1u < p.n_queries;
    for (uint32_t o = threadIdx.x * 16u; o < slice; o += blockDim.x * 16u)
        *reinterpret_cast<uint4 *>(smem + o) = *reinterpret_cast<const uint4 *>(profile + o);
    {
        uint32_t *st = record();
        if (g < 4) st[g] = 0u; // every group is due at block 0
    }
    __syncthreads();
    const uint32_t n_items = p.q_end - p.q_begin;

    CellsQQ<K> cells;
    cells.reset();
    uint32_t tok = 0u, m_out = Z, b_out = Z;
    uint32_t em_rot[2] = {Z, Z}, eb_rot[2] = {Z, Z}; // (written only by the DPP moves: the leader keeps score 0, see swg_diag_dyn_kernel)
    uint32_t zero_v = Z;
    asm volatile("" : "+v"(zero_v));
    uint32_t bc = Z; // the current sequence's bests on their way along its last row (see swg_diag_dyn_kernel)
    uint32_t go_v = p.go, ge_v = p.ge;
    uint32_t nlast = 0u;
    // v_perm selector of the leader: residue byte of X (.. 00) or Y (.. 01), zero, flags byte, zero; the blocks in
    // flight keep the selector they were loaded under (see swg_diag32q_kernel)
    uint32_t pick = 0x0C020C00u, pick_nxt = 0x0C020C00u, pick_cur = 0x0C020C00u;
    uint32_t T0 = 0u, T1 = 0u, T2 = 0u, T3 = 0u;
    const uint32_t *const zero_blk = reinterpret_cast<const uint32_t *>(p.tok + p.zero_block);
    const uint32_t *tp = zero_blk;
    uint32_t tstep = 0u;
    uint32_t blocks = 0u, next_event = 0u, drain = 0u;
    bool hot = false;
    uint32_t rank;
    {
        const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((11 << 11) | (4 << 6) | 4);
        const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        const uint32_t simd = (((hw >> 4) << 2) | (hw & 3u) | (xcc << 10)) & (SWG_DYN_SIMD_SLOTS - 1u);
        uint32_t r = 0u;
        if (lane == 0) r = atomicAdd(p.simd_ranks + simd, 1u);
        rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    }
    auto take_turn = [&]() {
        const uint32_t x = ((uint32_t)(wall_clock64() >> SWG_DYN_TURN_SHIFT) + rank) & 0xFFFFu;
        const uint32_t turn = p.turn_levels == 4u ? (x & 3u) : x - 3u * ((x * 0xAAABu) >> 17);
        if (turn == 0u) __builtin_amdgcn_s_setprio(0);
        else if (turn == 1u) __builtin_amdgcn_s_setprio(1);
        else if (turn == 2u) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    };

    // (the loop exists twice, for 16-lane groups and for wider ones, chosen once per launch: a run-time test of G inside the
    // row makes the compiler issue BOTH hand-overs on every row -- three wave_shr moves and three selects, then, for 16
    // lanes, three row_shr moves and five copies -- eleven instructions per row too many)
    auto main_loop = [&](auto gw_tag) {
    constexpr int GW = decltype(gw_tag)::value; // lanes per group, a constant inside the loop
    constexpr bool G16 = GW == 16;
    for (;;) {
        if (blocks == next_event) {
            __builtin_amdgcn_s_setprio(3);
            uint32_t *st = record();
            uint32_t end_at = st[0];
            uint32_t fl = st[1];
            if (leader && end_at == blocks) {
                uint32_t tried = fl >> 8;
                uint32_t item = SWG_DYN_NONE;
                while (tried < SWG_DYN_SHARDS) {
                    const uint32_t shard = (blockIdx.x + tried) & (SWG_DYN_SHARDS - 1u);
                    const uint32_t cand = shard + SWG_DYN_SHARDS * atomicAdd(queue + shard * SWG_DYN_SHARD_STRIDE, 1u);
                    if (cand < n_items) {
                        item = cand;
                        break;
                    }
                    ++tried;
                }
                fl = tried << 8;
                const uint32_t seq = item != SWG_DYN_NONE ? p.q_begin + item : SWG_DYN_NONE;
                if (seq != SWG_DYN_NONE && seq < p.seq_limit) {
                    const uint32_t pr = seq >> 1;
                    const uint32_t first = p.pair_off[pr];
                    const uint32_t len = p.pair_off[pr + 1u] - first;
                    tp = reinterpret_cast<const uint32_t *>(p.tok + first);
                    tstep = 4u;
                    end_at = blocks + len;
                    pick = 0x0C020C00u | (seq & 1u);
                    const uint32_t pushed = st[2];
                    st[SWG_DYN_RING + (pushed & (SWG_DYN_RING - 1u))] = seq;
                    st[2] = pushed + 1u;
                    if (len >= p.prio_blocks) fl |= SWG_DYN_HOT;
                } else {
                    end_at = SWG_DYN_NONE;
                    tp = zero_blk;
                    tstep = 0u;
                }
                st[0] = end_at;
                st[1] = fl;
            }
            next_event = SWG_DYN_NONE;
            for (int i = 0; i < 64; i += G)
                next_event = min(next_event, (uint32_t)__builtin_amdgcn_readlane((int)end_at, i));
            hot = __builtin_amdgcn_ballot_w64(leader && end_at != SWG_DYN_NONE && (fl & SWG_DYN_HOT) != 0u) != 0ull;
            if (!hot) take_turn();
        }
        if (next_event == SWG_DYN_NONE) {
            if (drain >= (uint32_t)G + 12u) break;
            drain += 4u;
        }
        pick_cur = pick_nxt;
        pick_nxt = pick;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t raw = r == 0 ? T0 : r == 1 ? T1 : r == 2 ? T2 : T3;
            const uint32_t fresh = __builtin_amdgcn_perm(raw, raw, pick_cur);
            uint32_t em, eb;
            constexpr int Gs = GW;
            if (G16) {
                tok = dpp_keep<DPP_ROW_SHR1>(fresh, tok);
                em = em_rot[r & 1] = dpp_keep<DPP_ROW_SHR1>(em_rot[r & 1], m_out);
                eb = eb_rot[r & 1] = dpp_keep<DPP_ROW_SHR1>(eb_rot[r & 1], b_out);
            } else {
                const uint32_t u0 = dpp_keep<DPP_WAVE_SHR1>(fresh, tok);
                const uint32_t u1 = em_rot[r & 1] = dpp_keep<DPP_WAVE_SHR1>(em_rot[r & 1], m_out);
                const uint32_t u2 = eb_rot[r & 1] = dpp_keep<DPP_WAVE_SHR1>(eb_rot[r & 1], b_out);
                if (Gs == 32) { // lane 32 starts a group too
                    tok = leader ? fresh : u0;
                    em = leader ? Z : u1;
                    eb = leader ? Z : u2;
                } else {
                    tok = u0;
                    em = u1;
                    eb = u2;
                }
            }
            if (r == 0) T0 = tp[0];
            else if (r == 1) T1 = tp[1];
            else if (r == 2) T2 = tp[2];
            else T3 = tp[3];
            const bool special = __builtin_amdgcn_ballot_w64(tok > 0xFFFFu) != 0ull;
            if (special) {
                uint32_t fm = 0u - ((tok >> 16) & 1u);
                asm volatile("" : "+v"(fm)); // (keeps this a branch)
                // (no wipe test: a batch runs on these cells only when no query of it can score 4096, let alone 32768)
                cells.best = (cells.best & ~fm) | (Z & fm);
                go_v = (go_v & ~fm) | (SWG_F16_BIG & fm);
                ge_v = (ge_v & ~fm) | (SWG_F16_BIG & fm);
            }
            const uint2 e = cells.template row<(K > 4)>(prof_addr<0>(base, tok), em, eb, go_v, ge_v, zero_v);
            if (special) {
                go_v = p.go;
                ge_v = p.ge;
                // the sequence's two scores (one per query of the pair) travel along its last row: see swg_diag_dyn_kernel
                {
                    uint32_t cin;
                    if (G16) {
                        cin = dpp_keep<DPP_ROW_SHR1>(Z, bc);
                    } else {
                        const uint32_t u3 = dpp_keep<DPP_WAVE_SHR1>(Z, bc);
                        cin = (GW == 32 && leader) ? Z : u3;
                    }
                    bc = pk_max3_f16(cin, cells.best, cells.best);
                }
                if (tail && (tok & SWG_TOK_LAST) != 0u) {
                    const uint32_t seq = record()[SWG_DYN_RING + (nlast & (SWG_DYN_RING - 1u))];
                    if (seq < p.seq_limit) {
                        scores_a[seq] = f16_score(f16_key(bc & 0xFFFFu));
                        if (have_b) scores_b[seq] = f16_score(f16_key(bc >> 16));
                    }
                    ++nlast;
                }
            }
            m_out = e.x;
            b_out = e.y;
        }
        tp += tstep;
        ++blocks;
        if (!hot) take_turn();
    }
    };
    if (G == 16) main_loop(std::integral_constant<int, 16>());
    else if (G == 32) main_loop(std::integral_constant<int, 32>());
    else main_loop(std::integral_constant<int, 64>());
}

#if SWG_HAS_PART(0)
// ---------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------
__global__ void swg_build_profile_kernel(const int8_t *sub, const int8_t *query, uint32_t lq,
                                         uint32_t ncols, int elem_size, uint32_t ch, uint32_t k_real,
                                         uint32_t k_padded, uint32_t swizzle_lanes, int f16, uint32_t qcol0, uint8_t *out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; // one (layout column, code)
    if (t >= ncols * 32u) return;
    const uint32_t col = t >> 5, code = t & 31u;
    // a lane's slice is k_padded layout columns of which the first k_real are query columns (qcol0: the query
    // column of layout column 0 -- the last pass of a long query may have a geometry of its own)
    const uint32_t j = col % k_padded, qcol = qcol0 + (col / k_padded) * k_real + j;
    const bool pad = (j >= k_real) || (qcol >= lq) || (code == 0u);
    const int v = pad ? 0 : (int)sub[(int)query[qcol] * 32 + (int)code];
    // row of this residue inside its chunk: the residue itself, or swizzled by the reading lane (SWG_LDS_SWIZZLE)
    const uint32_t row = swizzle_lanes ? code ^ (((col / k_padded) % swizzle_lanes) & 31u) : code;
    const size_t e = (size_t)(col / ch) * (32u * ch) + row * ch + (col % ch); // [col/ch][32][ch]
    if (elem_size == 2 && f16) // packed-f16 cells: the score as an f16 number, -65504 for padding
        reinterpret_cast<_Float16 *>(out)[e] = pad ? (_Float16)-65504.0f : (_Float16)(float)v;
    else if (elem_size == 2)
        reinterpret_cast<int16_t *>(out)[e] = pad ? (int16_t)-32768 : (int16_t)v;
    else
        reinterpret_cast<int32_t *>(out)[e] = pad ? -(1 << 29) : v;
}

// The same for several queries at once (swg_search_multi): grid.y = query; int16, 4-column chunks.
__global__ void swg_build_profiles_multi_kernel(const int8_t *sub, const int8_t *queries, const uint32_t *q_off,
                                                uint32_t ncols, uint32_t k_real, uint32_t k_padded, uint32_t swizzle_lanes,
                                                int f16, uint8_t *out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; // one (layout column, code)
    if (t >= ncols * 32u) return;
    const uint32_t qi = blockIdx.y;
    const int8_t *query = queries + q_off[qi];
    const uint32_t lq = q_off[qi + 1u] - q_off[qi];
    const uint32_t col = t >> 5, code = t & 31u;
    const uint32_t j = col % k_padded, qcol = (col / k_padded) * k_real + j;
    const bool pad = (j >= k_real) || (qcol >= lq) || (code == 0u);
    const int v = pad ? 0 : (int)sub[(int)query[qcol] * 32 + (int)code];
    const uint32_t row = swizzle_lanes ? code ^ (((col / k_padded) % swizzle_lanes) & 31u) : code;
    const size_t e = (size_t)(col / 4u) * 128u + row * 4u + (col % 4u); // [col/4][32][4]
    if (f16) reinterpret_cast<_Float16 *>(out + (size_t)qi * ncols * 64u)[e] = pad ? (_Float16)-65504.0f : (_Float16)(float)v;
    else reinterpret_cast<int16_t *>(out + (size_t)qi * ncols * 64u)[e] = pad ? (int16_t)-32768 : (int16_t)v;
}

// Profiles of query PAIRS for swg_diag_qq_kernel: grid.y = pair; entry (column, residue) = the two queries' scores as
// f16 numbers side by side (-65504 where a query has no such column), [col/2][32 residues][2 columns] x 4 bytes, rows
// swizzled by the reading lane like the other lane-group profiles.  Pair y = queries 2y and 2y+1 of `order` (an odd
// batch's last pair holds its query twice).
__global__ void swg_build_profiles_qq_kernel(const int8_t *sub, const int8_t *queries, const uint32_t *q_off, const uint32_t *order,
                                             uint32_t n_queries, uint32_t ncols, uint32_t k_real, uint32_t k_padded,
                                             uint32_t swizzle_lanes, uint8_t *out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; // one (layout column, code)
    if (t >= ncols * 32u) return;
    const uint32_t col = t >> 5, code = t & 31u;
    const uint32_t j = col % k_padded, qcol = (col / k_padded) * k_real + j;
    uint32_t word = 0u;
    for (uint32_t h = 0; h < 2u; ++h) {
        const uint32_t qi = order[min(2u * blockIdx.y + h, n_queries - 1u)];
        const int8_t *query = queries + q_off[qi];
        const uint32_t lq = q_off[qi + 1u] - q_off[qi];
        const bool pad = (j >= k_real) || (qcol >= lq) || (code == 0u);
        const _Float16 v = pad ? (_Float16)-65504.0f : (_Float16)(float)sub[(int)query[qcol] * 32 + (int)code];
        word |= (uint32_t)__builtin_bit_cast(unsigned short, v) << (16u * h);
    }
    const uint32_t row = swizzle_lanes ? code ^ (((col / k_padded) % swizzle_lanes) & 31u) : code;
    const size_t e = (size_t)(col / 2u) * 64u + row * 2u + (col % 2u); // [col/2][32][2]
    reinterpret_cast<uint32_t *>(out + (size_t)blockIdx.y * ncols * 128u)[e] = word;
}

hipError_t swg_launch_build_profiles_qq(const int8_t *d_sub, const int8_t *d_queries, const uint32_t *d_q_off, const uint32_t *d_order,
                                        uint32_t n_queries, uint32_t ncols, int k_real, int k_padded, uint8_t *d_profiles,
                                        hipStream_t stream, int swizzle_lanes)
{
    if (n_queries == 0 || ncols == 0) return hipSuccess;
    hipLaunchKernelGGL(swg_build_profiles_qq_kernel, dim3((ncols * 32u + 255u) / 256u, (n_queries + 1u) / 2u), dim3(256), 0, stream,
                       d_sub, d_queries, d_q_off, d_order, n_queries, ncols, (uint32_t)k_real, (uint32_t)k_padded,
                       (uint32_t)swizzle_lanes, d_profiles);
    return hipGetLastError();
}

hipError_t swg_launch_build_profiles_multi(const int8_t *d_sub, const int8_t *d_queries, const uint32_t *d_q_off,
                                           uint32_t n_queries, uint32_t ncols, int k_real, int k_padded,
                                           uint8_t *d_profiles, hipStream_t stream, int swizzle_lanes, int f16)
{
    if (n_queries == 0 || ncols == 0) return hipSuccess;
    hipLaunchKernelGGL(swg_build_profiles_multi_kernel, dim3((ncols * 32u + 255u) / 256u, n_queries), dim3(256), 0, stream,
                       d_sub, d_queries, d_q_off, ncols, (uint32_t)k_real, (uint32_t)k_padded, (uint32_t)swizzle_lanes,
                       f16, d_profiles);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// per-database layouts, built on the device from the uploaded residue bytes
// ---------------------------------------------------------------------------
// What goes over PCIe is one byte per residue (every sequence a run of whole dwords, the rest of
// its last dword holding the padding residue 0) and three words per sequence.  The layouts the
// fill kernels read are made from that here, at HBM speed, instead of on the host.
//
// Pair tokens (diagonal engine): thread b writes 4-row block b of the pair-major token array, one
// 32-bit token per row (see SWG_TOK_*).  Block k of a pair holds the rows 4k .. 4k+3 of its token
// stream = two reset rows, then one row per residue of the longer sequence X (row r: residue r - 2),
// i.e. residues 4k-2 .. 4k+1: the upper half of residue dword k-1 and the lower half of dword k.
// The last-row flag goes on the row of X's last residue -- for an empty pair on the second reset
// row, so that every pair is finished by the tail lane exactly once.
__global__ void swg_build_tokens_kernel(const uint32_t *codes, const uint64_t *code_off, const uint32_t *lens,
                                        const uint32_t *pair_off, uint32_t n_pairs, uint4 *tok)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_pairs == 0u || b >= (uint64_t)pair_off[n_pairs]) return;
    uint32_t lo = 0u, hi = n_pairs; // pair_off[lo] <= b < pair_off[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if ((uint64_t)pair_off[mid] <= b) lo = mid; else hi = mid;
    }
    const uint32_t k = (uint32_t)(b - pair_off[lo]);
    const uint32_t sx = 2u * lo, sy = sx + 1u;
    const uint32_t lx = lens[sx], ly = lens[sy];
    const uint32_t *cx = codes + code_off[sx], *cy = codes + code_off[sy];
    const uint32_t ndx = (lx + 3u) / 4u, ndy = (ly + 3u) / 4u;
    const uint32_t x1 = k < ndx ? cx[k] : 0u, x0 = (k >= 1u && k - 1u < ndx) ? cx[k - 1u] : 0u;
    const uint32_t y1 = k < ndy ? cy[k] : 0u, y0 = (k >= 1u && k - 1u < ndy) ? cy[k - 1u] : 0u;
    const uint32_t xw = (x1 << 16) | (x0 >> 16), yw = (y1 << 16) | (y0 >> 16); // residues 4k-2 .. 4k+1, one byte each
    uint32_t t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = ((xw >> (8 * r)) & 0xFFu) | (((yw >> (8 * r)) & 0xFFu) << 8);
    if (k == 0u) {
        t[0] |= SWG_TOK_RESET;
        t[1] |= SWG_TOK_RESET | SWG_TOK_RESET2;
    }
    const uint32_t last = lx + 1u; // row of X's last residue
    if (last / 4u == k) t[last & 3u] |= SWG_TOK_LAST;
    tok[b] = make_uint4(t[0], t[1], t[2], t[3]);
}

// The same image from reference-shaped 16-lane batches (swg_fill_batches16): the two sequences of a pair are two
// adjacent lanes of one batch, so a row's two residues are two adjacent bytes of the batch as the caller holds it.
__global__ void swg_build_tokens16_kernel(const uint8_t *stage, const uint64_t *pair_src, const uint32_t *pair_len,
                                          const uint32_t *pair_off, uint32_t n_pairs, uint4 *tok, uint32_t *bad)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_pairs == 0u || b >= (uint64_t)pair_off[n_pairs]) return;
    uint32_t lo = 0u, hi = n_pairs; // pair_off[lo] <= b < pair_off[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if ((uint64_t)pair_off[mid] <= b) lo = mid; else hi = mid;
    }
    const uint32_t k = (uint32_t)(b - pair_off[lo]);
    const uint64_t src = pair_src[lo] & ~(1ull << 63);
    const bool has_y = (pair_src[lo] >> 63) == 0ull;
    const uint32_t len = pair_len[lo];
    uint32_t t[4];
    bool wrong = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t row = 4u * k + (uint32_t)r; // rows 0, 1: reset rows; row r: residue r - 2
        t[r] = 0u;
        if (row >= 2u && row - 2u < len) {
            const uint8_t *q = stage + src + (uint64_t)(row - 2u) * 16u;
            const uint32_t x = q[0], y = has_y ? q[1] : 0u;
            wrong |= x < 1u || x > 31u || (has_y && (y < 1u || y > 31u));
            t[r] = ((x & 31u) << 3) | ((y & 31u) << 11);
        }
    }
    if (wrong) atomicOr(bad, 1u);
    if (k == 0u) {
        t[0] |= SWG_TOK_RESET;
        t[1] |= SWG_TOK_RESET | SWG_TOK_RESET2;
    }
    const uint32_t last = len + 1u;
    if (last / 4u == k) t[last & 3u] |= SWG_TOK_LAST;
    tok[b] = make_uint4(t[0], t[1], t[2], t[3]);
}

hipError_t swg_launch_build_tokens16(const uint8_t *d_stage, const uint64_t *d_pair_src, const uint32_t *d_pair_len,
                                     const uint32_t *d_pair_off, uint32_t n_pairs, uint64_t total_blocks, uint4 *d_tok,
                                     uint32_t *d_bad, hipStream_t stream)
{
    if (n_pairs == 0 || total_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(swg_build_tokens16_kernel, dim3((uint32_t)((total_blocks + 255) / 256)), dim3(256), 0, stream, d_stage,
                       d_pair_src, d_pair_len, d_pair_off, n_pairs, d_tok, d_bad);
    return hipGetLastError();
}

// Bin image (systolic engine, bin-based int32 kernel): one workgroup per bin, thread s = slot s of
// the bin; dword[blk*128 + SWG_BIN_COLUMN(s)] = residue dword blk of that sequence (0 past its end).
__global__ void swg_build_bins_kernel(const uint32_t *codes, const uint64_t *code_off, const uint32_t *lens,
                                      const uint64_t *bin_off, const uint32_t *bin_nblk, uint32_t *packed)
{
    const uint32_t bin = blockIdx.x, s = threadIdx.x;
    const uint32_t slot = bin * SWG_BIN + s;
    const uint32_t nd = (lens[slot] + 3u) / 4u, nblk = bin_nblk[bin];
    const uint32_t *c = codes + code_off[slot];
    uint32_t *out = packed + bin_off[bin] + SWG_BIN_COLUMN(s);
    for (uint32_t blk = 0; blk < nblk; ++blk) out[(size_t)blk * SWG_BIN] = blk < nd ? c[blk] : 0u;
}

hipError_t swg_launch_build_tokens(const uint32_t *d_codes, const uint64_t *d_code_off, const uint32_t *d_lens,
                                   const uint32_t *d_pair_off, uint32_t n_pairs, uint64_t total_blocks, uint4 *d_tok,
                                   hipStream_t stream)
{
    if (n_pairs == 0 || total_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(swg_build_tokens_kernel, dim3((uint32_t)((total_blocks + 255) / 256)), dim3(256), 0, stream,
                       d_codes, d_code_off, d_lens, d_pair_off, n_pairs, d_tok);
    return hipGetLastError();
}

hipError_t swg_launch_build_bins(const uint32_t *d_codes, const uint64_t *d_code_off, const uint32_t *d_lens,
                                 const uint64_t *d_bin_off, const uint32_t *d_bin_nblk, uint32_t n_bins,
                                 uint32_t *d_packed, hipStream_t stream)
{
    if (n_bins == 0) return hipSuccess;
    hipLaunchKernelGGL(swg_build_bins_kernel, dim3(n_bins), dim3(SWG_BIN), 0, stream, d_codes, d_code_off, d_lens,
                       d_bin_off, d_bin_nblk, d_packed);
    return hipGetLastError();
}

__global__ void swg_collect_saturated_kernel(const int32_t *scores, uint32_t n, int32_t ceiling, uint32_t *list,
                                             uint32_t *count, const uint32_t *lens, uint32_t *rows16)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && scores[i] >= ceiling) {
        list[atomicAdd(count, 1u)] = i;
        if (rows16) atomicAdd(rows16, (lens[i] + 15u) / 16u); // rows the re-score will walk, in units of 16
    }
}

#endif // part 0

// ---------------------------------------------------------------------------
// The int32 diagonal fill: 64 lanes share ONE sequence, exact recurrence
// ---------------------------------------------------------------------------
// Used for the sequences the int16 fill flagged as saturated (typically a handful
// of very long, very similar ones: exactly the shape that needs the shortest chain
// per row), for gap scores the packed form cannot express, and when forced.  A
// workgroup takes W consecutive entries of the work list, one per wavefront; all
// of them walk the query pass by pass (the int32 profile slice of a pass is shared
// in LDS), lane g holding K columns, the leader lane turning the bins' residue
// dwords into tokens on the fly.
template <int K> struct CellsDiag32 {
    static constexpr int CHUNK = 512; // 32 residues x 4 columns x int32
    int U[K], A[K], D[K];
    int best, ddl;

    DEVINL void reset()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) U[k] = A[k] = D[k] = 0;
        best = 0;
        ddl = 0;
    }

    DEVINL void row(const uint8_t *prof, uint32_t off, int el, int eb, int ed, int go, int ge, int &ol,
                    int &ob, int &od)
    {
        int dd = ddl;
        int ll = el, bl = eb;
#pragma unroll
        for (int c = 0; c < K / 4; ++c) {
            const int4 sv = *reinterpret_cast<const int4 *>(prof + off + c * CHUNK);
            const int s[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * c + u;
                const int h = imax(dd + s[u], 0);             // src/alignment.c:124-129
                const int a = imax3(U[k] + go, A[k] + ge, 0); // src/alignment.c:142-147
                const int b = imax3(ll + go, bl + ge, 0);     // src/alignment.c:156-161
                dd = D[k];
                U[k] = imax(h, b);
                ll = imax(h, a);
                D[k] = imax(ll, b);
                A[k] = a;
                bl = b;
                best = imax(best, h);                         // src/alignment.c:133
            }
        }
        ddl = ed;
        ol = ll;
        ob = bl;
        od = D[K - 1];
    }
};

DEVINL int dpp_wave_shr1(int keep, int src)
{
    return __builtin_amdgcn_update_dpp(keep, src, DPP_WAVE_SHR1, 0xf, 0xf, false);
}

#define SWG_PAD32 (-(1 << 29))

template <int K, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void swg_diag32_kernel(const SwgFillParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; // int32 profile slice + 2 words
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);
    const bool leader = lane == 0, tail = lane == 63;
    const uint32_t slice = 64u * K * 128u; // bytes of one pass: 64*K columns x 32 residues x 4 B
    uint32_t *wgword = reinterpret_cast<uint32_t *>(smem + slice);
    const uint32_t n_items = p.list_count ? *p.list_count : p.n_items; // sequences to score
    const uint32_t base = (uint32_t)lane * (K / 4) * 512u;
    const int npass = (int)p.npass;
    uint4 *sp = reinterpret_cast<uint4 *>(p.scratch) +
                ((size_t)blockIdx.x * W + w) * (p.scratch_wg_dwords / 4); // rows of this wavefront

    for (;;) {
        // ---- next batch of W sequences for this workgroup ----------------------
        __syncthreads();
        if (threadIdx.x == 0) wgword[0] = atomicAdd(p.queue, 1u);
        __syncthreads();
        const uint32_t batch = wgword[0];
        if ((uint64_t)batch * W >= n_items) break;
        const uint32_t idx = batch * W + w;
        const bool valid = idx < n_items;
        uint32_t rank = 0, nblk = 0;
        const uint32_t *rp = p.residues;
        if (valid) {
            rank = p.list ? p.list[idx] : idx;
            const uint32_t b = rank / SWG_BIN;
            nblk = rank < p.n_bins * SWG_BIN ? p.bin_nblk[b] : 0u;
            rp = p.residues + (nblk ? p.bin_off[b] : 0) + SWG_BIN_COLUMN(rank % SWG_BIN);
        }
        nblk = __builtin_amdgcn_readfirstlane(nblk);
        const uint32_t rows = nblk * 4u;
        // stream of this wavefront: block 0 = (pad, pad, pad, reset), blocks 1..nblk = residues
        const uint32_t nsteps = nblk ? (nblk + 1u) * 4u + 64u : 0u;

        for (int pass = 0; pass < npass; ++pass) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            {
                const uint8_t *src = p.profile + (size_t)pass * slice;
                for (uint32_t o = threadIdx.x * 16u; o < slice; o += blockDim.x * 16u)
                    *reinterpret_cast<uint4 *>(smem + o) = *reinterpret_cast<const uint4 *>(src + o);
            }
            __syncthreads();

            CellsDiag32<K> cells;
            cells.reset();
            int tok = 0, o_l = 0, o_b = 0, o_d = 0, c_out = 0;
            uint32_t cur = 0u, nxt = (leader && nblk > 0u) ? rp[0] : 0u;
            for (uint32_t s4 = 0; s4 < nsteps; s4 += 4u) {
                const uint32_t blk = s4 / 4u; // block 0 is the reset block
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int fresh;
                    if (blk == 0u)
                        fresh = r == 3 ? (int)SWG_TOK_RESET : 0;
                    else
                        fresh = (int)((cur >> (8 * r)) & 0xF8u) | ((blk == nblk && r == 3) ? (int)SWG_TOK_LAST : 0);
                    // edge of the column left of the query: zero, or what the tail lane
                    // spilled for this row in the previous pass
                    int lm = 0, lb = 0, ld = 0;
                    const uint32_t lrow = s4 + (uint32_t)r; // stream row handled by the leader now
                    if (pass > 0 && leader && lrow < rows + 4u) {
                        const uint4 v = load_edge_l2(sp + lrow);
                        lm = (int)v.x;
                        lb = (int)v.y;
                        ld = (int)v.z;
                    }
                    tok = dpp_wave_shr1(fresh, tok);
                    const int el = dpp_wave_shr1(lm, o_l);
                    const int eb = dpp_wave_shr1(lb, o_b);
                    const int ed = dpp_wave_shr1(ld, o_d);
                    const int cin = dpp_wave_shr1(0, c_out);
                    const bool rst = (tok & (int)SWG_TOK_RESET) != 0;
                    if (rst) cells.best = 0;
                    cells.row(smem, base + (((uint32_t)tok & 0xF8u) << 1), el, eb, ed, rst ? SWG_PAD32 : p.go,
                              rst ? SWG_PAD32 : p.ge, o_l, o_b, o_d);
                    c_out = imax(cin, cells.best);
                    if (tail) {
                        if ((tok & (int)SWG_TOK_LAST) && valid) atomicMax(p.scores + rank, c_out);
                        const uint32_t trow = lrow - 63u; // stream row handled by the tail lane now
                        if (pass + 1 < npass && trow < rows + 4u)
                            sp[trow] = make_uint4((uint32_t)o_l, (uint32_t)o_b, (uint32_t)o_d, 0u);
                    }
                }
                cur = nxt;
                nxt = (leader && blk + 1u < nblk) ? rp[(size_t)(blk + 1u) * SWG_BIN] : 0u;
            }
        }
    }
}

#if SWG_HAS_PART(0)
// ---------------------------------------------------------------------------
// device top-K: histogram -> threshold -> compaction of the few candidates
// ---------------------------------------------------------------------------
// Hits are ordered by (score desc, original index asc) = descending 64-bit key.
// Scores are small integers, so a 4096-bin histogram finds the score T of the
// K-th best hit; every entry with score >= T (K plus ties at T) is appended to a
// candidate list that the host sorts.  Scores >= 4095 share the last bin; if the
// threshold falls there, or the candidates do not fit, `status` tells the host
// to fall back to reading all scores.
#define SWG_TOPK_BINS 4096

// (blockIdx.y = query of a batch, swg_search_multi: its score row at scores + y * score_stride, its histogram at
// hist + y * SWG_TOPK_BINS, its threshold / status / count words at meta + y * 4 and its candidates at cand + y * cap;
// a single search is a grid of one row with its own pointers)
__global__ void swg_topk_hist_kernel(const int32_t *scores, const uint32_t *order, uint32_t n, uint32_t *hist, uint64_t score_stride)
{
    scores += (size_t)blockIdx.y * score_stride;
    hist += (size_t)blockIdx.y * SWG_TOPK_BINS;
    // These few wavefronts run BESIDE the next search's fill (their own stream): at the default priority the
    // persistent fill wavefronts starve them for milliseconds; at the top one they are done in microseconds.
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint32_t h[SWG_TOPK_BINS];
    for (int i = threadIdx.x; i < SWG_TOPK_BINS; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (order[i] == 0xFFFFFFFFu) continue;
        int v = scores[i];
        v = v < 0 ? 0 : (v > SWG_TOPK_BINS - 1 ? SWG_TOPK_BINS - 1 : v);
        atomicAdd(&h[v], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SWG_TOPK_BINS; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// One block of 256 threads, 16 bins each: out[0] = threshold score T, out[1] = status (0 ok, 1 fall back).
// It runs beside the NEXT search's fill on its own stream, and a fill at K=32 leaves every SIMD's register
// file 8 registers per lane short of full: a kernel that needs more cannot be placed until a fill workgroup
// retires (round 1's form, 20 registers, "took" 3.8-5 ms on config 3; the histogram and compaction kernels
// beside it, 8 and 6 registers, 20 us).  Hence the one-at-a-time loops and the register cap.
#define SWG_TOPK_THR_THREADS 256
__global__ __launch_bounds__(SWG_TOPK_THR_THREADS) __attribute__((amdgpu_num_vgpr(8))) void
swg_topk_threshold_kernel(const uint32_t *hist, uint32_t k, uint32_t cap, uint32_t *out)
{
    hist += (size_t)blockIdx.x * SWG_TOPK_BINS; // (one block per query of a batch)
    out += (size_t)blockIdx.x * 4u;
    __builtin_amdgcn_s_setprio(3);
    constexpr int PER = SWG_TOPK_BINS / SWG_TOPK_THR_THREADS;
    __shared__ uint32_t part[SWG_TOPK_THR_THREADS];
    const int t = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll 1
    for (int j = 0; j < PER; ++j) sum += hist[PER * t + j];
    part[t] = sum;
    __syncthreads();
    // suffix sums: part[t] = entries in bins >= PER * t
#pragma unroll 1
    for (int d = 1; d < SWG_TOPK_THR_THREADS; d <<= 1) {
        const uint32_t add = (t + d < SWG_TOPK_THR_THREADS) ? part[t + d] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    const uint32_t total = part[0];
    const uint32_t want = k < total ? k : total;
    const uint32_t above = (t + 1 < SWG_TOPK_THR_THREADS) ? part[t + 1] : 0u; // entries in bins >= PER * (t + 1)
    if (want > 0 && above < want && part[t] >= want) {
        uint32_t acc = above;
        int T = PER * t;
#pragma unroll 1
        for (int j = PER - 1; j >= 0; --j) {
            acc += hist[PER * t + j];
            if (acc >= want) {
                T = PER * t + j;
                break;
            }
        }
        out[0] = (uint32_t)T;
        out[1] = (T == SWG_TOPK_BINS - 1 || acc > cap) ? 1u : 0u;
    }
    if (want == 0 && t == 0) {
        out[0] = 0u;
        out[1] = 0u;
    }
}

__global__ void swg_topk_compact_kernel(const int32_t *scores, const uint32_t *order, uint32_t n,
                                        const uint32_t *thr, uint64_t *cand, uint32_t cap, uint32_t *count, uint64_t score_stride)
{
    scores += (size_t)blockIdx.y * score_stride;
    thr += (size_t)blockIdx.y * 4u;
    count += (size_t)blockIdx.y * 4u;
    cand += (size_t)blockIdx.y * cap;
    // These few wavefronts run BESIDE the next search's fill (their own stream): at the default priority the
    // persistent fill wavefronts starve them for milliseconds; at the top one they are done in microseconds.
    __builtin_amdgcn_s_setprio(3);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || thr[1] != 0u) return;
    const uint32_t oi = order[i];
    if (oi == 0xFFFFFFFFu) return;
    const int v = scores[i];
    if (v >= (int)thr[0]) {
        const uint32_t at = atomicAdd(count, 1u);
        if (at < cap) cand[at] = ((uint64_t)(uint32_t)(v < 0 ? 0 : v) << 32) | (uint64_t)(0xFFFFFFFFu - oi);
    }
}

hipError_t swg_launch_topk(const int32_t *d_scores, const uint32_t *d_order, uint32_t n_slots, uint32_t k,
                           uint32_t *d_hist, uint32_t *d_thr, uint64_t *d_cand, uint32_t cap, uint32_t *d_count,
                           hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_hist, 0, SWG_TOPK_BINS * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const int blocks = (int)((n_slots + 255) / 256 < 512 ? (n_slots + 255) / 256 : 512);
    hipLaunchKernelGGL(swg_topk_hist_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, d_scores, d_order,
                       n_slots, d_hist, (uint64_t)0);
    hipLaunchKernelGGL(swg_topk_threshold_kernel, dim3(1), dim3(SWG_TOPK_THR_THREADS), 0, stream, d_hist, k, cap, d_thr);
    hipLaunchKernelGGL(swg_topk_compact_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, stream, d_scores,
                       d_order, n_slots, d_thr, d_cand, cap, d_count, (uint64_t)0);
    return hipGetLastError();
}

// The same for the n_queries score rows of a batch (row y at d_scores + y * score_stride) in three launches:
// d_hist[n_queries][4096], d_meta[n_queries][4] = {threshold, status (0 ok / 1 fall back), candidates, -},
// d_cand[n_queries][cap].
hipError_t swg_launch_topk_multi(const int32_t *d_scores, uint64_t score_stride, const uint32_t *d_order, uint32_t n_slots,
                                 uint32_t n_queries, uint32_t k, uint32_t *d_hist, uint32_t *d_meta, uint64_t *d_cand, uint32_t cap,
                                 hipStream_t stream)
{
    if (n_queries == 0 || n_queries > 65535) return n_queries ? hipErrorInvalidValue : hipSuccess;
    hipError_t e = hipMemsetAsync(d_hist, 0, (size_t)n_queries * SWG_TOPK_BINS * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(d_meta, 0, (size_t)n_queries * 16, stream)) != hipSuccess) return e;
    const int blocks = (int)((n_slots + 255) / 256 < 64 ? (n_slots + 255) / 256 : 64);
    hipLaunchKernelGGL(swg_topk_hist_kernel, dim3(blocks > 0 ? blocks : 1, n_queries), dim3(256), 0, stream, d_scores, d_order,
                       n_slots, d_hist, score_stride);
    hipLaunchKernelGGL(swg_topk_threshold_kernel, dim3(n_queries), dim3(SWG_TOPK_THR_THREADS), 0, stream, d_hist, k, cap, d_meta);
    hipLaunchKernelGGL(swg_topk_compact_kernel, dim3((n_slots + 255) / 256, n_queries), dim3(256), 0, stream, d_scores, d_order,
                       n_slots, d_meta, d_cand, cap, d_meta + 2, score_stride);
    return hipGetLastError();
}

#endif // part 0

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
typedef void (*SwgDynKernel)(const SwgDiagDynParams);
typedef void (*SwgQ32Kernel)(const SwgDiagQ32Params);
typedef void (*SwgQQKernel)(const SwgDiagQQParams);
// kernel of a variant, from the part that instantiates it: which = 0 single pass / 1 one pass of several / 2 the same,
// wide form (int16 cells); 0 single pass / 1 one pass of several (f16 cells)
SwgDynKernel swg_dyn_kernel_i16(int variant, int which);
SwgDynKernel swg_dyn_kernel_f16(int variant, int which);

#if SWG_HAS_PART(1)
SwgDynKernel swg_dyn_kernel_i16(int variant, int which)
{
#define SWG_ROW(K, W) {swg_diag_dyn_kernel<K, W, false, 0>, swg_diag_dyn_kernel<K, W, true, 0>, swg_diag_dyn_kernel<K, W, true, 1>},
    static const SwgDynKernel t[][3] = {SWG_DIAG_VARIANTS(SWG_ROW)};
#undef SWG_ROW
    return t[variant][which];
}
#endif

#if SWG_HAS_PART(2)
SwgDynKernel swg_dyn_kernel_f16(int variant, int which)
{
#define SWG_ROW(K, W) {swg_diag_dyn_kernel<K, W, false, 2>, swg_diag_dyn_kernel<K, W, true, 2>},
    static const SwgDynKernel t[][2] = {SWG_DIAG_VARIANTS(SWG_ROW)};
#undef SWG_ROW
    return t[variant][which];
}
#endif

#if SWG_HAS_PART(3)
// (the exact cells hold more state per row: above 16 columns they are compiled for 12 wavefronts per CU, 170 registers)
template <int K, int MAXW> struct SwgX32Waves {
    static constexpr int value = K > 16 && MAXW > SWG_X32_WAVES_ABOVE16 ? SWG_X32_WAVES_ABOVE16 : MAXW;
};
hipError_t swg_launch_diag32q(int variant, bool edges, bool exact, int W, int workgroups, const SwgDiagQ32Params &p, hipStream_t stream)
{
#define SWG_ROW(K, W)                                                                                                              \
    {swg_diag32q_kernel<K, W, false>, swg_diag32q_kernel<K, W, true>, swg_diag32q_kernel<K, SwgX32Waves<K, W>::value, false, true>, \
     swg_diag32q_kernel<K, SwgX32Waves<K, W>::value, true, true>},
    static const SwgQ32Kernel t[][4] = {SWG_DIAG_VARIANTS(SWG_ROW)};
#undef SWG_ROW
    if (variant < 0 || variant >= swg_num_diag_variants() || W < 1 || W > swg_diag_variant_info(variant).max_waves || workgroups < 1 ||
        (p.G != 16 && p.G != 32 && p.G != 64) || p.q_end < p.q_begin)
        return hipErrorInvalidValue;
    const size_t lds = swg_diag32q_lds_bytes(swg_diag_variant_info(variant).K, (int)p.G, W);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto k = t[variant][(exact ? 2 : 0) + (edges ? 1 : 0)];
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(workgroups), dim3(W * 64), lds, stream, p);
    return hipGetLastError();
}
#endif

#if SWG_HAS_PART(4)
// (the query pairs' profile has the int32 profile's size: 4 bytes per column and residue)
hipError_t swg_launch_diag_qq(int variant, int W, int workgroups, int n_pairs, const SwgDiagQQParams &p, hipStream_t stream)
{
#define SWG_ROW(K, W) swg_diag_qq_kernel<K, W>,
    static const SwgQQKernel t[] = {SWG_DIAG_VARIANTS(SWG_ROW)};
#undef SWG_ROW
    if (variant < 0 || variant >= swg_num_diag_variants() || W < 1 || W > swg_diag_variant_info(variant).max_waves || workgroups < 1 ||
        n_pairs < 1 || n_pairs > 65535 || (p.G != 16 && p.G != 32 && p.G != 64) || p.q_end < p.q_begin)
        return hipErrorInvalidValue;
    const size_t lds = swg_diag32q_lds_bytes(swg_diag_variant_info(variant).K, (int)p.G, W);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto k = t[variant];
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(workgroups, n_pairs), dim3(W * 64), lds, stream, p);
    return hipGetLastError();
}
#endif

#if SWG_HAS_PART(0)
namespace {
struct Variant {
    SwgKernelInfo info;
    void (*kernel)(const SwgFillParams);
    void (*kernel_f16)(const SwgFillParams); // the same geometry on packed-f16 cells (int16 table only), or null
};

template <class Cells, int K, int MAXW, int BITS> Variant make_variant()
{
    Variant v;
    v.info.bits = BITS;
    v.info.K = K;
    v.info.max_waves = MAXW;
    v.info.nb = (int)(sizeof(typename Cells::edge_t) / 4);
    v.info.elem_size = Cells::ESZ;
    v.info.lds_per_wave = Cells::SLICE + 2 * SWG_ROWS_PER_BLK * 64 * sizeof(typename Cells::edge_t);
    v.info.lds_fixed = 2 * SWG_ROWS_PER_BLK * 64 * sizeof(typename Cells::edge_t) + (SWG_ITEM_RING + 2) * 4;
    v.kernel = swg_fill_kernel<Cells, K, MAXW>;
    v.kernel_f16 = nullptr;
    return v;
}
template <int K, int MAXW> Variant make_variant16()
{
    Variant v = make_variant<CellsI16<K>, K, MAXW, 16>();
    v.kernel_f16 = swg_fill_kernel<CellsSF16<K>, K, MAXW>;
    return v;
}

const Variant *variants16(int *n)
{
    static const Variant v[] = {
        make_variant16<32, 12>(),
        make_variant16<16, 16>(),
        make_variant16<48, 8>(),
        make_variant16<24, 16>(),
    };
    *n = (int)(sizeof(v) / sizeof(v[0]));
    return v;
}
const Variant *variants32(int *n)
{
    static const Variant v[] = {
        make_variant<CellsI32<32>, 32, 12, 32>(),
        make_variant<CellsI32<16>, 16, 16, 32>(),
    };
    *n = (int)(sizeof(v) / sizeof(v[0]));
    return v;
}
const Variant *variants(int bits, int *n) { return bits == 16 ? variants16(n) : variants32(n); }

// the fixed-stream diagonal fill: [0] single pass, [1] multi-pass, [2] multi-pass wide form (also runs one pass)
typedef void (*DiagKernel)(const SwgDiagParams);
struct DiagVariant {
    int K, max_waves;
    DiagKernel kernel[3];
};
const DiagVariant *diag_variants(int *n)
{
#define SWG_ROW(K, W) {K, W, {swg_diag_kernel<K, W, false>, swg_diag_kernel<K, W, true>, swg_diag_kernel<K, W, true, true>}},
    static const DiagVariant v[] = {SWG_DIAG_VARIANTS(SWG_ROW)};
#undef SWG_ROW
    *n = (int)(sizeof(v) / sizeof(v[0]));
    return v;
}
} // namespace

int swg_q32_padded_cols(int K) { return (K + 1) / 2 * 2; }

size_t swg_diag32q_lds_bytes(int K, int G, int W)
{
    return (size_t)G * swg_q32_padded_cols(K) * 128u + (size_t)W * (64 / G) * SWG_DYN_STATE * 4u;
}

int swg_num_diag_variants()
{
    int n;
    diag_variants(&n);
    return n;
}

SwgKernelInfo swg_diag_variant_info(int variant)
{
    int n;
    const DiagVariant &d = diag_variants(&n)[variant];
    SwgKernelInfo info;
    info.bits = 16;
    info.K = d.K;
    info.max_waves = d.max_waves;
    info.nb = 2;
    info.elem_size = 2;
    info.lds_per_wave = 0;
    info.lds_fixed = 0;
    return info;
}

hipError_t swg_launch_diag(int variant, bool multipass, bool wide, int W, int workgroups, size_t lds_bytes,
                           const SwgDiagParams &p, hipStream_t stream)
{
    int n;
    const DiagVariant *v = diag_variants(&n);
    if (variant < 0 || variant >= n || W < 1 || W > v[variant].max_waves || workgroups < 1 ||
        (p.G != 16 && p.G != 32 && p.G != 64))
        return hipErrorInvalidValue;
    auto k = wide ? v[variant].kernel[2] : v[variant].kernel[multipass ? 1 : 0];
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(workgroups), dim3(W * 64), lds_bytes, stream, p);
    return hipGetLastError();
}

int swg_diag_padded_cols(int K) { return (K + 3) / 4 * 4; }

size_t swg_diag_dyn_lds_bytes(int K, int G, int W)
{
    return (size_t)G * swg_diag_padded_cols(K) * 64u + (size_t)W * (64 / G) * SWG_DYN_STATE * 4u;
}

hipError_t swg_launch_diag_dyn(int variant, bool edges, int form, int W, int workgroups, const SwgDiagDynParams &p,
                               hipStream_t stream, int n_queries)
{
    if (n_queries < 1 || n_queries > 65535) return hipErrorInvalidValue;
    int n;
    const DiagVariant *v = diag_variants(&n);
    if (variant < 0 || variant >= n || W < 1 || W > v[variant].max_waves || workgroups < 1 ||
        (p.G != 16 && p.G != 32 && p.G != 64) || p.q_end < p.q_begin)
        return hipErrorInvalidValue;
    const size_t lds = swg_diag_dyn_lds_bytes(v[variant].K, (int)p.G, W);
    if (form < 0 || form > 2 || (form == 1 && !edges)) return hipErrorInvalidValue;
    auto k = form == 2 ? swg_dyn_kernel_f16(variant, edges ? 1 : 0) : swg_dyn_kernel_i16(variant, form == 1 ? 2 : edges ? 1 : 0);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(workgroups, n_queries), dim3(W * 64), lds, stream, p);
    return hipGetLastError();
}

hipError_t swg_launch_diag32(int W, int workgroups, const SwgFillParams &p, hipStream_t stream)
{
    constexpr int K = SWG_DIAG32_K, MAXW = 16;
    if (W < 1 || W > MAXW || workgroups < 1) return hipErrorInvalidValue;
    const size_t lds = 64u * K * 128u + 16;
    auto k = swg_diag32_kernel<K, MAXW>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(workgroups), dim3(W * 64), lds, stream, p);
    return hipGetLastError();
}

int swg_num_variants(int bits)
{
    int n;
    variants(bits, &n);
    return n;
}

SwgKernelInfo swg_variant_info(int bits, int variant)
{
    int n;
    const Variant *v = variants(bits, &n);
    return v[variant].info;
}

hipError_t swg_launch_fill(int bits, int variant, int W, int workgroups, const SwgFillParams &p,
                           hipStream_t stream, bool f16)
{
    int n;
    const Variant *v = variants(bits, &n);
    if (variant < 0 || variant >= n || W < 1 || W > v[variant].info.max_waves || workgroups < 1)
        return hipErrorInvalidValue;
    auto k = f16 ? v[variant].kernel_f16 : v[variant].kernel;
    if (!k) return hipErrorInvalidValue;
    const size_t lds = v[variant].info.lds_per_wave * (size_t)W + v[variant].info.lds_fixed;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(workgroups), dim3(W * 64), lds, stream, p);
    return hipGetLastError();
}

hipError_t swg_launch_build_profile(const int8_t *d_sub, const int8_t *d_query, uint32_t lq,
                                    uint32_t ncols, int elem_size, int chunk_cols, int k_real, int k_padded,
                                    uint8_t *d_profile, hipStream_t stream, int swizzle_lanes, int f16, uint32_t qcol0)
{
    const uint32_t n = ncols * 32u;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(swg_build_profile_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_sub,
                       d_query, lq, ncols, elem_size, (uint32_t)chunk_cols, (uint32_t)k_real, (uint32_t)k_padded,
                       (uint32_t)swizzle_lanes, f16, qcol0, d_profile);
    return hipGetLastError();
}

// zeroes two regions (sizes in 16-byte units) in ONE launch: a search's score array and its counters, which as
// two hipMemsetAsync were two runtime fill kernels plus the gaps between them ahead of every fill
__global__ void swg_zero2_kernel(uint4 *a, uint32_t na, uint4 *b, uint32_t nb)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    if (i < na) a[i] = z;
    else if (i - na < nb) b[i - na] = z;
}

hipError_t swg_launch_zero2(void *a, size_t a_bytes, void *b, size_t b_bytes, hipStream_t stream)
{
    if ((a_bytes | b_bytes) & 15u || ((a_bytes + b_bytes) >> 4) >= (1ull << 32)) return hipErrorInvalidValue;
    const uint32_t na = (uint32_t)(a_bytes >> 4), nb = (uint32_t)(b_bytes >> 4);
    if (na + nb == 0) return hipSuccess;
    hipLaunchKernelGGL(swg_zero2_kernel, dim3((na + nb + 255) / 256), dim3(256), 0, stream, static_cast<uint4 *>(a), na,
                       static_cast<uint4 *>(b), nb);
    return hipGetLastError();
}

// The same by pairs (ranks 2p, 2p+1 share a lane group of the 16-bit fill): pair p is listed once if either of its
// sequences reached the ceiling; *seqs counts the flagged sequences, *rows16 the rows of the listed pairs in
// units of 16.
__global__ void swg_collect_flagged_pairs_kernel(const int32_t *scores, uint32_t first_pair, uint32_t n_pairs, int32_t ceiling,
                                                 uint32_t *list, uint32_t *count, uint32_t *seqs, const uint32_t *lens, uint32_t *rows16)
{
    const uint32_t p = first_pair + blockIdx.x * blockDim.x + threadIdx.x; // (pairs before first_pair ran on other cells)
    if (p >= n_pairs) return;
    const uint32_t fx = scores[2u * p] >= ceiling, fy = scores[2u * p + 1u] >= ceiling;
    if (fx | fy) {
        list[atomicAdd(count, 1u)] = p;
        atomicAdd(seqs, fx + fy);
        atomicAdd(rows16, (lens[2u * p] + 17u) / 16u);
    }
}

hipError_t swg_launch_collect_flagged_pairs(const int32_t *d_scores, uint32_t first_pair, uint32_t n_pairs, int32_t ceiling,
                                            uint32_t *d_list, uint32_t *d_count, uint32_t *d_seqs, const uint32_t *d_lens,
                                            uint32_t *d_rows16, hipStream_t stream)
{
    if (n_pairs <= first_pair) return hipSuccess;
    hipLaunchKernelGGL(swg_collect_flagged_pairs_kernel, dim3((n_pairs - first_pair + 255) / 256), dim3(256), 0, stream, d_scores,
                       first_pair, n_pairs, ceiling, d_list, d_count, d_seqs, d_lens, d_rows16);
    return hipGetLastError();
}

hipError_t swg_launch_collect_saturated(const int32_t *d_scores, uint32_t n_slots, int32_t ceiling, uint32_t *d_list,
                                        uint32_t *d_count, const uint32_t *d_lens, uint32_t *d_rows16, hipStream_t stream)
{
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(swg_collect_saturated_kernel, dim3((n_slots + 255) / 256), dim3(256), 0,
                       stream, d_scores, n_slots, ceiling, d_list, d_count, d_lens, d_lens ? d_rows16 : nullptr);
    return hipGetLastError();
}
#endif // part 0
