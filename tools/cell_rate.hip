// Microbenchmark: cycles per two DP cells of the packed recurrence in its candidate forms, with the real
// dependency structure (K columns of register state, profile words from LDS, one row after another) but
// nothing else of the fill kernel.  What it answers: does the gfx950 three-operand packed f16 maximum
// (v_pk_maximum3_f16) turn the 10-instruction int16 cell into an 8.5-instruction one at the SAME cost per
// instruction?   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-sched-strategy=max-ilp -o cell_rate cell_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define DEVINL __device__ __forceinline__
#define BC(T, x) __builtin_bit_cast(T, x)

DEVINL uint32_t pk_add_i16_sat(uint32_t a, uint32_t b) { return BC(uint32_t, __builtin_elementwise_add_sat(BC(s16x2, a), BC(s16x2, b))); }
DEVINL uint32_t pk_sub_u16_sat(uint32_t a, uint32_t b) { return BC(uint32_t, __builtin_elementwise_sub_sat(BC(u16x2, a), BC(u16x2, b))); }
DEVINL uint32_t pk_max_i16(uint32_t a, uint32_t b) { return BC(uint32_t, __builtin_elementwise_max(BC(s16x2, a), BC(s16x2, b))); }
DEVINL uint32_t h_add(uint32_t a, uint32_t b) { return BC(uint32_t, BC(h2, a) + BC(h2, b)); }
DEVINL uint32_t h_sub(uint32_t a, uint32_t b) { return BC(uint32_t, BC(h2, a) - BC(h2, b)); }
DEVINL uint32_t h_max3(uint32_t a, uint32_t b, uint32_t c)
{
    return BC(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(BC(h2, a), BC(h2, b)), BC(h2, c)));
}
DEVINL uint2 lds_read_u2(uint32_t addr)
{
    const u32x2 v = *reinterpret_cast<__attribute__((address_space(3))) const u32x2 *>((uintptr_t)addr);
    return make_uint2(v.x, v.y);
}

// FORM 0: packed int16, 10 instructions per 2 cells (CellsDiag of swg_kernels.hip)
// FORM 1: packed f16 with three-operand maxima, 8.5 per 2 cells
// FORM 2: FORM 1 without the profile pairing (v_perm_b32): what a d16 LDS load form could reach (7.5)
template <int K, int FORM> struct Cells {
    static constexpr int NCH = K / 4;
    uint32_t M[K], G[K], A[K];
    uint32_t best, mdl;
    DEVINL void reset()
    {
        for (int k = 0; k < K; ++k) M[k] = G[k] = A[k] = 0u;
        best = mdl = 0u;
    }
    DEVINL uint2 row(uint32_t ax, uint32_t ay, uint32_t em, uint32_t eb, uint32_t go, uint32_t ge)
    {
        uint32_t md = mdl;
        uint32_t gl = FORM == 0 ? pk_sub_u16_sat(em, go) : h_sub(em, go);
        uint32_t bl = eb;
        uint2 nx = lds_read_u2(ax), ny = lds_read_u2(ay);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint2 wx = nx, wy = ny;
            if (c + 1 < NCH) {
                nx = lds_read_u2(ax + (c + 1) * 256);
                ny = lds_read_u2(ay + (c + 1) * 256);
            }
            uint32_t s[4];
            if (FORM == 2) {
                s[0] = wx.x; s[1] = wy.x; s[2] = wx.y; s[3] = wy.y;
            } else {
                s[0] = __builtin_amdgcn_perm(wy.x, wx.x, 0x05040100u);
                s[1] = __builtin_amdgcn_perm(wy.x, wx.x, 0x07060302u);
                s[2] = __builtin_amdgcn_perm(wy.y, wx.y, 0x05040100u);
                s[3] = __builtin_amdgcn_perm(wy.y, wx.y, 0x07060302u);
            }
            uint32_t mprev = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * c + u;
                if (FORM == 0) {
                    const uint32_t t = pk_add_i16_sat(md, s[u]);
                    md = M[k];
                    const uint32_t a = pk_max_i16(G[k], pk_sub_u16_sat(A[k], ge));
                    const uint32_t b = pk_max_i16(gl, pk_sub_u16_sat(bl, ge));
                    const uint32_t m = pk_max_i16(pk_max_i16(t, a), b);
                    M[k] = m;
                    A[k] = a;
                    gl = G[k] = pk_sub_u16_sat(m, go);
                    bl = b;
                    best = pk_max_i16(best, m);
                } else {
                    const uint32_t t = h_add(md, s[u]);
                    md = M[k];
                    const uint32_t a = h_max3(G[k], h_sub(A[k], ge), 0u);
                    const uint32_t b = h_max3(gl, h_sub(bl, ge), 0u);
                    const uint32_t m = h_max3(t, a, b);
                    M[k] = m;
                    A[k] = a;
                    gl = G[k] = h_sub(m, go);
                    bl = b;
                    if (u & 1) best = h_max3(best, mprev, m);
                    mprev = m;
                }
            }
            if (K > 16 && c + 1 < NCH) __builtin_amdgcn_sched_barrier(0);
        }
        mdl = em;
        return make_uint2(M[K - 1], bl);
    }
};

template <int K, int FORM> __global__ __launch_bounds__(768) void k(uint32_t *out, unsigned long long *cyc, int rows)
{
    extern __shared__ __attribute__((aligned(256))) uint8_t smem[];
    for (uint32_t o = threadIdx.x * 4u; o < (K / 4) * 256u * 16u; o += blockDim.x * 4u)
        *reinterpret_cast<uint32_t *>(smem + o) = FORM == 0 ? 0x00010001u * ((o >> 2) % 5u) : 0x3C003C00u; // small scores
    __syncthreads();
    Cells<K, FORM> cells;
    cells.reset();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem + (threadIdx.x & 15u) * (K / 4) * 256u;
    uint32_t tok = threadIdx.x * 0x0101u, em = 0u, eb = 0u;
    const uint32_t go = FORM == 0 ? 0x000B000Bu : 0x49804980u, ge = FORM == 0 ? 0x00010001u : 0x3C003C00u; // 11 / 1
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rows; ++r) {
        tok = tok * 5u + 1u;
        const uint2 e = cells.row(base + (tok & 0xF8u), base + ((tok >> 8) & 0xF8u), em, eb, go, ge);
        em = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)e.x, 0x111, 0xf, 0xf, true);
        eb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)e.y, 0x111, 0xf, 0xf, true);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = cells.best ^ em ^ eb;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int K, int FORM> void run(const char *name, int wps)
{
    const int rows = 20000, blocks = 256, threads = 64 * 4 * wps, lds = 64 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<K, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    uint32_t *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, 1024 * 1024 * 4);
    (void)hipMalloc(&cyc, 8192 * 8);
    hipLaunchKernelGGL((k<K, FORM>), dim3(blocks), dim3(threads), lds, 0, out, cyc, rows);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<K, FORM>), dim3(blocks), dim3(threads), lds, 0, out, cyc, rows);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    // wall cycles of a SIMD per 2 cells (one packed column-row) = duration / (rows * K * waves per SIMD)
    const double per2 = ms * 1e-3 * 2.4e9 / ((double)rows * K * wps);
    const double gcups = 1024.0 * wps * 64.0 * 2.0 * K * rows / (ms * 1e-3) / 1e9;
    printf("%-34s K=%2d wps=%d: %6.2f SIMD cycles per 2 cells (%.3f ms)  = %.0f GCUPS if the whole chip did only this\n", name, K,
           wps, per2, ms, gcups);
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main()
{
    for (int wps = 2; wps <= 3; ++wps) {
        run<16, 0>("int16, 10 instr / 2 cells", wps);
        run<16, 1>("f16 max3, 8.5 instr / 2 cells", wps);
        run<16, 2>("f16 max3 without v_perm, 7.5", wps);
        run<32, 0>("int16, 10 instr / 2 cells", wps);
        run<32, 1>("f16 max3, 8.5 instr / 2 cells", wps);
        run<32, 2>("f16 max3 without v_perm, 7.5", wps);
    }
    return 0;
}
