// What does rocprofv3's FETCH_SIZE report on gfx950 for the load shapes of the fill kernels?  MI355X_MICROARCH.md
// says it reports HALF the bytes of a wide coalesced streaming read (16 B per lane).  The multi-pass fill reads its
// tokens and edges with 4- and 8-byte buffer loads from 4 lanes of every lane group: does the same factor apply?
// Every kernel below reads a buffer of known size exactly once (cold: each launch has its own buffer) and sums it
// into one word so that nothing is optimised away.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_probe fetch_probe.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_probe      (FETCH_SIZE is in KiB)
// and compare the counter of each kernel with the bytes it read (printed here).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// 16 bytes per lane, consecutive lanes consecutive: the guide's case
__global__ void probe_wide16(const uint4 *src, size_t n16, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// 4 bytes per lane from lanes 0..3 of every 16-lane group, one 16-byte block per group and step (the token loads of
// swg_diag_dyn_kernel<EDGES>): raw_buffer_load_b32
__global__ void probe_quad_b32(const uint8_t *src, uint32_t bytes, uint32_t *out)
{
    const rsrc_t r = make_rsrc(src, bytes);
    const uint32_t lane = threadIdx.x & 63u, g = lane & 15u;
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / 16u, groups = gridDim.x * blockDim.x / 16u;
    uint32_t acc = 0;
    for (uint32_t blk = group; (uint64_t)blk * 16u < bytes; blk += groups) {
        const uint32_t off = g < 4u ? blk * 16u + g * 4u : 0xC0000000u; // (other lanes: out of bounds, reads zero)
        acc += __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// 8 bytes per lane from lanes 0..3 of every 16-lane group, one 32-byte row block per group and step (the edge loads)
__global__ void probe_quad_b64(const uint8_t *src, uint32_t bytes, uint32_t *out)
{
    const rsrc_t r = make_rsrc(src, bytes);
    const uint32_t lane = threadIdx.x & 63u, g = lane & 15u;
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / 16u, groups = gridDim.x * blockDim.x / 16u;
    uint32_t acc = 0;
    for (uint32_t blk = group; (uint64_t)blk * 32u < bytes; blk += groups) {
        const uint32_t off = g < 4u ? blk * 32u + g * 8u : 0xC0000000u;
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// 4 bytes per lane, all 64 lanes consecutive (the single-pass fill's token loads are per-lane dword loads of one
// leader per group; this is the plain dword stream for comparison)
__global__ void probe_dword(const uint32_t *src, size_t n4, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 0x12345678u) out[0] = acc;
}

// one leader lane per 16-lane group walks its own stream of 16-byte blocks, one dword per load (tp[0..3] of the
// single-pass fill): 4 loads of 4 bytes from the same 16-byte block, consecutive blocks per group
__global__ void probe_leader_dwords(const uint32_t *src, size_t n_blocks, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, g = lane & 15u;
    const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 16u, groups = (size_t)gridDim.x * blockDim.x / 16u;
    const size_t per = (n_blocks + groups - 1) / groups; // a contiguous run of blocks per group, as a pair's tokens are
    uint32_t acc = 0;
    if (g == 0)
        for (size_t b = group * per; b < (group + 1) * per && b < n_blocks; ++b) {
            const uint32_t *t = src + b * 4;
            acc += __builtin_nontemporal_load(t) ^ t[1] ^ t[2] ^ t[3];
        }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const size_t bytes = 512ull << 20; // 512 MiB per probe, each its own buffer: every byte comes from HBM once
    uint32_t *out;
    (void)hipMalloc(&out, 64);
    void *buf[5];
    for (auto &b : buf) {
        (void)hipMalloc(&b, bytes);
        (void)hipMemset(b, 1, bytes);
    }
    (void)hipDeviceSynchronize();
    const int wgs = 256 * 4, thr = 256;
    hipLaunchKernelGGL(probe_wide16, dim3(wgs), dim3(thr), 0, 0, (const uint4 *)buf[0], bytes / 16, out);
    hipLaunchKernelGGL(probe_quad_b32, dim3(wgs), dim3(thr), 0, 0, (const uint8_t *)buf[1], (uint32_t)bytes, out);
    hipLaunchKernelGGL(probe_quad_b64, dim3(wgs), dim3(thr), 0, 0, (const uint8_t *)buf[2], (uint32_t)bytes, out);
    hipLaunchKernelGGL(probe_dword, dim3(wgs), dim3(thr), 0, 0, (const uint32_t *)buf[3], bytes / 4, out);
    hipLaunchKernelGGL(probe_leader_dwords, dim3(wgs), dim3(thr), 0, 0, (const uint32_t *)buf[4], bytes / 16, out);
    (void)hipDeviceSynchronize();
    printf("every probe kernel read %zu bytes = %zu KiB exactly once\n", bytes, bytes / 1024);
    return 0;
}
