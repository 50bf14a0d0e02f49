// Microbenchmark: does the VGPR bank of the operands change the issue rate of packed-int16 ops?
// Physical registers are pinned in the asm text; wave-cycles per instruction at 4 waves/SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_bank valu_bank.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
template <int MODE> __global__ void k(uint32_t *out, int rep)
{
    // v16..v31 accumulate, v32..v35 second operands (banks 0..3)
    asm volatile("v_mov_b32 v32, 1\n v_mov_b32 v33, 2\n v_mov_b32 v34, 3\n v_mov_b32 v35, 4\n"
                 "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n"
                 "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n" ::: "v16","v17","v18","v19","v20","v21","v22","v23","v32","v33","v34","v35");
    for (int i = 0; i < rep; ++i) {
        if (MODE == 0) { // destination/first source and second source in the same bank
            asm volatile(REP8("v_pk_max_i16 v16, v16, v32\n v_pk_max_i16 v17, v17, v33\n v_pk_max_i16 v18, v18, v34\n v_pk_max_i16 v19, v19, v35\n") ::: "v16","v17","v18","v19");
        } else if (MODE == 1) { // different banks
            asm volatile(REP8("v_pk_max_i16 v16, v16, v33\n v_pk_max_i16 v17, v17, v34\n v_pk_max_i16 v18, v18, v35\n v_pk_max_i16 v19, v19, v32\n") ::: "v16","v17","v18","v19");
        } else if (MODE == 2) { // three distinct registers, all different banks (dst, src0, src1)
            asm volatile(REP8("v_pk_max_i16 v16, v21, v34\n v_pk_max_i16 v17, v22, v35\n v_pk_max_i16 v18, v23, v32\n v_pk_max_i16 v19, v20, v33\n") ::: "v16","v17","v18","v19");
        } else if (MODE == 3) { // src0 and src1 same bank, dst elsewhere
            asm volatile(REP8("v_pk_max_i16 v16, v21, v33\n v_pk_max_i16 v17, v22, v34\n v_pk_max_i16 v18, v23, v35\n v_pk_max_i16 v19, v20, v32\n") ::: "v16","v17","v18","v19");
        } else if (MODE == 4) { // second source an SGPR
            asm volatile(REP8("v_pk_max_i16 v16, v16, s4\n v_pk_max_i16 v17, v17, s4\n v_pk_max_i16 v18, v18, s4\n v_pk_max_i16 v19, v19, s4\n") ::: "v16","v17","v18","v19");
        } else if (MODE == 5) { // dependent chain on one register
            asm volatile(REP8("v_pk_max_i16 v16, v16, v33\n v_pk_max_i16 v16, v16, v34\n v_pk_max_i16 v16, v16, v35\n v_pk_max_i16 v16, v16, v33\n") ::: "v16");
        }
    }
    uint32_t r;
    asm volatile("v_add_u32 %0, v16, v17\n v_add_u32 %0, %0, v18\n v_add_u32 %0, %0, v19" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, int wps)
{
    uint32_t *out; hipMalloc(&out, 4u << 20);
    const int rep = 20000, blocks = 256, threads = 64 * 4 * wps;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, rep);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, rep);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s wps=%d: %.3f ms  -> %.2f cycles/instr/SIMD at 2.4 GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / (rep * 32.0 * wps));
    hipFree(out);
}

int main()
{
    for (int wps = 1; wps <= 4; wps *= 2) {
        run<0>("src1 in the bank of dst=src0", wps);
        run<1>("src1 in another bank", wps);
        run<2>("dst, src0, src1 all different banks", wps);
        run<3>("src0, src1 same bank", wps);
        run<4>("src1 an SGPR", wps);
        run<5>("dependent chain", wps);
    }
    return 0;
}
