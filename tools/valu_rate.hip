// Microbenchmark: issue rate of the VALU instructions the fill kernels are made of.
// Each wave runs REP x 16 independent instructions of one kind; cycles are read with
// s_memtime, for 1..4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define OPS16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int OP> __device__ __forceinline__ void op16(uint32_t (&x)[16], uint32_t y, uint32_t z)
{
#define A_PKADD(i) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(y));
#define A_PKSUB(i) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(y));
#define A_PKMAX(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define A_PKADDNC(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define A_ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define A_MAX32(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define A_MAX3(i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define A_MAX16(i) asm volatile("v_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define A_ADD16(i) asm volatile("v_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define A_SDWA(i) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x[i]) : "v"(y));
#define A_BFE(i) asm volatile("v_bfe_i32 %0, %0, 8, 8" : "+v"(x[i]));
#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define A_PKFMA(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define A_PKMAXS(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i]) : "s"(0x00100010));
#define A_MAXU16SDWA(i) asm volatile("v_max_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(x[i]) : "v"(y));
#define A_SAD(i) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define A_MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y));
#define A_ADDDPP(i) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y));

#define A_G(i, STR) asm volatile(STR : "+v"(x[i]) : "v"(y), "v"(z));
#define A_MAXF32(i) A_G(i, "v_max_f32 %0, %0, %1")
#define A_ADDF32(i) A_G(i, "v_add_f32 %0, %0, %1")
#define A_MAXU32(i) A_G(i, "v_max_u32 %0, %0, %1")
#define A_MINI32(i) A_G(i, "v_min_i32 %0, %0, %1")
#define A_MAXU16(i) A_G(i, "v_max_u16 %0, %0, %1")
#define A_SUBU16(i) A_G(i, "v_sub_u16 %0, %0, %1")
#define A_SUBU32(i) A_G(i, "v_sub_u32 %0, %0, %1")
#define A_AND(i) A_G(i, "v_and_b32 %0, %0, %1")
#define A_LSHL(i) A_G(i, "v_lshlrev_b32 %0, 3, %0")
#define A_MAX3F(i) A_G(i, "v_max3_f32 %0, %0, %1, %2")
#define A_PKMAXF16(i) A_G(i, "v_pk_max_f16 %0, %0, %1")
#define A_PKADDF16(i) A_G(i, "v_pk_add_f16 %0, %0, %1")
#define A_MAXF16(i) A_G(i, "v_max_f16 %0, %0, %1")
#define A_PKFMAF32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(xx[i]) : "v"(yy));
#define A_PKADDF32(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xx[i]) : "v"(yy));
#define A_MIX1(i) A_G(i, "v_pk_max_i16 %0, %0, %1\n\tv_add_u32 %0, %0, %2")
#define A_MIX2(i) A_G(i, "v_pk_max_i16 %0, %0, %1\n\tv_max_i16 %0, %0, %2")
#define A_ADDU16E64(i) A_G(i, "v_add_u16_e64 %0, %0, %1 clamp")
#define A_MAXI16OPSEL(i) A_G(i, "v_max_i16_e64 %0, %0, %1")
#define A_CNDMASK(i) A_G(i, "v_cndmask_b32 %0, %0, %1, vcc")
#define A_MEDI32(i) A_G(i, "v_med3_i32 %0, %0, %1, %2")
#define A_ADDCO(i) A_G(i, "v_add_co_u32 %0, vcc, %0, %1")
#define A_SUBREV(i) A_G(i, "v_subrev_u32 %0, %1, %0")
#define A_MULLO(i) A_G(i, "v_mul_u32_u24 %0, %0, %1")
#define A_ALIGNBIT(i) A_G(i, "v_alignbit_b32 %0, %0, %1, 16")
#define A_BFI(i) A_G(i, "v_bfi_b32 %0, %2, %0, %1")
#define A_ANDOR(i) A_G(i, "v_and_or_b32 %0, %0, %1, %2")
#define A_LSHLOR(i) A_G(i, "v_lshl_or_b32 %0, %0, 16, %1")
// round 3: the f16 cells' instructions (gfx950 three-operand packed maximum), alone and mixed as the cell mixes them
#define A_PKMAX3F16(i) A_G(i, "v_pk_maximum3_f16 %0, %0, %1, %2")
#define A_PKMIN3F16(i) A_G(i, "v_pk_minimum3_f16 %0, %0, %1, %2")
#define A_MIXF16A(i) A_G(i, "v_pk_add_f16 %0, %0, %1\n\tv_pk_maximum3_f16 %0, %0, %1, %2")
#define A_MIXF16B(i) A_G(i, "v_perm_b32 %0, %0, %1, %2\n\tv_pk_maximum3_f16 %0, %0, %1, %2")
#define A_MIXF16C(i) A_G(i, "v_pk_maximum3_f16 %0, %0, %1, %2\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
    if constexpr (OP == 0) { OPS16(A_PKADD) }
    if constexpr (OP == 1) { OPS16(A_PKSUB) }
    if constexpr (OP == 2) { OPS16(A_PKMAX) }
    if constexpr (OP == 3) { OPS16(A_PKADDNC) }
    if constexpr (OP == 4) { OPS16(A_PERM) }
    if constexpr (OP == 5) { OPS16(A_ADD32) }
    if constexpr (OP == 6) { OPS16(A_MAX32) }
    if constexpr (OP == 7) { OPS16(A_MAX3) }
    if constexpr (OP == 8) { OPS16(A_ADD3) }
    if constexpr (OP == 9) { OPS16(A_MAX16) }
    if constexpr (OP == 10) { OPS16(A_ADD16) }
    if constexpr (OP == 11) { OPS16(A_SDWA) }
    if constexpr (OP == 12) { OPS16(A_BFE) }
    if constexpr (OP == 13) { OPS16(A_FMA) }
    if constexpr (OP == 14) { OPS16(A_PKFMA) }
    if constexpr (OP == 15) { OPS16(A_PKMAXS) }
    if constexpr (OP == 16) { OPS16(A_MAXU16SDWA) }
    if constexpr (OP == 17) { OPS16(A_SAD) }
    if constexpr (OP == 18) { OPS16(A_MOVDPP) }
    if constexpr (OP == 19) { OPS16(A_ADDDPP) }
    if constexpr (OP == 100) { OPS16(A_MAXF32) }
    if constexpr (OP == 101) { OPS16(A_ADDF32) }
    if constexpr (OP == 102) { OPS16(A_MAXU32) }
    if constexpr (OP == 103) { OPS16(A_MINI32) }
    if constexpr (OP == 104) { OPS16(A_MAXU16) }
    if constexpr (OP == 105) { OPS16(A_SUBU16) }
    if constexpr (OP == 106) { OPS16(A_SUBU32) }
    if constexpr (OP == 107) { OPS16(A_AND) }
    if constexpr (OP == 108) { OPS16(A_LSHL) }
    if constexpr (OP == 109) { OPS16(A_MAX3F) }
    if constexpr (OP == 110) { OPS16(A_PKMAXF16) }
    if constexpr (OP == 111) { OPS16(A_PKADDF16) }
    if constexpr (OP == 112) { OPS16(A_MAXF16) }
    if constexpr (OP == 113) { OPS16(A_MIX1) }
    if constexpr (OP == 114) { OPS16(A_MIX2) }
    if constexpr (OP == 115) { OPS16(A_ADDU16E64) }
    if constexpr (OP == 116) { OPS16(A_MAXI16OPSEL) }
    if constexpr (OP == 117) { OPS16(A_CNDMASK) }
    if constexpr (OP == 118) { OPS16(A_MEDI32) }
    if constexpr (OP == 119) { OPS16(A_ADDCO) }
    if constexpr (OP == 120) { OPS16(A_SUBREV) }
    if constexpr (OP == 121) { OPS16(A_MULLO) }
    if constexpr (OP == 122) { OPS16(A_ALIGNBIT) }
    if constexpr (OP == 123) { OPS16(A_BFI) }
    if constexpr (OP == 124) { OPS16(A_ANDOR) }
    if constexpr (OP == 125) { OPS16(A_LSHLOR) }
    if constexpr (OP == 130) { OPS16(A_PKMAX3F16) }
    if constexpr (OP == 131) { OPS16(A_PKMIN3F16) }
    if constexpr (OP == 132) { OPS16(A_MIXF16A) }
    if constexpr (OP == 133) { OPS16(A_MIXF16B) }
    if constexpr (OP == 134) { OPS16(A_MIXF16C) }
}

template <int OP> __global__ void k(uint32_t *out, unsigned long long *cyc, int rep)
{
    uint32_t x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 7 + i;
    uint32_t y = threadIdx.x | 0x00010001u, z = 0x03020100u;
    __syncthreads();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) op16<OP>(x, y, z);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = 0;
    for (int i = 0; i < 16; ++i) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
        cyc[4096 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r1 - r0;
    }
}

extern int g_blocks, g_lds, g_quick;
template <int OP> void run(const char *name)
{
    const int rep = 60000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, g_lds);
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8192 * 8);
    printf("%-28s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {            // waves per SIMD (block = 4*wps waves, 1 block per CU)
        const int threads = 64 * 4 * wps, blocks = g_blocks;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), g_lds, 0, out, cyc, rep);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), g_lds, 0, out, cyc, rep);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(blocks * threads / 64), hr(blocks * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), cyc + 4096, hr.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0, avr = 0; for (auto v : h) avg += (double)v; avg /= h.size();
        for (auto v : hr) avr += (double)v; avr /= hr.size();
        const double ghz = avg / (avr * 10.0);   // memrealtime ticks at 100 MHz
        // cycles per wave-instruction per SIMD = wave cycles / (instr per wave) / (waves sharing the SIMD)
        const double per = avg / (rep * 16.0) / wps;
        printf("  wps=%d: %.2f cyc/instr/SIMD (%.2f ms, %.2f GHz, wall %.2f cyc)", wps, per, ms, ghz,
               ms * 1e-3 * 2.4e9 / (rep * 16.0 * wps));
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int g_blocks = 256, g_lds = 96 * 1024, g_quick = 0;
int main(int argc, char **argv)
{
    if (argc > 1) g_blocks = atoi(argv[1]);
    if (argc > 2) g_lds = atoi(argv[2]) * 1024;  // LDS per block: 96 KB = 1 block per CU, 64 KB = 2, 32 KB = 4
    if (argc > 3) g_quick = atoi(argv[3]);       // 1: packed int16 ops only; 2: those and the f16 cells' instructions
    printf("blocks=%d lds=%d\n", g_blocks, g_lds);
    run<0>("v_pk_add_i16 clamp");
    run<1>("v_pk_sub_u16 clamp");
    run<2>("v_pk_max_i16");
    run<3>("v_pk_add_u16");
    run<15>("v_pk_max_i16 (sgpr lit)");
    if (g_quick == 2) { // the f16 cells' instructions only (round 3); the MIX rows issue TWO instructions per slot: halve their figures
        run<111>("v_pk_add_f16");
        run<110>("v_pk_max_f16");
        run<130>("v_pk_maximum3_f16");
        run<131>("v_pk_minimum3_f16");
        run<4>("v_perm_b32");
        run<132>("MIX pk_add_f16 + pk_maximum3 (x2)");
        run<133>("MIX v_perm + pk_maximum3 (x2)");
        run<134>("MIX pk_maximum3 + mov_dpp (x2)");
        return 0;
    }
    if (g_quick) return 0;
    run<4>("v_perm_b32");
    run<5>("v_add_u32");
    run<6>("v_max_i32");
    run<7>("v_max3_i32");
    run<8>("v_add3_u32");
    run<9>("v_max_i16");
    run<10>("v_add_u16");
    run<11>("v_add_u32_sdwa");
    run<16>("v_max_i16_sdwa");
    run<12>("v_bfe_i32");
    run<13>("v_fma_f32");
    run<14>("v_pk_fma_f16");
    run<17>("v_sad_u8");
    run<18>("v_mov_b32_dpp row_shr");
    run<19>("v_add_u32_dpp row_shr");
    run<100>("MAXF32");
    run<101>("ADDF32");
    run<102>("MAXU32");
    run<103>("MINI32");
    run<104>("MAXU16");
    run<105>("SUBU16");
    run<106>("SUBU32");
    run<107>("AND");
    run<108>("LSHL");
    run<109>("MAX3F");
    run<110>("PKMAXF16");
    run<111>("PKADDF16");
    run<112>("MAXF16");
    run<113>("MIX1");
    run<114>("MIX2");
    run<115>("ADDU16E64");
    run<116>("MAXI16OPSEL");
    run<117>("CNDMASK");
    run<118>("MEDI32");
    run<119>("ADDCO");
    run<120>("SUBREV");
    run<121>("MULLO");
    run<122>("ALIGNBIT");
    run<123>("BFI");
    run<124>("ANDOR");
    run<125>("LSHLOR");
    return 0;
}
