// Issue cost of the vector instructions the random-overlap network is made of, on gfx950:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
// Each kernel runs REPS x 64 independent instances of one instruction per wavefront, 4 wavefronts per SIMD on every CU
// (the occupancy of k_rt_mix_species), and reports shader cycles per wave-instruction and SIMD from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REPS 256
#define R8(x) x x x x x x x x
#define BODY64(x) R8(R8(x))

#define PROBE(name, decl, body, sink)                                                     \
    __global__ void __launch_bounds__(256) k_##name(unsigned long long* out, int seed) {  \
        decl;                                                                             \
        unsigned long long t0 = __builtin_readcyclecounter();                             \
        for (int r = 0; r < REPS; r++) { body }                                           \
        unsigned long long t1 = __builtin_readcyclecounter();                             \
        sink;                                                                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                        \
    }

#define DECL8(T) T a0 = threadIdx.x * seed; T a1 = a0 + 1; T a2 = a0 + 2; T a3 = a0 + 3; T a4 = a0 + 4; T a5 = a0 + 5; T a6 = a0 + 6; T a7 = a0 + 7; T c = seed
#define U8 DECL8(unsigned)
#define D8 DECL8(double)
#define SINKU if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) out[1] = 1
#define SINKD if ((a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) == 0.12345) out[1] = 1
// eight independent chains, eight rounds: 64 instructions per loop trip
#define EACH8(INS) INS(a0) INS(a1) INS(a2) INS(a3) INS(a4) INS(a5) INS(a6) INS(a7)

#define I_MIN(a) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_MED3(a) asm volatile("v_med3_u32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
#define I_MOVDPP_Q(a) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));
#define I_MOVDPP_RM(a) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a));
#define I_MOVDPP_ROR(a) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a));
#define I_MINDPP(a) asm volatile("v_min_u32_dpp %0, %0, %1 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(a) : "v"(c));
#define I_MINDPP_Q(a) asm volatile("v_min_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));
#define I_CNDMASK(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c) : "vcc");
#define I_XOR(a) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_LSHLOR(a) asm volatile("v_lshl_or_b32 %0, %0, 9, %1" : "+v"(a) : "v"(c));
#define I_ALIGNBIT(a) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a) : "v"(c));
#define I_MAD24(a) asm volatile("v_mad_u32_u24 %0, %0, 3, %1" : "+v"(a) : "v"(c));
#define I_PERM16(a) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(c));
#define I_PERM32(a) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));
#define I_SWZ(a) asm volatile("ds_swizzle_b32 %0, %0 offset:0x401f" : "+v"(a));
#define I_BPERM(a) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a) : "v"(c));
#define I_ADDF64(a) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_MULF64(a) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_FMAF64(a) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(c));
#define I_CMPF64(a) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a), "v"(c) : "vcc");
#define I_MINF32NEG(a) asm volatile("v_min_f32_dpp %0, -%0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));
#define I_ADDF32(a) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_PKMIN(a) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_RCPF64(a) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));


#define I_MAXU(a) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_MINI(a) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_MINF32(a) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_MAXF32(a) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_MED3F32(a) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
#define I_MIN3U(a) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
#define I_MINU16(a) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_AND(a) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_OR(a) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_ADDU(a) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_SUBU(a) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_LSHL(a) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a));
#define I_LSHR(a) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a));
#define I_MOV(a) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(c));
#define I_CNDMASK64(a) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(c) : "s20", "s21");
#define I_CMPU(a) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" : : "v"(a), "v"(c) : "s20", "s21");
#define I_BFE(a) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a));
#define I_ANDOR(a) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
#define I_OR3(a) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
#define I_ADD3(a) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
#define I_LSHLADD(a) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(c));
#define I_BFI(a) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
#define I_XORDPP(a) asm volatile("v_xor_b32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));
#define I_ADDF32DPP(a) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));
#define I_MINF32SDWA(a) asm volatile("v_min_f32 %0, %0, -%1" : "+v"(a) : "v"(c));
#define I_SUBF32(a) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_MULF32(a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_FMAF32(a) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
#define I_PKADDF32(a) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_PKMINF16(a) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_MOVB64(a) asm volatile("v_mov_b64 %0, %1" : "+v"(a) : "v"(c));
#define I_CMPF64_64(a) asm volatile("v_cmp_gt_f64_e64 s[20:21], %0, %1" : : "v"(a), "v"(c) : "s20", "s21");
#define I_MAXF64(a) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(c));
#define I_DSREADB64(a) asm volatile("ds_read_b64 %0, %1" : "=v"(a) : "v"(lds_addr));

#define MK(name, DECL, INS, SINK) PROBE(name, DECL, R8(EACH8(INS)), SINK)
MK(min_u32, U8, I_MIN, SINKU)
MK(med3_u32, U8, I_MED3, SINKU)
MK(mov_dpp_quad, U8, I_MOVDPP_Q, SINKU)
MK(mov_dpp_rowmirror, U8, I_MOVDPP_RM, SINKU)
MK(mov_dpp_ror8, U8, I_MOVDPP_ROR, SINKU)
MK(min_u32_dpp_shl4, U8, I_MINDPP, SINKU)
MK(min_u32_dpp_quad, U8, I_MINDPP_Q, SINKU)
MK(min_f32_dpp_neg, U8, I_MINF32NEG, SINKU)
MK(cndmask, U8, I_CNDMASK, SINKU)
MK(xor_b32, U8, I_XOR, SINKU)
MK(lshl_or, U8, I_LSHLOR, SINKU)
MK(alignbit, U8, I_ALIGNBIT, SINKU)
MK(mad_u32_u24, U8, I_MAD24, SINKU)
MK(pk_min_u16, U8, I_PKMIN, SINKU)
MK(add_f32, U8, I_ADDF32, SINKU)
MK(permlane16_swap, U8, I_PERM16, SINKU)
MK(permlane32_swap, U8, I_PERM32, SINKU)
MK(ds_swizzle, U8, I_SWZ, SINKU)
MK(ds_bpermute, U8, I_BPERM, SINKU)
MK(add_f64, D8, I_ADDF64, SINKD)
MK(mul_f64, D8, I_MULF64, SINKD)
MK(fma_f64, D8, I_FMAF64, SINKD)
MK(cmp_gt_f64, D8, I_CMPF64, SINKD)
MK(rcp_f64, D8, I_RCPF64, SINKD)

MK(max_u32, U8, I_MAXU, SINKU)
MK(min_i32, U8, I_MINI, SINKU)
MK(min_f32, U8, I_MINF32, SINKU)
MK(max_f32, U8, I_MAXF32, SINKU)
MK(min_f32_neg, U8, I_MINF32SDWA, SINKU)
MK(med3_f32, U8, I_MED3F32, SINKU)
MK(min3_u32, U8, I_MIN3U, SINKU)
MK(min_u16, U8, I_MINU16, SINKU)
MK(and_b32, U8, I_AND, SINKU)
MK(or_b32, U8, I_OR, SINKU)
MK(add_u32, U8, I_ADDU, SINKU)
MK(sub_u32, U8, I_SUBU, SINKU)
MK(lshlrev, U8, I_LSHL, SINKU)
MK(lshrrev, U8, I_LSHR, SINKU)
MK(mov_b32, U8, I_MOV, SINKU)
MK(cndmask_e64, U8, I_CNDMASK64, SINKU)
MK(cmp_lt_u32_e64, U8, I_CMPU, SINKU)
MK(bfe_u32, U8, I_BFE, SINKU)
MK(and_or, U8, I_ANDOR, SINKU)
MK(or3, U8, I_OR3, SINKU)
MK(add3, U8, I_ADD3, SINKU)
MK(lshl_add, U8, I_LSHLADD, SINKU)
MK(bfi, U8, I_BFI, SINKU)
MK(xor_dpp, U8, I_XORDPP, SINKU)
MK(add_f32_dpp, U8, I_ADDF32DPP, SINKU)
MK(sub_f32, U8, I_SUBF32, SINKU)
MK(mul_f32, U8, I_MULF32, SINKU)
MK(fma_f32, U8, I_FMAF32, SINKU)
MK(pk_min_f16, U8, I_PKMINF16, SINKU)
MK(pk_add_f32, D8, I_PKADDF32, SINKD)
MK(mov_b64, D8, I_MOVB64, SINKD)
MK(cmp_gt_f64_e64, D8, I_CMPF64_64, SINKD)
MK(max_f64, D8, I_MAXF64, SINKD)

struct Entry { const char* name; void (*fn)(unsigned long long*, int); };
#define E(name) {#name, k_##name}
int main() {
    std::vector<Entry> es = {E(min_u32), E(med3_u32), E(mov_dpp_quad), E(mov_dpp_rowmirror), E(mov_dpp_ror8), E(min_u32_dpp_shl4),
                             E(min_u32_dpp_quad), E(min_f32_dpp_neg), E(cndmask), E(xor_b32), E(lshl_or), E(alignbit), E(mad_u32_u24),
                             E(pk_min_u16), E(add_f32), E(permlane16_swap), E(permlane32_swap), E(ds_swizzle), E(ds_bpermute),
                             E(add_f64), E(mul_f64), E(fma_f64), E(cmp_gt_f64), E(rcp_f64),
                             E(max_u32), E(min_i32), E(min_f32), E(max_f32), E(min_f32_neg), E(med3_f32), E(min3_u32), E(min_u16),
                             E(and_b32), E(or_b32), E(add_u32), E(sub_u32), E(lshlrev), E(lshrrev), E(mov_b32), E(cndmask_e64),
                             E(cmp_lt_u32_e64), E(bfe_u32), E(and_or), E(or3), E(add3), E(lshl_add), E(bfi), E(xor_dpp),
                             E(add_f32_dpp), E(sub_f32), E(mul_f32), E(fma_f32), E(pk_min_f16), E(pk_add_f32), E(mov_b64),
                             E(cmp_gt_f64_e64), E(max_f64)};
    unsigned long long* d;
    hipMalloc(&d, 16);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs; %d wave-instructions per wavefront, 16 wavefronts per CU (4 per SIMD)\n", p.gcnArchName, cus, REPS * 64);
    for (auto& e : es) {
        for (int wpc : {4, 16}) {   // workgroups of 256 threads = 4 wavefronts (one per SIMD); 1 or 4 of them per CU
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(e.fn, dim3(cus * wpc / 4), dim3(256), 0, 0, d, 3);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(e.fn, dim3(cus * wpc / 4), dim3(256), 0, 0, d, 3);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            unsigned long long cyc;
            hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
            const double n = (double)REPS * 64;
            // waves per SIMD = wpc / 4; cycles per instruction per SIMD = wave cycles / (n * waves per SIMD)
            // per-SIMD issue cost from the launch time: (ms - 0.020 ms of launch and loop overhead) x 2.4 GHz / instructions per SIMD
            printf("%-20s %2d waves/CU: kernel %.3f ms = %5.2f cycles per wave-instruction and SIMD at 2.4 GHz (s_memtime: %.2f ticks per instruction and wave)\n",
                   e.name, wpc, ms, (ms - 0.020) * 1e-3 * 2.4e9 / (n * (wpc / 4)), cyc / n);
        }
    }
    return 0;
}
