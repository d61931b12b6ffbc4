// Issue rates of the integer vector instructions the egocentric kernel chooses between, on gfx950: cycles per
// wave64 instruction and SIMD with 1, 2 and 4 waves per SIMD (s_memtime around 64 x 32 independent instructions).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(INSTR)                                                                                                  \
    uint32_t r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6,    \
             r7 = seed + 7;                                                                                          \
    const uint32_t b = seed * 3 + 1, c = seed ^ 0x5a5a;                                                              \
    __builtin_amdgcn_s_barrier();                                                                                    \
    const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                                \
    for (int it = 0; it < 64; ++it) {                                                                                \
        asm volatile(REP8(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7))                   \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                 \
                     : "v"(b), "v"(c)                                                                                \
                     : "vcc", "s10", "s11", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");                 \
    }                                                                                                                \
    const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                                \
    if ((threadIdx.x & 63) == 0) {                                                                                   \
        cycles[2 * (threadIdx.x >> 6)] = t0;                                                                         \
        cycles[2 * (threadIdx.x >> 6) + 1] = t1;                                                                     \
    }                                                                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;

#define I_ADD(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_ASHR(k) "v_ashrrev_i32 %" #k ", 10, %" #k "\n"
#define I_MED3(k) "v_med3_i32 %" #k ", %" #k ", %8, %9\n"
#define I_MAD24(k) "v_mad_i32_i24 %" #k ", %" #k ", %8, %9\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I_DOT2(k) "v_dot2_u32_u16 %" #k ", %" #k ", %8, %9\n"
#define I_PKMAX(k) "v_pk_max_i16 %" #k ", %" #k ", %8\n"
#define I_PKADD(k) "v_pk_add_u16 %" #k ", %" #k ", %8\n"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %9\n"
#define I_BFE(k) "v_bfe_i32 %" #k ", %" #k ", 10, 16\n"
#define I_ALIGNBIT(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 10\n"
#define I_MADU16(k) "v_mad_u32_u16 %" #k ", %" #k ", %8, %9\n"
#define I_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define I_OR3(k) "v_or3_b32 %" #k ", %" #k ", %8, %9\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define I_MULU24(k) "v_mul_u32_u24 %" #k ", %" #k ", %8\n"
#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_ADDF64(k) ""
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_PKMADI16(k) "v_pk_mad_i16 %" #k ", %" #k ", %8, %9\n"
#define I_CND64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[10:11]\n"
#define I_CMP(k) "v_cmp_lt_i32 vcc, %" #k ", %8\n"
#define I_CMP64(k) "v_cmp_lt_i32_e64 s[10:11], %" #k ", %8\n"
#define I_MIN(k) "v_min_i32 %" #k ", %" #k ", %8\n"
#define I_BFI(k) "v_bfi_b32 %" #k ", %" #k ", %8, %9\n"
#define I_MAD64(k) "v_mad_u64_u32 v[20:21], s[10:11], %" #k ", %8, v[22:23]\n"
#define I_LSHLADD64(k) "v_lshl_add_u64 v[20:21], v[22:23], 0, v[24:25]\n"
#define I_MULF64(k) "v_mul_f64 v[20:21], v[22:23], v[24:25]\n"
#define I_FMAF64(k) "v_fma_f64 v[20:21], v[22:23], v[24:25], v[26:27]\n"
#define I_RNDF64(k) "v_rndne_f64 v[20:21], v[22:23]\n"
#define I_CVTF64(k) "v_cvt_i32_f64 %" #k ", v[22:23]\n"
#define I_LDEXPF64(k) "v_ldexp_f64 v[20:21], v[22:23], 10\n"
#define I_READLANE(k) "v_readlane_b32 s10, %" #k ", 3\n"
#define I_SDWA(k) "v_add_u32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"

#define KERNEL(NAME, INSTR)                                                                        \
    __global__ void __launch_bounds__(1024) NAME(uint32_t seed, uint32_t* out, uint64_t* cycles)   \
    {                                                                                              \
        BODY(INSTR)                                                                                \
    }
KERNEL(k_add, I_ADD)
KERNEL(k_ashr, I_ASHR)
KERNEL(k_med3, I_MED3)
KERNEL(k_mad24, I_MAD24)
KERNEL(k_perm, I_PERM)
KERNEL(k_dot2, I_DOT2)
KERNEL(k_pkmax, I_PKMAX)
KERNEL(k_pkadd, I_PKADD)
KERNEL(k_add3, I_ADD3)
KERNEL(k_lshladd, I_LSHLADD)
KERNEL(k_bfe, I_BFE)
KERNEL(k_alignbit, I_ALIGNBIT)
KERNEL(k_madu16, I_MADU16)
KERNEL(k_andor, I_ANDOR)
KERNEL(k_or3, I_OR3)
KERNEL(k_mullo, I_MULLO)
KERNEL(k_mulu24, I_MULU24)
KERNEL(k_fma, I_FMA)
KERNEL(k_cndmask, I_CNDMASK)
KERNEL(k_mov, I_MOV)
KERNEL(k_pkmad, I_PKMADI16)
KERNEL(k_sdwa, I_SDWA)
KERNEL(k_cnd64, I_CND64)
KERNEL(k_cmp, I_CMP)
KERNEL(k_cmp64, I_CMP64)
KERNEL(k_min, I_MIN)
KERNEL(k_bfi, I_BFI)
KERNEL(k_mad64, I_MAD64)
KERNEL(k_lshladd64, I_LSHLADD64)
KERNEL(k_mulf64, I_MULF64)
KERNEL(k_fmaf64, I_FMAF64)
KERNEL(k_rndf64, I_RNDF64)
KERNEL(k_cvtf64, I_CVTF64)
KERNEL(k_ldexpf64, I_LDEXPF64)
KERNEL(k_readlane, I_READLANE)

int main()
{
    uint32_t* out;
    uint64_t* cyc;
    hipMalloc(&out, 1024 * 1024 * sizeof(uint32_t));
    hipMalloc(&cyc, 64 * sizeof(uint64_t));
    struct { const char* name; void (*fn)(uint32_t, uint32_t*, uint64_t*); } ks[] = {
        {"v_add_u32", k_add}, {"v_ashrrev_i32", k_ashr}, {"v_med3_i32", k_med3}, {"v_mad_i32_i24", k_mad24},
        {"v_perm_b32", k_perm}, {"v_dot2_u32_u16", k_dot2}, {"v_pk_max_i16", k_pkmax}, {"v_pk_add_u16", k_pkadd},
        {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshladd}, {"v_bfe_i32", k_bfe}, {"v_alignbit_b32", k_alignbit},
        {"v_mad_u32_u16", k_madu16}, {"v_and_or_b32", k_andor}, {"v_or3_b32", k_or3}, {"v_mul_lo_u32", k_mullo},
        {"v_mul_u32_u24", k_mulu24}, {"v_fma_f32", k_fma}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32", k_mov},
        {"v_pk_mad_i16", k_pkmad}, {"v_add_u32_sdwa", k_sdwa}, {"v_cndmask_b32 (sgpr pair)", k_cnd64},
        {"v_cmp_lt_i32 vcc", k_cmp}, {"v_cmp_lt_i32 sgpr pair", k_cmp64}, {"v_min_i32", k_min}, {"v_bfi_b32", k_bfi},
        {"v_mad_u64_u32", k_mad64}, {"v_lshl_add_u64", k_lshladd64}, {"v_mul_f64", k_mulf64}, {"v_fma_f64", k_fmaf64},
        {"v_rndne_f64", k_rndf64}, {"v_cvt_i32_f64", k_cvtf64}, {"v_ldexp_f64", k_ldexpf64}, {"v_readlane_b32", k_readlane}};
    printf("%-26s %10s %10s %10s   (cycles per wave64 instruction and SIMD; 4096 instructions per wave)\n", "instruction",
           "1 wave", "2 waves", "4 waves");
    for (auto& k : ks) {
        printf("%-26s", k.name);
        for (int waves_per_simd : {1, 2, 4}) {
            const int threads = 64 * 4 * waves_per_simd;   // one workgroup on one CU: waves spread over the 4 SIMDs
            uint64_t best = ~0ull;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(k.fn, dim3(1), dim3(threads), 0, 0, 12345u + rep, out, cyc);
                uint64_t t[64];
                hipMemcpy(t, cyc, sizeof(t), hipMemcpyDeviceToHost);
                uint64_t first = ~0ull, last = 0;   // the arbiter favours the oldest wave: first start to last end
                for (int w = 0; w < 4 * waves_per_simd; ++w) {
                    first = t[2 * w] < first ? t[2 * w] : first;
                    last = t[2 * w + 1] > last ? t[2 * w + 1] : last;
                }
                const uint64_t h = last - first;
                if (h < best) best = h;
            }
            printf(" %10.2f", (double)best / (4096.0 * waves_per_simd));
        }
        printf("\n");
    }
    return 0;
}
