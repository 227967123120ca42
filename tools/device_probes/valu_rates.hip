// valu_rates.hip -- what one vector instruction of the integer paths costs on gfx950: issue cycles per wave-instruction and SIMD at
// 1, 2, 4 and 8 wavefronts per SIMD (profiles/r04_unamb.md: the instruction budgets of unambiguous_kernel are priced with these).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rates tools/valu_rates.hip && tools/valu_rates
// Every kernel runs ITERS x 16 copies of ONE instruction on eight independent accumulators per lane; the shader clock comes from
// s_memtime against s_memrealtime (100 MHz) inside the same kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

struct Stamp { uint64_t cyc, real, r0, r1; };

// ONE asm statement per 16 instructions (hipcc puts an s_nop between separate asm statements): I(A) names the accumulator operand
#define I8(I) I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7)
#define D8(I) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0)  // one dependent chain: every instruction reads the previous one's result
#define BODY(I)                                                                                                                   \
    asm volatile(I8(I) I8(I)                                                                                                      \
                 : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7) \
                 : [b] "v"(b), [s] "s"(sv), [c] "v"(c), [m4] "v"(addr4), [m8] "v"(addr8)                                         \
                 : "s20", "s21", "vcc")

#define BODY_DEP(I)                                                                                                               \
    asm volatile(D8(I) D8(I)                                                                                                      \
                 : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7) \
                 : [b] "v"(b), [s] "s"(sv), [c] "v"(c), [m4] "v"(addr4), [m8] "v"(addr8)                                         \
                 : "s20", "s21", "vcc")

// A: the accumulator (read and written); [b], [c]: further vector operands; [s]: a scalar operand; [m4] / [m8]: lane-linear LDS byte
// addresses (4 and 8 bytes per lane)
#define BENCH(NAME, TYPE, INSTR) BENCH_(NAME, TYPE, INSTR, BODY)
#define BENCH_DEP(NAME, TYPE, INSTR) BENCH_(NAME, TYPE, INSTR, BODY_DEP)
#define BENCH_(NAME, TYPE, INSTR, BODYM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(TYPE *out, Stamp *st, uint32_t sv) {                      \
        __shared__ uint32_t lds[4096];                                                                    \
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 2654435761u;                          \
        __syncthreads();                                                                                  \
        const uint32_t t = threadIdx.x;                                                                   \
        TYPE a0 = (TYPE)t * 4, a1 = (TYPE)t * 4 + 1024, a2 = (TYPE)t * 4 + 2048, a3 = (TYPE)t * 4 + 3072;  \
        TYPE a4 = (TYPE)t * 4 + 4096, a5 = (TYPE)t * 4 + 5120, a6 = (TYPE)t * 4 + 6144, a7 = (TYPE)t * 4 + 7168;     \
        TYPE b = (TYPE)(sv | 5u);                                                                         \
        TYPE c = (TYPE)(sv * 7u + 3u);                                                                    \
        const uint32_t addr4 = t * 4u, addr8 = t * 8u;                                                    \
        const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();          \
        for (int i = 0; i < ITERS; ++i) {                                                                 \
            BODYM(INSTR);                                                                                 \
        }                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();          \
        out[blockIdx.x * 256 + t] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                \
        if ((t & 63u) == 0) st[blockIdx.x * 4 + (t >> 6)] = Stamp{t1 - t0, r1 - r0, r0, r1};                     \
    }

#define I_AND(A) "v_and_b32 %[" #A "], %[" #A "], %[b]\n"
BENCH(k_and, uint32_t, I_AND)
#define I_ADD(A) "v_add_u32 %[" #A "], %[" #A "], %[b]\n"
BENCH(k_add, uint32_t, I_ADD)
#define I_LSHR32(A) "v_lshrrev_b32 %[" #A "], 3, %[" #A "]\n"
BENCH(k_lshr32, uint32_t, I_LSHR32)
#define I_LSHL_OR(A) "v_lshl_or_b32 %[" #A "], %[" #A "], 1, %[b]\n"
BENCH(k_lshl_or, uint32_t, I_LSHL_OR)
#define I_AND_OR(A) "v_and_or_b32 %[" #A "], %[" #A "], %[b], %[c]\n"
BENCH(k_and_or, uint32_t, I_AND_OR)
#define I_OR3(A) "v_or3_b32 %[" #A "], %[" #A "], %[b], %[c]\n"
BENCH(k_or3, uint32_t, I_OR3)
#define I_BITOP3(A) "v_bitop3_b32 %[" #A "], %[" #A "], %[b], %[c] bitop3:0xc8\n"
BENCH(k_bitop3, uint32_t, I_BITOP3)
#define I_BITOP3_S(A) "v_bitop3_b32 %[" #A "], %[" #A "], %[s], %[c] bitop3:0xc8\n"
BENCH(k_bitop3_s, uint32_t, I_BITOP3_S)
#define I_ALIGNBIT(A) "v_alignbit_b32 %[" #A "], %[" #A "], %[b], %[c]\n"
BENCH(k_alignbit, uint32_t, I_ALIGNBIT)
#define I_ALIGNBIT_S(A) "v_alignbit_b32 %[" #A "], %[" #A "], %[b], %[s]\n"
BENCH(k_alignbit_s, uint32_t, I_ALIGNBIT_S)
#define I_PERM(A) "v_perm_b32 %[" #A "], %[" #A "], %[b], %[c]\n"
BENCH(k_perm, uint32_t, I_PERM)
#define I_BFREV(A) "v_bfrev_b32 %[" #A "], %[" #A "]\n"
BENCH(k_bfrev, uint32_t, I_BFREV)
#define I_BFE(A) "v_bfe_u32 %[" #A "], %[" #A "], 3, 9\n"
BENCH(k_bfe, uint32_t, I_BFE)
#define I_BFI(A) "v_bfi_b32 %[" #A "], %[b], %[" #A "], %[c]\n"
BENCH(k_bfi, uint32_t, I_BFI)
#define I_BCNT(A) "v_bcnt_u32_b32 %[" #A "], %[" #A "], %[b]\n"
BENCH(k_bcnt, uint32_t, I_BCNT)
#define I_FFBL(A) "v_ffbl_b32 %[" #A "], %[" #A "]\n"
BENCH(k_ffbl, uint32_t, I_FFBL)
#define I_MBCNT(A) "v_mbcnt_lo_u32_b32 %[" #A "], %[b], %[" #A "]\n"
BENCH(k_mbcnt, uint32_t, I_MBCNT)
#define I_CNDMASK(A) "v_cndmask_b32 %[" #A "], %[" #A "], %[b], vcc\n"
BENCH(k_cndmask, uint32_t, I_CNDMASK)
#define I_CNDMASK_S(A) "v_cndmask_b32_e64 %[" #A "], %[" #A "], %[b], s[20:21]\n"
BENCH(k_cndmask_s, uint32_t, I_CNDMASK_S)
#define I_CMP32(A) "v_cmp_lt_u32_e32 vcc, %[" #A "], %[b]\n"
BENCH(k_cmp32, uint32_t, I_CMP32)
#define I_CMP32S(A) "v_cmp_lt_u32_e64 s[20:21], %[" #A "], %[b]\n"
BENCH(k_cmp32s, uint32_t, I_CMP32S)
#define I_CMP_CND(A) "v_cmp_lt_u32_e32 vcc, %[" #A "], %[b]\n v_cndmask_b32_e32 %[" #A "], %[" #A "], %[c], vcc\n"
BENCH(k_cmp_cnd, uint32_t, I_CMP_CND)
#define I_ADD_CO(A) "v_add_co_u32_e32 %[" #A "], vcc, %[" #A "], %[b]\n"
BENCH(k_add_co, uint32_t, I_ADD_CO)
#define I_ADDC_CO(A) "v_addc_co_u32_e32 %[" #A "], vcc, %[" #A "], %[b], vcc\n"
BENCH(k_addc_co, uint32_t, I_ADDC_CO)
#define I_MIN(A) "v_min_u32_e32 %[" #A "], %[" #A "], %[b]\n"
BENCH(k_min, uint32_t, I_MIN)
#define I_ASHR(A) "v_ashrrev_i32_e32 %[" #A "], 31, %[" #A "]\n"
BENCH(k_ashr, uint32_t, I_ASHR)
#define I_XOR(A) "v_xor_b32_e32 %[" #A "], %[" #A "], %[b]\n"
BENCH(k_xor, uint32_t, I_XOR)
#define I_MUL_LO(A) "v_mul_lo_u32 %[" #A "], %[" #A "], %[b]\n"
BENCH(k_mul_lo, uint32_t, I_MUL_LO)
#define I_MUL_HI(A) "v_mul_hi_u32 %[" #A "], %[" #A "], %[b]\n"
BENCH(k_mul_hi, uint32_t, I_MUL_HI)
#define I_MUL_U24(A) "v_mul_u32_u24 %[" #A "], %[" #A "], %[b]\n"
BENCH(k_mul_u24, uint32_t, I_MUL_U24)
#define I_MAD_U24(A) "v_mad_u32_u24 %[" #A "], %[" #A "], %[b], %[c]\n"
BENCH(k_mad_u24, uint32_t, I_MAD_U24)
#define I_DOT4_U8(A) "v_dot4_u32_u8 %[" #A "], %[" #A "], %[b], %[c]\n"
BENCH(k_dot4_u8, uint32_t, I_DOT4_U8)
#define I_DOT8_U4(A) "v_dot8_u32_u4 %[" #A "], %[" #A "], %[b], %[c]\n"
BENCH(k_dot8_u4, uint32_t, I_DOT8_U4)
#define I_SAD_U8(A) "v_sad_u8 %[" #A "], %[" #A "], %[b], %[c]\n"
BENCH(k_sad_u8, uint32_t, I_SAD_U8)
#define I_READLANE(A) "v_readlane_b32 s20, %[" #A "], 5\n v_mov_b32 %[" #A "], s20\n"
BENCH(k_readlane, uint32_t, I_READLANE)
#define I_LSHR64(A) "v_lshrrev_b64 %[" #A "], 3, %[" #A "]\n"
BENCH(k_lshr64, uint64_t, I_LSHR64)
#define I_LSHR64_V(A) "v_lshrrev_b64 %[" #A "], %[m4], %[" #A "]\n"
BENCH(k_lshr64_v, uint64_t, I_LSHR64_V)
#define I_LSHL64_S(A) "v_lshlrev_b64 %[" #A "], %[s], %[" #A "]\n"
BENCH(k_lshl64_s, uint64_t, I_LSHL64_S)
#define I_ADD64(A) "v_lshl_add_u64 %[" #A "], %[" #A "], 0, %[b]\n"
BENCH(k_add64, uint64_t, I_ADD64)
#define I_BPERMUTE(A) "ds_bpermute_b32 %[" #A "], %[m4], %[" #A "]\n s_waitcnt lgkmcnt(7)\n"
BENCH(k_bpermute, uint32_t, I_BPERMUTE)
#define I_DS_READ32(A) "ds_read_b32 %[" #A "], %[m4]\n s_waitcnt lgkmcnt(7)\n"
BENCH(k_ds_read32, uint32_t, I_DS_READ32)
#define I_DS_READ64(A) "ds_read_b64 %[" #A "], %[m8]\n s_waitcnt lgkmcnt(7)\n"
BENCH(k_ds_read64, uint64_t, I_DS_READ64)
#define I_DS_READ_U16(A) "ds_read_u16 %[" #A "], %[m4]\n s_waitcnt lgkmcnt(7)\n"
BENCH(k_ds_read_u16, uint32_t, I_DS_READ_U16)
#define I_DS_WRITE16(A) "ds_write_b16 %[m4], %[" #A "]\n s_waitcnt lgkmcnt(7)\n"
BENCH(k_ds_write16, uint32_t, I_DS_WRITE16)
#define I_DS_WRITE32(A) "ds_write_b32 %[m4], %[" #A "]\n s_waitcnt lgkmcnt(7)\n"
BENCH(k_ds_write32, uint32_t, I_DS_WRITE32)

#define I_CMP64(A) "v_cmp_lt_u64_e32 vcc, %[" #A "], %[b]\n"
BENCH(k_cmp64, uint64_t, I_CMP64)
// the select of the fused XOR reducer's roll step (run_kernel.hpp): 64-bit compare, the wait state hipcc puts behind it, two selects
#define I_MIN64(A) "v_cmp_lt_u64_e32 vcc, %[" #A "], %[b]\n s_nop 1\n v_cndmask_b32_e32 %[m4], %[m4], %[m8], vcc\n v_cndmask_b32_e32 %[m8], %[m8], %[m4], vcc\n"
BENCH(k_min64, uint64_t, I_MIN64)
#define I_MIN64_NONOP(A) "v_cmp_lt_u64_e64 s[20:21], %[" #A "], %[b]\n v_and_b32 %[m4], %[m4], %[m8]\n v_and_b32 %[m8], %[m8], %[m4]\n"
BENCH(k_min64_nonop, uint64_t, I_MIN64_NONOP)
#define I_OR(A) "v_or_b32_e32 %[" #A "], %[" #A "], %[b]\n"
BENCH(k_or, uint32_t, I_OR)
#define I_SUB(A) "v_sub_u32_e32 %[" #A "], %[" #A "], %[b]\n"
BENCH(k_sub, uint32_t, I_SUB)
#define I_LSHL32(A) "v_lshlrev_b32_e32 %[" #A "], 3, %[" #A "]\n"
BENCH(k_lshl32, uint32_t, I_LSHL32)
#define I_LSHR32V(A) "v_lshrrev_b32_e32 %[" #A "], %[b], %[" #A "]\n"
BENCH(k_lshr32v, uint32_t, I_LSHR32V)
#define I_MOV(A) "v_mov_b32_e32 %[" #A "], %[b]\n"
BENCH(k_mov, uint32_t, I_MOV)
#define I_NOT(A) "v_not_b32_e32 %[" #A "], %[" #A "]\n"
BENCH(k_not, uint32_t, I_NOT)
#define I_AND_S(A) "v_and_b32_e32 %[" #A "], %[s], %[" #A "]\n"
BENCH(k_and_s, uint32_t, I_AND_S)
#define I_AND_LIT(A) "v_and_b32_e32 %[" #A "], 0x77777777, %[" #A "]\n"
BENCH(k_and_lit, uint32_t, I_AND_LIT)
#define I_ADD3(A) "v_add3_u32 %[" #A "], %[" #A "], %[b], %[c]\n"
BENCH(k_add3, uint32_t, I_ADD3)
#define I_LSHL_ADD(A) "v_lshl_add_u32 %[" #A "], %[" #A "], 2, %[b]\n"
BENCH(k_lshl_add, uint32_t, I_LSHL_ADD)
#define I_XAD(A) "v_xad_u32 %[" #A "], %[" #A "], %[b], %[c]\n"
BENCH(k_xad, uint32_t, I_XAD)
#define I_MAD64(A) "v_mad_u64_u32 %[" #A "], s[20:21], %[m4], %[m8], %[" #A "]\n"
BENCH(k_mad64, uint64_t, I_MAD64)
#define I_NOP(A) "s_nop 1\n"
BENCH(k_snop, uint32_t, I_NOP)
BENCH_DEP(k_dep_and, uint32_t, I_AND)
BENCH_DEP(k_dep_alignbit, uint32_t, I_ALIGNBIT)
BENCH_DEP(k_dep_lshr64, uint64_t, I_LSHR64)
BENCH_DEP(k_dep_add64, uint64_t, I_ADD64)
BENCH_DEP(k_dep_cmp_cnd, uint32_t, I_CMP_CND)

struct Entry {
    const char *name;
    void *fn;
    int bytes;
};

template <typename T>
static int run(const char *name, void (*fn)(T *, Stamp *, uint32_t), T *out, Stamp *st, std::vector<Stamp> &h) {
    printf("%-34s", name);
    for (int w : {1, 2, 4, 8}) {
        const int grid = 256 * w;  // 256-thread workgroups: one wavefront per SIMD and workgroup, w workgroups per CU
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, out, st, 11u);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, out, st, 11u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), st, sizeof(Stamp) * grid * 4, hipMemcpyDeviceToHost));
        // per wavefront: shader cycles and 100 MHz ticks between its first and its last instruction -> the clock.  For the launch:
        // first start to last end.  The SIMD's rate is instructions / SPAN: the arbiter prefers the oldest wavefront, so the
        // wavefronts of a SIMD finish one after the other and the mean of their lifetimes is shorter than the time the SIMD needed
        // (the first version of this tool divided by that mean and credited a SIMD with up to 1.8x what it issues).
        double cyc = 0, real = 0;
        uint64_t first = ~0ull, last = 0;
        for (int i = 0; i < grid * 4; ++i) {
            cyc += (double)h[i].cyc;
            real += (double)h[i].real;
            first = h[i].r0 < first ? h[i].r0 : first;
            last = h[i].r1 > last ? h[i].r1 : last;
        }
        const double ghz = cyc / (real * 10.0);
        const double span_cycles = (double)(last - first) * 10.0 * ghz;
        const double per = span_cycles / (ITERS * 16.0) / w;  // cycles of one SIMD per wave-instruction
        printf("  w%d %6.2f cyc (%.2f GHz)", w, per, ghz);
    }
    printf("\n");
    return 0;
}

int main() {
    uint32_t *out;
    Stamp *st;
    CHECK(hipMalloc(&out, 8ull * 256 * 2048 * 8));
    CHECK(hipMalloc(&st, sizeof(Stamp) * 2048 * 4 * 2));
    std::vector<Stamp> h(2048 * 4 * 2);
    printf("cycles of one SIMD per wave-instruction, at w wavefronts per SIMD (256 CUs, every SIMD busy)\n");
#define R32(NAME, INSTR) if (run<uint32_t>(INSTR, NAME, out, st, h)) return 1;
#define R64(NAME, INSTR) if (run<uint64_t>(INSTR, NAME, (uint64_t *)out, st, h)) return 1;
    R32(k_and, "v_and_b32")
    R32(k_or, "v_or_b32")
    R32(k_sub, "v_sub_u32")
    R32(k_lshl32, "v_lshlrev_b32 (const)")
    R32(k_lshr32v, "v_lshrrev_b32 (v shift)")
    R32(k_mov, "v_mov_b32")
    R32(k_not, "v_not_b32")
    R32(k_and_s, "v_and_b32 with an SGPR operand")
    R32(k_and_lit, "v_and_b32 with a literal")
    R32(k_add3, "v_add3_u32")
    R32(k_lshl_add, "v_lshl_add_u32")
    R32(k_xad, "v_xad_u32")
    R64(k_mad64, "v_mad_u64_u32")
    R32(k_add, "v_add_u32")
    R32(k_lshr32, "v_lshrrev_b32 (const)")
    R32(k_lshl_or, "v_lshl_or_b32")
    R32(k_and_or, "v_and_or_b32")
    R32(k_or3, "v_or3_b32")
    R32(k_bitop3, "v_bitop3_b32 (vvv)")
    R32(k_bitop3_s, "v_bitop3_b32 (vsv)")
    R32(k_alignbit, "v_alignbit_b32 (v shift)")
    R32(k_alignbit_s, "v_alignbit_b32 (s shift)")
    R32(k_perm, "v_perm_b32")
    R32(k_bfrev, "v_bfrev_b32")
    R32(k_bfe, "v_bfe_u32")
    R32(k_bfi, "v_bfi_b32")
    R32(k_bcnt, "v_bcnt_u32_b32")
    R32(k_ffbl, "v_ffbl_b32")
    R32(k_mbcnt, "v_mbcnt_lo_u32_b32")
    R32(k_cndmask, "v_cndmask_b32")
    R32(k_cndmask_s, "v_cndmask_b32_e64 (mask in s[20:21])")
    R32(k_cmp32, "v_cmp_lt_u32 -> vcc")
    R32(k_cmp32s, "v_cmp_lt_u32 -> s[20:21]")
    R32(k_cmp_cnd, "v_cmp + v_cndmask (2 instr)")
    R32(k_add_co, "v_add_co_u32 (carry out -> vcc)")
    R32(k_addc_co, "v_addc_co_u32 (carry in and out)")
    R32(k_min, "v_min_u32")
    R32(k_ashr, "v_ashrrev_i32")
    R32(k_xor, "v_xor_b32")
    R64(k_cmp64, "v_cmp_lt_u64 -> vcc")
    R64(k_min64, "v_cmp_lt_u64 + s_nop 1 + 2 v_cndmask (4 instr)")
    R64(k_min64_nonop, "v_cmp_lt_u64 -> sgpr + 2 v_and (3 instr)")
    R32(k_snop, "s_nop 1")
    R32(k_dep_and, "DEPENDENT chain: v_and_b32")
    R32(k_dep_alignbit, "DEPENDENT chain: v_alignbit_b32")
    R64(k_dep_lshr64, "DEPENDENT chain: v_lshrrev_b64")
    R64(k_dep_add64, "DEPENDENT chain: v_lshl_add_u64")
    R32(k_dep_cmp_cnd, "DEPENDENT chain: v_cmp + v_cndmask (2 instr)")
    R32(k_mul_lo, "v_mul_lo_u32")
    R32(k_mul_hi, "v_mul_hi_u32")
    R32(k_mul_u24, "v_mul_u32_u24")
    R32(k_mad_u24, "v_mad_u32_u24")
    R32(k_dot4_u8, "v_dot4_u32_u8")
    R32(k_dot8_u4, "v_dot8_u32_u4")
    R32(k_sad_u8, "v_sad_u8")
    R32(k_readlane, "v_readlane_b32 + v_mov (2 instr)")
    R64(k_lshr64, "v_lshrrev_b64 (const)")
    R64(k_lshr64_v, "v_lshrrev_b64 (v shift)")
    R64(k_lshl64_s, "v_lshlrev_b64 (s shift)")
    R64(k_add64, "v_lshl_add_u64")
    R32(k_bpermute, "ds_bpermute_b32")
    R32(k_ds_read32, "ds_read_b32 (lane-linear)")
    R64(k_ds_read64, "ds_read_b64 (lane-linear)")
    R32(k_ds_read_u16, "ds_read_u16 (lane-linear)")
    R32(k_ds_write16, "ds_write_b16 (lane-linear)")
    R32(k_ds_write32, "ds_write_b32 (lane-linear)")
    return 0;
}
