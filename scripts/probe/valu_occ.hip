// valu_occ.hip -- the SIMD's AGGREGATE issue rate for the instruction classes the symbol loops use, at 1 / 2 / 3 / 4 / 8
// waves per SIMD (gfx950).  valu_rates.hip ran one wave per SIMD only, i.e. it measured a lone wave's issue rhythm; the limit
// arguments of DESIGN.md need the rate of the SIMD.
//   * a workgroup is 4 x K waves (K per SIMD: a workgroup's own waves go to consecutive SIMDs, profiles/r04_residency_pairs.log),
//     one workgroup per CU (dynamic LDS > half of the CU's), 256 workgroups; K = 8: two workgroups of 16 waves per CU.
//   * every body is 64 x a group of four instructions on FOUR independent register chains (no dependent-issue stall inside a wave),
//   * cycles = s_memtime ticks (= shader cycles, MI355X_MICROARCH.md) of the slowest wave; printed: cycles per wave-instruction
//     PER SIMD = cycles / (K x trips x instructions per trip).  2.0 = a SIMD-32 issuing a wave64 instruction every 2 cycles.
//   * --half: EXEC = lanes 0..31 only (does a half-empty wave issue in half the time?)
// build: hipcc -O2 --offload-arch=gfx950 valu_occ.hip -o valu_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#define S4(x) x "\n" x "\n" x "\n" x "\n"
#define S16(x) S4(S4(x))
#define S64(x) S4(S16(x))
typedef unsigned long long u64;

#define KERNEL(name, body)                                                                                   \
__global__ __launch_bounds__(1024) void name(unsigned *out, u64 *cyc, int trips, int half) {                 \
    extern __shared__ unsigned lds[];                                                                         \
    unsigned a = threadIdx.x * 3u + 1u, b = threadIdx.x * 7u + 5u, c = a ^ 0x1234567u, d = b + 99u;           \
    unsigned e = a + 77u, f = b ^ 0x55u, g = c + 3u, h = d ^ 0x9999u;                                          \
    u64 A = ((u64)a << 32) | b, B = ((u64)c << 32) | d;                                                        \
    const unsigned la = (threadIdx.x * 8u) & 0x3fffu;                                                          \
    lds[threadIdx.x] = a; __syncthreads();                                                                    \
    const u64 t0 = clock64();                                                                                 \
    if (!half || (threadIdx.x & 63u) < 32u) for (int t = 0; t < trips; t++) { body }                          \
    const u64 t1 = clock64();                                                                                 \
    if ((threadIdx.x & 63u) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                        \
    out[blockIdx.x * 1024 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (unsigned)A ^ (unsigned)(A >> 32) ^ (unsigned)B ^ (unsigned)(B >> 32) ^ la; \
}
#define V4 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
// four chains: (a,b) (c,d) (e,f) (g,h)
#define Q(op) op " %0, %0, %1\n" op " %2, %2, %3\n" op " %4, %4, %5\n" op " %6, %6, %7"
KERNEL(k_add,      asm volatile(S16(Q("v_add_u32")) V4);)
KERNEL(k_and,      asm volatile(S16(Q("v_and_b32")) V4);)
KERNEL(k_lshl_add, asm volatile(S16("v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %2, %2, 1, %3\n v_lshl_add_u32 %4, %4, 1, %5\n v_lshl_add_u32 %6, %6, 1, %7") V4);)
KERNEL(k_mul_lo,   asm volatile(S16(Q("v_mul_lo_u32")) V4);)
KERNEL(k_mul_hi,   asm volatile(S16(Q("v_mul_hi_u32")) V4);)
KERNEL(k_mul24,    asm volatile(S16(Q("v_mul_u32_u24")) V4);)
KERNEL(k_mad24,    asm volatile(S16("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %2, %2, %3, %4\n v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %6, %6, %7, %0") V4);)
KERNEL(k_mad64,    asm volatile(S16("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %0, vcc, %6, %7, %0\n v_mad_u64_u32 %1, vcc, %3, %5, %1") : "+v"(A), "+v"(B), "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");)
KERNEL(k_sdwa,     asm volatile(S16("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2") V4);)
KERNEL(k_dpp,      asm volatile(S16("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %4, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %6, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") V4);)
KERNEL(k_perm,     asm volatile(S16("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %4, %4, %5, %6\n v_perm_b32 %6, %6, %7, %0") V4);)
KERNEL(k_bfi,      asm volatile(S16("v_bfi_b32 %0, %1, %0, %2\n v_bfi_b32 %2, %3, %2, %4\n v_bfi_b32 %4, %5, %4, %6\n v_bfi_b32 %6, %7, %6, %0") V4);)
KERNEL(k_pk,       asm volatile(S16("v_pk_sub_i16 %0, %0, %1\n v_pk_ashrrev_i16 %2, 7, %2\n v_pk_add_i16 %4, %4, %5\n v_pk_sub_i16 %6, %6, %7") V4);)
KERNEL(k_cnd,      asm volatile(S16("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %4, %4, %5, vcc\n v_cndmask_b32_e32 %6, %6, %7, vcc") V4 :: "vcc");)
KERNEL(k_alignbit, asm volatile(S16("v_alignbit_b32 %0, %0, %1, 15\n v_alignbit_b32 %2, %2, %3, 15\n v_alignbit_b32 %4, %4, %5, 15\n v_alignbit_b32 %6, %6, %7, 15") V4);)
KERNEL(k_and_or,   asm volatile(S16("v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %2, %2, %3, %4\n v_and_or_b32 %4, %4, %5, %6\n v_and_or_b32 %6, %6, %7, %0") V4);)
KERNEL(k_addc,     asm volatile(S16("v_add_co_u32_e32 %0, vcc, %0, %1\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n v_addc_co_u32_e32 %2, vcc, %2, %3, vcc") V4 :: "vcc");)
KERNEL(k_addc_b2b, asm volatile(S16("v_add_co_u32_e32 %0, vcc, %0, %1\n v_addc_co_u32_e32 %2, vcc, %2, %3, vcc\n v_add_co_u32_e32 %4, vcc, %4, %5\n v_addc_co_u32_e32 %6, vcc, %6, %7, vcc") V4 :: "vcc");)
KERNEL(k_addc_nop, asm volatile(S16("v_add_co_u32_e32 %0, vcc, %0, %1\n s_nop 1\n v_addc_co_u32_e32 %2, vcc, %2, %3, vcc\n v_add_co_u32_e32 %4, vcc, %4, %5\n s_nop 1\n v_addc_co_u32_e32 %6, vcc, %6, %7, vcc") V4 :: "vcc");)
KERNEL(k_cmp_cnd,  asm volatile(S16("v_cmp_gt_u32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cmp_gt_u32_e32 vcc, %4, %5\n v_cndmask_b32_e32 %6, %6, %7, vcc") V4 :: "vcc");)
KERNEL(k_cmp_2_cnd, asm volatile(S16("v_cmp_gt_u32_e32 vcc, %0, %1\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n v_cndmask_b32_e32 %2, %2, %3, vcc") V4 :: "vcc");)
KERNEL(k_valu_salu, asm volatile(S16("v_add_u32 %0, %0, %1\n s_and_b64 vcc, vcc, exec\n v_add_u32 %2, %2, %3\n s_or_b64 vcc, vcc, exec") V4 :: "vcc", "scc");)
// --- second batch: which encodings are on the 2-cycle path?
#define Q2(op) op " %0, %1, %0\n" op " %2, %3, %2\n" op " %4, %5, %4\n" op " %6, %7, %6"
KERNEL(k_lshlrev,  asm volatile(S16("v_lshlrev_b32_e32 %0, 1, %0\n v_lshlrev_b32_e32 %2, 3, %2\n v_lshlrev_b32_e32 %4, 1, %4\n v_lshlrev_b32_e32 %6, 2, %6") V4);)
KERNEL(k_lshrrev,  asm volatile(S16("v_lshrrev_b32_e32 %0, 1, %0\n v_lshrrev_b32_e32 %2, 3, %2\n v_lshrrev_b32_e32 %4, 1, %4\n v_lshrrev_b32_e32 %6, 2, %6") V4);)
KERNEL(k_ashrrev,  asm volatile(S16("v_ashrrev_i32_e32 %0, 1, %0\n v_ashrrev_i32_e32 %2, 3, %2\n v_ashrrev_i32_e32 %4, 1, %4\n v_ashrrev_i32_e32 %6, 2, %6") V4);)
KERNEL(k_sub,      asm volatile(S16(Q("v_sub_u32_e32")) V4);)
KERNEL(k_subrev,   asm volatile(S16(Q("v_subrev_u32_e32")) V4);)
KERNEL(k_xor,      asm volatile(S16(Q("v_xor_b32_e32")) V4);)
KERNEL(k_or,       asm volatile(S16(Q("v_or_b32_e32")) V4);)
KERNEL(k_min,      asm volatile(S16(Q("v_min_u32_e32")) V4);)
KERNEL(k_max,      asm volatile(S16(Q("v_max_u32_e32")) V4);)
KERNEL(k_mov,      asm volatile(S16("v_mov_b32_e32 %0, %1\n v_mov_b32_e32 %2, %3\n v_mov_b32_e32 %4, %5\n v_mov_b32_e32 %6, %7") V4);)
KERNEL(k_not,      asm volatile(S16("v_not_b32_e32 %0, %0\n v_not_b32_e32 %2, %2\n v_not_b32_e32 %4, %4\n v_not_b32_e32 %6, %6") V4);)
KERNEL(k_add_lit,  asm volatile(S16("v_add_u32_e32 %0, 0x12345, %0\n v_add_u32_e32 %2, 0x54321, %2\n v_and_b32_e32 %4, 0x7fff7fff, %4\n v_add_u32_e32 %6, 0x1111, %6") V4);)
KERNEL(k_add_e64,  asm volatile(S16("v_add_u32_e64 %0, %0, %1\n v_add_u32_e64 %2, %2, %3\n v_add_u32_e64 %4, %4, %5\n v_add_u32_e64 %6, %6, %7") V4);)
KERNEL(k_add_sgpr, asm volatile(S16("v_add_u32_e32 %0, s20, %0\n v_add_u32_e32 %2, s21, %2\n v_add_u32_e32 %4, s20, %4\n v_add_u32_e32 %6, s21, %6") V4 :: "s20", "s21");)
KERNEL(k_add3,     asm volatile(S16("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %6, %6, %7, %0") V4);)
KERNEL(k_bfe,      asm volatile(S16("v_bfe_u32 %0, %0, 1, 31\n v_bfe_u32 %2, %2, 1, 31\n v_bfe_u32 %4, %4, 1, 31\n v_bfe_u32 %6, %6, 1, 31") V4);)
KERNEL(k_lshl_or,  asm volatile(S16("v_lshl_or_b32 %0, %0, 1, %1\n v_lshl_or_b32 %2, %2, 1, %3\n v_lshl_or_b32 %4, %4, 1, %5\n v_lshl_or_b32 %6, %6, 1, %7") V4);)
KERNEL(k_cmp_e32,  asm volatile(S16("v_cmp_gt_u32_e32 vcc, %0, %1\n v_cmp_lt_u32_e32 vcc, %2, %3\n v_cmp_gt_u32_e32 vcc, %4, %5\n v_cmp_lt_u32_e32 vcc, %6, %7") V4 :: "vcc");)
KERNEL(k_cmp_e64,  asm volatile(S16("v_cmp_gt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %2, %3\n v_cmp_gt_u32_e64 s[20:21], %4, %5\n v_cmp_lt_u32_e64 s[22:23], %6, %7") V4 :: "s20", "s21", "s22", "s23");)
KERNEL(k_cnd_e64s, asm volatile(S16("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %4, %4, %5, s[22:23]\n v_cndmask_b32_e64 %6, %6, %7, s[22:23]") V4 :: "s20", "s21", "s22", "s23");)
KERNEL(k_cnd_e64v, asm volatile(S16("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %4, %4, %5, vcc\n v_cndmask_b32_e64 %6, %6, %7, vcc") V4 :: "vcc");)
KERNEL(k_cnd_add,  asm volatile(S16("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_add_u32_e32 %2, %2, %3\n v_cndmask_b32_e32 %4, %4, %5, vcc\n v_add_u32_e32 %6, %6, %7") V4 :: "vcc");)
KERNEL(k_cnd_dpp,  asm volatile("s_mov_b64 vcc, 0x5555\n s_nop 0\n" S16("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") V4 :: "vcc");)
KERNEL(k_cnd_sdwa, asm volatile(S16("v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_cndmask_b32_sdwa %2, %2, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_cndmask_b32_sdwa %4, %4, %5, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_cndmask_b32_sdwa %6, %6, %7, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") V4 :: "vcc");)
KERNEL(k_mov_dpp,  asm volatile(S16("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") V4);)
KERNEL(k_mbcnt,    asm volatile(S16("v_mbcnt_lo_u32_b32 %0, s20, %0\n v_mbcnt_hi_u32_b32 %2, s21, %2\n v_mbcnt_lo_u32_b32 %4, s20, %4\n v_mbcnt_hi_u32_b32 %6, s21, %6") V4 :: "s20", "s21");)
KERNEL(k_rfl,      asm volatile(S16("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %2\n v_readfirstlane_b32 s22, %4\n v_readfirstlane_b32 s23, %6") V4 :: "s20", "s21", "s22", "s23");)
KERNEL(k_add_u16,  asm volatile(S16(Q("v_add_u16_e32")) V4);)
KERNEL(k_subb,     asm volatile(S16("v_subb_co_u32_e32 %0, vcc, %0, %1, vcc\n v_add_u32_e32 %2, %2, %3\n v_add_u32_e32 %4, %4, %5\n v_add_u32_e32 %6, %6, %7") V4 :: "vcc");)
KERNEL(k_simple_complex, asm volatile(S16("v_add_u32_e32 %0, %0, %1\n v_bfi_b32 %2, %3, %2, %4\n v_and_b32_e32 %4, %4, %5\n v_perm_b32 %6, %6, %7, %0") V4);)
KERNEL(k_salu4,    asm volatile(S16("s_add_u32 s20, s20, s21\n s_and_b32 s22, s22, s23\n s_lshl_b32 s21, s21, 1\n s_or_b32 s23, s23, s20") ::: "s20", "s21", "s22", "s23", "scc");)
KERNEL(k_snop,     asm volatile(S16("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0"));)
KERNEL(k_add_lds1, asm volatile(S16("v_add_u32_e32 %0, %0, %1\n ds_read_b32 %2, %8\n v_add_u32_e32 %4, %4, %5\n ds_read_b32 %6, %8 offset:256") "\n s_waitcnt lgkmcnt(0)" V4 : "v"(la) : "memory");)
KERNEL(k_ds_b32,   asm volatile(S16("ds_read_b32 %0, %8\n ds_read_b32 %2, %8 offset:256\n ds_read_b32 %4, %8 offset:512\n ds_read_b32 %6, %8 offset:768") "\n s_waitcnt lgkmcnt(0)" V4 : "v"(la) : "memory");)
KERNEL(k_ds_u16,   asm volatile(S16("ds_read_u16 %0, %8\n ds_read_u16 %2, %8 offset:256\n ds_read_u16 %4, %8 offset:512\n ds_read_u16 %6, %8 offset:768") "\n s_waitcnt lgkmcnt(0)" V4 : "v"(la) : "memory");)
KERNEL(k_ds_w16,   asm volatile(S16("ds_write_b16 %8, %0\n ds_write_b16 %8, %2 offset:256\n ds_write_b16 %8, %4 offset:512\n ds_write_b16 %8, %6 offset:768") "\n s_waitcnt lgkmcnt(0)" V4 : "v"(la) : "memory");)
KERNEL(k_ds_w32,   asm volatile(S16("ds_write_b32 %8, %0\n ds_write_b32 %8, %2 offset:256\n ds_write_b32 %8, %4 offset:512\n ds_write_b32 %8, %6 offset:768") "\n s_waitcnt lgkmcnt(0)" V4 : "v"(la) : "memory");)
KERNEL(k_ds_b128,  asm volatile(S16("ds_read_b128 v[60:63], %0\n ds_read_b128 v[64:67], %0 offset:1024\n ds_read_b128 v[68:71], %0 offset:2048\n ds_read_b128 v[72:75], %0 offset:3072") "\n s_waitcnt lgkmcnt(0)" :: "v"(la * 2u) : "memory", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75");)
KERNEL(k_ds_perm,  asm volatile(S16("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %6, %8, %6") "\n s_waitcnt lgkmcnt(0)" V4 : "v"(la) : "memory");)
// --- third batch: what separates two v_cndmask_b32 (VOP2 / DPP / SDWA: implicit VCC) enough?
KERNEL(k_cd_nop,   asm volatile("s_mov_b64 vcc, 0x5555\n s_nop 0\n" S16("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 0\n v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 0\n v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 0\n v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 0") V4 :: "vcc");)
KERNEL(k_cd_mov,   asm volatile("s_mov_b64 vcc, 0x5555\n s_nop 0\n" S16("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_e32 %1, %0\n v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_e32 %3, %2\n v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_e32 %5, %4\n v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_e32 %7, %6") V4 :: "vcc");)
KERNEL(k_cd_22,    asm volatile("s_mov_b64 vcc, 0x5555\n s_nop 0\n" S16("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_e32 %1, %1, %0\n v_add_u32_e32 %3, %3, %2\n v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_e32 %5, %5, %4\n v_add_u32_e32 %7, %7, %6") V4 :: "vcc");)
KERNEL(k_cd_vcmp,  asm volatile("v_cmp_gt_u32_e32 vcc, %0, %1\n s_nop 1\n" S16("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") V4 :: "vcc");)
KERNEL(k_cd_salu,  asm volatile("s_mov_b64 vcc, 0x5555\n s_nop 0\n" S16("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_and_b32 s20, s20, s21\n v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_and_b32 s20, s20, s21\n v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_and_b32 s20, s20, s21\n v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_and_b32 s20, s20, s21") V4 :: "vcc", "s20", "s21", "scc");)
KERNEL(k_cd_bfi,   asm volatile("s_mov_b64 vcc, 0x5555\n s_nop 0\n" S16("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_bfi_b32 %1, %0, %1, %3\n v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_bfi_b32 %3, %2, %3, %5\n v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_bfi_b32 %5, %4, %5, %7\n v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_bfi_b32 %7, %6, %7, %1") V4 :: "vcc");)
KERNEL(k_movdpp_cnd64, asm volatile("s_mov_b64 s[20:21], 0x5555\n" S16("v_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_e64 %0, %1, %0, s[20:21]\n v_mov_b32_dpp %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_e64 %2, %3, %2, s[20:21]") V4 :: "s20", "s21");)
KERNEL(k_movdpp_bfi, asm volatile(S16("v_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_bfi_b32 %0, %7, %1, %0\n v_mov_b32_dpp %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_bfi_b32 %2, %7, %3, %2") V4);)
KERNEL(k_cnd32_b2b_2, asm volatile(S16("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_add_u32_e32 %4, %4, %5\n v_add_u32_e32 %6, %6, %7") V4 :: "vcc");)
KERNEL(k_cnd32_3,  asm volatile(S16("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_add_u32_e32 %2, %2, %3\n v_add_u32_e32 %4, %4, %5\n v_add_u32_e32 %6, %6, %7") V4 :: "vcc");)
// --- fourth batch: how long may a run of VCC-implicit v_cndmask be?
KERNEL(k_run3,  asm volatile(S16("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %4, %4, %5, vcc\n v_add_u32_e32 %6, %6, %7") V4 :: "vcc");)
KERNEL(k_run4,  asm volatile(S16("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %4, %4, %5, vcc\n v_cndmask_b32_e32 %6, %6, %7, vcc\n v_add_u32_e32 %1, %1, %0\n v_add_u32_e32 %3, %3, %2") V4 :: "vcc");)
KERNEL(k_run6,  asm volatile(S16("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %4, %4, %5, vcc\n v_cndmask_b32_e32 %6, %6, %7, vcc\n v_cndmask_b32_e32 %0, %0, %3, vcc\n v_cndmask_b32_e32 %2, %2, %5, vcc\n v_add_u32_e32 %1, %1, %0\n v_add_u32_e32 %3, %3, %2") V4 :: "vcc");)
KERNEL(k_run3b, asm volatile(S16("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %4, %4, %5, vcc\n v_bfi_b32 %6, %7, %6, %0") V4 :: "vcc");)
// LDS mixes (counted: the s_waitcnt at the end of the body drains them)
KERNEL(k_ds_u8_mix, asm volatile(S16("ds_read_u8 %0, %8\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7") "\n s_waitcnt lgkmcnt(0)" V4 : "v"(la) : "memory");)
KERNEL(k_ds_b64_mix, asm volatile(S16("ds_read_b64 %0, %7\n v_add_u32 %1, %1, %2\n v_add_u32 %3, %3, %4\n v_add_u32 %5, %5, %6") "\n s_waitcnt lgkmcnt(0)" : "+v"(A), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(la) : "memory");)
KERNEL(k_ds_u8,    asm volatile(S16("ds_read_u8 %0, %8\n ds_read_u8 %2, %8 offset:64\n ds_read_u8 %4, %8 offset:128\n ds_read_u8 %6, %8 offset:192") "\n s_waitcnt lgkmcnt(0)" V4 : "v"(la) : "memory");)
// the decoder's mix as it stands (per symbol pair ~ 22 VALU + 5 LDS): 4 VALU : 1 LDS
KERNEL(k_mix41,    asm volatile(S16("v_add_u32 %0, %0, %1\n v_and_b32 %2, %2, %3\n v_lshl_add_u32 %4, %4, 1, %5\n v_perm_b32 %6, %6, %7, %0\n ds_read_u16 %1, %8") "\n s_waitcnt lgkmcnt(0)" V4 : "v"(la) : "memory");)

struct Case { const char *name; void (*k)(unsigned *, u64 *, int, int); int per_trip; };
#define C(n, k, per) { n, k, per }
static Case cases[] = {
    C("add_u32", k_add, 64), C("and_b32", k_and, 64), C("lshl_add_u32", k_lshl_add, 64), C("mul_lo_u32", k_mul_lo, 64), C("mul_hi_u32", k_mul_hi, 64),
    C("mul_u32_u24", k_mul24, 64), C("mad_u32_u24", k_mad24, 64), C("mad_u64_u32", k_mad64, 64), C("add sdwa", k_sdwa, 64), C("add dpp quad_perm", k_dpp, 64),
    C("perm_b32", k_perm, 64), C("bfi_b32", k_bfi, 64), C("pk_*_i16", k_pk, 64), C("cndmask vcc", k_cnd, 64), C("alignbit", k_alignbit, 64),
    C("and_or_b32", k_and_or, 64), C("add_co,2add,addc", k_addc, 64), C("add_co,addc b2b", k_addc_b2b, 64), C("add_co,nop1,addc", k_addc_nop, 64),
    C("cmp,cnd b2b", k_cmp_cnd, 64), C("cmp,2add,cnd", k_cmp_2_cnd, 64), C("valu+salu 1:1", k_valu_salu, 64),
    C("ds_read_u8+3add", k_ds_u8_mix, 64), C("ds_read_b64+3add", k_ds_b64_mix, 64), C("ds_read_u8 x4", k_ds_u8, 64), C("4valu+ds_u16", k_mix41, 80),
    C("lshlrev_b32", k_lshlrev, 64), C("lshrrev_b32", k_lshrrev, 64), C("ashrrev_i32", k_ashrrev, 64), C("sub_u32", k_sub, 64), C("subrev_u32", k_subrev, 64),
    C("xor_b32", k_xor, 64), C("or_b32", k_or, 64), C("min_u32", k_min, 64), C("max_u32", k_max, 64), C("mov_b32", k_mov, 64), C("not_b32", k_not, 64),
    C("add/and literal", k_add_lit, 64), C("add_u32_e64", k_add_e64, 64), C("add_u32 sgpr src", k_add_sgpr, 64), C("add3_u32", k_add3, 64), C("bfe_u32", k_bfe, 64),
    C("lshl_or_b32", k_lshl_or, 64), C("cmp_e32 (vcc)", k_cmp_e32, 64), C("cmp_e64 (sgpr)", k_cmp_e64, 64), C("cnd_e64 sgpr", k_cnd_e64s, 64), C("cnd_e64 vcc", k_cnd_e64v, 64),
    C("cnd_e32,add", k_cnd_add, 64), C("cnd_dpp vcc b2b", k_cnd_dpp, 64), C("cnd_sdwa vcc b2b", k_cnd_sdwa, 64), C("mov_dpp", k_mov_dpp, 64), C("mbcnt", k_mbcnt, 64),
    C("readfirstlane", k_rfl, 64), C("add_u16", k_add_u16, 64), C("subb,3add", k_subb, 64), C("add,bfi,and,perm", k_simple_complex, 64), C("salu x4", k_salu4, 64),
    C("s_nop 0", k_snop, 64), C("add,ds_b32 1:1", k_add_lds1, 64), C("ds_read_b32 x4", k_ds_b32, 64), C("ds_read_u16 x4", k_ds_u16, 64), C("ds_write_b16 x4", k_ds_w16, 64),
    C("ds_write_b32 x4", k_ds_w32, 64), C("ds_read_b128 x4", k_ds_b128, 64), C("ds_bpermute x4", k_ds_perm, 64),
    C("cnd_dpp,s_nop (x8)", k_cd_nop, 128), C("cnd_dpp,v_mov (x8)", k_cd_mov, 128), C("2cnd_dpp,2add (x8)", k_cd_22, 128), C("cnd_dpp b2b, valu vcc", k_cd_vcmp, 64),
    C("cnd_dpp,s_and (x8)", k_cd_salu, 128), C("cnd_dpp,bfi (x8)", k_cd_bfi, 128), C("mov_dpp,cnd_e64", k_movdpp_cnd64, 64), C("mov_dpp,bfi", k_movdpp_bfi, 64),
    C("2cnd_e32,2add", k_cnd32_b2b_2, 64), C("cnd_e32,3add", k_cnd32_3, 64),
    C("3cnd_e32,add", k_run3, 64), C("4cnd_e32,2add", k_run4, 96), C("6cnd_e32,2add", k_run6, 128), C("3cnd_e32,bfi", k_run3b, 64),
};

int main(int argc, char **argv)
{
    int half = 0, trips = 1500; const char *only = nullptr;
    for (int i = 1; i < argc; i++) { if (!strcmp(argv[i], "--half")) half = 1; if (!strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i]; if (!strcmp(argv[i], "--trips") && i + 1 < argc) trips = atoi(argv[++i]); }
    unsigned *out; u64 *cyc; (void)hipMalloc(&out, 512 * 1024 * 4); (void)hipMalloc(&cyc, 512 * 16 * 8);
    std::vector<u64> h(512 * 16);
    const int KS[5] = { 1, 2, 3, 4, 8 };
    printf("# cycles per wave-instruction PER SIMD (slowest wave's s_memtime ticks / (K x trips x instructions)); %s\n", half ? "EXEC = lanes 0..31" : "EXEC = all 64 lanes");
    printf("%-20s", "class \\ waves/SIMD"); for (int k : KS) printf(" %7d", k); printf("   | one wave's cycles per instruction at K = 1 / 4 / 8\n");
    for (const Case &cs : cases) {
        if (only && !strstr(cs.name, only)) continue;
        printf("%-20s", cs.name);
        double lone[5];
        int idx = 0;
        for (int K : KS) {
            const int wgs_per_cu = K == 8 ? 2 : 1, waves = K == 8 ? 16 : 4 * K;
            const size_t lds = K == 8 ? 70 * 1024 : 100 * 1024;                 // one (two) workgroup(s) per CU
            (void)hipFuncSetAttribute((const void *)cs.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int grid = 256 * wgs_per_cu;
            (void)hipMemset(cyc, 0, 512 * 16 * 8);
            hipLaunchKernelGGL(cs.k, dim3(grid), dim3(64 * waves), lds, 0, out, cyc, trips, half);   // warm
            hipLaunchKernelGGL(cs.k, dim3(grid), dim3(64 * waves), lds, 0, out, cyc, trips, half);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), cyc, 512 * 16 * 8, hipMemcpyDeviceToHost);
            u64 mx = 0; for (int i = 0; i < grid; i++) for (int w = 0; w < waves; w++) if (h[i * 16 + w] > mx) mx = h[i * 16 + w];
            const double per = (double)mx / ((double)K * trips * cs.per_trip);
            lone[idx++] = (double)mx / ((double)trips * cs.per_trip);
            printf(" %7.2f", per);
        }
        printf("   | %.2f / %.2f / %.2f\n", lone[0], lone[3], lone[4]);
    }
    return 0;
}
