// valu_rates.hip -- issue cost of the integer VALU instructions the range coders lean on (gfx950).
// One wave per SIMD (1024 workgroups of 64), N independent-enough instructions per loop trip; prints cycles per
// instruction per wave (4.0 = full rate for wave64 on a 16-lane SIMD).
// build: hipcc -O2 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define S8(x) x "\n" x "\n" x "\n" x "\n" x "\n" x "\n" x "\n" x "\n"
#define S64(x) S8(S8(x))
#define KERNEL(name, body)                                                                     \
__global__ void name(unsigned *out, int trips) {                                               \
    unsigned a = threadIdx.x * 3u + 1u, b = threadIdx.x * 7u + 5u, c = a ^ 0x1234567u, d = b + 99u; \
    unsigned long long A = ((unsigned long long)a << 32) | b, B = ((unsigned long long)c << 32) | d;  \
    for (int t = 0; t < trips; t++) { body }                                            \
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d ^ (unsigned)A ^ (unsigned)(A >> 32) ^ (unsigned)B ^ (unsigned)(B >> 32);              \
}
KERNEL(k_add,      asm volatile(S64("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_mul_lo,   asm volatile(S64("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_mul_hi,   asm volatile(S64("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %3") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_mad24,    asm volatile(S64("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %2, %2, %3, %0") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_mad64,    asm volatile(S64("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %3, vcc, %2, %1, %3") : "+v"(A), "+v"(a), "+v"(b), "+v"(B) :: "vcc");)
KERNEL(k_shl64,    asm volatile(S64("v_lshlrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1") : "+v"(A), "+v"(B));)
KERNEL(k_add64,    asm volatile(S64("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0") : "+v"(A), "+v"(B));)
KERNEL(k_cmp64,    asm volatile(S64("v_cmp_gt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %0, %1") : "+v"(A), "+v"(B) :: "vcc");)
KERNEL(k_cmp32,    asm volatile(S64("v_cmp_gt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %0, %1") : "+v"(a), "+v"(b) :: "vcc");)
KERNEL(k_addc,     asm volatile(S64("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_pk,       asm volatile(S64("v_pk_sub_u16 %0, %0, %1\n v_pk_ashrrev_i16 %2, 5, %2") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_alignbit, asm volatile(S64("v_alignbit_b32 %0, %0, %1, 15\n v_alignbit_b32 %2, %2, %3, 15") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_cndmask,  asm volatile(S64("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_salu_mix, asm volatile(S64("v_add_u32 %0, %0, %1\n s_and_b64 vcc, vcc, exec\n v_add_u32 %2, %2, %3\n s_or_b64 vcc, vcc, exec") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc", "scc");)
KERNEL(k_cnd_e64,  asm volatile(S64("v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s10", "s11");)
KERNEL(k_cnd_dep,  asm volatile(S64("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %0, %1, %0, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_cmp_cnd,  asm volatile(S64("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32_e32 %2, %2, %3, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_cmp_x_cnd, asm volatile(S64("v_cmp_gt_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %3, %3, %1\n v_cndmask_b32_e32 %2, %2, %3, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_sdwa,     asm volatile(S64("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_dpp,      asm volatile(S64("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_ballot,   asm volatile(S64("v_cmp_gt_u32 vcc, %0, %1\n s_cmp_eq_u64 vcc, 0\n v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc", "scc");)
KERNEL(k_cnd_vop3vcc, asm volatile(S64("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_cnd_alt,   asm volatile(S64("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_add_u32 %2, %2, %3") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_cnd_22,    asm volatile(S64("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_add_u32 %1, %1, %0\n v_add_u32 %3, %3, %2") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_cnd_init,  asm volatile("s_mov_b64 vcc, 0x5555\n" S64("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_cnd_indep, asm volatile(S64("v_cndmask_b32_e32 %0, %1, %3, vcc\n v_cndmask_b32_e32 %2, %3, %1, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_addc_chain, asm volatile(S64("v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %3, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_cmp_cnd2,  asm volatile(S64("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %3, %3, %2, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");)
KERNEL(k_cmps_cnd2, asm volatile(S64("v_cmp_gt_u32 s[10:11], %0, %1\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n v_cndmask_b32_e64 %3, %3, %2, s[10:11]") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s10", "s11");)
KERNEL(k_dep_add,  asm volatile(S64("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_dep_mul,  asm volatile(S64("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
template <class K> static void run(const char *name, K k, int per_body)
{
    unsigned *out; hipMalloc(&out, 1024 * 64 * 4);
    const int trips = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, trips);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, trips); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)trips * 64 * per_body;
    printf("%-12s %.3f ms  %.2f ns per instruction per wave (x clock GHz = cycles)\n", name, ms, ms * 1e6 / inst);
    hipFree(out);
}
int main()
{
    run("add", k_add, 2); run("mul_lo", k_mul_lo, 2); run("mul_hi", k_mul_hi, 2); run("mad_u32_u24", k_mad24, 2);
    run("mad_u64_u32", k_mad64, 2); run("shift64", k_shl64, 2); run("lshl_add_u64", k_add64, 2); run("cmp_u64", k_cmp64, 2);
    run("cmp_u32", k_cmp32, 2); run("add_co/addc", k_addc, 2); run("pk_u16", k_pk, 2); run("alignbit", k_alignbit, 2);
    run("cndmask", k_cndmask, 2); run("valu+salu", k_salu_mix, 4); run("cnd e64 sgpr", k_cnd_e64, 2); run("cnd dep", k_cnd_dep, 2); run("cmp+cnd", k_cmp_cnd, 2); run("cmp,2add,cnd", k_cmp_x_cnd, 4); run("sdwa add", k_sdwa, 2); run("dpp add", k_dpp, 2); run("cmp,scmp,2add", k_ballot, 4); run("cnd vop3 vcc", k_cnd_vop3vcc, 2); run("cnd,add", k_cnd_alt, 2); run("2cnd,2add", k_cnd_22, 4); run("cnd vcc init", k_cnd_init, 2); run("cnd indep", k_cnd_indep, 2); run("addc chain", k_addc_chain, 2); run("cmp,2cnd vcc", k_cmp_cnd2, 3); run("cmp,2cnd sgpr", k_cmps_cnd2, 3); run("dep add", k_dep_add, 2); run("dep mul_lo", k_dep_mul, 2);
    return 0;
}
