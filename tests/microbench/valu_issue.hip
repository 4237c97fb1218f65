// valu_issue.hip -- gfx950 microbenchmark: how many wave64 VALU instructions does one SIMD issue per cycle, for the
// instruction kinds k_poa_dp_pk is made of, at 1 / 2 / 4 / 6 / 8 resident waves per SIMD?
// ("is wave64 int32 VALU issue 2 or 4 cycles here?")  Build: hipcc -O2 --offload-arch=gfx950 valu_issue.hip -o valu_issue
// Every block is 256 threads = one wave per SIMD of its CU; `b` blocks per CU are made resident by launching 256*b blocks.
// Per kind: 8 independent register chains, 32 instructions per loop trip, so a single wave is never latency-bound.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#define R8(OP, TAIL) \
    OP " %0, %0, %8" TAIL "\n\t" OP " %1, %1, %8" TAIL "\n\t" OP " %2, %2, %8" TAIL "\n\t" OP " %3, %3, %8" TAIL "\n\t" \
    OP " %4, %4, %8" TAIL "\n\t" OP " %5, %5, %8" TAIL "\n\t" OP " %6, %6, %8" TAIL "\n\t" OP " %7, %7, %8" TAIL "\n\t"
#define R32(OP, TAIL) R8(OP, TAIL) R8(OP, TAIL) R8(OP, TAIL) R8(OP, TAIL)
#define BODY2(OP, TAIL) asm volatile(R32(OP, TAIL) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))

enum { K_ADD, K_MAX, K_MAXDPP, K_PKADD, K_PKMAX, K_MAX3, K_LSHLOR, K_PERM, K_SDWA, K_CNDMASK, K_CMPCND, K_READLANE, K_DEP, K_MIXSALU, K_MOVDPP, K_BFE, K_SUB, K_AND, K_OR, K_XOR, K_LSHL, K_ASHR, K_MIN, K_MINU, K_ADD3, K_OR3, K_ANDOR, K_LSHLADD, K_MAD24, K_MADU24, K_MED3, K_PKSUB, K_PKMIN, K_PKMAD, K_PKLSHR, K_ADDF, K_FMA, K_MULLO, K_ALIGNBIT, K_SAD, K_ADDCO, K_SUBREV, K_MOV, K_MAXU16, K_ADD_SGPR, K_MAX_SGPR, K_CND_VCC_SALU, K_CND_SPAIR_SALU, K_CND_SPAIR_VALU, K_CND_ROT4, K_ADD_LIT, K_MAX_INL, K_SUBB_CHAIN, K_SSXX, K_S4X4, K_S3X1, K_ADD_INL, K_SX_DIFFWAVE, K_MAX3_SALU11, K_MAX3_SALU21, K_MAX3_SALU12, K_MAX3_BR, K_NKINDS };
static const char *kind_name[] = {"v_add_u32", "v_max_i32", "v_max_i32_dpp row_shr:1", "v_pk_add_i16", "v_pk_max_i16", "v_max3_i32", "v_lshl_or_b32",
                                  "v_perm_b32", "v_sub_u32_sdwa BYTE_0", "v_cndmask_b32 (vcc)", "v_cmp_gt_i32 + v_cndmask_b32 pairs", "v_readlane_b32",
                                  "v_add_u32 dependent chain", "v_add_u32 + s_add_u32 1:1", "v_mov_b32_dpp row_shr:1", "v_bfe_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_min_i32", "v_min_u32", "v_add3_u32", "v_or3_b32", "v_and_or_b32", "v_lshl_add_u32", "v_mad_i32_i24", "v_mad_u32_u24", "v_med3_i32", "v_pk_sub_i16", "v_pk_min_i16", "v_pk_mad_u16", "v_pk_lshrrev_b16", "v_add_f32", "v_fma_f32", "v_mul_lo_u32", "v_alignbit_b32", "v_sad_u32", "v_add_co_u32", "v_subrev_u32", "v_mov_b32", "v_max_u16", "v_add_u32 v, s, v (SGPR operand)", "v_max_i32 v, s, v (SGPR operand)", "v_cndmask_b32 vcc (vcc written by SALU)", "v_cndmask_b32_e64 mask in s[a:b] written by SALU", "v_cndmask_b32_e64 mask in s[a:b] written by v_cmp", "v_cndmask_b32_e64 rotating over 4 SGPR-pair masks (v_cmp written)", "v_add_u32 v, 0x12345, v (literal)", "v_max_i32 v, 5, v (inline const)", "v_sub_u32 then v_min_i32 alternating", "v_sub,v_sub,v_min,v_min repeating", "v_sub x4 then v_min x4 repeating", "v_sub x3 then v_min x1 repeating", "v_add_u32 v, 4, v (inline const)", "(unused)", "v_max3_i32 + s_add_u32 1:1 (VALU rate)", "v_max3_i32 x2 + s_add_u32 x1 (VALU rate)", "v_max3_i32 x1 + s_add_u32 x2 (VALU rate)", "v_max3_i32 x4 + s_cmp/s_cbranch (not taken) (VALU rate)"};

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(int iters, unsigned long long *cyc, int *sink)
{
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int b = threadIdx.x * 3 + 1;
    int s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    unsigned long long m64 = 0x5555aaaa3333ccccull + (unsigned long long)iters;
    asm volatile("" : "+s"(m64));
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; i++) {
        if constexpr (KIND == K_ADD) BODY2("v_add_u32", "");
        else if constexpr (KIND == K_MAX) BODY2("v_max_i32", "");
        else if constexpr (KIND == K_MAXDPP) {
            asm volatile(R32("v_max_i32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if constexpr (KIND == K_PKADD) BODY2("v_pk_add_i16", "");
        else if constexpr (KIND == K_PKMAX) BODY2("v_pk_max_i16", "");
        else if constexpr (KIND == K_MAX3) BODY2("v_max3_i32", ", %8");
        else if constexpr (KIND == K_LSHLOR) BODY2("v_lshl_or_b32", ", 3");
        else if constexpr (KIND == K_PERM) BODY2("v_perm_b32", ", %8");
        else if constexpr (KIND == K_SDWA) {
            asm volatile(R32("v_sub_u32_sdwa", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if constexpr (KIND == K_CNDMASK) {
            asm volatile("v_cmp_gt_i32 vcc, %8, %0\n\t" R32("v_cndmask_b32", ", vcc") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        } else if constexpr (KIND == K_CMPCND) {
#define CC(n) "v_cmp_gt_i32 vcc, %8, %" #n "\n\tv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n\t"
            asm volatile(CC(0) CC(1) CC(2) CC(3) CC(4) CC(5) CC(6) CC(7) CC(0) CC(1) CC(2) CC(3) CC(4) CC(5) CC(6) CC(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        } else if constexpr (KIND == K_READLANE) {
#define RL(s, v) "v_readlane_b32 %" #s ", %" #v ", 5\n\t"
            asm volatile(RL(0, 4) RL(1, 5) RL(2, 6) RL(3, 7) RL(0, 8) RL(1, 9) RL(2, 10) RL(3, 11) RL(0, 4) RL(1, 5) RL(2, 6) RL(3, 7) RL(0, 8) RL(1, 9) RL(2, 10) RL(3, 11)
                         RL(0, 4) RL(1, 5) RL(2, 6) RL(3, 7) RL(0, 8) RL(1, 9) RL(2, 10) RL(3, 11) RL(0, 4) RL(1, 5) RL(2, 6) RL(3, 7) RL(0, 8) RL(1, 9) RL(2, 10) RL(3, 11)
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if constexpr (KIND == K_DEP) {
#define D8 "v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t"
            asm volatile(D8 D8 D8 D8 : "+v"(a0) : "v"(b));
        } else if constexpr (KIND == K_MIXSALU) {
#define VS(n, s) "v_add_u32 %" #n ", %" #n ", %12\n\ts_add_u32 %" #s ", %" #s ", 3\n\t"
            asm volatile(VS(0, 8) VS(1, 9) VS(2, 10) VS(3, 11) VS(4, 8) VS(5, 9) VS(6, 10) VS(7, 11) VS(0, 8) VS(1, 9) VS(2, 10) VS(3, 11) VS(4, 8) VS(5, 9) VS(6, 10) VS(7, 11)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b) : "scc");
        } else if constexpr (KIND == K_MOVDPP) {
#define MD(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(MD(0) MD(1) MD(2) MD(3) MD(4) MD(5) MD(6) MD(7) MD(0) MD(1) MD(2) MD(3) MD(4) MD(5) MD(6) MD(7) MD(0) MD(1) MD(2) MD(3) MD(4) MD(5) MD(6) MD(7) MD(0) MD(1) MD(2) MD(3) MD(4) MD(5) MD(6) MD(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == K_BFE) BODY2("v_bfe_u32", ", 5");
        else if constexpr (KIND == K_SUB) BODY2("v_sub_u32", "");
        else if constexpr (KIND == K_AND) BODY2("v_and_b32", "");
        else if constexpr (KIND == K_OR) BODY2("v_or_b32", "");
        else if constexpr (KIND == K_XOR) BODY2("v_xor_b32", "");
        else if constexpr (KIND == K_LSHL) BODY2("v_lshlrev_b32", "");
        else if constexpr (KIND == K_ASHR) BODY2("v_ashrrev_i32", "");
        else if constexpr (KIND == K_MIN) BODY2("v_min_i32", "");
        else if constexpr (KIND == K_MINU) BODY2("v_min_u32", "");
        else if constexpr (KIND == K_ADD3) BODY2("v_add3_u32", ", %8");
        else if constexpr (KIND == K_OR3) BODY2("v_or3_b32", ", %8");
        else if constexpr (KIND == K_ANDOR) BODY2("v_and_or_b32", ", %8");
        else if constexpr (KIND == K_LSHLADD) BODY2("v_lshl_add_u32", ", 3");
        else if constexpr (KIND == K_MAD24) BODY2("v_mad_i32_i24", ", %8");
        else if constexpr (KIND == K_MADU24) BODY2("v_mad_u32_u24", ", %8");
        else if constexpr (KIND == K_MED3) BODY2("v_med3_i32", ", %8");
        else if constexpr (KIND == K_PKSUB) BODY2("v_pk_sub_i16", "");
        else if constexpr (KIND == K_PKMIN) BODY2("v_pk_min_i16", "");
        else if constexpr (KIND == K_PKMAD) BODY2("v_pk_mad_u16", ", %8");
        else if constexpr (KIND == K_PKLSHR) BODY2("v_pk_lshrrev_b16", "");
        else if constexpr (KIND == K_ADDF) BODY2("v_add_f32", "");
        else if constexpr (KIND == K_FMA) BODY2("v_fma_f32", ", %8");
        else if constexpr (KIND == K_MULLO) BODY2("v_mul_lo_u32", "");
        else if constexpr (KIND == K_ALIGNBIT) BODY2("v_alignbit_b32", ", 8");
        else if constexpr (KIND == K_SAD) BODY2("v_sad_u32", ", %8");
        else if constexpr (KIND == K_SUBREV) BODY2("v_subrev_u32", "");
        else if constexpr (KIND == K_MAXU16) BODY2("v_max_u16", "");
        else if constexpr (KIND == K_ADD_SGPR) {
            asm volatile(R32("v_add_u32", "") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0));
        } else if constexpr (KIND == K_MAX_SGPR) {
            asm volatile(R32("v_max_i32", "") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0));
        } else if constexpr (KIND == K_CND_VCC_SALU) {
            asm volatile("s_mov_b64 vcc, %9\n\ts_nop 4\n\t" R32("v_cndmask_b32", ", vcc") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m64) : "vcc");
        } else if constexpr (KIND == K_CND_SPAIR_SALU) {
            asm volatile(R32("v_cndmask_b32_e64", ", %9") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m64));
        } else if constexpr (KIND == K_CND_SPAIR_VALU) {
            unsigned long long mk;
            asm volatile("v_cmp_gt_i32_e64 %0, %1, %2\n\ts_nop 4" : "=s"(mk) : "v"(b), "v"(a0));
            asm volatile(R32("v_cndmask_b32_e64", ", %9") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(mk));
        } else if constexpr (KIND == K_CND_ROT4) {
            unsigned long long k0, k1, k2, k3;
            asm volatile("v_cmp_gt_i32_e64 %0, %4, %5\n\tv_cmp_gt_i32_e64 %1, %4, %6\n\tv_cmp_gt_i32_e64 %2, %4, %7\n\tv_cmp_gt_i32_e64 %3, %4, %8\n\ts_nop 4"
                         : "=s"(k0), "=s"(k1), "=s"(k2), "=s"(k3) : "v"(b), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
#define CR(n, m) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, %" #m "\n\t"
            asm volatile(CR(0, 9) CR(1, 10) CR(2, 11) CR(3, 12) CR(4, 9) CR(5, 10) CR(6, 11) CR(7, 12) CR(0, 9) CR(1, 10) CR(2, 11) CR(3, 12) CR(4, 9) CR(5, 10) CR(6, 11) CR(7, 12)
                         CR(0, 9) CR(1, 10) CR(2, 11) CR(3, 12) CR(4, 9) CR(5, 10) CR(6, 11) CR(7, 12) CR(0, 9) CR(1, 10) CR(2, 11) CR(3, 12) CR(4, 9) CR(5, 10) CR(6, 11) CR(7, 12)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(k0), "s"(k1), "s"(k2), "s"(k3));
        } else if constexpr (KIND == K_ADD_LIT) {
#define AL(n) "v_add_u32 %" #n ", 0x12345, %" #n "\n\t"
            asm volatile(AL(0) AL(1) AL(2) AL(3) AL(4) AL(5) AL(6) AL(7) AL(0) AL(1) AL(2) AL(3) AL(4) AL(5) AL(6) AL(7) AL(0) AL(1) AL(2) AL(3) AL(4) AL(5) AL(6) AL(7) AL(0) AL(1) AL(2) AL(3) AL(4) AL(5) AL(6) AL(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == K_MAX_INL) {
#define MI(n) "v_max_i32 %" #n ", 5, %" #n "\n\t"
            asm volatile(MI(0) MI(1) MI(2) MI(3) MI(4) MI(5) MI(6) MI(7) MI(0) MI(1) MI(2) MI(3) MI(4) MI(5) MI(6) MI(7) MI(0) MI(1) MI(2) MI(3) MI(4) MI(5) MI(6) MI(7) MI(0) MI(1) MI(2) MI(3) MI(4) MI(5) MI(6) MI(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == K_SUBB_CHAIN) {
#define SM(n) "v_sub_u32 %" #n ", %" #n ", %8\n\tv_min_i32 %" #n ", %" #n ", %8\n\t"
            asm volatile(SM(0) SM(1) SM(2) SM(3) SM(4) SM(5) SM(6) SM(7) SM(0) SM(1) SM(2) SM(3) SM(4) SM(5) SM(6) SM(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        }
        else if constexpr (KIND == K_SSXX) {
#define S_(n) "v_sub_u32 %" #n ", %" #n ", %8\n\t"
#define X_(n) "v_min_i32 %" #n ", %" #n ", %8\n\t"
            asm volatile(S_(0) S_(1) X_(2) X_(3) S_(4) S_(5) X_(6) X_(7) S_(0) S_(1) X_(2) X_(3) S_(4) S_(5) X_(6) X_(7) S_(0) S_(1) X_(2) X_(3) S_(4) S_(5) X_(6) X_(7) S_(0) S_(1) X_(2) X_(3) S_(4) S_(5) X_(6) X_(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if constexpr (KIND == K_S4X4) {
            asm volatile(S_(0) S_(1) S_(2) S_(3) X_(4) X_(5) X_(6) X_(7) S_(0) S_(1) S_(2) S_(3) X_(4) X_(5) X_(6) X_(7) S_(0) S_(1) S_(2) S_(3) X_(4) X_(5) X_(6) X_(7) S_(0) S_(1) S_(2) S_(3) X_(4) X_(5) X_(6) X_(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if constexpr (KIND == K_S3X1) {
            asm volatile(S_(0) S_(1) S_(2) X_(3) S_(4) S_(5) S_(6) X_(7) S_(0) S_(1) S_(2) X_(3) S_(4) S_(5) S_(6) X_(7) S_(0) S_(1) S_(2) X_(3) S_(4) S_(5) S_(6) X_(7) S_(0) S_(1) S_(2) X_(3) S_(4) S_(5) S_(6) X_(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if constexpr (KIND == K_ADD_INL) {
#define AI(n) "v_add_u32 %" #n ", 4, %" #n "\n\t"
            asm volatile(AI(0) AI(1) AI(2) AI(3) AI(4) AI(5) AI(6) AI(7) AI(0) AI(1) AI(2) AI(3) AI(4) AI(5) AI(6) AI(7) AI(0) AI(1) AI(2) AI(3) AI(4) AI(5) AI(6) AI(7) AI(0) AI(1) AI(2) AI(3) AI(4) AI(5) AI(6) AI(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
        else if constexpr (KIND == K_MAX3_SALU11) {
#define M3(n) "v_max3_i32 %" #n ", %" #n ", %12, %12\n\t"
#define SA(s) "s_add_u32 %" #s ", %" #s ", 3\n\t"
            asm volatile(M3(0) SA(8) M3(1) SA(9) M3(2) SA(10) M3(3) SA(11) M3(4) SA(8) M3(5) SA(9) M3(6) SA(10) M3(7) SA(11) M3(0) SA(8) M3(1) SA(9) M3(2) SA(10) M3(3) SA(11) M3(4) SA(8) M3(5) SA(9) M3(6) SA(10) M3(7) SA(11)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b) : "scc");
        } else if constexpr (KIND == K_MAX3_SALU21) {
            asm volatile(M3(0) M3(1) SA(8) M3(2) M3(3) SA(9) M3(4) M3(5) SA(10) M3(6) M3(7) SA(11) M3(0) M3(1) SA(8) M3(2) M3(3) SA(9) M3(4) M3(5) SA(10) M3(6) M3(7) SA(11)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b) : "scc");
        } else if constexpr (KIND == K_MAX3_SALU12) {
            asm volatile(M3(0) SA(8) SA(9) M3(1) SA(10) SA(11) M3(2) SA(8) SA(9) M3(3) SA(10) SA(11) M3(4) SA(8) SA(9) M3(5) SA(10) SA(11) M3(6) SA(8) SA(9) M3(7) SA(10) SA(11) M3(0) SA(8) SA(9) M3(1) SA(10) SA(11) M3(2) SA(8) SA(9) M3(3) SA(10) SA(11) M3(4) SA(8) SA(9) M3(5) SA(10) SA(11) M3(6) SA(8) SA(9) M3(7) SA(10) SA(11)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b) : "scc");
        } else if constexpr (KIND == K_MAX3_BR) {
#define BR(s, l) "s_cmp_eq_u32 %" #s ", 0x7fffffff\n\ts_cbranch_scc1 1f\n\t"
            asm volatile(M3(0) M3(1) M3(2) M3(3) BR(8, 1) M3(4) M3(5) M3(6) M3(7) BR(9, 2) M3(0) M3(1) M3(2) M3(3) BR(10, 3) M3(4) M3(5) M3(6) M3(7) BR(11, 4) "1:\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b) : "scc");
        } else if constexpr (KIND == K_ADDCO) {
#define AC(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n\t"
            asm volatile(AC(0) AC(1) AC(2) AC(3) AC(4) AC(5) AC(6) AC(7) AC(0) AC(1) AC(2) AC(3) AC(4) AC(5) AC(6) AC(7) AC(0) AC(1) AC(2) AC(3) AC(4) AC(5) AC(6) AC(7) AC(0) AC(1) AC(2) AC(3) AC(4) AC(5) AC(6) AC(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        } else if constexpr (KIND == K_MOV) {
#define MV(n, m) "v_mov_b32 %" #n ", %" #m "\n\t"
            asm volatile(MV(0, 1) MV(1, 2) MV(2, 3) MV(3, 4) MV(4, 5) MV(5, 6) MV(6, 7) MV(7, 0) MV(0, 1) MV(1, 2) MV(2, 3) MV(3, 4) MV(4, 5) MV(5, 6) MV(6, 7) MV(7, 0)
                         MV(0, 1) MV(1, 2) MV(2, 3) MV(3, 4) MV(4, 5) MV(5, 6) MV(6, 7) MV(7, 0) MV(0, 1) MV(1, 2) MV(2, 3) MV(3, 4) MV(4, 5) MV(5, 6) MV(6, 7) MV(7, 0)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0 + s1 + s2 + s3 == 0x7fffffff) sink[0] = 1;
}

static int insts_per_trip(int kind) { return kind == K_MIXSALU || kind == K_MAX3_SALU11 || kind == K_MAX3_SALU21 || kind == K_MAX3_SALU12 || kind == K_MAX3_BR ? 16 : 32; }  // VALU instructions per loop trip (K_CMPCND: 32 = 16 cmp + 16 cndmask)

template <int KIND>
static void run(int cus, unsigned long long *d_cyc, int *d_sink)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int b : {1, 2, 6}) {
        const int blocks = cus * b;
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, 2000, d_cyc, d_sink);  // warm-up (clocks)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, iters, d_cyc, d_sink);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc((size_t)blocks * 4);
        CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        std::sort(cyc.begin(), cyc.end());
        const double med = (double)cyc[cyc.size() / 2];
        const double n = (double)iters * insts_per_trip(KIND);
        // per SIMD: b waves each issue n instructions in `med` shader cycles (s_memtime) => cycles per wave-instruction of the SIMD
        const double cyc_per_inst_simd = med / (n * b);
        const double ginst_s = n * b * 4.0 * cus / (ms * 1e-3) / 1e9;  // wave-instructions per second, whole chip (wall clock)
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_inst_one_wave\": %.3f, \"cycles_per_wave_inst_per_simd\": %.3f, "
               "\"chip_Gwaveinst_per_s\": %.1f, \"ms\": %.3f, \"eff_clock_GHz\": %.3f}\n",
               kind_name[KIND], b, med / n, cyc_per_inst_simd, ginst_s, ms, med / (ms * 1e-3) / 1e9);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char **argv)
{
    const bool only_new = argc > 1;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    fprintf(stderr, "%s, %d CUs, clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    unsigned long long *d_cyc; int *d_sink;
    CK(hipMalloc(&d_cyc, (size_t)cus * 8 * 4 * 8));
    CK(hipMalloc(&d_sink, 4));
    if (only_new) {
        run<K_SUB>(cus, d_cyc, d_sink); run<K_MIN>(cus, d_cyc, d_sink); run<K_SUBB_CHAIN>(cus, d_cyc, d_sink); run<K_SSXX>(cus, d_cyc, d_sink); run<K_S4X4>(cus, d_cyc, d_sink); run<K_S3X1>(cus, d_cyc, d_sink); run<K_ADD_INL>(cus, d_cyc, d_sink);
        run<K_MAX3>(cus, d_cyc, d_sink); run<K_MAX3_SALU11>(cus, d_cyc, d_sink); run<K_MAX3_SALU21>(cus, d_cyc, d_sink); run<K_MAX3_SALU12>(cus, d_cyc, d_sink); run<K_MAX3_BR>(cus, d_cyc, d_sink);
        return 0;
    }
    run<K_ADD>(cus, d_cyc, d_sink);
    run<K_MAX>(cus, d_cyc, d_sink);
    run<K_MAXDPP>(cus, d_cyc, d_sink);
    run<K_MOVDPP>(cus, d_cyc, d_sink);
    run<K_PKADD>(cus, d_cyc, d_sink);
    run<K_PKMAX>(cus, d_cyc, d_sink);
    run<K_MAX3>(cus, d_cyc, d_sink);
    run<K_LSHLOR>(cus, d_cyc, d_sink);
    run<K_BFE>(cus, d_cyc, d_sink);
    run<K_PERM>(cus, d_cyc, d_sink);
    run<K_SDWA>(cus, d_cyc, d_sink);
    run<K_CNDMASK>(cus, d_cyc, d_sink);
    run<K_CMPCND>(cus, d_cyc, d_sink);
    run<K_READLANE>(cus, d_cyc, d_sink);
    run<K_DEP>(cus, d_cyc, d_sink);
    run<K_MIXSALU>(cus, d_cyc, d_sink);
    run<K_SUB>(cus, d_cyc, d_sink);
    run<K_AND>(cus, d_cyc, d_sink);
    run<K_OR>(cus, d_cyc, d_sink);
    run<K_XOR>(cus, d_cyc, d_sink);
    run<K_LSHL>(cus, d_cyc, d_sink);
    run<K_ASHR>(cus, d_cyc, d_sink);
    run<K_MIN>(cus, d_cyc, d_sink);
    run<K_MINU>(cus, d_cyc, d_sink);
    run<K_ADD3>(cus, d_cyc, d_sink);
    run<K_OR3>(cus, d_cyc, d_sink);
    run<K_ANDOR>(cus, d_cyc, d_sink);
    run<K_LSHLADD>(cus, d_cyc, d_sink);
    run<K_MAD24>(cus, d_cyc, d_sink);
    run<K_MADU24>(cus, d_cyc, d_sink);
    run<K_MED3>(cus, d_cyc, d_sink);
    run<K_PKSUB>(cus, d_cyc, d_sink);
    run<K_PKMIN>(cus, d_cyc, d_sink);
    run<K_PKMAD>(cus, d_cyc, d_sink);
    run<K_PKLSHR>(cus, d_cyc, d_sink);
    run<K_ADDF>(cus, d_cyc, d_sink);
    run<K_FMA>(cus, d_cyc, d_sink);
    run<K_MULLO>(cus, d_cyc, d_sink);
    run<K_ALIGNBIT>(cus, d_cyc, d_sink);
    run<K_SAD>(cus, d_cyc, d_sink);
    run<K_ADDCO>(cus, d_cyc, d_sink);
    run<K_SUBREV>(cus, d_cyc, d_sink);
    run<K_MOV>(cus, d_cyc, d_sink);
    run<K_MAXU16>(cus, d_cyc, d_sink);
    run<K_ADD_SGPR>(cus, d_cyc, d_sink);
    run<K_MAX_SGPR>(cus, d_cyc, d_sink);
    run<K_CND_VCC_SALU>(cus, d_cyc, d_sink);
    run<K_CND_SPAIR_SALU>(cus, d_cyc, d_sink);
    run<K_CND_SPAIR_VALU>(cus, d_cyc, d_sink);
    run<K_CND_ROT4>(cus, d_cyc, d_sink);
    run<K_ADD_LIT>(cus, d_cyc, d_sink);
    run<K_MAX_INL>(cus, d_cyc, d_sink);
    run<K_SUBB_CHAIN>(cus, d_cyc, d_sink);
    return 0;
}
