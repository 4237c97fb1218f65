// vga_poa_t4.hpp -- K4 "t4": the banded POA DP of k_poa_dp_pk (same algorithm, same outputs, bit-exact against
// oracle/og_poa.c) re-formulated for fewer VALU instructions per cell.  gfx950 issues the instruction kinds this kernel
// is made of at one wave64 instruction per ~4 cycles per SIMD whatever the mix (profiles/r02_valu_issue_microbench.txt),
// so the instruction count is the only lever.  What changed against k_poa_dp_pk:
//   * every score is carried SCALED BY 4; the two free low bits hold an argmax TAG at the two places a maximum of three
//     candidates is taken:   Ht' = max3(4M + 2, 4E1 + 1, 4E2 + 0),   H'' = max3(4Ht + 3, 4F1 + 1, 4F2 + 0).
//     One v_max3_i32 yields the value (bits 31..2) and, with the spec's tie order M > E1 > E2 and Ht > F1 > F2, the
//     source (bits 1..0): no compare / select chain.  The tags ride on constants that are added anyway.
//   * row state in LDS as separate planes: Hs[col] = 4H (int32) and Gs[col] = G1 | G2 << 8 with G1 = 4 g1 - 1,
//     G2 = 4 g2  (g_k = H - E_k(next row) in [e_k, o_k + e_k]): the E candidates of the next row are single SDWA
//     subtractions of a byte (v_sub_u32_sdwa src1_sel:BYTE_n) and arrive tagged; nothing is packed or unpacked.
//   * the new G bytes and the four direction bytes of a lane are DEPOSITED by the instruction that computes them
//     (dst_sel:BYTE_n dst_unused:UNUSED_PRESERVE); the one-bit flags are shifted into the direction byte by
//     v_cmp + v_addc_co_u32 pairs (code = 2 code + flag).
//   * "E_k opened here" is a property of the PREDECESSOR cell (H - (o+e) >= E - e there), so it is computed where that
//     cell's G byte is computed and stored in that cell's direction byte; the traceback reads it on arrival.
//   * the cross-wave part of the max-plus scan reads the other waves' totals with broadcast LDS reads and uniform
//     code instead of eight v_readlane.
// Direction byte (ENC 1 of poa_traceback_*):  [7:6] tag of H'' (3 Ht, 1 F1, 0 F2), [5:4] tag of Ht' (2 M, 1 E1, 0 E2),
// [3] E1 of a successor opens from this cell, [2] same for E2, [1] F1 of this cell did NOT open at j-1, [0] same for F2.
// Value rows in HBM (node-end rows, rows wider than the LDS window): int32 4H plane [W] then uint16 G plane [W].
// Limits (the host falls back to k_poa_dp_pk / k_poa_dp_lds otherwise): 4 (o1+e1) - 1 <= 255, 4 (o2+e2) <= 255, e1 >= 1,
// match + mismatch >= 0.
#pragma once

#define T4_NEG (4 * POA_NEG)

template <int B>
__device__ __forceinline__ int t4_sub_byte(int a, int g)  // a - byte B of g
{
    int r;
    if constexpr (B == 0) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(g));
    if constexpr (B == 1) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(g));
    if constexpr (B == 2) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(g));
    if constexpr (B == 3) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(g));
    return r;
}
// byte B of dst = min(t, c) (unsigned; c <= 255), and  acc = 2 acc + (t >= c)
template <int B>
__device__ __forceinline__ void t4_gap_byte(int &dst, int &acc, int t, int c)
{
    if constexpr (B == 0) asm("v_min_u32_sdwa %0, %2, %3 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\tv_cmp_le_u32 vcc, %3, %2\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc" : "+v"(dst), "+v"(acc) : "v"(t), "s"(c) : "vcc");
    if constexpr (B == 1) asm("v_min_u32_sdwa %0, %2, %3 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\tv_cmp_le_u32 vcc, %3, %2\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc" : "+v"(dst), "+v"(acc) : "v"(t), "s"(c) : "vcc");
    if constexpr (B == 2) asm("v_min_u32_sdwa %0, %2, %3 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\tv_cmp_le_u32 vcc, %3, %2\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc" : "+v"(dst), "+v"(acc) : "v"(t), "s"(c) : "vcc");
    if constexpr (B == 3) asm("v_min_u32_sdwa %0, %2, %3 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\tv_cmp_le_u32 vcc, %3, %2\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc" : "+v"(dst), "+v"(acc) : "v"(t), "s"(c) : "vcc");
}
__device__ __forceinline__ void t4_flag_ne(int &acc, int a, int b)  // acc = 2 acc + (a != b)
{
    asm("v_cmp_ne_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
// byte B of dst = low byte of (2 acc + (a != b))
template <int B>
__device__ __forceinline__ void t4_flag_ne_dep(int &dst, int acc, int a, int b)
{
    if constexpr (B == 0) asm("v_cmp_ne_u32 vcc, %2, %3\n\tv_addc_co_u32_sdwa %0, vcc, %1, %1, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(acc), "v"(a), "v"(b) : "vcc");
    if constexpr (B == 1) asm("v_cmp_ne_u32 vcc, %2, %3\n\tv_addc_co_u32_sdwa %0, vcc, %1, %1, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(acc), "v"(a), "v"(b) : "vcc");
    if constexpr (B == 2) asm("v_cmp_ne_u32 vcc, %2, %3\n\tv_addc_co_u32_sdwa %0, vcc, %1, %1, vcc dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(acc), "v"(a), "v"(b) : "vcc");
    if constexpr (B == 3) asm("v_cmp_ne_u32 vcc, %2, %3\n\tv_addc_co_u32_sdwa %0, vcc, %1, %1, vcc dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(acc), "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ int t4_max3(int a, int b, int c)
{
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// lanes 1..63: max(v of the lane below, pre); lane 0: pre   (wave_shr:1, bound_ctrl off: lane 0 keeps the old value)
__device__ __forceinline__ int t4_shr1_max(int v, int pre)
{
    int r = pre;
    asm("s_nop 1\n\tv_max_i32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v), "v"(pre));  // (s_nop: the compiler does not see that %1 is read through DPP)
    return r;
}
__device__ __forceinline__ int t4_shr1_mov(int v, int first)  // lanes 1..63: v of the lane below; lane 0: first
{
    int r = first;
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
    return r;
}

static inline size_t poa_t4_lds_bytes(uint32_t hg_cols, uint32_t lds_cols, int nt)
{
    const int nw = nt / 64;
    return std::max<size_t>(6ull * hg_cols + ((lds_cols / 2 + 15u) & ~15u), sizeof(tb_lds)) + (size_t)(3 * nw + 1 + 4 + 6 + 1) * 16 + 16;
}

template <int NT, bool DEF>
__global__ __launch_bounds__(NT, 4) void k_poa_dp_t4(
    const poa_prob *__restrict__ probs, const char *__restrict__ queries, const uint4 *__restrict__ node_tab,
    const uint32_t *__restrict__ seq32, const uint32_t *__restrict__ preds, poa_dev_params P, poa_row *rows, uint8_t *pool_arg,
    unsigned long long *pool_next_arg, uint64_t pool_size_arg, poa_out *__restrict__ outs, uint32_t lds_cols, uint32_t hg_cols,
    uint32_t win_mask, uint8_t *__restrict__ tb_ops, uint32_t *__restrict__ tb_orow, uint32_t n_arenas, uint64_t arena_size,
    unsigned long long *arena_ctr, uint32_t *arena_flag)
{
    constexpr int NW = NT / 64;
    constexpr int STEP = NT * 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int4 *sX = (int4 *)smem;                  // [2][NW] {scan1, scan2, last1, last2} per wave
    int4 *sRed = sX + 2 * NW;                 // [NW] {row max, -leftmost, rightmost, 0} per wave
    int32_t *edgeW = (int32_t *)(sRed + NW);  // [2] (+2 pad)
    int4 *sRow = (int4 *)(edgeW + 4);         // [4] the row's parameters, written by wave 0
    int4 *sLead = sRow + 4;                   // [6] wave 0's allocator state and counters
    int32_t *sSink = (int32_t *)(sLead + 6);  // [1] (+3 pad)
    constexpr int HDR = (3 * NW + 1 + 4 + 6 + 1) * 16;
    int32_t *Hs = (int32_t *)(smem + HDR);                                // [hg_cols] 4 H
    uint16_t *Gs = (uint16_t *)(smem + HDR + 4ull * hg_cols);             // [hg_cols] G1 | G2 << 8
    uint16_t *Qn = (uint16_t *)(smem + HDR + 6ull * hg_cols);             // [lds_cols / 4] four one-hot column codes per halfword
    const int edge_idx = (int)(edgeW - Hs);

    const uint64_t t_begin = __builtin_amdgcn_s_memrealtime();
    const poa_prob pb = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qlen = (int)pb.qlen;
    const char *query = queries + pb.q0;
    const uint4 *ntab = node_tab + pb.node0;
    const uint32_t *plist = preds + pb.pred0;
    poa_row *R = rows + pb.row0;

    // ---- pool: classic (chunks of the launch's segment) or arena mode, as in k_poa_dp_pk
    uint8_t *pool = pool_arg;
    unsigned long long *pool_next = pool_next_arg;
    uint64_t pool_size = pool_size_arg;
    uint32_t arena = 0;
    if (n_arenas) {
        int got = -1;
        if (!(pb.flags & 1u)) {
            if (tid == 0) {
                uint32_t a = (uint32_t)(((uint64_t)blockIdx.x * 2654435761ull) % n_arenas);
                for (uint32_t tries = 0; tries < (1u << 24); tries++) {
                    if (atomicCAS(&arena_flag[a], 0u, 1u) == 0u) { got = (int)a; break; }
                    a = a + 1 == n_arenas ? 0 : a + 1;
                    if ((tries & 15u) == 15u) __builtin_amdgcn_s_sleep(64);
                }
                if (got >= 0) (void)atomicExch(&arena_ctr[got], 0ull);
                sSink[1] = got;
            }
            __syncthreads();
            got = __builtin_amdgcn_readfirstlane(sSink[1]);
        }
        if (got < 0) {
            if (tid == 0) {
                poa_out &O = outs[blockIdx.x];
                O.t_begin = t_begin; O.t_end = t_begin; O.cells = 0; O.vcells = 0; O.maxw = 0; O.nops = 0;
                O.score = POA_NEG; O.row = 0; O.status = POA_ST_POOL;
            }
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        arena = (uint32_t)got;
        pool = pool_arg + (uint64_t)arena * arena_size;
        pool_next = arena_ctr + arena;
        pool_size = arena_size;
    }

    const int o1 = DEF ? 4 : P.o1, e1 = DEF ? 2 : P.e1, o2 = DEF ? 24 : P.o2, e2 = DEF ? 1 : P.e2;
    const int C1 = 4 * (o1 + e1) - 1, C2 = 4 * (o2 + e2);  // the G bytes of an opened gap
    const int bw = (int)pb.w;

    const bool leader = wv == 0;
    struct lead_t {
        uint64_t dcur, dend, vcur, vendp, wide_scratch, cells, vcells;
        int maxw, failed;
        uint64_t ring_base;
        uint32_t ring_head, ring_size;
        int sink_best, sink_have;
        uint32_t sink_row;
    };
    auto lead_load = [&]() -> lead_t {
        const int4 a = sLead[0], b = sLead[1], c = sLead[2], d = sLead[3], e = sLead[4], f = sLead[5];
        auto u64 = [](int lo, int hi) { return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; };
        return {u64(a.x, a.y), u64(a.z, a.w), u64(b.x, b.y), u64(b.z, b.w), u64(c.x, c.y), u64(c.z, c.w), u64(d.x, d.y), d.z, d.w,
                u64(e.x, e.y), (uint32_t)e.z, (uint32_t)e.w, f.x, f.y, (uint32_t)f.z};
    };
    auto lead_store = [&](const lead_t &L) {
        if (lane == 0) {
            sLead[0] = make_int4((int)(uint32_t)L.dcur, (int)(uint32_t)(L.dcur >> 32), (int)(uint32_t)L.dend, (int)(uint32_t)(L.dend >> 32));
            sLead[1] = make_int4((int)(uint32_t)L.vcur, (int)(uint32_t)(L.vcur >> 32), (int)(uint32_t)L.vendp, (int)(uint32_t)(L.vendp >> 32));
            sLead[2] = make_int4((int)(uint32_t)L.wide_scratch, (int)(uint32_t)(L.wide_scratch >> 32), (int)(uint32_t)L.cells, (int)(uint32_t)(L.cells >> 32));
            sLead[3] = make_int4((int)(uint32_t)L.vcells, (int)(uint32_t)(L.vcells >> 32), L.maxw, L.failed);
            sLead[4] = make_int4((int)(uint32_t)L.ring_base, (int)(uint32_t)(L.ring_base >> 32), (int)L.ring_head, (int)L.ring_size);
            sLead[5] = make_int4(L.sink_best, L.sink_have, (int)L.sink_row, 0);
        }
    };
    // bump allocation out of chunks; a request larger than a chunk takes whole chunks of its own
    auto alloc = [&](lead_t &L, uint64_t &cur, uint64_t &end, uint64_t bytes) -> uint64_t {
        bytes = (bytes + 15ull) & ~15ull;
        if (cur + bytes > end) {
            const uint64_t need = bytes > POA_CHUNK ? (bytes + POA_CHUNK - 1) & ~(POA_CHUNK - 1) : POA_CHUNK;
            unsigned long long bv = 0;
            if (lane == 0) bv = atomicAdd(pool_next, (unsigned long long)need);
            const uint64_t b = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bv);
            if (b + need > pool_size) L.failed = 1;
            cur = b;
            end = b + need;
        }
        const uint64_t r = cur;
        cur += bytes;
        return r;
    };

    // column codes, one-hot: nibble j = 1/2/4/8 for query[j-1] = A/C/G/T, 0 for anything else (and for column 0)
    int non_acgt = 0;
    for (int t = tid; t < (int)(lds_cols / 4); t += NT) {
        uint32_t hw = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = 4 * t + k;
            uint32_t code = 0;
            if (j >= 1 && j <= qlen) {
                const char ch = query[j - 1];
                code = ch == 'A' ? 1u : (ch == 'C' ? 2u : (ch == 'G' ? 4u : (ch == 'T' ? 8u : 0u)));
                non_acgt |= code == 0;
            }
            hw |= code << (4 * k);
        }
        Qn[t] = (uint16_t)hw;
    }
    const bool q_plain = __syncthreads_or(non_acgt) == 0;

    if (leader) {
        lead_t L = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, POA_NEG, 0, 0};
        if (win_mask != 0xFFFFFFFFu) L.wide_scratch = alloc(L, L.vcur, L.vendp, 2ull * 6ull * lds_cols);
        {
            const uint64_t maxrow = (6ull * (uint64_t)((qlen + 8) & ~3) + 15ull) & ~15ull;
            const uint64_t rb = (maxrow * (uint64_t)pb.ring_rows + POA_CHUNK - 1) & ~(POA_CHUNK - 1);
            unsigned long long bv = 0;
            if (lane == 0) bv = atomicAdd(pool_next, (unsigned long long)rb);
            const uint64_t b = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bv);
            if (b + rb > pool_size || rb >= (1ull << 32)) L.failed = 1;
            L.ring_base = b;
            L.ring_size = (uint32_t)maxrow;  // bytes per slot of the ring
            L.ring_head = 0;                 // the slot the next node-end row takes
        }
        lead_store(L);
    }
    int prev_beg = 0, prev_end = -1;
    bool prev_lds = true;
    uint32_t seq_word = 0, seq_word_idx = 0xFFFFFFFFu;
    bool stop = false;

    for (uint32_t v = 0; v < pb.n_nodes && !stop; v++) {
    const uint4 nt = ntab[v];
    const uint32_t nlen = nt.y & 0xFFFFFFu;
    for (uint32_t tn = 0; tn < nlen && !stop; tn++) {
        POA_MARK("row_topo");
        const uint32_t r = nt.x + tn;
        const bool first = tn == 0 && v > 0;
        const bool last = tn + 1 == nlen;
        const bool is_sink = last && (nt.z >> 31) != 0;
        const int np = v == 0 ? 0 : (tn == 0 ? (int)(nt.y >> 24) : 1);
        const uint32_t ps = nt.w;
        uint8_t gb = 0;
        if (v > 0) {
            const uint32_t bi = r - 1;
            if ((bi & 3u) == 0 || (bi >> 2) != seq_word_idx) { seq_word_idx = bi >> 2; seq_word = seq32[(pb.seq0 >> 2) + seq_word_idx]; }
            gb = (uint8_t)(seq_word >> (8u * (bi & 3u)));
        }
        bool far = r > 0 && !prev_lds;
        if (first) {
            if (np == 1) far |= ps != r - 1;
            else
                for (int t = 0; t < np; t++) far |= plist[ps + t] != r - 1;
        }
        if (__builtin_expect(far, 0)) __syncthreads();  // vmcnt(0) + barrier: value rows / row records of far predecessors have landed
        const bool single = r > 0 && np == 1;
        const uint32_t sp = first ? ps : r - 1;
        const bool sp_near = sp == r - 1 && prev_lds;
        POA_MARK("row_leader");
        // the wave that sets the row up rotates with the row (its state is in LDS): one wave per SIMD, so the set-up
        // work is spread over the CU's four SIMDs instead of making the first one the bottleneck
        if (wv == (int)(r % (uint32_t)NW)) {
            // (the other waves of the workgroup wait for this block: let it win the issue arbitration against the waves of
            // the other workgroups on its SIMD)
            __builtin_amdgcn_s_setprio(3);
            // the previous row's maximum: combine the waves' results (only this wave needs them: the band, and the row record)
            int prev_lmax = 0, prev_rmax = 0;
            if (r > 0) {
                int4 rw = make_int4(INT32_MIN, INT32_MIN, INT32_MIN, 0);
                if (lane < NW) rw = sRed[lane];
                int b = rw.x, t;
                t = poa_dpp<0x111, 0xf>(INT32_MIN, b); b = t > b ? t : b;
                if (NW > 2) { t = poa_dpp<0x112, 0xf>(INT32_MIN, b); b = t > b ? t : b; }
                if (NW > 4) { t = poa_dpp<0x114, 0xf>(INT32_MIN, b); b = t > b ? t : b; }
                if (NW > 8) { t = poa_dpp<0x118, 0xf>(INT32_MIN, b); b = t > b ? t : b; }
                const int rbest = __builtin_amdgcn_readlane(b, NW - 1);
                int lm = rw.x == rbest ? rw.y : INT32_MIN, rm = rw.x == rbest ? rw.z : INT32_MIN;
                t = poa_dpp<0x111, 0xf>(INT32_MIN, lm); lm = t > lm ? t : lm;
                t = poa_dpp<0x111, 0xf>(INT32_MIN, rm); rm = t > rm ? t : rm;
                if (NW > 2) {
                    t = poa_dpp<0x112, 0xf>(INT32_MIN, lm); lm = t > lm ? t : lm;
                    t = poa_dpp<0x112, 0xf>(INT32_MIN, rm); rm = t > rm ? t : rm;
                }
                if (NW > 4) {
                    t = poa_dpp<0x114, 0xf>(INT32_MIN, lm); lm = t > lm ? t : lm;
                    t = poa_dpp<0x114, 0xf>(INT32_MIN, rm); rm = t > rm ? t : rm;
                }
                if (NW > 8) {  // (up to 16 waves: 1 024 threads for the very long problems)
                    t = poa_dpp<0x118, 0xf>(INT32_MIN, lm); lm = t > lm ? t : lm;
                    t = poa_dpp<0x118, 0xf>(INT32_MIN, rm); rm = t > rm ? t : rm;
                }
                prev_lmax = -__builtin_amdgcn_readlane(lm, NW - 1);
                prev_rmax = __builtin_amdgcn_readlane(rm, NW - 1);
                if (lane == 0) { R[r - 1].lmax = prev_lmax; R[r - 1].rmax = prev_rmax; }
            }
            lead_t L = lead_load();
            const int remain = (int)(nt.z & 0x3fffffffu) + (int)(nlen - 1 - tn);
            if (lane == 0) sSink[0] = 0;
            int mpl, mpr;
            if (r == 0) { mpl = 0; mpr = 0; }
            else if (!first) { mpl = prev_lmax + 1; mpr = prev_rmax + 1; }
            else {
                mpl = INT32_MAX; mpr = 0;
                for (int t = 0; t < np; t++) {
                    const uint32_t p = np == 1 ? ps : plist[ps + t];
                    int lm, rm;
                    if (p == r - 1) { lm = prev_lmax + 1; rm = prev_rmax + 1; }
                    else {
                        lm = __builtin_amdgcn_readfirstlane(R[p].lmax) + 1;
                        rm = __builtin_amdgcn_readfirstlane(R[p].rmax) + 1;
                    }
                    mpl = lm < mpl ? lm : mpl;
                    mpr = rm > mpr ? rm : mpr;
                }
            }
            int beg, end;
            if (!P.banded) { beg = 0; end = qlen; }
            else {
                const int diag = qlen - remain;
                const int lo = mpl < diag ? mpl : diag;
                const int hi = mpr > diag ? mpr : diag;
                beg = lo - bw; if (beg < 0) beg = 0;
                end = hi + bw; if (end > qlen) end = qlen;
            }
            const int W = (end - (beg & ~3) + 1 + 3) & ~3;
            L.maxw = W > L.maxw ? W : L.maxw;
            const bool wide = (uint32_t)W + 8u > hg_cols;
            if (r > 0) L.cells += (uint64_t)(end - beg + 1);
            if (last || wide) L.vcells += (uint64_t)(end - beg + 1);
            const uint64_t doff = alloc(L, L.dcur, L.dend, (uint64_t)W * (np > 1 ? 4u : 1u));
            uint64_t voff = 0;
            if (last && !L.failed) {
                if (r == 0 || (nt.z & 0x40000000u)) voff = alloc(L, L.vcur, L.vendp, 6ull * (uint64_t)W);
                else {
                    // fixed slots of one worst-case row: the rows of the last ring_rows node ends survive whatever their
                    // widths (a byte ring that wraps when a row does not fit can overwrite the row written two slots ago)
                    voff = L.ring_base + (uint64_t)L.ring_head * L.ring_size;
                    L.ring_head = L.ring_head + 1 == pb.ring_rows ? 0 : L.ring_head + 1;
                }
            } else if (wide) voff = L.wide_scratch + (r & 1u) * 6ull * lds_cols;
            int pbeg = prev_beg, pend = prev_end;
            uint64_t vpo = 0;
            if (single && !sp_near) {
                pbeg = __builtin_amdgcn_readfirstlane(R[sp].beg);
                pend = __builtin_amdgcn_readfirstlane(R[sp].end);
                const uint64_t vo = R[sp].voff;
                vpo = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(vo >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vo);
            }
            lead_store(L);
            if (lane == 0) {
                if (!L.failed) {
                    R[r].beg = beg; R[r].end = end;
                    R[r].doff = doff; R[r].voff = voff;
                    R[r].pred = ps; R[r].npred = first ? (uint32_t)np : 0u;
                    R[r].base = 0;
                }
                sRow[0] = make_int4(beg, end, (int)(uint32_t)doff, (int)(uint32_t)(doff >> 32));
                sRow[1] = make_int4((int)(uint32_t)voff, (int)(uint32_t)(voff >> 32), pbeg, pend);
                sRow[2] = make_int4((int)(uint32_t)vpo, (int)(uint32_t)(vpo >> 32), L.failed, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        POA_LDS_BARRIER();
        POA_MARK("row_pickup");
        const int4 rw0 = sRow[0], rw1 = sRow[1], rw2 = sRow[2];
        const int beg = __builtin_amdgcn_readfirstlane(rw0.x), end = __builtin_amdgcn_readfirstlane(rw0.y);
        const uint64_t doff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rw0.w) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(rw0.z);
        const uint64_t voff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rw1.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(rw1.x);
        const int pbeg = __builtin_amdgcn_readfirstlane(rw1.z), pend = __builtin_amdgcn_readfirstlane(rw1.w);
        const uint64_t vpo = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rw2.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(rw2.x);
        if (__builtin_amdgcn_readfirstlane(rw2.z)) { stop = true; break; }
        const int bal = beg & ~3;
        const int W = (end - bal + 1 + 3) & ~3;
        const bool wide = (uint32_t)W + 8u > hg_cols;
        const bool keep = last || wide;
        uint8_t *Vrow = pool + voff;
        uint8_t *drow = pool + doff;
        const int gcode = gb == 'A' ? 0 : (gb == 'C' ? 1 : (gb == 'G' ? 2 : (gb == 'T' ? 3 : 4)));
        const int sc_eq = gcode == 4 ? 0 : P.match, sc_ne = gcode == 4 ? 0 : -P.mismatch;
        const int gsh = gcode & 3;
        const int ne4t = 4 * sc_ne + 2, mm4 = 4 * (sc_eq - sc_ne);  // M candidates carry tag 2
        const uint8_t *Vp = pool + vpo;
        const int balp = pbeg & ~3;

        POA_MARK("row_steps");
        // uniform per wave, held in vector registers: running maxima of the max-plus scan over the previous steps of this
        // row (carry) and the scan value of the last column of the previous step (left)
        int carry1 = POA_IDENT, carry2 = POA_IDENT, left1 = POA_IDENT, left2 = POA_IDENT;
        int best = INT32_MIN, lpos = beg, rpos = beg;
        int buf = 0;
        for (int c0 = 0; c0 < W; c0 += STEP, buf ^= 1) {
            const int c = c0 + 4 * tid;
            const int j0 = bal + c;
            const bool lane_act = j0 <= end;
            const int nw_step = (W - c0 + 255) / 256 < NW ? (W - c0 + 255) / 256 : NW;
            const bool wave_act = wv < nw_step;
            // carried from phase 1 to phase 2, per cell: Ht' (tagged), 4 Ht, E1' (tag 1), E2' (tag 0)
            int htt[4], ht4[4], e1t[4], e2t[4], pmeta[4];
            int agg1 = POA_IDENT, agg2 = POA_IDENT, alast1 = POA_IDENT, alast2 = POA_IDENT;
#pragma unroll
            for (int k = 0; k < 4; k++) { htt[k] = T4_NEG + 2; ht4[k] = T4_NEG; e1t[k] = T4_NEG + 1; e2t[k] = T4_NEG; pmeta[k] = 0; }
            uint32_t qn = 0u;
            if (wave_act) qn = (uint32_t)Qn[j0 >> 2];
            // the mask-free path and its edge patches (lp / rp / lq): see k_poa_dp_pk
            const int jw0 = bal + c0 + 256 * wv, jw1 = jw0 + 255;
            const bool lp = jw0 < beg;
            const bool rp = jw1 > end || jw1 > pend;
            const bool lq = (lp ? beg : jw0) <= pbeg;
            const bool fastw = single && q_plain && wave_act && (!lq || sp_near) && (!rp || (sp_near && end <= pend + 1));
            const int base1 = 4 * e1 * j0, base2 = 4 * e2 * j0;  // the scan runs on lane-relative values in the fast path
        POA_MARK("p1_fast");
            if (__builtin_expect(fastw, 1)) {
                int4 hv;
                uint2 gg;
                int hprev;
                if (__builtin_expect(sp_near, 1)) {
                    hv = *(const int4 *)(Hs + (j0 & win_mask));
                    gg = *(const uint2 *)(Gs + (j0 & win_mask));
                    if (tid == NT - 1) edgeW[buf] = hv.w;
                    {
                        // one LDS read through an index (a pointer select would become a flat load)
                        const bool edge = tid == 0 && c0 > 0;
                        int w = Hs[edge ? edge_idx + (buf ^ 1) : ((j0 > 0 ? j0 - 1 : 0) & win_mask)];
                        asm volatile("" : "+v"(w));
                        hprev = w;
                    }
                } else {
                    const int Wp = (pend - balp + 1 + 3) & ~3;
                    const int idx = j0 - balp;
                    hv = *(const int4 *)((const int32_t *)Vp + idx);
                    gg = *(const uint2 *)(Vp + 4ll * Wp + 2ll * idx);
                    hprev = ((const int32_t *)Vp)[idx > 0 ? idx - 1 : 0];
                    // consume the loads inside this branch (their s_waitcnt vmcnt must not land in shared code)
                    asm volatile("" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w), "+v"(gg.x), "+v"(gg.y), "+v"(hprev));
                }
                auto phase1 = [&](auto edge_c) {
                    constexpr bool EDGE = decltype(edge_c)::value;
                    int hj[4] = {hv.x, hv.y, hv.z, hv.w};
                    int ga = (int)gg.x, gbb = (int)gg.y;
                    if constexpr (EDGE) {
                        if (rp) {
#pragma unroll
                            for (int k = 0; k < 4; k++) hj[k] = j0 + k > pend ? T4_NEG : hj[k];
                            ga = j0 > pend ? 0 : (j0 + 1 > pend ? (ga & 0xffff) : ga);
                            gbb = j0 + 2 > pend ? 0 : (j0 + 3 > pend ? (gbb & 0xffff) : gbb);
                        }
                        if (lq) {
#pragma unroll
                            for (int k = 0; k < 4; k++) hj[k] = j0 + k < pbeg ? T4_NEG : hj[k];
                            ga = j0 + 1 < pbeg ? 0 : (j0 < pbeg ? (int)((uint32_t)ga & 0xffff0000u) : ga);
                            gbb = j0 + 3 < pbeg ? 0 : (j0 + 2 < pbeg ? (int)((uint32_t)gbb & 0xffff0000u) : gbb);
                        }
                    }
                    const uint32_t eqb = qn >> gsh;
                    int hp = hprev;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int g = k < 2 ? ga : gbb;
                        const int ev1 = (k & 1) ? t4_sub_byte<2>(hj[k], g) : t4_sub_byte<0>(hj[k], g);
                        const int ev2 = (k & 1) ? t4_sub_byte<3>(hj[k], g) : t4_sub_byte<1>(hj[k], g);
                        int m = (int)__umul24(__builtin_amdgcn_ubfe(eqb, 4u * k, 1u), (uint32_t)mm4) + (hp + ne4t);
                        if constexpr (EDGE) {
                            if (lq) m = j0 + k - 1 < pbeg ? T4_NEG + 2 : m;
                        }
                        const int h = t4_max3(m, ev1, ev2);
                        htt[k] = h;
                        ht4[k] = h & ~3;
                        e1t[k] = ev1;
                        e2t[k] = ev2;
                        hp = hj[k];
                    }
                    if constexpr (EDGE) {
                        if (lp) {  // cells left of beg stay out of the scan
#pragma unroll
                            for (int k = 0; k < 4; k++) ht4[k] = j0 + k < beg ? POA_IDENT : ht4[k];
                        }
                    }
                    const int r11 = ht4[1] + 4 * e1, r12 = ht4[2] + 8 * e1, r13 = ht4[3] + 12 * e1;
                    const int r21 = ht4[1] + 4 * e2, r22 = ht4[2] + 8 * e2, r23 = ht4[3] + 12 * e2;
                    const int a1 = t4_max3(ht4[0], r11, r12), a2 = t4_max3(ht4[0], r21, r22);
                    agg1 = (a1 > r13 ? a1 : r13) + base1;
                    agg2 = (a2 > r23 ? a2 : r23) + base2;
                    alast1 = r13 + base1;
                    alast2 = r23 + base2;
                };
                if (__builtin_expect(lp || rp || lq, 0)) phase1(std::true_type{});
                else phase1(std::false_type{});
        POA_MARK("p1_lean");
            } else if (wave_act && single) {
                // ---------------- lean path, phase 1 (cold: band limits re-derived here, not hoisted into scalar registers)
                int pbeg_ = pbeg, pend_ = pend, beg_ = beg, end_ = end, balp_ = balp, gsh_ = gsh;
                asm volatile("" : "+s"(pbeg_), "+s"(pend_), "+s"(beg_), "+s"(end_), "+s"(balp_), "+s"(gsh_));
                const int pbeg = pbeg_, pend = pend_, beg = beg_, end = end_, balp = balp_, gsh = gsh_;
                const unsigned span = (unsigned)(end - beg), pspan = (unsigned)(pend - pbeg);
                int hj[4], wm0;
                uint32_t g16[4];
                if (sp_near) {
                    const int4 hv = *(const int4 *)(Hs + (j0 & win_mask));
                    const uint2 gg = *(const uint2 *)(Gs + (j0 & win_mask));
                    hj[0] = hv.x; hj[1] = hv.y; hj[2] = hv.z; hj[3] = hv.w;
                    g16[0] = gg.x & 0xffffu; g16[1] = gg.x >> 16; g16[2] = gg.y & 0xffffu; g16[3] = gg.y >> 16;
                    if (tid == NT - 1) edgeW[buf] = hj[3];
                    const bool edge = tid == 0 && c0 > 0;
                    int w = Hs[edge ? edge_idx + (buf ^ 1) : ((j0 > 0 ? j0 - 1 : 0) & win_mask)];
                    asm volatile("" : "+v"(w));
                    wm0 = w;
                } else {
                    const int idx = j0 - balp;
                    const int Wp = (pend - balp + 1 + 3) & ~3;
                    int4 hv = make_int4(0, 0, 0, 0);
                    uint2 gg = make_uint2(0u, 0u);
                    if (idx >= 0 && idx < Wp) {
                        hv = *(const int4 *)((const int32_t *)Vp + idx);
                        gg = *(const uint2 *)(Vp + 4ll * Wp + 2ll * idx);
                    }
                    wm0 = (idx >= 1 && idx - 1 < Wp) ? ((const int32_t *)Vp)[idx - 1] : 0;
                    asm volatile("" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w), "+v"(gg.x), "+v"(gg.y), "+v"(wm0));
                    hj[0] = hv.x; hj[1] = hv.y; hj[2] = hv.z; hj[3] = hv.w;
                    g16[0] = gg.x & 0xffffu; g16[1] = gg.x >> 16; g16[2] = gg.y & 0xffffu; g16[3] = gg.y >> 16;
                }
                bool inprev = j0 >= 1 && (unsigned)(j0 - 1 - pbeg) <= pspan;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = j0 + k;
                    const bool inj = (unsigned)(j - pbeg) <= pspan;
                    const bool actk = (unsigned)(j - beg) <= span;
                    const int qc = (int)((qn >> (4 * k)) & 15u);
                    const int s = ((qc >> gsh) & 1) ? sc_eq : (qc == 0 ? 0 : sc_ne);
                    const int wm = k == 0 ? wm0 : hj[k - 1];
                    const int m = inprev ? wm + 4 * s + 2 : T4_NEG + 2;
                    const int ev1 = inj ? hj[k] - (int)(g16[k] & 255u) : T4_NEG + 1;
                    const int ev2 = inj ? hj[k] - (int)(g16[k] >> 8) : T4_NEG;
                    const int h = t4_max3(m, ev1, ev2);
                    htt[k] = h;
                    ht4[k] = h & ~3;
                    e1t[k] = ev1;
                    e2t[k] = ev2;
                    const int a1 = actk ? ht4[k] + 4 * e1 * j : POA_IDENT, a2 = actk ? ht4[k] + 4 * e2 * j : POA_IDENT;
                    agg1 = a1 > agg1 ? a1 : agg1;
                    agg2 = a2 > agg2 ? a2 : agg2;
                    if (k == 3) { alast1 = a1; alast2 = a2; }
                    inprev = inj;
                }
        POA_MARK("p1_general");
            } else if (wave_act) {
                // ---------------- general path, phase 1: the source row and rows with several predecessors
                int beg_ = beg, end_ = end, gsh_ = gsh;
                asm volatile("" : "+s"(beg_), "+s"(end_), "+s"(gsh_));
                const int beg = beg_, end = end_, gsh = gsh_;
                const unsigned span = (unsigned)(end - beg);
                if (r == 0) {
#pragma unroll
                    for (int k = 0; k < 4; k++) { ht4[k] = (j0 + k == 0) ? 0 : T4_NEG; htt[k] = ht4[k] + 2; }
                } else if (lane_act) {
                    int m[4], ev1[4], ev2[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) { m[k] = T4_NEG + 2; ev1[k] = T4_NEG + 1; ev2[k] = T4_NEG; }
                    for (int t = 0; t < np; t++) {
                        const uint32_t p = plist[ps + t];
                        int hj[4], wm0 = 0;
                        uint32_t g16[4];
                        int bp, ep;
                        if (p == r - 1 && prev_lds) {
                            bp = prev_beg; ep = prev_end;
                            const int4 hv = *(const int4 *)(Hs + (j0 & win_mask));
                            const uint2 gg = *(const uint2 *)(Gs + (j0 & win_mask));
                            hj[0] = hv.x; hj[1] = hv.y; hj[2] = hv.z; hj[3] = hv.w;
                            g16[0] = gg.x & 0xffffu; g16[1] = gg.x >> 16; g16[2] = gg.y & 0xffffu; g16[3] = gg.y >> 16;
                            if (tid == NT - 1) edgeW[buf] = hj[3];
                            const bool edge = tid == 0 && c0 > 0;
                            int w = Hs[edge ? edge_idx + (buf ^ 1) : ((j0 > 0 ? j0 - 1 : 0) & win_mask)];
                            asm volatile("" : "+v"(w));
                            wm0 = w;
                        } else {
                            bp = __builtin_amdgcn_readfirstlane(R[p].beg);
                            ep = __builtin_amdgcn_readfirstlane(R[p].end);
                            const uint64_t vo = R[p].voff;
                            const uint64_t vos = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(vo >> 32)) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vo);
                            const uint8_t *Vq = pool + vos;
                            const int balq = bp & ~3;
                            const int Wq = (ep - balq + 1 + 3) & ~3;
                            const int idx = j0 - balq;
                            int4 hv = make_int4(0, 0, 0, 0);
                            uint2 gg = make_uint2(0u, 0u);
                            if (idx >= 0 && idx < Wq) {
                                hv = *(const int4 *)((const int32_t *)Vq + idx);
                                gg = *(const uint2 *)(Vq + 4ll * Wq + 2ll * idx);
                            }
                            wm0 = (idx >= 1 && idx - 1 < Wq) ? ((const int32_t *)Vq)[idx - 1] : 0;
                            asm volatile("" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w), "+v"(gg.x), "+v"(gg.y), "+v"(wm0));
                            hj[0] = hv.x; hj[1] = hv.y; hj[2] = hv.z; hj[3] = hv.w;
                            g16[0] = gg.x & 0xffffu; g16[1] = gg.x >> 16; g16[2] = gg.y & 0xffffu; g16[3] = gg.y >> 16;
                        }
                        const unsigned pspan = (unsigned)(ep - bp);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int j = j0 + k;
                            const bool actk = (unsigned)(j - beg) <= span;
                            const int qc = (int)((qn >> (4 * k)) & 15u);
                            const int s = ((qc >> gsh) & 1) ? sc_eq : (qc == 0 ? 0 : sc_ne);
                            const int wm = k == 0 ? wm0 : hj[k - 1];
                            if (actk && j >= 1 && (unsigned)(j - 1 - bp) <= pspan) {
                                const int cnd = wm + 4 * s + 2;
                                if (cnd > m[k]) { m[k] = cnd; pmeta[k] = (pmeta[k] & ~255) | t; }
                            }
                            if (actk && (unsigned)(j - bp) <= pspan) {
                                const int c1 = hj[k] - (int)(g16[k] & 255u);
                                if (c1 > ev1[k]) { ev1[k] = c1; pmeta[k] = (pmeta[k] & ~0xff00) | (t << 8); }
                                const int c2 = hj[k] - (int)(g16[k] >> 8);
                                if (c2 > ev2[k]) { ev2[k] = c2; pmeta[k] = (pmeta[k] & ~0xff0000) | (t << 16); }
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int h = t4_max3(m[k], ev1[k], ev2[k]);
                        htt[k] = h;
                        ht4[k] = h & ~3;
                        e1t[k] = ev1[k];
                        e2t[k] = ev2[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = j0 + k;
                    const bool actk = (unsigned)(j - beg) <= span;
                    const int a1 = actk ? ht4[k] + 4 * e1 * j : POA_IDENT, a2 = actk ? ht4[k] + 4 * e2 * j : POA_IDENT;
                    agg1 = a1 > agg1 ? a1 : agg1;
                    agg2 = a2 > agg2 ? a2 : agg2;
                    if (k == 3) { alast1 = a1; alast2 = a2; }
                }
            }
        POA_MARK("scan");
            int i1 = POA_IDENT, i2 = POA_IDENT;
            if (wave_act) {
                i1 = poa_wave_scan_max(agg1);
                i2 = poa_wave_scan_max(agg2);
            }
            if (lane == 63) sX[buf * NW + wv] = make_int4(i1, i2, alast1, alast2);
            POA_LDS_BARRIER();
        POA_MARK("exchange");
            // cross-wave part of the scan, out of the waves' totals (broadcast LDS reads, uniform code): this wave's prefix and
            // the value left of its first column; the row's running maximum only if another step follows (that step is
            // then full, i.e. every wave is active here)
            int pre1 = carry1, pre2 = carry2, pl1 = left1, pl2 = left2;
            {
                const bool more = c0 + STEP < W;
                int all1 = carry1, all2 = carry2;
#pragma unroll
                for (int q = 0; q < NW; q++) {
                    if ((q < wv && wave_act) || more) {
                        const int4 x = sX[buf * NW + q];
                        if (q < wv) {
                            pre1 = x.x > pre1 ? x.x : pre1;
                            pre2 = x.y > pre2 ? x.y : pre2;
                            pl1 = x.z;
                            pl2 = x.w;
                        }
                        all1 = x.x > all1 ? x.x : all1;
                        all2 = x.y > all2 ? x.y : all2;
                        if (q == NW - 1) { left1 = x.z; left2 = x.w; }
                    }
                }
                carry1 = all1; carry2 = all2;  // (only used when another step follows)
            }
            if (wave_act) {
                const int run1_ = t4_shr1_max(i1, pre1), run2_ = t4_shr1_max(i2, pre2);
                const int la1_ = t4_shr1_mov(alast1, pl1), la2_ = t4_shr1_mov(alast2, pl2);
        POA_MARK("p2_fast");
                if (__builtin_expect(fastw, 1)) {
                    auto phase2 = [&](auto edge_c) {
                        constexpr bool EDGE = decltype(edge_c)::value;
                        int R1 = run1_ - base1, R2 = run2_ - base2, L1 = la1_ - base1, L2 = la2_ - base2;
                        int h4[4];
                        int dirq = 0, ga = 0, gbb = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int f1 = R1 - (4 * (o1 + e1 * k) - 1), f2 = R2 - 4 * (o2 + e2 * k);
                            const int hh = t4_max3(ht4[k] | 3, f1, f2);
                            const int h = hh & ~3;
                            h4[k] = h;
                            // [3:2] tag of H'', [1:0] tag of Ht'; what lies above bit 3 is shifted out of the byte below
                            int acc = ((hh << 2) & ~3) | (htt[k] & 3);
                            const int t1 = (h + 4 * e1) - e1t[k], t2 = (h + 4 * e2) - e2t[k];
                            if (k == 0) { t4_gap_byte<0>(ga, acc, t1, C1); t4_gap_byte<1>(ga, acc, t2, C2); }
                            if (k == 1) { t4_gap_byte<2>(ga, acc, t1, C1); t4_gap_byte<3>(ga, acc, t2, C2); }
                            if (k == 2) { t4_gap_byte<0>(gbb, acc, t1, C1); t4_gap_byte<1>(gbb, acc, t2, C2); }
                            if (k == 3) { t4_gap_byte<2>(gbb, acc, t1, C1); t4_gap_byte<3>(gbb, acc, t2, C2); }
                            t4_flag_ne(acc, R1, L1);
                            if (k == 0) t4_flag_ne_dep<0>(dirq, acc, R2, L2);
                            if (k == 1) t4_flag_ne_dep<1>(dirq, acc, R2, L2);
                            if (k == 2) t4_flag_ne_dep<2>(dirq, acc, R2, L2);
                            if (k == 3) t4_flag_ne_dep<3>(dirq, acc, R2, L2);
                            L1 = ht4[k] + 4 * e1 * k; L2 = ht4[k] + 4 * e2 * k;
                            R1 = L1 > R1 ? L1 : R1;
                            R2 = L2 > R2 ? L2 : R2;
                        }
                        // row maximum, leftmost / rightmost column
                        {
                            int hb[4] = {h4[0], h4[1], h4[2], h4[3]};
                            if constexpr (EDGE) {
#pragma unroll
                                for (int k = 0; k < 4; k++) hb[k] = j0 + k > end ? INT32_MIN : hb[k];
                            }
                            const int m3 = t4_max3(hb[0], hb[1], hb[2]);
                            const int m4 = m3 > hb[3] ? m3 : hb[3];
                            if (m4 >= best) {
                                const int kf = hb[0] == m4 ? 0 : (hb[1] == m4 ? 1 : (hb[2] == m4 ? 2 : 3));
                                const int kl = hb[3] == m4 ? 3 : (hb[2] == m4 ? 2 : (hb[1] == m4 ? 1 : 0));
                                if (m4 > best) { best = m4; lpos = j0 + kf; }
                                rpos = j0 + kl;
                            }
                        }
                        if (__builtin_expect(is_sink, 0)) {
                            const int kq = qlen - j0;
#pragma unroll
                            for (int k = 0; k < 4; k++)
                                if (kq == k) sSink[0] = h4[k];
                        }
                        if (!EDGE || j0 <= end) {
                            const int4 hq = make_int4(h4[0], h4[1], h4[2], h4[3]);
                            const uint2 gq = make_uint2((uint32_t)ga, (uint32_t)gbb);
                            if (!wide) {
                                *(int4 *)(Hs + (j0 & win_mask)) = hq;
                                *(uint2 *)(Gs + (j0 & win_mask)) = gq;
                            }
                            *(uint32_t *)(drow + c) = (uint32_t)dirq;
                            if (keep) {
                                *(int4 *)((int32_t *)Vrow + c) = hq;
                                *(uint2 *)(Vrow + 4ll * W + 2ll * c) = gq;
                            }
                        }
                    };
                    if (__builtin_expect(lp || rp, 0)) phase2(std::true_type{});
                    else phase2(std::false_type{});
        POA_MARK("p2_slow");
                } else if (lane_act) {
                    int beg_ = beg, end_ = end;
                    asm volatile("" : "+s"(beg_), "+s"(end_));
                    const int beg = beg_;
                    const unsigned span = (unsigned)(end_ - beg_);
                    int run1 = run1_, run2 = run2_, la1 = la1_, la2 = la2_;
                    int h4[4];
                    uint32_t g16[4], codev[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int j = j0 + k;
                        const bool actk = (unsigned)(j - beg) <= span;
                        // at the first column run1 / run2 are still POA_IDENT, which keeps F below everything
                        const int f1 = run1 - 4 * (o1 + e1 * j) + 1, f2 = run2 - 4 * (o2 + e2 * j);
                        const int hh = t4_max3(ht4[k] | 3, f1, f2);
                        const int h = hh & ~3;
                        h4[k] = h;
                        const uint32_t t1 = (uint32_t)((h + 4 * e1) - e1t[k]), t2 = (uint32_t)((h + 4 * e2) - e2t[k]);
                        const uint32_t G1 = t1 < (uint32_t)C1 ? t1 : (uint32_t)C1, G2 = t2 < (uint32_t)C2 ? t2 : (uint32_t)C2;
                        g16[k] = G1 | (G2 << 8);
                        codev[k] = ((uint32_t)(hh & 3) << 6) | ((uint32_t)(htt[k] & 3) << 4) | (t1 >= (uint32_t)C1 ? 8u : 0u) | (t2 >= (uint32_t)C2 ? 4u : 0u) |
                                   (run1 != la1 ? 2u : 0u) | (run2 != la2 ? 1u : 0u);
                        const int hb = actk ? h : INT32_MIN;
                        if (hb > best) { best = hb; lpos = j; rpos = j; }
                        else if (actk && hb == best) rpos = j;
                        const int a1 = actk ? ht4[k] + 4 * e1 * j : POA_IDENT, a2 = actk ? ht4[k] + 4 * e2 * j : POA_IDENT;
                        run1 = a1 > run1 ? a1 : run1;
                        run2 = a2 > run2 ? a2 : run2;
                        la1 = actk ? a1 : la1; la2 = actk ? a2 : la2;
                    }
                    if (__builtin_expect(is_sink, 0)) {
                        const int kq = qlen - j0;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (kq == k) sSink[0] = h4[k];
                    }
                    const int Wl = (end_ - bal + 1 + 3) & ~3;  // (recomputed here: keeps the plane addresses out of scalar registers)
                    {
                        const int4 hq = make_int4(h4[0], h4[1], h4[2], h4[3]);
                        const uint2 gq = make_uint2(g16[0] | (g16[1] << 16), g16[2] | (g16[3] << 16));
                        if (!wide) {
                            *(int4 *)(Hs + (j0 & win_mask)) = hq;
                            *(uint2 *)(Gs + (j0 & win_mask)) = gq;
                        }
                        *(uint32_t *)(drow + c) = codev[0] | (codev[1] << 8) | (codev[2] << 16) | (codev[3] << 24);
                        if (keep) {
                            *(int4 *)((int32_t *)Vrow + c) = hq;
                            *(uint2 *)(Vrow + 4ll * Wl + 2ll * c) = gq;
                        }
                        if (__builtin_expect(np > 1, 0)) {
                            *(uint32_t *)(drow + (uint64_t)Wl + c) = (uint32_t)(pmeta[0] & 255) | ((uint32_t)(pmeta[1] & 255) << 8) | ((uint32_t)(pmeta[2] & 255) << 16) | ((uint32_t)(pmeta[3] & 255) << 24);
                            *(uint32_t *)(drow + 2ull * Wl + c) = (uint32_t)((pmeta[0] >> 8) & 255) | ((uint32_t)((pmeta[1] >> 8) & 255) << 8) | ((uint32_t)((pmeta[2] >> 8) & 255) << 16) | ((uint32_t)((pmeta[3] >> 8) & 255) << 24);
                            *(uint32_t *)(drow + 3ull * Wl + c) = (uint32_t)((pmeta[0] >> 16) & 255) | ((uint32_t)((pmeta[1] >> 16) & 255) << 8) | ((uint32_t)((pmeta[2] >> 16) & 255) << 16) | ((uint32_t)((pmeta[3] >> 16) & 255) << 24);
                        }
                    }
                }
            }
        POA_MARK("step_end");
        }
        POA_MARK("row_reduce");
        {
            int wb = poa_wave_scan_max(best);
            wb = __builtin_amdgcn_readlane(wb, 63);
            int lm = best == wb ? -lpos : INT32_MIN;
            int rm = best == wb ? rpos : INT32_MIN;
            lm = poa_wave_scan_max(lm);
            rm = poa_wave_scan_max(rm);
            if (lane == 63) sRed[wv] = make_int4(wb, lm, rm, 0);
        }
        POA_LDS_BARRIER();
        POA_MARK("row_end");
        if (__builtin_expect(is_sink, 0) && wv == (int)((r + 1) % (uint32_t)NW)) {  // (the wave that sets the next row up: program order)
            lead_t L = lead_load();
            int val = POA_NEG;
            if (qlen >= beg && qlen <= end) val = sSink[0] >> 2;
            if (!L.sink_have || val > L.sink_best) { L.sink_best = val; L.sink_row = r; L.sink_have = 1; }
            lead_store(L);
        }
        prev_beg = beg; prev_end = end;
        prev_lds = !wide;
    }
    }
    __syncthreads();
    if (tid >= 64) return;
    int status = POA_ST_OK;
    uint32_t start_row = 0;
    {
        const lead_t L = lead_load();
        const bool failed = L.failed != 0;
        poa_out &O = outs[blockIdx.x];
        if (failed) status = POA_ST_POOL;
        else {
            start_row = L.sink_row;
            status = (L.sink_have != 0 && L.sink_best > POA_NEG / 2) ? POA_ST_OK : POA_ST_NOALN;
        }
        if (tid == 0) {
            O.t_begin = t_begin;
            O.cells = L.cells;
            O.vcells = L.vcells;
            O.maxw = (uint32_t)L.maxw;
            O.score = failed ? POA_NEG : L.sink_best;
            O.row = start_row;
            O.status = status;
        }
    }
    if (tb_ops) {
        status = __builtin_amdgcn_readfirstlane(status);
        start_row = (uint32_t)__builtin_amdgcn_readfirstlane((int)start_row);
        poa_traceback_wave<1>(*(tb_lds *)(smem + HDR), tid, pb, rows, preds, pool, outs[blockIdx.x], tb_ops, tb_orow, 0, status, start_row);
    }
    if (tid == 0) {
        outs[blockIdx.x].t_end = __builtin_amdgcn_s_memrealtime();
        if (n_arenas) {
            const unsigned long long used = atomicAdd(pool_next, 0ull);
            (void)atomicAdd(pool_next_arg, used < arena_size ? used : (unsigned long long)arena_size);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            (void)atomicExch(&arena_flag[arena], 0u);
        }
    }
}
