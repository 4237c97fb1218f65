// vga_poa_t7.hpp -- K4 "t7": k_poa_dp_t6's row (eight columns per lane, a row loop whose scalar state fits the scalar registers,
// row maximum resolved once per row) for bands that do not fit a wave: NT / 64 waves per problem, the row above in LDS by absolute
// column (k_poa_dp_t5's window, 6 bytes per column), one workgroup barrier per step for the cross-wave part of the max-plus scan and
// one at the end of a row.  Same recurrences, direction dwords, value rows, row records, chunk pool and fused traceback as
// k_poa_dp_t5 / k_poa_dp_t6 (bit-exact against oracle/og_poa.c), so the kernels are interchangeable problem by problem.
//   * A lane owns 8 consecutive columns of a step (two quads), a wave 512, a step NT * 8.  Per cell that halves what a step pays for
//     the wave scan, the cross-wave exchange, the addresses of its loads and stores and the loop around it.
//   * The row above is read before a step's barrier and written back after it (in place, as t5); the first lane of a later step
//     takes the column left of it from a word the last lane parked.
//   * Band edges: a wave whose 512 columns lie inside the predecessor's band reads it as it is; the (at most two) others mask the
//     cells outside it.  Cells right of `end` are written back far below every real score, so the row maximum needs no band test.
//   * Row maximum: a lane keeps its best word and the first / last step it saw it in; the columns are found once per row from the
//     lane's own cells in LDS, the wave's triple goes to LDS, every wave combines the NW triples at the top of the next row (t5).
// Handed back with POA_ST_RETRY (and re-run by k_poa_dp_t5 at once): a row wider than the LDS window, classic pool mode.
#pragma once

template <int NT, bool DEF>
__global__ __launch_bounds__(NT) void k_poa_dp_t7(const poa_prob *__restrict__ probs, const char *__restrict__ queries,
                                                  const uint4 *__restrict__ node_tab, const uint32_t *__restrict__ seq32,
                                                  const uint32_t *__restrict__ preds, const poa_t5_args A)
{
    constexpr int NW = NT / 64;
    constexpr int CPL = 8, Q = 2;
    constexpr int STEP = NT * CPL;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int4 *sX = (int4 *)smem;                  // [2][NW] {scan1, scan2, last1, last2} per wave
    int4 *sRed = sX + 2 * NW;                 // [NW] {row max, leftmost, rightmost, 0} per wave
    int32_t *edgeW = (int32_t *)(sRed + NW);  // [2] (+2 pad) the word left of a later step's first column
    int32_t *sSink = edgeW + 4;               // [0] H of the sink column, [1] state slot hand-over, [2] best sink value, [3] its row + 1
    uint64_t *sChunk = (uint64_t *)(sSink + 4);  // [4] a new chunk's index
    constexpr int HDR = (3 * NW + 1 + 1 + 2) * 16;
    const uint32_t lds_cols = t5_own(A.lds_cols), hg_cols = t5_own(A.hg_cols);
    const int wmask = (int)(hg_cols - 1u);  // (a power of two: the host sees to it)
    int32_t *Hs = (int32_t *)(smem + HDR);                                // [hg_cols] 4 H + 1
    uint16_t *Gs = (uint16_t *)(smem + HDR + 4ull * hg_cols);             // [hg_cols] G1 | G2 << 8
    uint16_t *Qn = (uint16_t *)(smem + HDR + 6ull * hg_cols);             // [lds_cols / 4] four one-hot column codes per halfword

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) A.outs[blockIdx.x].t_begin = __builtin_amdgcn_s_memrealtime();
    if (A.prio) __builtin_amdgcn_s_setprio(3);
    const poa_prob pb = probs[blockIdx.x];
    const int qlen = t5_own((int)pb.qlen);
    const char *query = queries + pb.q0;
    const uint4 *ntab = node_tab + t5_own(pb.node0);
    const uint32_t *plist = preds + t5_own(pb.pred0);
    const uint32_t *seqw = seq32 + t5_own(pb.seq0 >> 2);
    poa_row *R = A.rows + t5_own(pb.row0);
    const uint32_t n_nodes = t5_own(pb.n_nodes), ring_rows = t5_own(pb.ring_rows);
    const int bw = t5_own((int)pb.w);
    const int banded = t5_own(A.P.banded);
    const int p_match = t5_own(A.P.match), p_mismatch = t5_own(A.P.mismatch);
    const int o1 = DEF ? 4 : t5_own(A.P.o1), e1 = DEF ? 2 : t5_own(A.P.e1), o2 = DEF ? 24 : t5_own(A.P.o2), e2 = DEF ? 1 : t5_own(A.P.e2);
    const int D1 = 4 * o1, D2 = 4 * o2 + 1;  // (the gap-byte arithmetic of k_poa_dp_t5)
    const uint32_t g_bias = (uint32_t)(4 * e1 | (4 * e2) << 8) * 0x00010001u;
    const uint32_t e_probe = (uint32_t)((128 - D1) | (128 - D2) << 8) * 0x00010001u;

    // ---- a state region (ring of value rows) and chunks for the direction rows: as k_poa_dp_t5 in chunk-pool mode
    int status = POA_ST_OK;
    int got = -1;
    if (A.cp.n_slots != 0 && !(pb.flags & 1u)) {
        if (tid == 0) sSink[1] = poa_slot_acquire(A.cp.slot_flag, A.cp.n_slots, blockIdx.x);
        __syncthreads();
        got = __builtin_amdgcn_readfirstlane(sSink[1]);
    }
    if (got < 0) {
        if (tid == 0) {
            poa_out &O = A.outs[blockIdx.x];
            O.t_end = O.t_begin; O.cells = 0; O.vcells = 0; O.maxw = 0; O.nops = 0;
            O.score = POA_NEG; O.row = 0; O.status = A.cp.n_slots ? POA_ST_POOL : POA_ST_RETRY;
        }
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t state_slot = (uint32_t)got;
    const uint64_t state_lo = (uint64_t)A.cp.state_base + (uint64_t)state_slot * A.cp.state_size;
    uint32_t own_head = POA_NIL, own_tail = POA_NIL, own_chunks = 0;
    uint64_t dcur = 0;
    uint32_t drem = 0;
    bool failed = false;
    // direction rows (and the value rows that outlive the ring) out of 1 MiB chunks: one thread pops, the workgroup hears of it
    // through LDS -- every wave reaches this branch in the same row (its condition is replicated state)
    auto alloc = [&](uint32_t bytes_asked) -> uint64_t {
        const uint32_t bytes = (bytes_asked + 15u) & ~15u;
        if (__builtin_expect(bytes > drem, 0)) {
            if (bytes > POA_CHUNK) { failed = true; return dcur; }
            if (tid == 0) {
                const uint32_t idx = poa_chunk_pop(A.cp, blockIdx.x);
                if (idx != POA_NIL) __hip_atomic_store(A.cp.next + idx, own_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sChunk[0] = idx == POA_NIL ? 0ull : poa_chunk_addr(A.cp, idx);
                sChunk[1] = idx;
            }
            __syncthreads();
            const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sChunk[1]);
            const uint64_t a = poa_uniform_u64(sChunk[0]);
            __syncthreads();
            if (idx == POA_NIL) { failed = true; return dcur; }
            own_head = idx;
            if (own_tail == POA_NIL) own_tail = idx;
            own_chunks++;
            dcur = a;
            drem = (uint32_t)POA_CHUNK;
        }
        const uint64_t r = dcur;
        dcur += bytes;
        drem -= bytes;
        return r;
    };

    // ---- column codes (one-hot nibbles, four columns per halfword): column j stands for query[j - 1]
    int non_acgt = 0;
    for (int t = tid; t < (int)(lds_cols / 4); t += NT) {
        uint32_t hw = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = 4 * t + k;
            uint32_t code = 0;
            if (j >= 1 && j <= qlen) {
                const char ch = query[j - 1];
                code = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') ? 1u << (((uint32_t)ch >> 1) & 3u) : 0u;
                non_acgt |= code == 0;
            }
            hw |= code << (4 * k);
        }
        Qn[t] = (uint16_t)hw;
    }
    if (tid == 0) { sSink[2] = POA_NEG; sSink[3] = 0; }
    const bool q_plain = __builtin_amdgcn_readfirstlane(__syncthreads_or(non_acgt)) == 0;
    // a row of this kernel is never wider than the window, nor than the query
    const uint32_t ring_size = (6u * ((hg_cols < (uint32_t)qlen + 8u ? hg_cols : (uint32_t)qlen + 8u) + 8u) + 15u) & ~15u;
    const uint64_t ring_base = state_lo;
    if ((uint64_t)ring_size * ring_rows > A.cp.state_size) { failed = true; status = POA_ST_RETRY; }
    uint32_t ring_head = 0;

    int prev_beg = 0, prev_end = -1;
    uint32_t seq_word = 0, seq_word_idx = 0xFFFFFFFFu;
    uint64_t cells = 0, vcells = 0;
    int maxw = 0;

    for (uint32_t v = 0; v < n_nodes && !failed; v++) {
    const uint4 nt = ntab[v];
    const uint32_t nlen = nt.y & 0xFFFFFFu;
    const int np_node = (int)(nt.y >> 24);
    for (uint32_t tn = 0; tn < nlen && !failed; tn++) {
        POA_MARK("t7_row");
        const uint32_t r = nt.x + tn;
        const bool last = tn + 1 == nlen;
        const bool is_sink = last && (nt.z >> 31) != 0;
        const uint32_t ps = nt.w;
        uint32_t gb = 0;
        if (v > 0) {
            const uint32_t bi = r - 1;
            if ((bi & 3u) == 0 || (bi >> 2) != seq_word_idx) { seq_word_idx = bi >> 2; seq_word = seqw[seq_word_idx]; }
            gb = (seq_word >> (8u * (bi & 3u))) & 0xffu;
        }
        const bool simple = r > 0 && (tn > 0 || (np_node == 1 && ps == r - 1));
        const bool first = tn == 0 && v > 0;
        const int np = v == 0 ? 0 : (tn == 0 ? np_node : 1);
        const int remain = (int)(nt.z & 0x3fffffffu) + (int)(nlen - 1 - tn);
        // ---- the previous row's maximum: every wave combines the waves' triples (k_poa_dp_t5)
        int prev_lmax = 0, prev_rmax = 0;
        if (r > 0) {
            int4 rw = make_int4(INT32_MIN, INT32_MAX, INT32_MIN, 0);
            if (lane < NW) rw = sRed[lane];
            int b = rw.x, t;
            t = poa_dpp<0x111, 0xf>(INT32_MIN, b); b = t > b ? t : b;
            if (NW > 2) { t = poa_dpp<0x112, 0xf>(INT32_MIN, b); b = t > b ? t : b; }
            if (NW > 4) { t = poa_dpp<0x114, 0xf>(INT32_MIN, b); b = t > b ? t : b; }
            if (NW > 8) { t = poa_dpp<0x118, 0xf>(INT32_MIN, b); b = t > b ? t : b; }
            const int rbest = __builtin_amdgcn_readlane(b, NW - 1);
            const uint64_t tie = __builtin_amdgcn_ballot_w64(rw.x == rbest);
            if (__builtin_expect(__builtin_popcountll(tie) == 1, 1)) {
                const int w1 = __builtin_ctzll(tie);
                prev_lmax = __builtin_amdgcn_readlane(rw.y, w1);
                prev_rmax = __builtin_amdgcn_readlane(rw.z, w1);
            } else {
                int lm = rw.x == rbest ? rw.y : INT32_MAX, rm = rw.x == rbest ? rw.z : INT32_MIN;
                t = poa_dpp<0x111, 0xf>(INT32_MAX, lm); lm = t < lm ? t : lm;
                t = poa_dpp<0x111, 0xf>(INT32_MIN, rm); rm = t > rm ? t : rm;
                if (NW > 2) {
                    t = poa_dpp<0x112, 0xf>(INT32_MAX, lm); lm = t < lm ? t : lm;
                    t = poa_dpp<0x112, 0xf>(INT32_MIN, rm); rm = t > rm ? t : rm;
                }
                if (NW > 4) {
                    t = poa_dpp<0x114, 0xf>(INT32_MAX, lm); lm = t < lm ? t : lm;
                    t = poa_dpp<0x114, 0xf>(INT32_MIN, rm); rm = t > rm ? t : rm;
                }
                if (NW > 8) {
                    t = poa_dpp<0x118, 0xf>(INT32_MAX, lm); lm = t < lm ? t : lm;
                    t = poa_dpp<0x118, 0xf>(INT32_MIN, rm); rm = t > rm ? t : rm;
                }
                prev_lmax = __builtin_amdgcn_readlane(lm, NW - 1);
                prev_rmax = __builtin_amdgcn_readlane(rm, NW - 1);
            }
            // (rows a later row can name as a far predecessor: node ends -- their record keeps the columns)
            if (tn == 0 && tid == 0) { R[r - 1].lmax = prev_lmax; R[r - 1].rmax = prev_rmax; }
        }
        // ---- band
        int mpl, mpr;
        if (r == 0) { mpl = 0; mpr = 0; }
        else if (simple) { mpl = prev_lmax + 1; mpr = prev_rmax + 1; }
        else {
            __syncthreads();  // vmcnt(0) + barrier: value rows and row records of far predecessors have landed
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
        if (!banded) { beg = 0; end = qlen; }
        else {
            const int diag = qlen - remain;
            const int lo = mpl < diag ? mpl : diag;
            const int hi = mpr > diag ? mpr : diag;
            beg = lo - bw; if (beg < 0) beg = 0;
            end = hi + bw; if (end > qlen) end = qlen;
        }
        const int bal = beg & ~3;
        const int W = (end - bal + 1 + 3) & ~3;
        const int nbase = beg & ~7;  // the first lane's first column
        if ((uint32_t)(end - nbase + 1 + 8) > hg_cols) { failed = true; status = POA_ST_RETRY; break; }  // (wider than the LDS window)
        const uint64_t doff = alloc((uint32_t)W * (np > 1 ? 4u : 1u));
        uint64_t voff = 0;
        if (last && !failed) {
            if (r == 0 || (nt.z & 0x40000000u)) voff = alloc(6u * (uint32_t)W);
            else {
                voff = ring_base + (uint64_t)ring_head * ring_size;
                ring_head = ring_head + 1 == ring_rows ? 0 : ring_head + 1;
            }
        }
        if (__builtin_expect(failed, 0)) break;
        if (r > 0) cells += (uint64_t)(end - beg + 1);
        if (last) vcells += (uint64_t)(end - beg + 1);
        maxw = W > maxw ? W : maxw;
        if (tid == 0) {
            *(int4 *)&R[r].beg = make_int4(beg, end, (int)(uint32_t)doff, (int)(uint32_t)(doff >> 32));
            *(uint2 *)&R[r].pred = make_uint2(ps, first ? (uint32_t)np : 0u);
            if (last) R[r].voff = voff;
        }
        uint8_t *drow = (uint8_t *)doff;
        uint8_t *Vrow = (uint8_t *)voff;

        // ---- the row's base
        const uint32_t gd = gb - (uint32_t)'A';
        const bool acgt = gd < 20u && ((0x80045u >> gd) & 1u);
        const int sc_eq = acgt ? p_match : 0, sc_ne = acgt ? -p_mismatch : 0;
        const int gsh = (int)((gb >> 1) & 3u);
        const int ne4t = 4 * sc_ne + 1, mm4 = 4 * (sc_eq - sc_ne);

        int pbeg = prev_beg, pend = prev_end;  // where the row above is defined
        if (!simple && r > 0) {
            POA_MARK("t7_stage");
            // ---- STAGING (k_poa_dp_t5's): the virtual predecessor row of the predecessors' value rows, into LDS
            // (a predecessor without a value row -- vga_poa_t5.hpp, staging: the problem is given up, by every wave alike)
            for (int t = 0; t < np; t++) failed |= poa_uniform_u64(R[np == 1 ? ps : plist[ps + t]].voff) == 0;
            if (__builtin_expect(failed, 0)) break;
            for (int c0 = 0; nbase + c0 <= end; c0 += STEP) {
                const int jl = nbase + c0 + CPL * tid;
                if (jl > end) continue;
                int hl = T4_NEG;
                if (np == 1) {
                    const int bp = __builtin_amdgcn_readfirstlane(R[ps].beg), ep = __builtin_amdgcn_readfirstlane(R[ps].end);
                    const uint64_t vq_off = poa_uniform_u64(R[ps].voff);
                    const uint8_t *Vq = (const uint8_t *)vq_off;
                    const int balq = bp & ~3;
                    const int Wq = vq_off != 0 ? (ep - balq + 1 + 3) & ~3 : 0;
                    const unsigned pspan = (unsigned)(ep - bp);
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const int j0 = jl + 4 * q;
                        const int idx = j0 - balq;
                        int4 hv = make_int4(T4_NEG + 1, T4_NEG + 1, T4_NEG + 1, T4_NEG + 1);
                        uint2 gg = make_uint2(0u, 0u);
                        if (idx >= 0 && idx < Wq) {
                            hv = *(const int4 *)((const int32_t *)Vq + idx);
                            gg = *(const uint2 *)(Vq + 4ll * Wq + 2ll * idx);
                        }
                        const bool in0 = (unsigned)(j0 - bp) <= pspan, in1 = (unsigned)(j0 + 1 - bp) <= pspan, in2 = (unsigned)(j0 + 2 - bp) <= pspan,
                                   in3 = (unsigned)(j0 + 3 - bp) <= pspan;
                        hv.x = in0 ? hv.x : T4_NEG + 1; hv.y = in1 ? hv.y : T4_NEG + 1; hv.z = in2 ? hv.z : T4_NEG + 1; hv.w = in3 ? hv.w : T4_NEG + 1;
                        gg.x = (in0 ? gg.x & 0xffffu : 0u) | (in1 ? gg.x & 0xffff0000u : 0u);
                        gg.y = (in2 ? gg.y & 0xffffu : 0u) | (in3 ? gg.y & 0xffff0000u : 0u);
                        *(int4 *)(Hs + (j0 & wmask)) = hv;
                        *(uint2 *)(Gs + (j0 & wmask)) = gg;
                    }
                    if (c0 == 0 && tid == 0 && nbase > 0) {
                        const int idx = jl - 1 - balq;
                        const int wl = (idx >= 0 && idx < Wq) ? ((const int32_t *)Vq)[idx] : T4_NEG + 1;
                        Hs[(nbase - 1) & wmask] = (unsigned)(jl - 1 - bp) <= pspan ? wl : T4_NEG + 1;
                    }
                } else {
                    int hm[Q][4], x1[Q][4], x2[Q][4];
                    uint32_t ah[Q], a1[Q], a2[Q];
                    uint32_t ahl = 0;
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        ah[q] = 0; a1[q] = 0; a2[q] = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) { hm[q][k] = T4_NEG; x1[q][k] = T4_NEG; x2[q][k] = T4_NEG; }
                    }
                    for (int t = 0; t < np; t++) {
                        const uint32_t p = plist[ps + t];
                        const int bp = __builtin_amdgcn_readfirstlane(R[p].beg), ep = __builtin_amdgcn_readfirstlane(R[p].end);
                        const uint64_t vq_off = poa_uniform_u64(R[p].voff);
                        const uint8_t *Vq = (const uint8_t *)vq_off;
                        const int balq = bp & ~3;
                        const int Wq = vq_off != 0 ? (ep - balq + 1 + 3) & ~3 : 0;
                        const unsigned pspan = (unsigned)(ep - bp);
#pragma unroll
                        for (int q = 0; q < Q; q++) {
                            const int j0 = jl + 4 * q;
                            const int idx = j0 - balq;
                            int4 hv = make_int4(0, 0, 0, 0);
                            uint2 gg = make_uint2(0u, 0u);
                            if (idx >= 0 && idx < Wq) {
                                hv = *(const int4 *)((const int32_t *)Vq + idx);
                                gg = *(const uint2 *)(Vq + 4ll * Wq + 2ll * idx);
                            }
                            const int hj[4] = {hv.x, hv.y, hv.z, hv.w};
                            const uint32_t g16[4] = {gg.x & 0xffffu, gg.x >> 16, gg.y & 0xffffu, gg.y >> 16};
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if ((unsigned)(j0 + k - bp) <= pspan) {
                                    const int h = hj[k], c1 = h - (int)(g16[k] & 255u), c2 = h - (int)(g16[k] >> 8);
                                    if (h > hm[q][k]) { hm[q][k] = h; ah[q] = (ah[q] & ~(255u << (8 * k))) | ((uint32_t)t << (8 * k)); }
                                    if (c1 > x1[q][k]) { x1[q][k] = c1; a1[q] = (a1[q] & ~(255u << (8 * k))) | ((uint32_t)t << (8 * k)); }
                                    if (c2 > x2[q][k]) { x2[q][k] = c2; a2[q] = (a2[q] & ~(255u << (8 * k))) | ((uint32_t)t << (8 * k)); }
                                }
                            }
                        }
                        {
                            const int idx = jl - 1 - balq;
                            const int wl = (idx >= 0 && idx < Wq) ? ((const int32_t *)Vq)[idx] : 0;
                            if (jl >= 1 && (unsigned)(jl - 1 - bp) <= pspan && wl > hl) { hl = wl; ahl = (uint32_t)t; }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const int j0 = jl + 4 * q;
                        uint32_t gv[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) gv[k] = (uint32_t)(hm[q][k] - x1[q][k]) | ((uint32_t)(hm[q][k] - x2[q][k]) << 8);
                        *(int4 *)(Hs + (j0 & wmask)) = make_int4(hm[q][0], hm[q][1], hm[q][2], hm[q][3]);
                        *(uint2 *)(Gs + (j0 & wmask)) = make_uint2(gv[0] | (gv[1] << 16), gv[2] | (gv[3] << 16));
                        const int c = j0 - bal;
                        if (c >= 0 && c < W) {
                            const uint32_t left = q == 0 ? ahl : (ah[q > 0 ? q - 1 : 0] >> 24);
                            *(uint32_t *)(drow + (uint32_t)(W + c)) = left | (ah[q] << 8);
                            *(uint32_t *)(drow + (uint32_t)(2 * W + c)) = a1[q];
                            *(uint32_t *)(drow + (uint32_t)(3 * W + c)) = a2[q];
                        }
                    }
                    if (c0 == 0 && tid == 0 && nbase > 0) Hs[(nbase - 1) & wmask] = hl;
                }
            }
            POA_LDS_BARRIER();
            // the virtual row is defined on every column the row can look at
            pbeg = nbase > 0 ? nbase - 1 : 0;
            pend = INT32_MAX / 2;
        }

        // ---- steps of NT * 8 columns
        int best = INT32_MIN, bfirst = 0, blast = 0;
        int carry1 = POA_IDENT, carry2 = POA_IDENT, left1 = POA_IDENT, left2 = POA_IDENT;
        int buf = 0;
        for (int c0 = 0; nbase + c0 <= end; c0 += STEP, buf ^= 1) {
            const int jw0 = nbase + c0 + 64 * CPL * wv;  // the wave's first column
            const int j0 = jw0 + CPL * lane;
            const bool wave_act = jw0 <= end;
            const bool more = nbase + c0 + STEP <= end;  // another step follows (then every wave is active in this one)
            const int base1 = 4 * e1 * j0, base2 = 4 * e2 * j0;  // (the scan runs in absolute "a-space" across waves and steps)
            int H[Q][4];
            uint32_t Ga[Q], Gb[Q];
            int htt[Q][4], ht4[Q][4], e1t[Q][4], e2t[Q][4];
            int agg1 = POA_IDENT, agg2 = POA_IDENT, alast1 = POA_IDENT, alast2 = POA_IDENT;
            POA_MARK("t7_p1");
            if (wave_act) {
                if (__builtin_expect(r > 0, 1)) {
                    // the row above: eight words, eight gap-byte pairs, the codes of the eight columns
                    {
                        const int4 h0 = *(const int4 *)(Hs + (j0 & wmask)), h1 = *(const int4 *)(Hs + ((j0 + 4) & wmask));
                        const uint4 g = *(const uint4 *)(Gs + (j0 & wmask));
                        H[0][0] = h0.x; H[0][1] = h0.y; H[0][2] = h0.z; H[0][3] = h0.w;
                        H[1][0] = h1.x; H[1][1] = h1.y; H[1][2] = h1.z; H[1][3] = h1.w;
                        Ga[0] = g.x; Gb[0] = g.y; Ga[1] = g.z; Gb[1] = g.w;
                    }
                    const uint32_t qq = *(const uint32_t *)(Qn + (j0 >> 2));
                    // the word left of the wave: the last word of the wave before (same step: not written yet), or -- first wave of
                    // a later step -- what the last lane parked
                    int left0;
                    if (wv == 0 && c0 > 0) left0 = edgeW[buf ^ 1];
                    else left0 = Hs[(jw0 > 0 ? jw0 - 1 : 0) & wmask];
                    if (more && tid == NT - 1) edgeW[buf] = H[1][3];
                    // band edges of the row above (uniform per wave)
                    if (__builtin_expect(jw0 <= pbeg || jw0 + 64 * CPL - 1 > pend, 0)) {
#pragma unroll
                        for (int q = 0; q < Q; q++)
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const int j = j0 + 4 * q + k;
                                H[q][k] = (j < pbeg || j > pend) ? T4_NEG + 1 : H[q][k];
                            }
                        left0 = (jw0 - 1 < pbeg || jw0 - 1 > pend || jw0 == 0) ? T4_NEG + 1 : left0;
                    }
                    int hp = t4_shr1_mov(H[1][3], left0);
                    // (a query character other than A / C / G / T scores 0 against anything: its column code is 0, and the copy of
                    // the loop for such queries adds the mismatch term only where the code is not)
                    auto cells = [&](auto plain_c) {
                        constexpr bool PLAIN = decltype(plain_c)::value;
#pragma unroll
                        for (int q = 0; q < Q; q++) {
                            const uint32_t q4 = q == 0 ? qq & 0xffffu : qq >> 16;
                            const uint32_t eqb = q4 >> gsh;
                            const uint32_t anyb = PLAIN ? 0u : (q4 | (q4 >> 1) | (q4 >> 2) | (q4 >> 3));
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const int g = (int)(k < 2 ? Ga[q] : Gb[q]);
                                const int ev1 = (k & 1) ? t4_sub_byte<2>(H[q][k], g) : t4_sub_byte<0>(H[q][k], g);
                                const int ev2 = (k & 1) ? t4_sub_byte<3>(H[q][k], g) : t4_sub_byte<1>(H[q][k], g);
                                int m;
                                if constexpr (PLAIN) m = (int)__umul24(__builtin_amdgcn_ubfe(eqb, 4u * k, 1u), (uint32_t)mm4) + (hp + ne4t);
                                else
                                    m = (int)__umul24(__builtin_amdgcn_ubfe(eqb, 4u * k, 1u), (uint32_t)mm4) +
                                        (int)__builtin_amdgcn_ubfe(anyb, 4u * k, 1u) * (ne4t - 1) + (hp + 1);
                                htt[q][k] = t4_max3(m, ev1, ev2);
                                e1t[q][k] = ev1;
                                e2t[q][k] = ev2;
                                hp = H[q][k];
                            }
                        }
                    };
                    if (__builtin_expect(q_plain, 1)) cells(std::true_type{});
                    else cells(std::false_type{});
                } else {
                    // the source row: H(0, 0) = 0, everything else comes out of the insertion scan
#pragma unroll
                    for (int q = 0; q < Q; q++)
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            htt[q][k] = (j0 + 4 * q + k == 0 ? 0 : T4_NEG) + 2;
                            e1t[q][k] = T4_NEG + 1;
                            e2t[q][k] = T4_NEG;
                        }
                }
                POA_MARK("t7_scan");
                int a1 = POA_IDENT, a2 = POA_IDENT;
                const int sb = beg - nbase;  // cells of the very first lane left of beg stay out of the scan
#pragma unroll
                for (int q = 0; q < Q; q++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int c = 4 * q + k;
                        int h4 = htt[q][k] & ~3;
                        if (c < CPL - 1) h4 = (tid == 0 && c0 == 0 && c < sb) ? POA_IDENT : h4;
                        ht4[q][k] = h4;
                        const int r1 = h4 + 4 * e1 * c, r2 = h4 + 4 * e2 * c;
                        a1 = r1 > a1 ? r1 : a1;
                        a2 = r2 > a2 ? r2 : a2;
                        if (c == CPL - 1) { alast1 = r1 + base1; alast2 = r2 + base2; }
                    }
                agg1 = a1 + base1;
                agg2 = a2 + base2;
            }
            int i1 = POA_IDENT, i2 = POA_IDENT;
            if (wave_act) { i1 = poa_wave_scan_max(agg1); i2 = poa_wave_scan_max(agg2); }
            if (lane == 63) sX[buf * NW + wv] = make_int4(i1, i2, alast1, alast2);
            POA_LDS_BARRIER();
            POA_MARK("t7_exchange");
            int pre1 = carry1, pre2 = carry2, pl1 = left1, pl2 = left2;
            if ((wave_act && wv > 0) || more) {
                int4 x = make_int4(POA_IDENT, POA_IDENT, POA_IDENT, POA_IDENT);
                if (lane < NW) x = sX[buf * NW + lane];
                int s1 = x.x, s2 = x.y, t;
                t = poa_dpp<0x111, 0xf>(INT32_MIN, s1); s1 = t > s1 ? t : s1;
                t = poa_dpp<0x111, 0xf>(INT32_MIN, s2); s2 = t > s2 ? t : s2;
                if (NW > 2) {
                    t = poa_dpp<0x112, 0xf>(INT32_MIN, s1); s1 = t > s1 ? t : s1;
                    t = poa_dpp<0x112, 0xf>(INT32_MIN, s2); s2 = t > s2 ? t : s2;
                }
                if (NW > 4) {
                    t = poa_dpp<0x114, 0xf>(INT32_MIN, s1); s1 = t > s1 ? t : s1;
                    t = poa_dpp<0x114, 0xf>(INT32_MIN, s2); s2 = t > s2 ? t : s2;
                }
                if (NW > 8) {
                    t = poa_dpp<0x118, 0xf>(INT32_MIN, s1); s1 = t > s1 ? t : s1;
                    t = poa_dpp<0x118, 0xf>(INT32_MIN, s2); s2 = t > s2 ? t : s2;
                }
                if (wv > 0) {
                    const int a = __builtin_amdgcn_readlane(s1, wv - 1), b = __builtin_amdgcn_readlane(s2, wv - 1);
                    pre1 = a > pre1 ? a : pre1;
                    pre2 = b > pre2 ? b : pre2;
                    pl1 = __builtin_amdgcn_readlane(x.z, wv - 1);
                    pl2 = __builtin_amdgcn_readlane(x.w, wv - 1);
                }
                if (more) {
                    const int a = __builtin_amdgcn_readlane(s1, NW - 1), b = __builtin_amdgcn_readlane(s2, NW - 1);
                    carry1 = a > carry1 ? a : carry1;
                    carry2 = b > carry2 ? b : carry2;
                    left1 = __builtin_amdgcn_readlane(x.z, NW - 1);
                    left2 = __builtin_amdgcn_readlane(x.w, NW - 1);
                }
            }
            if (wave_act) {
                POA_MARK("t7_p2");
                const int run1_ = t4_shr1_max(i1, pre1), run2_ = t4_shr1_max(i2, pre2);
                const int la1_ = t4_shr1_mov(alast1, pl1), la2_ = t4_shr1_mov(alast2, pl2);
                int R1 = run1_ - base1, R2 = run2_ - base2, L1 = la1_ - base1, L2 = la2_ - base2;
                uint32_t dirs[Q];
                const bool edge_out = jw0 + 64 * CPL - 1 > end;  // the wave that holds `end`: what lies right of it is written back clean
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    int dirq = 0, ga = 0, gbb = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int c = 4 * q + k;
                        const int h4 = ht4[q][k];
                        const int f1 = R1 - (4 * (o1 + e1 * c) - 1), f2 = R2 - 4 * (o2 + e2 * c);
                        const int hh = t4_max3(h4 | 3, f1, f2);
                        const int h = (hh & ~3) | 1;
                        int acc = (hh << 2) | (htt[q][k] & 3);
                        const int u1 = h - e1t[q][k], u2 = h - e2t[q][k];
                        if (k == 0) { t5_min_byte<0>(ga, u1, D1); t5_min_byte<1>(ga, u2, D2); }
                        if (k == 1) { t5_min_byte<2>(ga, u1, D1); t5_min_byte<3>(ga, u2, D2); }
                        if (k == 2) { t5_min_byte<0>(gbb, u1, D1); t5_min_byte<1>(gbb, u2, D2); }
                        if (k == 3) { t5_min_byte<2>(gbb, u1, D1); t5_min_byte<3>(gbb, u2, D2); }
                        t4_flag_ne(acc, R1, L1);
                        if (k == 0) t4_flag_ne_dep<0>(dirq, acc, R2, L2);
                        if (k == 1) t4_flag_ne_dep<1>(dirq, acc, R2, L2);
                        if (k == 2) t4_flag_ne_dep<2>(dirq, acc, R2, L2);
                        if (k == 3) t4_flag_ne_dep<3>(dirq, acc, R2, L2);
                        L1 = h4 + 4 * e1 * c; L2 = h4 + 4 * e2 * c;
                        R1 = L1 > R1 ? L1 : R1;
                        R2 = L2 > R2 ? L2 : R2;
                        H[q][k] = h;
                    }
                    {
                        const uint32_t ya = (uint32_t)ga + e_probe, yb = ((uint32_t)gbb + e_probe) >> 1;
                        const uint32_t e8 = (ya & 0x80808080u) | (yb & ~0x80808080u);
                        dirq = (int)((e8 & 0xC0C0C0C0u) | ((uint32_t)dirq & ~0xC0C0C0C0u));
                        Ga[q] = (uint32_t)ga + g_bias;
                        Gb[q] = (uint32_t)gbb + g_bias;
                    }
                    dirs[q] = (uint32_t)dirq;
                }
                if (__builtin_expect(edge_out, 0)) {
#pragma unroll
                    for (int q = 0; q < Q; q++)
#pragma unroll
                        for (int k = 0; k < 4; k++) H[q][k] = j0 + 4 * q + k > end ? T4_NEG + 1 : H[q][k];
                }
                if (__builtin_expect(is_sink, 0)) {
                    const int kq = qlen - j0;
#pragma unroll
                    for (int q = 0; q < Q; q++)
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (kq == 4 * q + k) sSink[0] = H[q][k];
                }
                // the lane's best word and the first / last step it was seen in
                {
                    const int m3 = t4_max3(H[0][0], H[0][1], H[0][2]), n3 = t4_max3(H[1][0], H[1][1], H[1][2]);
                    const int m4 = t4_max3(m3, n3, H[0][3]);
                    const int m8 = m4 > H[1][3] ? m4 : H[1][3];
                    const bool gt = m8 > best;
                    best = gt ? m8 : best;
                    bfirst = gt ? c0 : bfirst;
                    blast = m8 >= best ? c0 : blast;
                }
                POA_MARK("t7_stores");
                if (j0 <= end) {
                    *(int4 *)(Hs + (j0 & wmask)) = make_int4(H[0][0], H[0][1], H[0][2], H[0][3]);
                    *(int4 *)(Hs + ((j0 + 4) & wmask)) = make_int4(H[1][0], H[1][1], H[1][2], H[1][3]);
                    *(uint4 *)(Gs + (j0 & wmask)) = make_uint4(Ga[0], Gb[0], Ga[1], Gb[1]);
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const int c = j0 + 4 * q - bal;
                        if (c >= 0 && c < W) {
                            *(uint32_t *)(drow + (uint32_t)c) = dirs[q];
                            if (last) {
                                *(int4 *)(Vrow + 4u * (uint32_t)c) = make_int4(H[q][0], H[q][1], H[q][2], H[q][3]);
                                *(uint2 *)(Vrow + (uint32_t)(4 * W + 2 * c)) = make_uint2(Ga[q], Gb[q]);
                            }
                        }
                    }
                }
            }
        }
        POA_MARK("t7_rowmax");
        // ---- the wave's (maximum, leftmost, rightmost column): from the lane's own cells of the steps it saw its best word in
        {
            const int wb = __builtin_amdgcn_readlane(poa_wave_scan_max(best), 63);
            int lcol = INT32_MAX, rcol = INT32_MIN;
            if (best == wb && wb != INT32_MIN) {
                const int jf = nbase + bfirst + 64 * CPL * wv + CPL * lane, jl2 = nbase + blast + 64 * CPL * wv + CPL * lane;
                const int4 f0 = *(const int4 *)(Hs + (jf & wmask)), f1 = *(const int4 *)(Hs + ((jf + 4) & wmask));
                const int4 l0 = *(const int4 *)(Hs + (jl2 & wmask)), l1 = *(const int4 *)(Hs + ((jl2 + 4) & wmask));
                const int hf[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                const int hl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
                int cf = 0, cl = 0;
#pragma unroll
                for (int c = 7; c >= 0; c--) cf = hf[c] == wb ? c : cf;
#pragma unroll
                for (int c = 0; c < 8; c++) cl = hl[c] == wb ? c : cl;
                lcol = jf + cf;
                rcol = jl2 + cl;
            }
            lcol = poa_wave_scan_min(lcol);
            rcol = poa_wave_scan_max(rcol);
            if (lane == 63) sRed[wv] = make_int4(wb, lcol, rcol, 0);
        }
        POA_LDS_BARRIER();
        if (__builtin_expect(is_sink, 0) && tid == 0) {  // (the next sink row's phase 2 is at least one barrier away)
            const int val = (qlen >= beg && qlen <= end) ? sSink[0] >> 2 : POA_NEG;
            if (sSink[3] == 0 || val > sSink[2]) { sSink[2] = val; sSink[3] = (int)r + 1; }
        }
        prev_beg = beg; prev_end = end;
    }
    }
    __syncthreads();
    // (the last row of the problem: its maximum's columns go into its record like every node end's -- nothing reads them)
    if (tid >= 64) return;
    // ---- epilogue, first wave: the result, the traceback (k_poa_dp_t5's), the pool
    poa_out &O = A.outs[blockIdx.x];
    uint32_t start_row = 0;
    int sink_best = POA_NEG;
    if (failed) { if (status == POA_ST_OK) status = POA_ST_POOL; }
    else {
        sink_best = __builtin_amdgcn_readfirstlane(sSink[2]);
        const int sr = __builtin_amdgcn_readfirstlane(sSink[3]);
        start_row = sr ? (uint32_t)(sr - 1) : 0u;
        status = (sr != 0 && sink_best > POA_NEG / 2) ? POA_ST_OK : POA_ST_NOALN;
    }
    if (tid == 0) {
        O.cells = failed ? 0 : cells; O.vcells = failed ? 0 : vcells; O.maxw = failed ? 0u : (uint32_t)maxw;
        O.score = failed ? POA_NEG : sink_best;
        O.row = start_row;
        O.status = status;
    }
    if (A.tb_ops) poa_traceback_wave<2>(*(tb_lds *)(smem + HDR), tid, pb, A.rows, preds, nullptr, O, A.tb_ops, A.tb_orow, 0, status, start_row);
    if (tid == 0) {
        O.t_end = __builtin_amdgcn_s_memrealtime();
        if (own_head != POA_NIL) poa_chunk_push(A.cp, blockIdx.x, own_head, own_tail);
        (void)atomicAdd(A.pool_next, (unsigned long long)own_chunks * POA_CHUNK + (uint64_t)ring_size * ring_rows);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        (void)atomicExch(&A.cp.slot_flag[state_slot], 0u);
    }
}
