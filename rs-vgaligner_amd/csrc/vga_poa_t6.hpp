// vga_poa_t6.hpp -- K4 "t6": ONE WAVE PER PROBLEM for narrow bands (config 5: mean band 340 columns, widest rows 350-700).
// Same recurrences, direction dwords, value rows, row records and fused traceback as k_poa_dp_t5 (bit-exact against
// oracle/og_poa.c), but the row state never leaves the registers and nothing is exchanged between waves:
//   * a lane owns CPL consecutive columns (CPL = 8: two quads), the wave a WINDOW of 64 CPL columns anchored at a multiple of
//     CPL at or below the band's first column; H words and G bytes of the row above live in 3 CPL / 2 vector registers.  When
//     the anchor moves (every CPL rows while the band slides along the diagonal) the state moves across lanes: one DPP
//     wave_shl per register for the common step of one lane, ds_bpermute for anything else;
//   * a row wider than the window (config 5: one problem in eight has a stretch of them) runs a SECOND STEP of the same code on
//     a second register set that continues the window (128 virtual lanes); the scan totals of the first step are its carry;
//   * no barriers, no LDS row, no cross-wave scan: the max-plus scan of a row is the lane's serial scan, one DPP wave scan and
//     one wave_shr; the row maximum is a DPP reduction and two ballots;
//   * every row is written CLEAN: cells outside [beg, end] leave the row far below every real score, so no row ever masks what
//     it reads -- the per-cell band tests of k_poa_dp_t5's edge wave-steps (every wave-step is one when a row is a single
//     step) shrink to one select per cell on the way out and one on the way into the scan;
//   * rows whose predecessors are not the row above (first row of a node behind a bubble, several predecessors) build the
//     virtual predecessor row of k_poa_dp_t5's staging pass straight into the registers from the predecessors' value rows;
//   * the row loop's scalar state is small enough to stay in scalar registers (k_poa_dp_t5 keeps ~200 scalars alive and
//     pays ~45 v_readlane / v_writelane per wave and row for the ones that spill).
// What it does not do is handed back with POA_ST_RETRY and runs in k_poa_dp_t5: a row whose band does not fit two windows,
// a query with characters other than A / C / G / T, anything outside chunk-pool mode.
#pragma once

template <int CPL>
static inline size_t poa_t6_lds_bytes(uint32_t lds_cols)
{
    return std::max<size_t>(((size_t)lds_cols / 2 + 15u) & ~(size_t)15u, sizeof(tb_lds)) + 64;
}

// lanes 0..62: v of the lane above; lane 63: fill   (wave_shl:1, bound_ctrl off: the last lane keeps the old value)
__device__ __forceinline__ int t6_shl1(int v, int fill)
{
    int r = fill;
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
    return r;
}
// any lane distance d (new[l] = old[l + d], `fill` where l + d leaves the wave)
__device__ __forceinline__ int t6_shift(int v, int fill, int src_lane_bytes, bool in_range)
{
    const int t = __builtin_amdgcn_ds_bpermute(src_lane_bytes, v);
    return in_range ? t : fill;
}

template <int CPL, bool DEF>
__global__ __launch_bounds__(64) void k_poa_dp_t6(const poa_prob *__restrict__ probs, const char *__restrict__ queries,
                                                  const uint4 *__restrict__ node_tab, const uint32_t *__restrict__ seq32,
                                                  const uint32_t *__restrict__ preds, const poa_t5_args A)
{
    static_assert(CPL == 8 || CPL == 12 || CPL == 16, "two to four quads per lane");
    constexpr int Q = CPL / 4;
    constexpr int WIN = 64 * CPL;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *Qn = (uint16_t *)smem;  // [lds_cols / 4] four one-hot column codes per halfword (as k_poa_dp_t5); the traceback's staging area later

    const int lane = threadIdx.x;
    if (lane == 0) A.outs[blockIdx.x].t_begin = __builtin_amdgcn_s_memrealtime();
    const poa_prob pb = probs[blockIdx.x];
    const int qlen = t5_own((int)pb.qlen);
    const char *query = queries + pb.q0;
    const uint4 *ntab = node_tab + t5_own(pb.node0);
    const uint32_t *plist = preds + t5_own(pb.pred0);
    const uint32_t *seqw = seq32 + t5_own(pb.seq0 >> 2);
    poa_row *R = A.rows + t5_own(pb.row0);
    const uint32_t n_nodes = t5_own(pb.n_nodes), ring_rows = t5_own(pb.ring_rows);
    const uint32_t lds_cols = t5_own(A.lds_cols);
    const int bw = t5_own((int)pb.w);
    const int banded = t5_own(A.P.banded);
    const int p_match = t5_own(A.P.match), p_mismatch = t5_own(A.P.mismatch);
    const int o1 = DEF ? 4 : t5_own(A.P.o1), e1 = DEF ? 2 : t5_own(A.P.e1), o2 = DEF ? 24 : t5_own(A.P.o2), e2 = DEF ? 1 : t5_own(A.P.e2);
    const int D1 = 4 * o1, D2 = 4 * o2 + 1;  // (the gap-byte arithmetic of k_poa_dp_t5)
    const uint32_t g_bias = (uint32_t)(4 * e1 | (4 * e2) << 8) * 0x00010001u;
    const uint32_t e_probe = (uint32_t)((128 - D1) | (128 - D2) << 8) * 0x00010001u;

    // ---- a state region (ring of value rows) and, for the direction rows, chunks from the pool: as k_poa_dp_t5 in chunk-pool mode
    int status = POA_ST_OK;
    int got = -1;
    if (A.cp.n_slots != 0 && !(pb.flags & 1u)) {
        if (lane == 0) got = poa_slot_acquire(A.cp.slot_flag, A.cp.n_slots, blockIdx.x);
        got = __builtin_amdgcn_readfirstlane(got);
    }
    if (got < 0) {
        if (lane == 0) {
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
    // direction rows (and the value rows that outlive the ring) out of 1 MiB chunks; lane 0 pops, the wave hears of it by readfirstlane
    auto alloc = [&](uint32_t bytes_asked) -> uint64_t {
        const uint32_t bytes = (bytes_asked + 15u) & ~15u;
        if (__builtin_expect(bytes > drem, 0)) {
            uint32_t idx = POA_NIL;
            if (bytes <= POA_CHUNK) {
                if (lane == 0) {
                    idx = poa_chunk_pop(A.cp, blockIdx.x);
                    if (idx != POA_NIL) __hip_atomic_store(A.cp.next + idx, own_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
            }
            if (idx == POA_NIL) { failed = true; return dcur; }
            own_head = idx;
            if (own_tail == POA_NIL) own_tail = idx;
            own_chunks++;
            uint64_t a = 0;
            if (lane == 0) a = poa_chunk_addr(A.cp, idx);
            dcur = poa_uniform_u64(a);
            drem = (uint32_t)POA_CHUNK;
        }
        const uint64_t r = dcur;
        dcur += bytes;
        drem -= bytes;
        return r;
    };

    // ---- column codes (one-hot nibbles, four columns per halfword): column j stands for query[j - 1]
    int non_acgt = 0;
    for (int t = lane; t < (int)(lds_cols / 4); t += 64) {
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool q_plain = __builtin_amdgcn_ballot_w64(non_acgt != 0) == 0;
    // a row of this kernel never has more than 2 WIN + 4 storage columns: the ring's slots are sized for that
    const uint32_t ring_size = (6u * (uint32_t)(2 * WIN + 8) + 15u) & ~15u;
    const uint64_t ring_base = state_lo;
    if ((uint64_t)ring_size * ring_rows > A.cp.state_size || !q_plain) { failed = true; status = POA_ST_RETRY; }
    uint32_t ring_head = 0;

    // ---- the row above, in registers: lane l of set s holds columns wbase + CPL (64 s + l) .. + CPL - 1.  Set 0 is the window
    // every row runs in; set 1 continues it for the rows that are wider (a second step of the same code): it only holds anything
    // while b_live
    int H[2][Q][4];             // cell words 4 H + 1
    uint32_t Ga[2][Q], Gb[2][Q];  // G1 | G2 << 8 of cells 0, 1 / 2, 3 of each quad
    uint32_t qn[2][Q];          // the column codes of the lane's quads (they only change when the window moves)
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int q = 0; q < Q; q++) {
            qn[s][q] = 0; Ga[s][q] = 0; Gb[s][q] = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) H[s][q][k] = T4_NEG + 1;
        }
    bool b_live = false;
    int wbase = 0;  // first column of the window the registers hold (a multiple of CPL)
    bool have_codes = false;
    const int lane_e1 = 4 * e1 * CPL * lane, lane_e2 = 4 * e2 * CPL * lane;  // a-space offset of the lane's first column in its window

    int prev_lmax = 0, prev_rmax = 0;
    uint32_t seq_word = 0, seq_word_idx = 0xFFFFFFFFu;
    int sink_val = POA_NEG;
    uint32_t sink_row1 = 0;  // best sink row + 1 (0: none yet)
    uint64_t cells = 0, vcells = 0;
    int maxw = 0;

    for (uint32_t v = 0; v < n_nodes && !failed; v++) {
    const uint4 nt = ntab[v];
    const uint32_t nlen = nt.y & 0xFFFFFFu;
    const int np_node = (int)(nt.y >> 24);
    for (uint32_t tn = 0; tn < nlen && !failed; tn++) {
        POA_MARK("t6_row");
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
        // ---- band
        int mpl, mpr;
        if (r == 0) { mpl = 0; mpr = 0; }
        else if (simple) { mpl = prev_lmax + 1; mpr = prev_rmax + 1; }
        else {
            // value rows and row records of far predecessors were stored by this wave: they have landed once its stores have
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        const int nbase = (int)((uint32_t)beg / (uint32_t)CPL) * CPL;  // the window this row needs
        if (end - nbase >= 2 * WIN) { failed = true; status = POA_ST_RETRY; break; }
        const bool two = end - nbase >= WIN;  // a second step
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
        if (lane == 0) {
            // what the traceback reads of every row: band, direction row, predecessor (a row inside a node: npred 0 = the row above)
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

        POA_MARK("t6_state");
        int hp0[2];  // the predecessor's word at the column left of the lane's first, per set
        hp0[1] = T4_NEG + 1;
        if (two && !b_live) {
            // the second set comes into use: nothing of the row above lies there
#pragma unroll
            for (int q = 0; q < Q; q++) {
                Ga[1][q] = 0; Gb[1][q] = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) H[1][q][k] = T4_NEG + 1;
            }
            b_live = true;
        }
        if (__builtin_expect(simple, 1)) {
            // ---- the row above is in the registers: move it if the window moves
            hp0[0] = t4_shr1_mov(H[0][Q - 1][3], T4_NEG + 1);
            if (b_live) hp0[1] = t4_shr1_mov(H[1][Q - 1][3], __builtin_amdgcn_readlane(H[0][Q - 1][3], 63));
            const int dl = (nbase - wbase) / CPL;
            if (dl != 0) {
                if (dl == 1 && !b_live) {
                    hp0[0] = t6_shl1(hp0[0], T4_NEG + 1);
#pragma unroll
                    for (int q = 0; q < Q; q++) {
#pragma unroll
                        for (int k = 0; k < 4; k++) H[0][q][k] = t6_shl1(H[0][q][k], T4_NEG + 1);
                        Ga[0][q] = (uint32_t)t6_shl1((int)Ga[0][q], 0);
                        Gb[0][q] = (uint32_t)t6_shl1((int)Gb[0][q], 0);
                    }
                } else if (dl == 1) {
                    // lane 63 of the first set takes over lane 0 of the second
                    hp0[0] = t6_shl1(hp0[0], __builtin_amdgcn_readlane(hp0[1], 0));
                    hp0[1] = t6_shl1(hp0[1], T4_NEG + 1);
#pragma unroll
                    for (int q = 0; q < Q; q++) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            H[0][q][k] = t6_shl1(H[0][q][k], __builtin_amdgcn_readlane(H[1][q][k], 0));
                            H[1][q][k] = t6_shl1(H[1][q][k], T4_NEG + 1);
                        }
                        Ga[0][q] = (uint32_t)t6_shl1((int)Ga[0][q], __builtin_amdgcn_readlane((int)Ga[1][q], 0));
                        Gb[0][q] = (uint32_t)t6_shl1((int)Gb[0][q], __builtin_amdgcn_readlane((int)Gb[1][q], 0));
                        Ga[1][q] = (uint32_t)t6_shl1((int)Ga[1][q], 0);
                        Gb[1][q] = (uint32_t)t6_shl1((int)Gb[1][q], 0);
                    }
                } else {
                    // any distance: lane l of set s takes what virtual lane 64 s + l + dl held (of 128, the second set only if live)
                    const int s0 = lane + dl, s1 = lane + 64 + dl;
                    const int sb0 = (s0 & 63) << 2, sb1 = (s1 & 63) << 2;
                    const int w0 = s0 >> 6, w1 = s1 >> 6;  // 0: first set, 1: second set, anything else: outside
                    const bool live = b_live;
                    auto move = [&](int &a, int &b, int fill) {
                        const int a0 = __builtin_amdgcn_ds_bpermute(sb0, a), a1 = __builtin_amdgcn_ds_bpermute(sb1, a);
                        int b0 = fill, b1 = fill;
                        if (live) { b0 = __builtin_amdgcn_ds_bpermute(sb0, b); b1 = __builtin_amdgcn_ds_bpermute(sb1, b); }
                        a = w0 == 0 ? a0 : (w0 == 1 ? b0 : fill);
                        b = w1 == 0 ? a1 : (w1 == 1 ? b1 : fill);
                    };
                    move(hp0[0], hp0[1], T4_NEG + 1);
#pragma unroll
                    for (int q = 0; q < Q; q++) {
#pragma unroll
                        for (int k = 0; k < 4; k++) move(H[0][q][k], H[1][q][k], T4_NEG + 1);
                        int x, y;
                        x = (int)Ga[0][q]; y = (int)Ga[1][q]; move(x, y, 0); Ga[0][q] = (uint32_t)x; Ga[1][q] = (uint32_t)y;
                        x = (int)Gb[0][q]; y = (int)Gb[1][q]; move(x, y, 0); Gb[0][q] = (uint32_t)x; Gb[1][q] = (uint32_t)y;
                    }
                }
                have_codes = false;
            }
        } else if (r > 0) {
            // ---- STAGING (k_poa_dp_t5's, into registers): the virtual predecessor row of the predecessors' value rows
            have_codes = false;
            auto stage = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                const int jl = nbase + CPL * (lane + 64 * s);
                if (np == 1) {
                    const int bp = __builtin_amdgcn_readfirstlane(R[ps].beg), ep = __builtin_amdgcn_readfirstlane(R[ps].end);
                    const uint64_t vq_off = poa_uniform_u64(R[ps].voff);
                    const uint8_t *Vq = (const uint8_t *)vq_off;
                    if (__builtin_expect(vq_off == 0, 0)) failed = true;  // (a predecessor without a value row: vga_poa_t5.hpp, staging)
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
                        H[s][q][0] = in0 ? hv.x : T4_NEG + 1; H[s][q][1] = in1 ? hv.y : T4_NEG + 1; H[s][q][2] = in2 ? hv.z : T4_NEG + 1; H[s][q][3] = in3 ? hv.w : T4_NEG + 1;
                        Ga[s][q] = (in0 ? gg.x & 0xffffu : 0u) | (in1 ? gg.x & 0xffff0000u : 0u);
                        Gb[s][q] = (in2 ? gg.y & 0xffffu : 0u) | (in3 ? gg.y & 0xffff0000u : 0u);
                    }
                    {
                        const int idx = jl - 1 - balq;
                        const int wl = (idx >= 0 && idx < Wq) ? ((const int32_t *)Vq)[idx] : T4_NEG + 1;
                        hp0[s] = (unsigned)(jl - 1 - bp) <= pspan ? wl : T4_NEG + 1;
                    }
                } else {
                    int hm[Q][4], x1[Q][4], x2[Q][4];
                    uint32_t ah[Q], a1[Q], a2[Q];  // which predecessor won: a byte per cell (planes of the direction row)
                    int hl = T4_NEG;
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
                        if (__builtin_expect(vq_off == 0, 0)) failed = true;
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
                        uint32_t gv[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            H[s][q][k] = hm[q][k];
                            gv[k] = (uint32_t)(hm[q][k] - x1[q][k]) | ((uint32_t)(hm[q][k] - x2[q][k]) << 8);
                        }
                        Ga[s][q] = gv[0] | (gv[1] << 16);
                        Gb[s][q] = gv[2] | (gv[3] << 16);
                        // predecessor-choice planes: M of column j looks at column j - 1 of the predecessors
                        const int c = jl + 4 * q - bal;
                        if (c >= 0 && c < W) {
                            const uint32_t left = q == 0 ? ahl : (ah[q > 0 ? q - 1 : 0] >> 24);
                            *(uint32_t *)(drow + (uint32_t)(W + c)) = left | (ah[q] << 8);
                            *(uint32_t *)(drow + (uint32_t)(2 * W + c)) = a1[q];
                            *(uint32_t *)(drow + (uint32_t)(3 * W + c)) = a2[q];
                        }
                    }
                    hp0[s] = hl;
                }
            };
            stage(std::integral_constant<int, 0>{});
            if (two) stage(std::integral_constant<int, 1>{});
        }
        wbase = nbase;
        if (!have_codes) {
            // (a lane's first column is a multiple of 4: the quads' halfwords are adjacent)
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const int t = ((nbase + CPL * (lane + 64 * s)) >> 2) + q;
                    qn[s][q] = t < (int)(lds_cols / 4) ? (uint32_t)Qn[t] : 0u;
                }
            have_codes = true;
        }

        // right of `end` the row is cleaned on the way out: virtual lanes above le, and in lane le the cells above se
        const int le = (end - nbase) / CPL, se = (end - nbase) - le * CPL;
        const int sb = beg - nbase;  // 0 .. CPL - 1: cells of the first lane left of beg stay out of the scan
        int bestv[2] = {INT32_MIN, INT32_MIN};
        int tot1 = POA_IDENT, tot2 = POA_IDENT, lst1 = POA_IDENT, lst2 = POA_IDENT;  // what a step hands to the next one
        auto step = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const int jl = nbase + CPL * (lane + 64 * s);
            const int le1 = lane_e1 + 4 * e1 * WIN * s, le2 = lane_e2 + 4 * e2 * WIN * s;
            int htt[Q][4], ht4[Q][4], e1t[Q][4], e2t[Q][4];
            POA_MARK("t6_p1");
            // ---- phase 1: M / E1 / E2 from the row above, Ht' (tagged); the lane's part of the max-plus scan
            if (__builtin_expect(r > 0, 1)) {
                int hp = hp0[s];
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const uint32_t eqb = qn[s][q] >> gsh;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int g = (int)(k < 2 ? Ga[s][q] : Gb[s][q]);
                        const int ev1 = (k & 1) ? t4_sub_byte<2>(H[s][q][k], g) : t4_sub_byte<0>(H[s][q][k], g);
                        const int ev2 = (k & 1) ? t4_sub_byte<3>(H[s][q][k], g) : t4_sub_byte<1>(H[s][q][k], g);
                        const int m = (int)__umul24(__builtin_amdgcn_ubfe(eqb, 4u * k, 1u), (uint32_t)mm4) + (hp + ne4t);
                        htt[q][k] = t4_max3(m, ev1, ev2);
                        e1t[q][k] = ev1;
                        e2t[q][k] = ev2;
                        hp = H[s][q][k];
                    }
                }
            } else {
                // the source row: H(0, 0) = 0, everything else comes out of the insertion scan
#pragma unroll
                for (int q = 0; q < Q; q++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        htt[q][k] = (jl + 4 * q + k == 0 ? 0 : T4_NEG) + 2;
                        e1t[q][k] = T4_NEG + 1;
                        e2t[q][k] = T4_NEG;
                    }
            }
            POA_MARK("t6_scan");
            int agg1, agg2, alast1 = POA_IDENT, alast2 = POA_IDENT;
            {
                int a1 = POA_IDENT, a2 = POA_IDENT;
#pragma unroll
                for (int q = 0; q < Q; q++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int c = 4 * q + k;
                        int h4 = htt[q][k] & ~3;
                        // (cells left of beg: only the first lane has any, and never its last cell; their tags do not matter)
                        if (s == 0 && c < CPL - 1) h4 = (lane == 0 && c < sb) ? POA_IDENT : h4;
                        ht4[q][k] = h4;
                        const int r1 = h4 + 4 * e1 * c, r2 = h4 + 4 * e2 * c;
                        a1 = r1 > a1 ? r1 : a1;
                        a2 = r2 > a2 ? r2 : a2;
                        if (c == CPL - 1) { alast1 = r1 + le1; alast2 = r2 + le2; }
                    }
                agg1 = a1 + le1;
                agg2 = a2 + le2;
            }
            const int i1 = poa_wave_scan_max(agg1), i2 = poa_wave_scan_max(agg2);
            const int run1_ = t4_shr1_max(i1, tot1), run2_ = t4_shr1_max(i2, tot2);
            const int la1_ = t4_shr1_mov(alast1, lst1), la2_ = t4_shr1_mov(alast2, lst2);
            if (s == 0 && two) {
                const int a = __builtin_amdgcn_readlane(i1, 63), b = __builtin_amdgcn_readlane(i2, 63);
                tot1 = a; tot2 = b;
                lst1 = __builtin_amdgcn_readlane(alast1, 63);
                lst2 = __builtin_amdgcn_readlane(alast2, 63);
            }
            POA_MARK("t6_p2");
            // ---- phase 2: H'' = max3(Ht, F1, F2), the cell words, gap bytes and direction dwords; the row maximum
            int R1 = run1_ - le1, R2 = run2_ - le2, L1 = la1_ - le1, L2 = la2_ - le2;
            int best = INT32_MIN;
            uint32_t dirs[Q];
            const int lg = lane + 64 * s;
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
                    // clean on the way out
                    const bool off = lg > le || (lg == le && c > se);
                    H[s][q][k] = off ? T4_NEG + 1 : h;
                }
                {
                    const uint32_t ya = (uint32_t)ga + e_probe, yb = ((uint32_t)gbb + e_probe) >> 1;
                    const uint32_t e8 = (ya & 0x80808080u) | (yb & ~0x80808080u);
                    dirq = (int)((e8 & 0xC0C0C0C0u) | ((uint32_t)dirq & ~0xC0C0C0C0u));
                    Ga[s][q] = (uint32_t)ga + g_bias;
                    Gb[s][q] = (uint32_t)gbb + g_bias;
                }
                dirs[q] = (uint32_t)dirq;
                const int m3 = t4_max3(H[s][q][0], H[s][q][1], H[s][q][2]);
                const int m4 = m3 > H[s][q][3] ? m3 : H[s][q][3];
                best = m4 > best ? m4 : best;
            }
            bestv[s] = best;
            POA_MARK("t6_stores");
            // ---- stores: direction dwords, and the value row of a node's last base (what a far successor reads)
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int c = jl + 4 * q - bal;
                if (c >= 0 && c < W) {
                    *(uint32_t *)(drow + (uint32_t)c) = dirs[q];
                    if (last) {
                        *(int4 *)(Vrow + 4u * (uint32_t)c) = make_int4(H[s][q][0], H[s][q][1], H[s][q][2], H[s][q][3]);
                        *(uint2 *)(Vrow + (uint32_t)(4 * W + 2 * c)) = make_uint2(Ga[s][q], Gb[s][q]);
                    }
                }
            }
        };
        step(std::integral_constant<int, 0>{});
        if (__builtin_expect(two, 0)) step(std::integral_constant<int, 1>{});
        b_live = two;
        POA_MARK("t6_rowmax");
        // ---- the row maximum and its leftmost / rightmost column (cells outside the band are far below it)
        {
            const int wb0 = __builtin_amdgcn_readlane(poa_wave_scan_max(bestv[0]), 63);
            int wb = wb0, wb1 = INT32_MIN;
            if (two) { wb1 = __builtin_amdgcn_readlane(poa_wave_scan_max(bestv[1]), 63); wb = wb1 > wb ? wb1 : wb; }
            // one compare per slot gives the holders of the maximum as a lane mask per slot: the first / last holding lane of any slot,
            // then the first / last slot of that lane, are scalar bit tests
            auto first_of = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                uint64_t m[CPL], any = 0;
#pragma unroll
                for (int q = 0; q < Q; q++)
#pragma unroll
                    for (int k = 0; k < 4; k++) { m[4 * q + k] = __builtin_amdgcn_ballot_w64(H[s][q][k] == wb); any |= m[4 * q + k]; }
                const int lf = __builtin_ctzll(any);
                int cf = 0;
#pragma unroll
                for (int c = CPL - 1; c >= 0; c--)
                    if ((m[c] >> lf) & 1ull) cf = c;
                return nbase + CPL * (lf + 64 * s) + cf;
            };
            auto last_of = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                uint64_t m[CPL], any = 0;
#pragma unroll
                for (int q = 0; q < Q; q++)
#pragma unroll
                    for (int k = 0; k < 4; k++) { m[4 * q + k] = __builtin_amdgcn_ballot_w64(H[s][q][k] == wb); any |= m[4 * q + k]; }
                const int lr = 63 - __builtin_clzll(any);
                int cr = 0;
#pragma unroll
                for (int c = 0; c < CPL; c++)
                    if ((m[c] >> lr) & 1ull) cr = c;
                return nbase + CPL * (lr + 64 * s) + cr;
            };
            if (__builtin_expect(!two, 1)) {
                prev_lmax = first_of(std::integral_constant<int, 0>{});
                prev_rmax = last_of(std::integral_constant<int, 0>{});
            } else {
                prev_lmax = wb0 == wb ? first_of(std::integral_constant<int, 0>{}) : first_of(std::integral_constant<int, 1>{});
                prev_rmax = wb1 == wb ? last_of(std::integral_constant<int, 1>{}) : last_of(std::integral_constant<int, 0>{});
            }
            if (last && lane == 0) { R[r].lmax = prev_lmax; R[r].rmax = prev_rmax; }
        }
        POA_MARK("t6_sink");
        if (__builtin_expect(is_sink, 0)) {
            int val = POA_NEG;
            if (qlen >= beg && qlen <= end) {
                const int lq = (qlen - nbase) / CPL, cs = (qlen - nbase) - lq * CPL;
                int w = 0;
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int q = 0; q < Q; q++)
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (cs == 4 * q + k && (lq >> 6) == s) w = __builtin_amdgcn_readlane(H[s][q][k], lq & 63);
                val = w >> 2;
            }
            if (sink_row1 == 0 || val > sink_val) { sink_val = val; sink_row1 = r + 1; }
        }
    }
    }
    // ---- epilogue: the result, the traceback (k_poa_dp_t5's, out of the same records and direction rows), the pool
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    poa_out &O = A.outs[blockIdx.x];
    uint32_t start_row = 0;
    if (failed) { if (status == POA_ST_OK) status = POA_ST_POOL; }
    else {
        start_row = sink_row1 ? sink_row1 - 1 : 0u;
        status = (sink_row1 != 0 && sink_val > POA_NEG / 2) ? POA_ST_OK : POA_ST_NOALN;
    }
    if (lane == 0) {
        O.cells = failed ? 0 : cells; O.vcells = failed ? 0 : vcells; O.maxw = failed ? 0u : (uint32_t)maxw;
        O.score = failed ? POA_NEG : sink_val;
        O.row = start_row;
        O.status = status;
    }
    if (A.tb_ops) poa_traceback_wave<2>(*(tb_lds *)smem, lane, pb, A.rows, preds, nullptr, O, A.tb_ops, A.tb_orow, 0, status, start_row);
    if (lane == 0) {
        O.t_end = __builtin_amdgcn_s_memrealtime();
        if (own_head != POA_NIL) poa_chunk_push(A.cp, blockIdx.x, own_head, own_tail);
        (void)atomicAdd(A.pool_next, (unsigned long long)own_chunks * POA_CHUNK + (uint64_t)ring_size * ring_rows);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        (void)atomicExch(&A.cp.slot_flag[state_slot], 0u);
    }
}
