// vga_poa_w1.hpp -- K4 "w1": the banded POA DP with ONE WAVE PER PROBLEM and the row state in registers.
//
// Why.  k_poa_dp_t4 / k_poa_dp_pk spread a row (~2 000 band cells at 10 kbp) over four waves: eight cells per thread between
// two synchronisations.  Stamps and PMC counts show what that costs: per row ~5 600 instructions of
// which ~1 400 are DP arithmetic -- the rest is the machinery around it (one wave setting the row up while three wait, LDS
// round trips for the descriptor, the cross-wave scan exchange and the row maximum, ~4.5 barriers) and every wave repeats
// the scalar bookkeeping.  Here a problem is a single wave:
//   * column j lives in lane j & 63 of "block" j >> 6; block kk uses slot kk mod 96 of 96 state slots, each slot a fixed
//     physical VGPR for 4 H (one column per lane) plus half a VGPR for the two G bytes (vga_poa_t4.hpp has the
//     representation: scores scaled by 4, argmax tags, G = 4 g - 1 / 4 g).  A band of up to 96 x 64 = 6 144 columns stays in
//     registers from row to row; as the band moves the blocks simply wrap around the slots.  No LDS for the row state.
//   * a row is a loop over the blocks of its band (~31), each block one pass of phase 1 -> wave scan -> phase 2 over 64
//     cells with a scalar carry to the next block.  No barrier, no cross-wave exchange, no leader / follower split; the
//     row's scalar work (band, pool space, row record) is done once, by the wave itself, from the static row information
//     k_poa_rowprep leaves in the row records.
//   * the state registers are managed by hand: the compiler is confined to v0..v103 (amdgpu_num_vgpr) and never sees
//     v104..v247; a block reads its slot into ordinary registers and writes it back with four v_mov_b32 under
//     s_set_gpr_idx_on (gfx9's VGPR index mode: the register number is offset by the slot).  One copy of the block code
//     serves every slot -- 96 per-slot copies (115 KB of code) thrashed the instruction cache and ran 2.7x slower.
//   * rows whose predecessor is not the row directly above (first rows of nodes behind a bubble, rows with several
//     predecessors) read their predecessors' value rows from HBM -- every such predecessor is the last row of a node and
//     those rows are always kept -- in code shared by all slots.
// 256 VGPRs per wave: two waves per SIMD, eight problems per CU.  A row wider than 6 144 columns ends the problem with
// POA_ST_WIDE and the host runs it with k_poa_dp_t4.  Direction bytes, value rows, row records and the fused traceback are
// those of k_poa_dp_t4 (ENC 1).  Default gap penalties only (the constants are folded into the bodies).
#pragma once

// Static part of every row record, written once per launch before k_poa_dp_w1: predecessor (list), remain, POA_RF_* flags.
// One workgroup per problem, one thread per node.
__global__ __launch_bounds__(256) void k_poa_rowprep(
    const poa_prob *__restrict__ probs, const uint4 *__restrict__ node_tab, const uint32_t *__restrict__ seq32,
    const uint32_t *__restrict__ preds, poa_row *__restrict__ rows)
{
    const poa_prob pb = probs[blockIdx.x];
    const uint4 *ntab = node_tab + pb.node0;
    const uint32_t *plist = preds + pb.pred0;
    const uint8_t *bases = (const uint8_t *)seq32 + pb.seq0;  // row r is byte r - 1
    poa_row *R = rows + pb.row0;
    for (uint32_t v = threadIdx.x; v < pb.n_nodes; v += 256) {
        const uint4 nt = ntab[v];
        const uint32_t nlen = nt.y & 0xFFFFFFu, deg = nt.y >> 24;
        for (uint32_t tn = 0; tn < nlen; tn++) {
            const uint32_t r = nt.x + tn;
            const bool first = tn == 0 && v > 0, last = tn + 1 == nlen;
            const uint32_t np = v == 0 ? 0u : (tn == 0 ? deg : 1u);
            bool far = false;
            if (first) {
                if (np == 1) far = nt.w != r - 1;
                else
                    for (uint32_t t = 0; t < np; t++) far |= plist[nt.w + t] != r - 1;
            }
            const uint8_t gb = v > 0 ? bases[r - 1] : 0;
            const uint32_t gcode = gb == 'A' ? 0u : (gb == 'C' ? 1u : (gb == 'G' ? 2u : (gb == 'T' ? 3u : 4u)));
            const uint32_t fl = (first ? POA_RF_FIRST : 0u) | (last ? POA_RF_LAST : 0u) | ((last && (nt.z >> 31)) ? POA_RF_SINK : 0u) |
                                ((nt.z & 0x40000000u) ? POA_RF_KEEP : 0u) | (far ? POA_RF_FAR : 0u) | (gcode << 8) | (np << 16);
            R[r].pred = nt.w;
            R[r].npred = first ? np : 0u;
            R[r].base = (int32_t)((nt.z & 0x3fffffffu) + (nlen - 1 - tn));
            R[r].hmax = (int32_t)fl;
        }
    }
}

// what a block body sees: the row's uniform values, the block's, the lane's, and what is carried from block to block
struct w1_blk {
    // row (uniform)
    int beg, end, pbeg, pend, bal, W;
    int ne4t, mm4, gsh, C1, C2, np;
    bool general, keep;
    uint8_t *drow, *Vrow;
    // block (uniform)
    bool edge, sink_blk;
    int blk_e1, blk_e2, sink_lane;
    // lane
    int j, lane_e1, lane_e2;
    uint32_t qb;
    int g_htt, g_e1t, g_e2t, g_pm;  // general rows: the candidates of this block's cells, computed by the shared code
    // carried
    int prev_edge, carry1, carry2, last1, last2, sink_val;
    int best, jf, jl;
};

#define W1_SLOTS 96              // 96 x 64 = 6 144 columns
#define W1_FIRST_STATE_VGPR 104  // v104..v199: 4 H per slot; v200..v247: the G byte pairs, slot s in half s / 48 of register s % 48
#define W1_H0 "v104"
#define W1_G0 "v200"
#define W1_LAST_STATE_VGPR "v247"

// state register file, indexed by slot (gfx9 VGPR index mode; M0 holds the index while it is on)
__device__ __forceinline__ int w1_rd_h(int slot)
{
    int r;
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\ts_nop 0\n\tv_mov_b32 %0, " W1_H0 "\n\ts_set_gpr_idx_off" : "=v"(r) : "s"(slot));
    return r;
}
__device__ __forceinline__ int w1_rd_g(int idx)
{
    int r;
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\ts_nop 0\n\tv_mov_b32 %0, " W1_G0 "\n\ts_set_gpr_idx_off" : "=v"(r) : "s"(idx));
    return r;
}
__device__ __forceinline__ void w1_wr_h(int slot, int v)
{
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(DST)\n\ts_nop 0\n\tv_mov_b32 " W1_H0 ", %0\n\ts_set_gpr_idx_off" :: "v"(v), "s"(slot));
}
__device__ __forceinline__ void w1_wr_g(int idx, int v)
{
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(DST)\n\ts_nop 0\n\tv_mov_b32 " W1_G0 ", %0\n\ts_set_gpr_idx_off" :: "v"(v), "s"(idx));
}

// one column block (64 columns, one per lane) of one DP row, for the block's slot
__device__ __forceinline__ void w1_block(w1_blk &c, const int slot)
{
    const int gidx = slot >= W1_SLOTS / 2 ? slot - W1_SLOTS / 2 : slot;
    const bool ghi = slot >= W1_SLOTS / 2;
    const int hj = w1_rd_h(slot);
    const int gpair = w1_rd_g(gidx);
    // the old value of this block's last column is the "column to the left" of the next block's first lane
    const int old63 = __builtin_amdgcn_readlane(hj, 63);
    int htt, e1t, e2t;
    if (__builtin_expect(!c.general, 1)) {
        // ---- phase 1 out of the registers: the row directly above
        const int g16 = (int)((uint32_t)gpair >> (ghi ? 16 : 0));
        int hp = t4_shr1_mov(hj, c.prev_edge);
        e1t = t4_sub_byte<0>(hj, g16);
        e2t = t4_sub_byte<1>(hj, g16);
        const uint32_t eq = __builtin_amdgcn_ubfe(c.qb, (uint32_t)c.gsh, 1u);
        int m = (int)__umul24(eq, (uint32_t)c.mm4) + (hp + c.ne4t);
        if (__builtin_expect(c.edge, 0)) {
            // cells whose predecessor cell lies outside the predecessor's band read it as minus infinity
            const bool inj = (unsigned)(c.j - c.pbeg) <= (unsigned)(c.pend - c.pbeg);
            const bool inprev = c.j >= 1 && (unsigned)(c.j - 1 - c.pbeg) <= (unsigned)(c.pend - c.pbeg);
            e1t = inj ? e1t : T4_NEG + 1;
            e2t = inj ? e2t : T4_NEG;
            m = inprev ? m : T4_NEG + 2;
        }
        htt = t4_max3(m, e1t, e2t);
    } else {
        htt = c.g_htt; e1t = c.g_e1t; e2t = c.g_e2t;
    }
    const int ht4 = htt & ~3;
    // ---- the insertion recurrence: max-plus prefix scan over the row, block by block (carry = the running maximum so far)
    int a1 = ht4 + c.lane_e1 + c.blk_e1, a2 = ht4 + c.lane_e2 + c.blk_e2;
    bool act = true;
    if (__builtin_expect(c.edge, 0)) {
        act = (unsigned)(c.j - c.beg) <= (unsigned)(c.end - c.beg);
        a1 = act ? a1 : POA_IDENT;
        a2 = act ? a2 : POA_IDENT;
    }
    const int i1 = poa_wave_scan_max(a1), i2 = poa_wave_scan_max(a2);
    int cv1 = c.carry1, cv2 = c.carry2, lv1 = c.last1, lv2 = c.last2;  // uniform values in vector registers
    asm volatile("" : "+v"(cv1), "+v"(cv2), "+v"(lv1), "+v"(lv2));
    const int R1 = t4_shr1_max(i1, cv1), R2 = t4_shr1_max(i2, cv2);   // max over the columns left of j (incl. earlier blocks)
    const int L1 = t4_shr1_mov(a1, lv1), L2 = t4_shr1_mov(a2, lv2);   // the value of column j - 1
    // ---- phase 2
    const int f1 = R1 - (c.lane_e1 + c.blk_e1) - (4 * 4 - 1), f2 = R2 - (c.lane_e2 + c.blk_e2) - 4 * 24;  // - 4 (o + e j) (+ tag 1 / 0)
    const int hh = t4_max3(ht4 | 3, f1, f2);
    const int h4 = hh & ~3;
    int acc = ((hh << 2) & ~3) | (htt & 3);
    int gnew = 0;
    {
        const int t1 = (h4 + 4 * 2) - e1t, t2 = (h4 + 4 * 1) - e2t;
        t4_gap_byte<0>(gnew, acc, t1, c.C1);
        t4_gap_byte<1>(gnew, acc, t2, c.C2);
    }
    t4_flag_ne(acc, R1, L1);
    t4_flag_ne(acc, R2, L2);
    // write the slot back: H, and this slot's half of the G pair
    w1_wr_h(slot, h4);
    w1_wr_g(gidx, (int)__builtin_amdgcn_perm((uint32_t)gnew, (uint32_t)gpair, ghi ? 0x05040100u : 0x03020504u));
    // ---- row maximum: per lane the best value and the first / last column that reached it (columns ascend with the blocks)
    {
        const int hb = act ? h4 : INT32_MIN;
        const bool gt = hb > c.best, ge = hb >= c.best;
        c.jf = gt ? c.j : c.jf;
        c.jl = ge ? c.j : c.jl;
        c.best = gt ? hb : c.best;
    }
    // ---- stores: direction byte (+ the three predecessor-choice bytes of a row with several predecessors), value row
    {
        const uint32_t jrel = (uint32_t)(c.j - c.bal);
        if (__builtin_expect(!c.edge, 1) || act) {
            c.drow[jrel] = (uint8_t)acc;
            if (c.keep) {
                ((int32_t *)c.Vrow)[jrel] = h4;
                ((uint16_t *)(c.Vrow + 4ll * c.W))[jrel] = (uint16_t)gnew;
            }
            if (__builtin_expect(c.np > 1, 0)) {
                c.drow[(uint64_t)c.W + jrel] = (uint8_t)c.g_pm;
                c.drow[2ull * c.W + jrel] = (uint8_t)(c.g_pm >> 8);
                c.drow[3ull * c.W + jrel] = (uint8_t)(c.g_pm >> 16);
            }
        }
    }
    if (__builtin_expect(c.sink_blk, 0)) c.sink_val = __builtin_amdgcn_readlane(h4, c.sink_lane);
    // ---- hand over to the next block
    c.prev_edge = old63;
    {
        const int t1 = __builtin_amdgcn_readlane(i1, 63), t2 = __builtin_amdgcn_readlane(i2, 63);
        c.carry1 = t1 > c.carry1 ? t1 : c.carry1;
        c.carry2 = t2 > c.carry2 ? t2 : c.carry2;
        c.last1 = __builtin_amdgcn_readlane(a1, 63);
        c.last2 = __builtin_amdgcn_readlane(a2, 63);
    }
}

static inline size_t poa_w1_lds_bytes(uint32_t max_q) { return std::max<size_t>((size_t)max_q + 1 + 192, sizeof(tb_lds)) + 64; }

__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(W1_FIRST_STATE_VGPR))) void k_poa_dp_w1(
    const poa_prob *__restrict__ probs, const char *__restrict__ queries, const uint32_t *__restrict__ preds, poa_dev_params P,
    poa_row *rows, uint8_t *pool_arg, unsigned long long *pool_next_arg, uint64_t pool_size_arg, poa_out *__restrict__ outs,
    uint8_t *__restrict__ tb_ops, uint32_t *__restrict__ tb_orow, uint32_t n_arenas, uint64_t arena_size,
    unsigned long long *arena_ctr, uint32_t *arena_flag, const uint4 *__restrict__ rinfo_all)
{
    constexpr int C = W1_SLOTS;
    asm volatile("" ::: W1_LAST_STATE_VGPR);  // the kernel's register count must cover the hand-managed state registers
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *Qs = smem;  // [qlen + 1 + 192] one-hot code of query[j - 1] at index j (0 for j = 0, non-ACGT and beyond the query)

    const uint64_t t_begin = __builtin_amdgcn_s_memrealtime();
    const poa_prob pb = probs[blockIdx.x];
    const int lane = threadIdx.x;
    const int qlen = (int)pb.qlen;
    const char *query = queries + pb.q0;
    const uint32_t *plist = preds + pb.pred0;
    poa_row *R = rows + pb.row0;
    const uint4 *__restrict__ rinfo = rinfo_all + 3ull * pb.row0 + 2;  // static quad of record r: uint4 number 3 r + 2

    // ---- pool: classic (chunks of the launch's segment) or arena mode, as in k_poa_dp_t4
    uint8_t *pool = pool_arg;
    unsigned long long *pool_next = pool_next_arg;
    uint64_t pool_size = pool_size_arg;
    uint32_t arena = 0;
    if (n_arenas) {
        int got = -1;
        if (!(pb.flags & 1u)) {
            if (lane == 0) {
                uint32_t a = (uint32_t)(((uint64_t)blockIdx.x * 2654435761ull) % n_arenas);
                for (uint32_t tries = 0; tries < (1u << 24); tries++) {
                    if (atomicCAS(&arena_flag[a], 0u, 1u) == 0u) { got = (int)a; break; }
                    a = a + 1 == n_arenas ? 0 : a + 1;
                    if ((tries & 15u) == 15u) __builtin_amdgcn_s_sleep(64);
                }
                if (got >= 0) (void)atomicExch(&arena_ctr[got], 0ull);
            }
            got = __builtin_amdgcn_readfirstlane(got);
        }
        if (got < 0) {
            if (lane == 0) {
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
    auto rfl64 = [](uint64_t x) -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    };
    bool failed = false, too_wide = false;
    auto take = [&](uint64_t bytes) -> uint64_t {  // whole chunks from the pool
        unsigned long long bv = 0;
        if (lane == 0) bv = atomicAdd(pool_next, (unsigned long long)bytes);
        const uint64_t b = rfl64(bv);
        if (b + bytes > pool_size) failed = true;
        return b;
    };
    auto bump = [&](uint64_t &cur, uint64_t &end, uint64_t bytes) -> uint64_t {
        bytes = (bytes + 15ull) & ~15ull;
        if (cur + bytes > end) {
            const uint64_t need = bytes > POA_CHUNK ? (bytes + POA_CHUNK - 1) & ~(POA_CHUNK - 1) : POA_CHUNK;
            cur = take(need);
            end = cur + need;
        }
        const uint64_t r = cur;
        cur += bytes;
        return r;
    };

    // column codes, one-hot per column: 1/2/4/8 for query[j-1] = A/C/G/T, 0 otherwise
    int non_acgt = 0;
    for (int j = lane; j < qlen + 1 + 192; j += 64) {
        uint32_t code = 0;
        if (j >= 1 && j <= qlen) {
            const char ch = query[j - 1];
            code = ch == 'A' ? 1u : (ch == 'C' ? 2u : (ch == 'G' ? 4u : (ch == 'T' ? 8u : 0u)));
            non_acgt |= code == 0;
        }
        Qs[j] = (uint8_t)code;
    }
    const bool q_plain = __builtin_amdgcn_ballot_w64(non_acgt != 0) == 0ull;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the codes are in LDS (one wave: no barrier needed)
    __builtin_amdgcn_wave_barrier();

    // allocators: direction rows, kept value rows (source row, rows read far ahead), the ring of node-end value rows
    uint64_t dcur = 0, dend = 0, vcur = 0, vend = 0;
    const uint64_t maxrow = (6ull * (uint64_t)((qlen + 8) & ~3) + 15ull) & ~15ull;
    const uint64_t ring_bytes = (maxrow * (uint64_t)pb.ring_rows + POA_CHUNK - 1) & ~(POA_CHUNK - 1);
    const uint64_t ring_base = take(ring_bytes);
    if (ring_bytes >= (1ull << 32)) failed = true;
    uint32_t ring_head = 0;  // the slot of the ring the next node-end row takes (a slot = one worst-case row)

    int sink_best = POA_NEG, sink_have = 0;
    uint32_t sink_row = 0;
    int prev_beg = 0, prev_end = -1, prev_lmax = 0, prev_rmax = 0;
    bool prev_last = false;
    uint4 linfo = rinfo[0];
    // the mask-free path scores every query base as match or mismatch: a query with other characters goes to k_poa_dp_t4
    if (!q_plain) too_wide = true;

    w1_blk c;
    c.C1 = 4 * (4 + 2) - 1;
    c.C2 = 4 * (24 + 1);
    c.lane_e1 = 4 * 2 * lane;
    c.lane_e2 = 4 * 1 * lane;
    c.sink_val = 0;

#ifdef W1_DEBUG_TICKS
    unsigned long long dbg_gen = 0, dbg_fast = 0, dbg_ngen = 0, dbg_setup = 0;
#endif
    for (uint32_t r = 0; r <= pb.N && !failed && !too_wide; r++) {
        POA_MARK("w1_row");
#ifdef W1_DEBUG_TICKS
        unsigned long long dbg_t0;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dbg_t0)::"memory");
#endif
        const uint4 info = linfo;
        if (r < pb.N) linfo = rinfo[3ull * (r + 1)];  // (requested a row ahead)
        const uint32_t sfl = info.w, ps = info.x;
        const bool first = (sfl & POA_RF_FIRST) != 0, last = (sfl & POA_RF_LAST) != 0, is_sink = (sfl & POA_RF_SINK) != 0;
        const int np = (int)((sfl >> 16) & 255u);
        const int gcode = (int)((sfl >> 8) & 7u);
        // only the last row of a node can be a far predecessor: the others' maxima are not kept
        if (r > 0 && prev_last && lane == 0) *(int2 *)&R[r - 1].lmax = make_int2(prev_lmax, prev_rmax);
        const bool general = r == 0 || np != 1 || (sfl & POA_RF_FAR) != 0;
        if (general && r > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own value rows / row records have landed
        // ---- band
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
            const int diag = qlen - (int)info.z;
            const int lo = mpl < diag ? mpl : diag;
            const int hi = mpr > diag ? mpr : diag;
            beg = lo - (int)pb.w; if (beg < 0) beg = 0;
            end = hi + (int)pb.w; if (end > qlen) end = qlen;
        }
        const int bal = beg & ~3;
        const int W = (end - bal + 1 + 3) & ~3;
        const int kb = beg >> 6, ke = end >> 6;
        if (ke - kb + 1 > C) { too_wide = true; break; }
        // ---- pool space, row record
        const uint64_t doff = bump(dcur, dend, (uint64_t)W * (np > 1 ? 4u : 1u));
        uint64_t voff = 0;
        if (last) {
            if (r == 0 || (sfl & POA_RF_KEEP)) voff = bump(vcur, vend, 6ull * (uint64_t)W);
            else {
                voff = ring_base + (uint64_t)ring_head * maxrow;  // fixed slots (see k_poa_dp_t4)
                ring_head = ring_head + 1 == pb.ring_rows ? 0 : ring_head + 1;
            }
        }
        if (failed) break;
        if (lane == 0) {
            *(int4 *)&R[r].beg = make_int4(beg, end, (int)(uint32_t)doff, (int)(uint32_t)(doff >> 32));
            if (last) R[r].voff = voff;
        }
        // ---- the row's constants
        const int sc_eq = gcode == 4 ? 0 : P.match, sc_ne = gcode == 4 ? 0 : -P.mismatch;
        c.beg = beg; c.end = end; c.bal = bal; c.W = W;
        c.pbeg = prev_beg; c.pend = prev_end;
        c.ne4t = 4 * sc_ne + 2; c.mm4 = 4 * (sc_eq - sc_ne); c.gsh = gcode & 3;
        c.np = np;
        c.general = general;
        c.keep = last;
        c.drow = pool + doff;
        c.Vrow = pool + voff;
        c.carry1 = POA_IDENT; c.carry2 = POA_IDENT; c.last1 = POA_IDENT; c.last2 = POA_IDENT;
        c.best = INT32_MIN; c.jf = beg; c.jl = beg;
        c.prev_edge = T4_NEG;
        if (!c.general && (kb << 6) - 1 >= prev_beg && (kb << 6) - 1 <= prev_end) c.prev_edge = __builtin_amdgcn_readlane(w1_rd_h((kb + C - 1) % C), 63);
        const int sink_kk = is_sink && qlen >= beg && qlen <= end ? (qlen >> 6) : -1;
        c.sink_lane = qlen & 63;
#ifdef W1_DEBUG_TICKS
        unsigned long long dbg_t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dbg_t1)::"memory");
        dbg_setup += dbg_t1 - dbg_t0;
#endif
        uint32_t qb_next = Qs[(kb << 6) + lane];
        for (int kk = kb; kk <= ke; kk++) {
            POA_MARK("w1_block");
            const int slot = kk % C;
            c.j = (kk << 6) + lane;
            c.qb = qb_next;
            qb_next = Qs[((kk + 1) << 6) + lane];
            c.blk_e1 = 4 * 2 * (kk << 6);
            c.blk_e2 = 4 * 1 * (kk << 6);
            c.edge = c.general || (kk << 6) < beg || (kk << 6) + 63 > end || (kk << 6) - 1 < prev_beg || (kk << 6) + 63 > prev_end;
            c.sink_blk = kk == sink_kk;
            if (__builtin_expect(c.general, 0)) {
                // ---- phase 1 of a general row, shared by all slots: candidates from the predecessors' value rows in HBM
                const int j = c.j;
                const bool actk = (unsigned)(j - beg) <= (unsigned)(end - beg);
                int m = T4_NEG + 2, ev1 = T4_NEG + 1, ev2 = T4_NEG, pm = 0;
                if (r == 0) m = (j == 0 ? 0 : T4_NEG) + 2;
                else {
                    const int qc = (int)(c.qb & 15u);
                    const int s = ((qc >> c.gsh) & 1) ? sc_eq : (qc == 0 ? 0 : sc_ne);
                    for (int t = 0; t < np; t++) {
                        const uint32_t p = np == 1 ? ps : plist[ps + t];
                        const int bp = __builtin_amdgcn_readfirstlane(R[p].beg), ep = __builtin_amdgcn_readfirstlane(R[p].end);
                        const uint8_t *Vq = pool + rfl64(R[p].voff);
                        const int balq = bp & ~3;
                        const int Wq = (ep - balq + 1 + 3) & ~3;
                        const int idx = j - balq;
                        if (actk && j >= 1 && (unsigned)(j - 1 - bp) <= (unsigned)(ep - bp)) {
                            const int cnd = ((const int32_t *)Vq)[idx - 1] + 4 * s + 2;
                            if (cnd > m) { m = cnd; pm = (pm & ~255) | t; }
                        }
                        if (actk && (unsigned)(j - bp) <= (unsigned)(ep - bp)) {
                            const int hj = ((const int32_t *)Vq)[idx];
                            const uint32_t g16 = ((const uint16_t *)(Vq + 4ll * Wq))[idx];
                            const int c1 = hj - (int)(g16 & 255u);
                            if (c1 > ev1) { ev1 = c1; pm = (pm & ~0xff00) | (t << 8); }
                            const int c2 = hj - (int)(g16 >> 8);
                            if (c2 > ev2) { ev2 = c2; pm = (pm & ~0xff0000) | (t << 16); }
                        }
                    }
                }
                c.g_htt = t4_max3(m, ev1, ev2);
                c.g_e1t = ev1;
                c.g_e2t = ev2;
                c.g_pm = pm;
            }
            w1_block(c, slot);
        }
        if (too_wide) break;
#ifdef W1_DEBUG_TICKS
        {
            unsigned long long dbg_t2;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dbg_t2)::"memory");
            if (general) { dbg_gen += dbg_t2 - dbg_t1; dbg_ngen++; } else dbg_fast += dbg_t2 - dbg_t1;
        }
#endif
        // ---- row maximum: leftmost / rightmost column
        {
            int wb = poa_wave_scan_max(c.best);
            wb = __builtin_amdgcn_readlane(wb, 63);
            int lm = c.best == wb ? -c.jf : INT32_MIN;
            int rm = c.best == wb ? c.jl : INT32_MIN;
            lm = poa_wave_scan_max(lm);
            rm = poa_wave_scan_max(rm);
            prev_lmax = -__builtin_amdgcn_readlane(lm, 63);
            prev_rmax = __builtin_amdgcn_readlane(rm, 63);
        }
        if (__builtin_expect(is_sink, 0)) {
            // the sink takes the first largest H[qlen] among its predecessors (rows come in the order of the sink list)
            const int val = sink_kk >= 0 ? (c.sink_val >> 2) : POA_NEG;
            if (!sink_have || val > sink_best) { sink_best = val; sink_row = r; sink_have = 1; }
        }
        prev_beg = beg; prev_end = end;
        prev_last = last;
    }
    // ---- the result record, then the traceback of this problem out of LDS (the direction rows were written by this wave)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int status = POA_ST_OK;
    uint32_t start_row = 0;
    {
        poa_out &O = outs[blockIdx.x];
        if (too_wide) status = POA_ST_WIDE;
        else if (failed) status = POA_ST_POOL;
        else {
            start_row = sink_row;
            status = (sink_have != 0 && sink_best > POA_NEG / 2) ? POA_ST_OK : POA_ST_NOALN;
        }
        // statistics of the problem (band cells, cells of the kept value rows, widest row), from the row records
        uint64_t cells = 0, vcells = 0;
        uint32_t maxw = 0;
        if (status == POA_ST_OK || status == POA_ST_NOALN) {
            for (uint32_t rr = (uint32_t)lane; rr <= pb.N; rr += 64) {
                const int2 be = *(const int2 *)&R[rr].beg;
                const uint32_t fl = (uint32_t)R[rr].hmax;
                const uint32_t wd = (uint32_t)(be.y - be.x + 1);
                const uint32_t Wr = (uint32_t)((be.y - (be.x & ~3) + 1 + 3) & ~3);
                if (rr > 0) cells += wd;
                if (fl & POA_RF_LAST) vcells += wd;
                maxw = Wr > maxw ? Wr : maxw;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                cells += ((uint64_t)(uint32_t)__shfl_xor((int)(cells >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)cells, o);
                vcells += ((uint64_t)(uint32_t)__shfl_xor((int)(vcells >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)vcells, o);
                const uint32_t om = (uint32_t)__shfl_xor((int)maxw, o);
                maxw = om > maxw ? om : maxw;
            }
        }
#ifdef W1_DEBUG_TICKS
        cells = dbg_gen; vcells = dbg_fast; maxw = (uint32_t)dbg_ngen; (void)dbg_setup;
        if (lane == 0) O.nops = (uint32_t)(dbg_setup >> 10);
#endif
        if (lane == 0) {
            O.t_begin = t_begin;
            O.cells = cells;
            O.vcells = vcells;
            O.maxw = maxw;
            O.score = (status == POA_ST_OK || status == POA_ST_NOALN) ? sink_best : POA_NEG;
            O.row = start_row;
            O.status = status;
        }
    }
    if (tb_ops) {
        status = __builtin_amdgcn_readfirstlane(status);
        start_row = (uint32_t)__builtin_amdgcn_readfirstlane((int)start_row);
        poa_traceback_wave<1>(*(tb_lds *)smem, lane, pb, rows, preds, pool, outs[blockIdx.x], tb_ops, tb_orow, 0, status, start_row);
    }
    if (lane == 0) {
        outs[blockIdx.x].t_end = __builtin_amdgcn_s_memrealtime();
        if (n_arenas) {
            const unsigned long long used = atomicAdd(pool_next, 0ull);
            (void)atomicAdd(pool_next_arg, used < arena_size ? used : (unsigned long long)arena_size);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            (void)atomicExch(&arena_flag[arena], 0u);
        }
    }
}
