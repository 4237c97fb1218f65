// vga_poa_t5.hpp -- K4 "t5": k_poa_dp_t4's row phases (same recurrences, same direction bytes, same value rows, bit-exact
// against oracle/og_poa.c) under a LEADERLESS row loop.  k_poa_dp_t4 spends more than half of its issue slots outside the
// interior cell path, most of it per row: one wave (the "leader") reduces the previous row's maximum, runs the bump
// allocator out of a state block in LDS, writes the row record and broadcasts ~12 scalars through LDS that every wave
// then picks up with v_readfirstlane, with a workgroup barrier in between -- ~716 VALU + ~1 780 SALU instructions per row,
// 170 of the VALU being v_readlane / v_writelane spills of the ~200 scalars live across the row loop.
// Here every wave derives the row by itself, in scalar registers:
//   * the node table, the node bases and the predecessor lists come through scalar loads, so band, width, allocation and
//     scoring constants of a row are SALU arithmetic on values every wave already holds -- no leader block, no state block
//     in LDS, no pick-up, and one workgroup barrier per row less;
//   * the only cross-wave input of a row is the previous row's maximum: the waves leave (max, leftmost, rightmost) in
//     LDS at the end of a row (as before) and each wave combines the NW entries itself at the top of the next one;
//   * the bump allocator's state (two chunk cursors, the ring head) is replicated in every wave's scalar registers; only
//     when a chunk runs out (once per MiB of direction bytes) one lane takes the next chunk with an atomic and the
//     workgroup exchanges its offset through LDS -- every wave reaches that branch in the same row;
//   * the row record and the previous row's maximum are stored by lane 0 of the wave whose turn it is (r mod NW);
//   * counters (cells, value cells, widest row) and the best sink row are scalars of every wave.
// Pool handling (classic launch segment or arenas), value-row ring, wide-row scratch, LDS window and the fused traceback
// are those of k_poa_dp_t4, so the two kernels are interchangeable problem by problem (VGA_POA_KERNEL=t4 selects the old one).
#pragma once

// byte B of dst = min(t, c)   (unsigned; c <= 255)
template <int B>
__device__ __forceinline__ void t5_min_byte(int &dst, int t, int c)
{
    if constexpr (B == 0) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(t), "s"(c));
    if constexpr (B == 1) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(t), "s"(c));
    if constexpr (B == 2) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(t), "s"(c));
    if constexpr (B == 3) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(t), "s"(c));
}
__device__ __forceinline__ uint64_t t5_uniform64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

static inline size_t poa_t5_lds_bytes(uint32_t hg_cols, uint32_t lds_cols, int nt)
{
    const int nw = nt / 64;
    return std::max<size_t>(6ull * hg_cols + ((lds_cols / 2 + 15u) & ~15u), sizeof(tb_lds)) + (size_t)(3 * nw + 1 + 1 + 2) * 16 + 16;
}

// The launch arguments as one by-value struct: the row loop copies the few it needs into scalars of their own, and the
// epilogue reads the rest again through the kernarg pointer, so that nothing of it has to stay in registers over the rows.
struct poa_t5_args {
    const poa_prob *probs;
    const char *queries;
    const uint4 *node_tab;
    const uint32_t *seq32, *preds;
    poa_row *rows;
    uint8_t *pool;
    unsigned long long *pool_next;
    uint64_t pool_size;
    poa_out *outs;
    uint8_t *tb_ops;
    uint32_t *tb_orow;
    poa_chunk_pool cp;  // cp.n_slots != 0: direction rows out of the chunk pool, the rest out of a state region
    uint32_t lds_cols, hg_cols, win_mask;
    poa_dev_params P;
    uint32_t prio;  // != 0: the waves of this launch run at raised issue priority (the launch of a call's longest problems: their
                    // sequential rows decide how long the call takes, so they should not share issue slots evenly with the bulk)
};
// a scalar of its own: cuts a uniform value loose from the (wide) load that produced it
__device__ __forceinline__ int t5_own(int v)
{
    asm volatile("" : "+v"(v));  // (through a vector register: the prologue can afford it, and nothing can be folded away)
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ uint32_t t5_own(uint32_t v) { return (uint32_t)t5_own((int)v); }
__device__ __forceinline__ uint64_t t5_own(uint64_t v) { return ((uint64_t)t5_own((uint32_t)(v >> 32)) << 32) | t5_own((uint32_t)v); }
template <typename T>
__device__ __forceinline__ T *t5_own(T *p) { return (T *)t5_own((uint64_t)p); }

// (the read-only inputs stay kernel parameters of their own: only a __restrict__ PARAMETER tells the compiler that no store of
// the kernel can change them, which is what lets their loads be scalar loads)
#define POA_T5_ARGS_OFFSET 40  // the struct's place in the kernarg segment: behind the five pointers
template <int NT, bool DEF>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(5)))  // (at most 96 vector registers: five waves per SIMD)
void k_poa_dp_t5(const poa_prob *__restrict__ probs, const char *__restrict__ queries,
                                                       const uint4 *__restrict__ node_tab, const uint32_t *__restrict__ seq32,
                                                       const uint32_t *__restrict__ preds, const poa_t5_args A)
{
    poa_row *rows = A.rows;
    uint8_t *pool_arg = A.pool;
    unsigned long long *pool_next_arg = A.pool_next;
    const uint64_t pool_size_arg = A.pool_size;
    const uint32_t lds_cols = t5_own(A.lds_cols), hg_cols = t5_own(A.hg_cols), win_mask = t5_own(A.win_mask);
    // Conditions that hold for the whole problem or are carried from row to row live as bits of ONE scalar register, laundered at
    // the top of every row: as `bool`s the compiler keeps each of them as a 64-bit lane mask (two scalar registers, or two
    // spill lanes and two v_readlane per use) for as long as it lives
    enum : uint32_t { ST_CHUNKED = 1u, ST_QPLAIN = 2u, ST_FAILED = 4u, ST_PREV_LDS = 8u, ST_PREV_FAR_USE = 16u };
    uint32_t st = (A.cp.n_slots != 0 ? ST_CHUNKED : 0u) | ST_PREV_LDS;
#define chunked ((st & ST_CHUNKED) != 0)
#define failed ((st & ST_FAILED) != 0)
#define q_plain ((st & ST_QPLAIN) != 0)
#define prev_lds ((st & ST_PREV_LDS) != 0)
#define prev_far_use ((st & ST_PREV_FAR_USE) != 0)
    struct { int match, mismatch, o1, e1, o2, e2, banded; } P = {t5_own(A.P.match), t5_own(A.P.mismatch), A.P.o1, A.P.e1, A.P.o2, A.P.e2, t5_own(A.P.banded)};
    constexpr int NW = NT / 64;
    constexpr int STEP = NT * 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int4 *sX = (int4 *)smem;                  // [2][NW] {scan1, scan2, last1, last2} per wave
    int4 *sRed = sX + 2 * NW;                 // [NW] {row max, -leftmost, rightmost, 0} per wave
    int32_t *edgeW = (int32_t *)(sRed + NW);  // [2] (+2 pad)
    int32_t *sSink = edgeW + 4;               // [0] H of the sink column, [1] arena hand-over (+2 pad)
    uint64_t *sChunk = (uint64_t *)(sSink + 4);  // [4] a new chunk's offset: direction rows / value rows / the ring
    constexpr int HDR = (3 * NW + 1 + 1 + 2) * 16;
    int32_t *Hs = (int32_t *)(smem + HDR);                                // [hg_cols] 4 H
    uint16_t *Gs = (uint16_t *)(smem + HDR + 4ull * hg_cols);             // [hg_cols] G1 | G2 << 8
    uint16_t *Qn = (uint16_t *)(smem + HDR + 6ull * hg_cols);             // [lds_cols / 4] four one-hot column codes per halfword
    const int edge_idx = (int)(edgeW - Hs);

    if (threadIdx.x == 0) A.outs[blockIdx.x].t_begin = __builtin_amdgcn_s_memrealtime();
    if (A.prio) __builtin_amdgcn_s_setprio(3);
    const poa_prob pb = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qlen = t5_own((int)pb.qlen);
    const char *query = queries + pb.q0;
    const uint4 *ntab = node_tab + t5_own(pb.node0);  // (the pointers keep their provenance: scalar loads need it)
    const uint32_t *plist = preds + t5_own(pb.pred0);
    const uint32_t *seqw = seq32 + t5_own(pb.seq0 >> 2);
    poa_row *R = rows + t5_own(pb.row0);
    const uint32_t n_nodes = t5_own(pb.n_nodes), ring_rows = t5_own(pb.ring_rows);

    // ---- pool: classic (bump allocation out of the launch's segment, offsets relative to it) or the chunk pool
    // (vga_poa_kernels.hpp: a state region for what must be contiguous, 1 MiB chunks for the direction rows; offsets are
    // device addresses, pool base 0)
    uint8_t *pool = pool_arg;
    unsigned long long *pool_next = pool_next_arg;
    uint64_t pool_size = pool_size_arg;
    uint32_t state_slot = 0;
    uint64_t state_lo = 0, state_hi = 0;
    if (chunked) {
        int got = -1;
        if (!(pb.flags & 1u)) {
            if (tid == 0) {
                got = poa_slot_acquire(A.cp.slot_flag, A.cp.n_slots, blockIdx.x);
                sSink[1] = got;
            }
            __syncthreads();
            got = __builtin_amdgcn_readfirstlane(sSink[1]);
        }
        if (got < 0) {
            if (tid == 0) {
                poa_out &O = A.outs[blockIdx.x];
                O.t_end = O.t_begin; O.cells = 0; O.vcells = 0; O.maxw = 0; O.nops = 0;
                O.score = POA_NEG; O.row = 0; O.status = POA_ST_POOL;
            }
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        state_slot = (uint32_t)got;
        pool = nullptr;
        state_lo = (uint64_t)A.cp.state_base + (uint64_t)state_slot * A.cp.state_size;
        state_hi = state_lo + A.cp.state_size;
    }
    pool_size = t5_own(pool_size);

    const int o1 = DEF ? 4 : P.o1, e1 = DEF ? 2 : P.e1, o2 = DEF ? 24 : P.o2, e2 = DEF ? 1 : P.e2;
    // Row state of this kernel (its own: k_poa_dp_t4's differs by one): a cell's word is 4 H + 1, its gap bytes are
    // G1 = 4 g1 (in [4 e1, 4 (o1 + e1)]) and G2 = 4 g2 + 1, so that  word - G1 = 4 E1 + 1  and  word - G2 = 4 E2 + 0  arrive
    // tagged as before, and  word - E_k(tagged)  is never negative: the bytes are produced as  min(word - E_k, 4 o_k (+ 1))
    // = G_k - 4 e_k  with one v_min_u32_sdwa each, and the 4 e_k come back with ONE packed add per two cells.  Before that
    // add, "byte == its clamp" (the gap of a successor opens from this cell) is bit 7 of  byte + (128 - clamp): the eight
    // flags of four cells cost two packed adds, a shift and two v_bfi instead of eight compare / add-with-carry pairs.
    const int C1 = 4 * (o1 + e1), C2 = 4 * (o2 + e2) + 1;  // the G bytes of an opened gap
    const int D1 = 4 * o1, D2 = 4 * o2 + 1;                // ... less 4 e_k: what the biased bytes are clamped to
    const uint32_t g_bias = (uint32_t)(4 * e1 | (4 * e2) << 8) * 0x00010001u;
    const uint32_t e_probe = (uint32_t)((128 - D1) | (128 - D2) << 8) * 0x00010001u;
    const int bw = t5_own((int)pb.w);

    // ---- allocator state, identical in every wave (scalar registers).  Bump allocation out of 1 MiB chunks; a request
    // larger than a chunk takes whole chunks of its own.  A new chunk is taken by one lane and handed round through LDS:
    // every wave reaches this branch in the same row (the condition only depends on replicated state), `slot` keeps the
    // two requests of one row apart.
    uint32_t own_head = POA_NIL, own_tail = POA_NIL, own_chunks = 0;  // chunk pool: the chunks this workgroup holds (a list through cp.next)
    // (`rem`: the bytes left of the piece `cur` points into -- a 32-bit count, so that the test of every row is one scalar compare;
    // a 64-bit "cur + bytes > end" goes through the vector ALU.  No request reaches 4 GiB: a row has at most 2^24 + 8 columns of
    // at most 6 bytes, the ring at most POA_RING_SPAN + 1 = 33 such rows, 3.3 GB.)
    auto alloc = [&](uint64_t &cur, uint32_t &rem, uint32_t bytes_asked, int slot) -> uint64_t {
        const uint32_t bytes = (bytes_asked + 15u) & ~15u;
        if (__builtin_expect(bytes > rem, 0)) {
            if (chunked) {
                if (slot == 0 && bytes <= POA_CHUNK) {
                    // direction rows: the next chunk from the free list, chained in front of the ones this workgroup holds
                    if (tid == 0) {
                        const uint32_t idx = poa_chunk_pop(A.cp, blockIdx.x);
                        if (idx != POA_NIL) __hip_atomic_store(A.cp.next + idx, own_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sChunk[0] = idx;
                    }
                    __syncthreads();
                    const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sChunk[0]);
                    __syncthreads();
                    if (idx == POA_NIL) st |= ST_FAILED;
                    else {
                        own_head = idx;
                        if (own_tail == POA_NIL) own_tail = idx;
                        own_chunks++;
                        cur = poa_uniform_u64(poa_chunk_addr(A.cp, idx));
                        rem = (uint32_t)POA_CHUNK;
                    }
                } else
                    st |= ST_FAILED;  // (the state region is exhausted, or a row larger than a chunk: the classic pass takes the problem)
                if (failed) return cur;
            } else {
                const uint64_t need = bytes > POA_CHUNK ? ((uint64_t)bytes + POA_CHUNK - 1) & ~(POA_CHUNK - 1) : POA_CHUNK;
                if (tid == 0) sChunk[slot] = atomicAdd(pool_next, (unsigned long long)need);
                __syncthreads();
                const uint64_t b = t5_uniform64(sChunk[slot]);
                __syncthreads();  // (rare path: the slot may be written again as soon as every wave has read it)
                if (b + need > pool_size) st |= ST_FAILED;
                cur = b;
                rem = (uint32_t)need;
            }
        }
        const uint64_t r = cur;
        cur += bytes;
        rem -= bytes;
        return r;
    };

    // column codes, one-hot: nibble j has bit (ch >> 1) & 3 set for ch = query[j-1] in A/C/T/G (bits 0/1/2/3), 0 for anything else
    // (and for column 0)
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
    if (__builtin_amdgcn_readfirstlane(__syncthreads_or(non_acgt)) == 0) st |= ST_QPLAIN;

    uint64_t dcur = 0, vcur = state_lo, wide_scratch = 0, ring_base = 0;
    uint32_t dend = 0, vendp = (uint32_t)(state_hi - state_lo);  // (bytes left, see alloc)
    uint32_t ring_head = 0;  // the slot the next node-end row takes
    uint32_t ring_size;      // bytes per slot of the ring: one worst-case row
    if (win_mask != 0xFFFFFFFFu) wide_scratch = alloc(vcur, vendp, 2u * 6u * lds_cols, 1);
    {
        const uint64_t maxrow = (6ull * (uint64_t)((qlen + 8) & ~3) + 15ull) & ~15ull;
        ring_size = (uint32_t)maxrow;
        if (chunked) {
            ring_base = alloc(vcur, vendp, (uint32_t)(maxrow * (uint64_t)ring_rows), 1);  // (out of the state region)
        } else {
            const uint64_t rb = (maxrow * (uint64_t)ring_rows + POA_CHUNK - 1) & ~(POA_CHUNK - 1);
            if (tid == 0) sChunk[2] = atomicAdd(pool_next, (unsigned long long)rb);
            __syncthreads();
            const uint64_t b = t5_uniform64(sChunk[2]);
            if (b + rb > pool_size || rb >= (1ull << 32)) st |= ST_FAILED;
            ring_base = b;
        }
    }
    if (tid == 0) { sSink[2] = POA_NEG; sSink[3] = 0; }  // best sink value so far / its row + 1 (0: none yet)
    int prev_beg = 0, prev_end = -1;
    uint32_t seq_word = 0, seq_word_idx = 0xFFFFFFFFu;

    for (uint32_t v = 0; v < n_nodes && !failed; v++) {
    const uint4 nt = ntab[v];
    const uint32_t nlen = nt.y & 0xFFFFFFu;
    for (uint32_t tn = 0; tn < nlen && !failed; tn++) {
        asm volatile("" : "+s"(st));
        POA_MARK("row_topo");
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
        // a SIMPLE row: its only predecessor is the row directly above, and that row's state is in LDS -- every row inside a
        // node and the first row of a node that continues a linear chain (~9 rows in 10 on the HLA graphs).  Such a row
        // needs no predecessor list, no row record of another row and no far-row barrier: its set-up is straight scalar code
        const bool simple = r > 0 && prev_lds && (tn > 0 || ((nt.y >> 24) == 1u && ps == r - 1));
        const bool writer = wv == (int)(r % (uint32_t)NW);  // the wave that stores this row's record
        POA_MARK("row_prevmax");
        // ---- the previous row's maximum: every wave combines the waves' results (the band needs it)
        int prev_lmax = 0, prev_rmax = 0;
        if (r > 0) {
            // lane q < NW holds wave q's (row maximum, leftmost, rightmost column).  Nearly always one wave holds the maximum
            // alone: its entry is read straight from its lane; otherwise leftmost / rightmost are reduced over the tied waves
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
                if (NW > 8) {  // (up to 16 waves: 1 024 threads for the very long problems)
                    t = poa_dpp<0x118, 0xf>(INT32_MAX, lm); lm = t < lm ? t : lm;
                    t = poa_dpp<0x118, 0xf>(INT32_MIN, rm); rm = t > rm ? t : rm;
                }
                prev_lmax = __builtin_amdgcn_readlane(lm, NW - 1);
                prev_rmax = __builtin_amdgcn_readlane(rm, NW - 1);
            }
            // only rows that a later row can name as a far predecessor need it in their record: node ends and the source
            if (prev_far_use && writer && lane == 0) { R[r - 1].lmax = prev_lmax; R[r - 1].rmax = prev_rmax; }
        }
        POA_MARK("row_setup");
        const int remain = (int)(nt.z & 0x3fffffffu) + (int)(nlen - 1 - tn);
        int beg, end, np, pbeg = prev_beg, pend = prev_end;
        bool first, single, sp_near;
        uint32_t sp = r - 1;
        uint64_t vpo = 0;
        if (__builtin_expect(simple, 1)) {
            first = tn == 0;
            np = 1; single = true; sp_near = true;
            const int mpl = prev_lmax + 1, mpr = prev_rmax + 1;
            if (!P.banded) { beg = 0; end = qlen; }
            else {
                const int diag = qlen - remain;
                const int lo = mpl < diag ? mpl : diag;
                const int hi = mpr > diag ? mpr : diag;
                beg = lo - bw; if (beg < 0) beg = 0;
                end = hi + bw; if (end > qlen) end = qlen;
            }
        } else {
            // ---- the general case: the source row, the first row of a node with several or far predecessors, a row below a
            // row that was too wide for the LDS window
            first = tn == 0 && v > 0;
            np = v == 0 ? 0 : (tn == 0 ? (int)(nt.y >> 24) : 1);
            bool far = r > 0 && !prev_lds;
            if (first) {
                if (np == 1) far |= ps != r - 1;
                else
                    for (int t = 0; t < np; t++) far |= plist[ps + t] != r - 1;
            }
            single = r > 0 && np == 1;
            sp = first ? ps : r - 1;
            sp_near = sp == r - 1 && prev_lds;
            // vmcnt(0) + barrier: value rows / row records of far predecessors have landed
            if (far) __syncthreads();
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
            if (!P.banded) { beg = 0; end = qlen; }
            else {
                const int diag = qlen - remain;
                const int lo = mpl < diag ? mpl : diag;
                const int hi = mpr > diag ? mpr : diag;
                beg = lo - bw; if (beg < 0) beg = 0;
                end = hi + bw; if (end > qlen) end = qlen;
            }
            if (single && !sp_near) {
                pbeg = __builtin_amdgcn_readfirstlane(R[sp].beg);
                pend = __builtin_amdgcn_readfirstlane(R[sp].end);
                vpo = t5_uniform64(R[sp].voff);
            }
        }
        const int bal = beg & ~3;
        const int W = (end - bal + 1 + 3) & ~3;
        const bool wide = (uint32_t)W + 8u > hg_cols;
        const bool keep = last || wide;
        const uint64_t doff = alloc(dcur, dend, (uint32_t)W * (np > 1 ? 4u : 1u), 0);
        uint64_t voff = 0;
        if (last && !failed) {
            // (a value row that outlives the ring: chunk-pool mode keeps it among the direction rows, whose chunks stay with the
            // workgroup to the end; the state region only holds the ring and the scratch rows)
            if (r == 0 || (nt.z & 0x40000000u)) voff = chunked ? alloc(dcur, dend, 6u * (uint32_t)W, 0) : alloc(vcur, vendp, 6u * (uint32_t)W, 1);
            else {
                // fixed slots of one worst-case row: the rows of the last ring_rows node ends survive whatever their
                // widths (a byte ring that wraps when a row does not fit can overwrite the row written two slots ago)
                voff = ring_base + (uint64_t)ring_head * ring_size;
                ring_head = ring_head + 1 == ring_rows ? 0 : ring_head + 1;
            }
        } else if (wide) voff = wide_scratch + (r & 1u) * 6ull * lds_cols;
        if (__builtin_expect(failed, 0)) break;
        if (writer && lane == 0) {
            R[r].beg = beg; R[r].end = end;
            R[r].doff = doff; R[r].voff = voff;
            R[r].pred = ps; R[r].npred = first ? (uint32_t)np : 0u;
            R[r].base = keep ? 1 : 0;  // (the counters of the result are summed up from the records at the end)
        }
        uint8_t *Vrow = pool + voff;
        uint8_t *drow = pool + doff;
        // the row's base: a column code (Qn) has bit (ch >> 1) & 3 set for ch = A / C / T / G; anything else scores 0
        const uint32_t gd = gb - (uint32_t)'A';
        const bool acgt = gd < 20u && ((0x80045u >> gd) & 1u);
        const int sc_eq = acgt ? P.match : 0, sc_ne = acgt ? -P.mismatch : 0;
        const int gsh = (int)((gb >> 1) & 3u);
        const int ne4t = 4 * sc_ne + 1, mm4 = 4 * (sc_eq - sc_ne);  // M candidates carry tag 2 (the cell word brings 1 of it)
        // ---- STAGING: a row with predecessors that are not the LDS row (a bubble's arms meeting, the first row behind a
        // bubble, a row below a row that was too wide for the window) first builds ONE virtual predecessor row in LDS out of
        // its predecessors' value rows in HBM, and then runs as a hot row against it.  Per column j over the predecessors
        // whose band holds j:  Hv = max H,  X1 = max (H - G1),  X2 = max (H - G2)  and  Gv = Hv - X  (a valid gap byte: for
        // the predecessor with the largest H, Hv - X <= its G, and X is some predecessor's H - G with that H <= Hv) -- the
        // M / E1 / E2 candidates of the row come out as if every predecessor had been looked at; columns that no
        // predecessor holds read 4 POA_NEG.  Which predecessor won each maximum (first in list order on a tie) is what the
        // traceback needs: three byte planes behind the direction row, as before.  The predecessors' rows are complete
        // (the far-row barrier above has waited for them).
        const bool staged = !simple && r > 0 && !wide && q_plain;
        if (staged) {
            POA_MARK("row_stage");
            // Chunk-pool mode addresses the pool from 0: a predecessor whose record holds no value row -- only a row that is not
            // the last base of a node can be one, and no predecessor list names such a row -- would be read at address 0.  A
            // variant build that sent every row through the single-predecessor paths took `ps` of a several-predecessor node
            // (an index into the predecessor list) for a row and faulted exactly there ("the nomulti fault").  Such a
            // problem is given up instead, and the loads below test the address.
            if (chunked) {
                bool no_row = false;
                for (int t = 0; t < np; t++) no_row |= t5_uniform64(R[np == 1 ? sp : plist[ps + t]].voff) == 0;
                if (__builtin_expect(no_row, 0)) st |= ST_FAILED;
            }
            for (int c0 = 0; c0 < W; c0 += STEP) {
                const int c = c0 + 4 * tid;
                const int j0 = bal + c;
                if (j0 <= end && np == 1) {
                    // one predecessor: its cells as they are, 4 POA_NEG outside its band
                    const int bp = __builtin_amdgcn_readfirstlane(R[sp].beg), ep = __builtin_amdgcn_readfirstlane(R[sp].end);
                    const uint64_t vq_off = t5_uniform64(R[sp].voff);
                    const uint8_t *Vq = pool + vq_off;
                    // (the address is tested through the offset: `pool` is a null pointer in chunk-pool mode, and a compiler may
                    // decide what a comparison of null + offset with null gives)
                    const bool vq_ok = !chunked || vq_off != 0;
                    const int balq = bp & ~3;
                    const int Wq = (ep - balq + 1 + 3) & ~3;
                    const int idx = j0 - balq;
                    int4 hv = make_int4(T4_NEG, T4_NEG, T4_NEG, T4_NEG);
                    uint2 gg = make_uint2(0u, 0u);
                    if (idx >= 0 && idx < Wq && vq_ok) {
                        hv = *(const int4 *)((const int32_t *)Vq + idx);
                        gg = *(const uint2 *)(Vq + 4ll * Wq + 2ll * idx);
                    }
                    const unsigned pspan = (unsigned)(ep - bp);
                    const bool in0 = (unsigned)(j0 - bp) <= pspan, in1 = (unsigned)(j0 + 1 - bp) <= pspan, in2 = (unsigned)(j0 + 2 - bp) <= pspan,
                               in3 = (unsigned)(j0 + 3 - bp) <= pspan;
                    hv.x = in0 ? hv.x : T4_NEG; hv.y = in1 ? hv.y : T4_NEG; hv.z = in2 ? hv.z : T4_NEG; hv.w = in3 ? hv.w : T4_NEG;
                    gg.x = (in0 ? gg.x & 0xffffu : 0u) | (in1 ? gg.x & 0xffff0000u : 0u);
                    gg.y = (in2 ? gg.y & 0xffffu : 0u) | (in3 ? gg.y & 0xffff0000u : 0u);
                    *(int4 *)(Hs + (j0 & win_mask)) = hv;
                    *(uint2 *)(Gs + (j0 & win_mask)) = gg;
                    if (c == 0 && bal > 0) {
                        const int wl = (idx >= 1 && idx - 1 < Wq && vq_ok) ? ((const int32_t *)Vq)[idx - 1] : 0;
                        Hs[(bal - 1) & win_mask] = (unsigned)(j0 - 1 - bp) <= pspan ? wl : T4_NEG;
                    }
                } else if (j0 <= end) {
                    int hm[4], x1[4], x2[4], ah[4], a1[4], a2[4];
                    int hl = T4_NEG, ahl = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) { hm[k] = T4_NEG; x1[k] = T4_NEG; x2[k] = T4_NEG; ah[k] = 0; a1[k] = 0; a2[k] = 0; }
                    for (int t = 0; t < np; t++) {
                        const uint32_t p = np == 1 ? sp : plist[ps + t];
                        const int bp = __builtin_amdgcn_readfirstlane(R[p].beg), ep = __builtin_amdgcn_readfirstlane(R[p].end);
                        const uint64_t vq_off = t5_uniform64(R[p].voff);
                        const uint8_t *Vq = pool + vq_off;
                        const int balq = bp & ~3;
                        const int Wq = (!chunked || vq_off != 0) ? (ep - balq + 1 + 3) & ~3 : 0;
                        const int idx = j0 - balq;
                        int4 hv = make_int4(0, 0, 0, 0);
                        uint2 gg = make_uint2(0u, 0u);
                        if (idx >= 0 && idx < Wq) {
                            hv = *(const int4 *)((const int32_t *)Vq + idx);
                            gg = *(const uint2 *)(Vq + 4ll * Wq + 2ll * idx);
                        }
                        int wl = (idx >= 1 && idx - 1 < Wq) ? ((const int32_t *)Vq)[idx - 1] : 0;
                        const unsigned pspan = (unsigned)(ep - bp);
                        const int hj[4] = {hv.x, hv.y, hv.z, hv.w};
                        const uint32_t g16[4] = {gg.x & 0xffffu, gg.x >> 16, gg.y & 0xffffu, gg.y >> 16};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if ((unsigned)(j0 + k - bp) <= pspan) {
                                const int h = hj[k], c1 = h - (int)(g16[k] & 255u), c2 = h - (int)(g16[k] >> 8);
                                if (h > hm[k]) { hm[k] = h; ah[k] = t; }
                                if (c1 > x1[k]) { x1[k] = c1; a1[k] = t; }
                                if (c2 > x2[k]) { x2[k] = c2; a2[k] = t; }
                            }
                        }
                        if (j0 >= 1 && (unsigned)(j0 - 1 - bp) <= pspan && wl > hl) { hl = wl; ahl = t; }
                    }
                    const int4 hq = make_int4(hm[0], hm[1], hm[2], hm[3]);
                    uint32_t gv[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) gv[k] = (uint32_t)(hm[k] - x1[k]) | ((uint32_t)(hm[k] - x2[k]) << 8);
                    *(int4 *)(Hs + (j0 & win_mask)) = hq;
                    *(uint2 *)(Gs + (j0 & win_mask)) = make_uint2(gv[0] | (gv[1] << 16), gv[2] | (gv[3] << 16));
                    if (c == 0 && bal > 0) Hs[(bal - 1) & win_mask] = hl;
                    if (np > 1) {
                        // predecessor-choice planes: M of column j looks at column j - 1 of the predecessors
                        *(uint32_t *)(drow + (uint64_t)W + c) = (uint32_t)ahl | ((uint32_t)ah[0] << 8) | ((uint32_t)ah[1] << 16) | ((uint32_t)ah[2] << 24);
                        *(uint32_t *)(drow + 2ull * W + c) = (uint32_t)a1[0] | ((uint32_t)a1[1] << 8) | ((uint32_t)a1[2] << 16) | ((uint32_t)a1[3] << 24);
                        *(uint32_t *)(drow + 3ull * W + c) = (uint32_t)a2[0] | ((uint32_t)a2[1] << 8) | ((uint32_t)a2[2] << 16) | ((uint32_t)a2[3] << 24);
                    }
                }
            }
            POA_LDS_BARRIER();
            // the virtual row is defined on every column the row can look at
            pbeg = bal > 0 ? bal - 1 : 0;
            pend = bal + W - 1;
        }
        const uint8_t *Vp = pool + vpo;
        const int balp = pbeg & ~3;

        POA_MARK("row_steps");
        // the row's uniform conditions as bits of ONE scalar register, tested where they are used (as booleans of their own
        // the compiler keeps each of them as a 64-bit lane mask over the whole step loop: ~25 scalar registers, and the
        // spills that follow); the copy taken inside a step cannot be hoisted out of the loop
        enum : uint32_t { F_SINGLE = 1u, F_NEAR = 2u, F_PLAIN = 4u, F_WIDE = 8u, F_KEEP = 16u, F_SINK = 32u, F_MULTI = 64u, F_SRC = 128u };
        const uint32_t rowf = (single ? F_SINGLE : 0u) | (sp_near ? F_NEAR : 0u) | (q_plain ? F_PLAIN : 0u) | (wide ? F_WIDE : 0u) |
                              (keep ? F_KEEP : 0u) | (is_sink ? F_SINK : 0u) | (np > 1 ? F_MULTI : 0u) | (r == 0 ? F_SRC : 0u);
        // uniform per wave, held in vector registers: running maxima of the max-plus scan over the previous steps of this
        // row (carry) and the scan value of the last column of the previous step (left)
        int best = INT32_MIN, lpos = beg, rpos = beg;
        // HOT rows -- a simple row (above) that fits the LDS window, with a query of plain ACGT: more than eight rows in ten --
        // run a copy of the step loop in which every active wave takes the mask-free path (interior or edge-patched) and
        // nothing else exists; the other copy keeps every path.  (Cells that no predecessor cell reaches hold values near
        // 4 POA_NEG; only that they stay far below every real score matters, not their exact value.)
        const bool hot = (simple || staged) && !wide && q_plain;
        const int rlim = end < pend ? end : pend;
        auto run_steps = [&](auto hot_c) __attribute__((always_inline)) {
        constexpr bool HOT = decltype(hot_c)::value;
        int carry1 = POA_IDENT, carry2 = POA_IDENT, left1 = POA_IDENT, left2 = POA_IDENT;
        int next_left = 0;
        int buf = 0;
        for (int c0 = 0; c0 < W; c0 += STEP, buf ^= 1) {
            uint32_t f = HOT ? (rowf & (F_KEEP | F_SINK)) : rowf;
            asm volatile("" : "+s"(f));
            if (HOT) f |= F_SINGLE | F_NEAR | F_PLAIN;
            const int c = c0 + 4 * tid;
            const int j0 = bal + c;
            const bool lane_act = j0 <= end;
            // the mask-free path and its edge patches (lp / rp / lq).  Kept as integers: a boolean made of several compares
            // lives in a 64-bit lane mask, with a select and an AND in front of every branch on it
            const int jw0 = bal + c0 + 256 * wv, jw1 = jw0 + 255;
            const bool wave_act = jw0 <= end;  // (this wave has columns in this step: 256 wv < W - c0)
            int d_lp = jw0 - beg, d_rp = rlim - jw1, d_lq = (jw0 > beg ? jw0 : beg) - pbeg - 1;  // each negative when its patch applies
            asm volatile("" : "+s"(d_lp), "+s"(d_rp), "+s"(d_lq));
            const uint32_t e_lp = (uint32_t)d_lp >> 31;  // jw0 < beg
            const uint32_t e_rp = (uint32_t)d_rp >> 31;  // jw1 > min(end, pend)
            const uint32_t e_lq = (uint32_t)d_lq >> 31;  // max(jw0, beg) <= pbeg
            const bool lp = e_lp != 0, rp = e_rp != 0, lq = e_lq != 0;
            // carried from phase 1 to phase 2, per cell: Ht' (tagged), 4 Ht, E1' (tag 1), E2' (tag 0)
            int htt[4], ht4[4], e1t[4], e2t[4], pmeta[4];
            int agg1 = POA_IDENT, agg2 = POA_IDENT, alast1 = POA_IDENT, alast2 = POA_IDENT;
#pragma unroll
            for (int k = 0; k < 4; k++) { htt[k] = T4_NEG + 2; ht4[k] = T4_NEG; e1t[k] = T4_NEG + 1; e2t[k] = T4_NEG; pmeta[k] = 0; }
            uint32_t qn = 0u;
            if (wave_act) qn = (uint32_t)Qn[j0 >> 2];
            const bool fastw = HOT ? wave_act
                                   : (f & (F_SINGLE | F_PLAIN)) == (F_SINGLE | F_PLAIN) && wave_act && (!lq || (f & F_NEAR)) && (!rp || ((f & F_NEAR) && end <= pend + 1));
            const int base1 = 4 * e1 * j0, base2 = 4 * e2 * j0;  // the scan runs on lane-relative values in the fast path
        if constexpr (HOT) POA_MARK("hot_p1_fast"); else POA_MARK("p1_fast");
            if (__builtin_expect(fastw, 1)) {
                int4 hv;
                uint2 gg;
                int hprev;
                if constexpr (HOT) {
                    // the column left of a lane's four is the last word of the lane below (DPP); lane 0 of a wave takes it
                    // from LDS with a broadcast read: column jw0 - 1, or -- first wave of a later step, whose left neighbour
                    // the last lane overwrote in the step before -- the word the first wave parked then (next_left)
                    hv = *(const int4 *)(Hs + (j0 & win_mask));
                    gg = *(const uint2 *)(Gs + (j0 & win_mask));
                    int left0;
                    if (wv == 0 && c0 > 0) left0 = next_left;
                    else left0 = Hs[(jw0 > 0 ? jw0 - 1 : 0) & (int)win_mask];
                    if (wv == 0 && c0 + STEP < W) next_left = Hs[(jw0 + STEP - 1) & (int)win_mask];
                    hprev = t4_shr1_mov(hv.w, left0);
                } else if (__builtin_expect((f & F_NEAR) != 0, 1)) {
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
                    if constexpr (EDGE && HOT) {
                        // a band that moves right by more than one column: lanes beyond pend + 1 have no cell to their
                        // left in the predecessor either (what the LDS holds there is an older row's)
                        if (rp) hp = j0 - 1 > pend ? T4_NEG : hp;
                    }
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
                if (__builtin_expect((e_lp | e_rp | e_lq) != 0, 0)) phase1(std::true_type{});
                else phase1(std::false_type{});
        if constexpr (HOT) POA_MARK("hot_p1_lean"); else POA_MARK("p1_lean");
            } else if (wave_act && (f & F_SINGLE)) {
                // ---------------- lean path, phase 1 (cold: band limits re-derived here, not hoisted into scalar registers)
                int pbeg_ = pbeg, pend_ = pend, beg_ = beg, end_ = end, balp_ = balp, gsh_ = gsh;
                asm volatile("" : "+s"(pbeg_), "+s"(pend_), "+s"(beg_), "+s"(end_), "+s"(balp_), "+s"(gsh_));
                const int pbeg = pbeg_, pend = pend_, beg = beg_, end = end_, balp = balp_, gsh = gsh_;
                const unsigned span = (unsigned)(end - beg), pspan = (unsigned)(pend - pbeg);
                int hj[4], wm0;
                uint32_t g16[4];
                if (f & F_NEAR) {
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
                    const int m = inprev ? wm + 4 * s + 1 : T4_NEG + 2;
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
        if constexpr (HOT) POA_MARK("hot_p1_general"); else POA_MARK("p1_general");
            } else if (wave_act) {
                // ---------------- general path, phase 1: the source row and rows with several predecessors
                int beg_ = beg, end_ = end, gsh_ = gsh;
                asm volatile("" : "+s"(beg_), "+s"(end_), "+s"(gsh_));
                const int beg = beg_, end = end_, gsh = gsh_;
                const unsigned span = (unsigned)(end - beg);
                if (f & F_SRC) {
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
                                const int cnd = wm + 4 * s + 1;
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
        if constexpr (HOT) POA_MARK("hot_scan"); else POA_MARK("scan");
            int i1 = POA_IDENT, i2 = POA_IDENT;
            if (wave_act) {
                i1 = poa_wave_scan_max(agg1);
                i2 = poa_wave_scan_max(agg2);
            }
            if (lane == 63) sX[buf * NW + wv] = make_int4(i1, i2, alast1, alast2);
            POA_LDS_BARRIER();
        if constexpr (HOT) POA_MARK("hot_exchange"); else POA_MARK("exchange");
            // cross-wave part of the scan: lane q of every wave picks up wave q's entry, a DPP max-scan over those NW lanes
            // gives every wave's inclusive prefix, and a wave reads its own exclusive prefix (and the value left of its first
            // column) from the lane below its index.  The row's running maximum is only needed if another step follows (that
            // step is then full, i.e. every wave is active here).  carry / left are scalars.
            int pre1 = carry1, pre2 = carry2, pl1 = left1, pl2 = left2;
            {
                const bool more = c0 + STEP < W;
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
            }
            if (wave_act) {
                const int run1_ = t4_shr1_max(i1, pre1), run2_ = t4_shr1_max(i2, pre2);
                const int la1_ = t4_shr1_mov(alast1, pl1), la2_ = t4_shr1_mov(alast2, pl2);
        if constexpr (HOT) POA_MARK("hot_p2_fast"); else POA_MARK("p2_fast");
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
                            const int h = (hh & ~3) | 1;  // the cell's word
                            h4[k] = h;
                            // [3:2] tag of H'', [1:0] tag of Ht'; bits 7:6 of the byte below are replaced at the end
                            int acc = (hh << 2) | (htt[k] & 3);
                            const int u1 = h - e1t[k], u2 = h - e2t[k];
                            if (k == 0) { t5_min_byte<0>(ga, u1, D1); t5_min_byte<1>(ga, u2, D2); }
                            if (k == 1) { t5_min_byte<2>(ga, u1, D1); t5_min_byte<3>(ga, u2, D2); }
                            if (k == 2) { t5_min_byte<0>(gbb, u1, D1); t5_min_byte<1>(gbb, u2, D2); }
                            if (k == 3) { t5_min_byte<2>(gbb, u1, D1); t5_min_byte<3>(gbb, u2, D2); }
                            t4_flag_ne(acc, R1, L1);
                            if (k == 0) t4_flag_ne_dep<0>(dirq, acc, R2, L2);
                            if (k == 1) t4_flag_ne_dep<1>(dirq, acc, R2, L2);
                            if (k == 2) t4_flag_ne_dep<2>(dirq, acc, R2, L2);
                            if (k == 3) t4_flag_ne_dep<3>(dirq, acc, R2, L2);
                            L1 = ht4[k] + 4 * e1 * k; L2 = ht4[k] + 4 * e2 * k;
                            R1 = L1 > R1 ? L1 : R1;
                            R2 = L2 > R2 ? L2 : R2;
                        }
                        {
                            // "E_k of a successor opens from this cell": bit 7 of byte + (128 - clamp).  Cells 0 / 1 keep bit 7 of
                            // their two bytes, cells 2 / 3 move to bit 6 (tb_decode2 knows)
                            const uint32_t ya = (uint32_t)ga + e_probe, yb = ((uint32_t)gbb + e_probe) >> 1;
                            const uint32_t e8 = (ya & 0x80808080u) | (yb & ~0x80808080u);
                            dirq = (int)((e8 & 0xC0C0C0C0u) | ((uint32_t)dirq & ~0xC0C0C0C0u));
                            ga = (int)((uint32_t)ga + g_bias);
                            gbb = (int)((uint32_t)gbb + g_bias);
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
                        if (__builtin_expect((f & F_SINK) != 0, 0)) {
                            const int kq = qlen - j0;
#pragma unroll
                            for (int k = 0; k < 4; k++)
                                if (kq == k) sSink[0] = h4[k];
                        }
                        if (!EDGE || j0 <= end) {
                            const int4 hq = make_int4(h4[0], h4[1], h4[2], h4[3]);
                            const uint2 gq = make_uint2((uint32_t)ga, (uint32_t)gbb);
                            if (!(f & F_WIDE)) {
                                *(int4 *)(Hs + (j0 & win_mask)) = hq;
                                *(uint2 *)(Gs + (j0 & win_mask)) = gq;
                            }
                            *(uint32_t *)(drow + c) = (uint32_t)dirq;
                            if (f & F_KEEP) {
                                *(int4 *)((int32_t *)Vrow + c) = hq;
                                *(uint2 *)(Vrow + 4ll * W + 2ll * c) = gq;
                            }
                        }
                    };
                    if (__builtin_expect((e_lp | e_rp) != 0, 0)) phase2(std::true_type{});
                    else phase2(std::false_type{});
        if constexpr (HOT) POA_MARK("hot_p2_slow"); else POA_MARK("p2_slow");
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
                        const int h = (hh & ~3) | 1;  // the cell's word
                        h4[k] = h;
                        const uint32_t t1 = (uint32_t)((h + 4 * e1) - e1t[k]), t2 = (uint32_t)((h + 4 * e2) - e2t[k]);
                        const uint32_t G1 = t1 < (uint32_t)C1 ? t1 : (uint32_t)C1, G2 = t2 < (uint32_t)C2 ? t2 : (uint32_t)C2;
                        g16[k] = G1 | (G2 << 8);
                        // the direction dword of four cells (tb_decode2): byte k = [5:4] tag of H'', [3:2] tag of Ht', [1] / [0] F1 / F2
                        // did not open; bits 7 (cells 0, 1) / 6 (cells 2, 3) of bytes 2 (k & 1), 2 (k & 1) + 1: E1 / E2 open from here
                        codev[k] = (((uint32_t)(hh & 3) << 4) | ((uint32_t)(htt[k] & 3) << 2) | (run1 != la1 ? 2u : 0u) | (run2 != la2 ? 1u : 0u)) << (8 * k);
                        codev[k] |= ((t1 >= (uint32_t)C1 ? 1u : 0u) | (t2 >= (uint32_t)C2 ? 256u : 0u)) << (16 * (k & 1) + (k < 2 ? 7 : 6));
                        const int hb = actk ? h : INT32_MIN;  // (cell words: as in the mask-free path)
                        if (hb > best) { best = hb; lpos = j; rpos = j; }
                        else if (actk && hb == best) rpos = j;
                        const int a1 = actk ? ht4[k] + 4 * e1 * j : POA_IDENT, a2 = actk ? ht4[k] + 4 * e2 * j : POA_IDENT;
                        run1 = a1 > run1 ? a1 : run1;
                        run2 = a2 > run2 ? a2 : run2;
                        la1 = actk ? a1 : la1; la2 = actk ? a2 : la2;
                    }
                    if (__builtin_expect((f & F_SINK) != 0, 0)) {
                        const int kq = qlen - j0;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (kq == k) sSink[0] = h4[k];
                    }
                    const int Wl = (end_ - bal + 1 + 3) & ~3;  // (recomputed here: keeps the plane addresses out of scalar registers)
                    {
                        const int4 hq = make_int4(h4[0], h4[1], h4[2], h4[3]);
                        const uint2 gq = make_uint2(g16[0] | (g16[1] << 16), g16[2] | (g16[3] << 16));
                        if (!(f & F_WIDE)) {
                            *(int4 *)(Hs + (j0 & win_mask)) = hq;
                            *(uint2 *)(Gs + (j0 & win_mask)) = gq;
                        }
                        *(uint32_t *)(drow + c) = codev[0] | codev[1] | codev[2] | codev[3];
                        if (f & F_KEEP) {
                            *(int4 *)((int32_t *)Vrow + c) = hq;
                            *(uint2 *)(Vrow + 4ll * Wl + 2ll * c) = gq;
                        }
                        if (__builtin_expect((f & F_MULTI) != 0, 0)) {
                            *(uint32_t *)(drow + (uint64_t)Wl + c) = (uint32_t)(pmeta[0] & 255) | ((uint32_t)(pmeta[1] & 255) << 8) | ((uint32_t)(pmeta[2] & 255) << 16) | ((uint32_t)(pmeta[3] & 255) << 24);
                            *(uint32_t *)(drow + 2ull * Wl + c) = (uint32_t)((pmeta[0] >> 8) & 255) | ((uint32_t)((pmeta[1] >> 8) & 255) << 8) | ((uint32_t)((pmeta[2] >> 8) & 255) << 16) | ((uint32_t)((pmeta[3] >> 8) & 255) << 24);
                            *(uint32_t *)(drow + 3ull * Wl + c) = (uint32_t)((pmeta[0] >> 16) & 255) | ((uint32_t)((pmeta[1] >> 16) & 255) << 8) | ((uint32_t)((pmeta[2] >> 16) & 255) << 16) | ((uint32_t)((pmeta[3] >> 16) & 255) << 24);
                        }
                    }
                }
            }
        if constexpr (HOT) POA_MARK("hot_step_end"); else POA_MARK("step_end");
        }
        };
        if (__builtin_expect(hot, 1)) run_steps(std::true_type{});
        else run_steps(std::false_type{});
        POA_MARK("row_reduce");
        {
            // the wave's (maximum, leftmost, rightmost column): when a single lane holds the maximum -- nearly always -- that
            // lane stores its own three values; otherwise leftmost / rightmost are reduced over the tied lanes
            const int wb = __builtin_amdgcn_readlane(poa_wave_scan_max(best), 63);
            const bool mine = best == wb;
            const uint64_t tie = __builtin_amdgcn_ballot_w64(mine);
            int *red = (int *)(sRed + wv);
            if (__builtin_expect(__builtin_popcountll(tie) == 1, 1)) {
                if (mine) { red[0] = best; red[1] = lpos; red[2] = rpos; }
            } else {
                int lm = mine ? lpos : INT32_MAX, rm = mine ? rpos : INT32_MIN;
                lm = poa_wave_scan_min(lm);
                rm = poa_wave_scan_max(rm);
                if (lane == 63) { red[0] = wb; red[1] = lm; red[2] = rm; }
            }
        }
        POA_LDS_BARRIER();
        POA_MARK("row_end");
        if (__builtin_expect(is_sink, 0) && tid == 0) {  // (the next sink row's phase 2 is at least one barrier away)
            const int val = (qlen >= beg && qlen <= end) ? sSink[0] >> 2 : POA_NEG;
            if (sSink[3] == 0 || val > sSink[2]) { sSink[2] = val; sSink[3] = (int)r + 1; }
        }
        prev_beg = beg; prev_end = end;
        st = (st & ~(ST_PREV_LDS | ST_PREV_FAR_USE)) | (wide ? 0u : ST_PREV_LDS) | (last ? ST_PREV_FAR_USE : 0u);  // (row 0 is the last row of the source entry)
    }
    }
    __syncthreads();
    if (tid >= 64) return;
    // ---- epilogue, first wave: everything it needs is read again (nothing of it was kept in registers over the rows)
    const poa_t5_args *ap = (const poa_t5_args *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + POA_T5_ARGS_OFFSET);
    asm volatile("" : "+s"(ap));
    const poa_t5_args E = *ap;
    const poa_prob pe = E.probs[blockIdx.x];
    poa_out &O = E.outs[blockIdx.x];
    int status = POA_ST_OK;
    uint32_t start_row = 0;
    int sink_best = POA_NEG;
    if (failed) status = POA_ST_POOL;
    else {
        sink_best = __builtin_amdgcn_readfirstlane(sSink[2]);
        const int sr = __builtin_amdgcn_readfirstlane(sSink[3]);
        start_row = sr ? (uint32_t)(sr - 1) : 0u;
        status = (sr != 0 && sink_best > POA_NEG / 2) ? POA_ST_OK : POA_ST_NOALN;
        // the counters of the result, from the row records (the final barrier's vmcnt(0) has made them visible)
        const poa_row *Rr = E.rows + pe.row0;
        uint64_t cells = 0, vcells = 0;
        int maxw = 0;
        for (uint32_t r = (uint32_t)lane; r <= pe.N; r += 64) {
            const int b = Rr[r].beg, e = Rr[r].end, kp = Rr[r].base;
            const int W = (e - (b & ~3) + 1 + 3) & ~3;
            maxw = W > maxw ? W : maxw;
            if (r > 0) cells += (uint64_t)(e - b + 1);
            if (kp) vcells += (uint64_t)(e - b + 1);
        }
        for (int sh = 32; sh > 0; sh >>= 1) {
            cells += __shfl_down(cells, sh);
            vcells += __shfl_down(vcells, sh);
            const int m2 = __shfl_down(maxw, sh);
            maxw = m2 > maxw ? m2 : maxw;
        }
        if (tid == 0) { O.cells = cells; O.vcells = vcells; O.maxw = (uint32_t)maxw; }
    }
    if (tid == 0) {
        if (failed) { O.cells = 0; O.vcells = 0; O.maxw = 0; }
        O.score = failed ? POA_NEG : sink_best;
        O.row = start_row;
        O.status = status;
    }
    if (E.tb_ops) {
        uint8_t *pool_e = E.cp.n_slots ? nullptr : E.pool;
        poa_traceback_wave<2>(*(tb_lds *)(smem + HDR), tid, pe, E.rows, E.preds, pool_e, O, E.tb_ops, E.tb_orow, 0, status, start_row);
    }
    if (tid == 0) {
        O.t_end = __builtin_amdgcn_s_memrealtime();
        if (E.cp.n_slots) {
            // the chunks go back in one step, then the state region; what this problem took feeds the host's footprint scale
            if (own_head != POA_NIL) poa_chunk_push(E.cp, blockIdx.x, own_head, own_tail);
            (void)atomicAdd(E.pool_next, (unsigned long long)own_chunks * POA_CHUNK + (vcur - state_lo));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            (void)atomicExch(&E.cp.slot_flag[state_slot], 0u);
        }
    }
}
#undef chunked
#undef failed
#undef q_plain
#undef prev_lds
#undef prev_far_use
