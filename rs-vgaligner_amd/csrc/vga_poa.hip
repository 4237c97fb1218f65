// vga_poa.hip -- banded partial-order alignment on gfx950: the replacement for the reference's only
// native component, abPOA behind  AbpoaAligner::create_align_safe(nodes, edges, query, Global)
// (src/align.rs:173-203; result fields consumed at src/align.rs:1107,1152-1165).
//
//   K4  k_poa_dp_lds<NT,4>  one workgroup per (read, subgraph) problem.  Rows = graph bases in topological
//                        order, processed one after the other because the adaptive band of a row depends on
//                        where its predecessors' maxima fell (that dependency is also why an anti-diagonal
//                        wavefront cannot be used: a row's band is unknown until the rows above are complete).
//                        Lanes run across the band's columns, four adjacent columns per lane.
//   K4b k_poa_traceback  one lane per problem walking the 1-byte direction codes.
//
// HBM layout per problem (all carved from one persistent pool by a device-side bump allocator, 1 MiB chunks;
// every row is stored from the 4-aligned column  bal = beg & ~3  with a width rounded up to 4):
//   direction row  : 1 byte per cell (+3 predecessor-choice bytes per cell on the rare rows with more than one
//                    predecessor) -- the only per-cell data kept for the traceback;
//   value row      : int32 H + two 1-byte clamped gap deltas per cell, kept only for the LAST base of every graph
//                    node (the only rows a later, non-adjacent row can depend on) and the source row;
//   row arrays     : 40 B per row (beg, end, direction offset, value offset, leftmost/rightmost max column,
//                    predecessor row or predecessor-list slice).
// The deltas: a successor only ever needs max(H - (O+E), Ek - E); storing d = min(H - Ek, O) keeps exactly that
// quantity (H - E - d) and the open/extend decision (d == O) in one byte.
//
// Host -> device description of a problem is per NODE (16 B each) plus the node sequences; rows are generated
// on the device.  Numerics are 32-bit integer and bit-exact against oracle/og_poa.c (see its header for the
// specification: recurrences, tie order, band rule).
#include "vga_common.hpp"
#include "vga_poa_internal.hpp"

#include <algorithm>
#include <cstddef>
#include <type_traits>
#include <chrono>
#include <thread>

#define POA_NEG (-(1 << 21))  // "minus infinity"; H stays inside 23 signed bits (see k_poa_dp_pk)
#define POA_IDENT (INT32_MIN / 2)
#define POA_CHUNK (1ull << 20)
#define POA_RING_SPAN 32  // value rows read within this many nodes live in the per-problem ring, the others are kept
#define POA_SLOTS 4  // most sub-batches in flight (VGA_POA_SLOTS; default 2): own stream, pool segment and staging buffers each

#define POA_ST_OK 0
#define POA_ST_POOL 1
#define POA_ST_NOALN 2
#define POA_ST_TRACE 3
#define POA_ST_RANGE 4  // 16-bit storage: a score came near the representable range, the problem is re-run in 32 bits
#define POA_ST_WIDE 5   // k_poa_dp_w1: a row wider than its register file (or a query with non-ACGT characters): re-run with k_poa_dp_t4

struct poa_prob {
    uint64_t node0;  // first entry of the node table (entry 0 of a problem is the virtual source)
    uint64_t pred0;  // first entry of the predecessor list (row ids)
    uint64_t sink0;  // first entry of the sink predecessor list
    uint64_t q0;     // first query byte
    uint64_t ops0;   // first entry of the traceback output
    uint64_t row0;   // first entry of the per-row arrays (rows 0..N)
    uint64_t seq0;   // first byte of the node sequences (row r is byte r-1)
    uint32_t n_sink;
    uint32_t qlen;
    uint32_t N;
    uint32_t w;      // adaptive band half-width: wb + floor(wf * qlen), computed on the host in double
    uint32_t n_nodes;  // node-table entries incl. the source
    uint32_t ring_rows;  // value rows of node-end rows live in a ring of this many worst-case rows (k_poa_dp_pk)
    uint32_t flags;      // bit 0: too large for an arena (k_poa_dp_pk in arena mode reports POA_ST_POOL at once)
    uint32_t pad;
};

struct poa_row {          // per DP row, 48 B
    int32_t beg, end;     // band
    uint64_t doff, voff;  // direction row / value row in the pool
    int32_t lmax, rmax;   // leftmost / rightmost column of the row maximum
    // the last four words form one aligned 16-byte group: k_poa_rowprep fills them for k_poa_dp_w1, which reads them with a
    // single scalar load per row
    uint32_t pred, npred; // predecessor row or predecessor-list slice; npred != 0 only on the first row of a node
    int32_t base, hmax;   // k_poa_dp_pk<.., H16>: the row's values are stored relative to `base`, hmax = the row maximum.
                          // k_poa_dp_w1: base = graph bases after this row on the longest path to the sink ("remain"),
                          // hmax = static flags of the row (POA_RF_*), both written by k_poa_rowprep
};
static_assert(sizeof(poa_row) == 48 && offsetof(poa_row, pred) == 32, "poa_row layout");
#define POA_RF_FIRST 1u    // first base of a node (other than the source)
#define POA_RF_LAST 2u     // last base of a node
#define POA_RF_SINK 4u     // ... of a node without successors: the row feeds the sink
#define POA_RF_KEEP 8u     // its value row is read more than POA_RING_SPAN nodes ahead: kept outside the ring
#define POA_RF_FAR 16u     // a predecessor is not the row directly above
                           // bits 8..10: code of the row's base (A C G T other), bits 16..23: number of predecessors

struct poa_out {          // per problem, 56 B
    int32_t score;
    uint32_t row;         // sink predecessor the traceback starts from
    int32_t status;
    uint32_t maxw;        // widest row (storage columns)
    uint64_t cells, vcells;
    uint32_t nops, pad;
    uint64_t t_begin, t_end;  // s_memrealtime (100 MHz) when the DP workgroup started / finished: occupancy diagnostics
};

struct poa_dev_params {
    int32_t match, mismatch, o1, e1, o2, e2, banded;
};

// ---------------------------------------------------------------------------------------------------------
// K4.  * The row that was just filled stays in LDS, indexed by ABSOLUTE query column and overwritten in place
//        by the next row:  Hs[j] int32 (H), Ds[j] uint16 (d1 | d2 << 8); the common predecessor (the row directly
//        above) costs three vector LDS reads per four cells instead of an L2 round trip.
//      * Every lane owns four adjacent, 4-aligned columns: the insertion recurrence
//        Fk[j] = max_{j'<j} Ht[j'] - Ok - Ek (j - j')  runs serially inside the lane and only the per-lane
//        aggregates go through the wave scan (DPP row_shr / row_bcast, no LDS permutes).
//      * Workgroup barriers wait for LDS only (s_waitcnt lgkmcnt(0); s_barrier): direction bytes and node-end
//        value rows are fire-and-forget dword / dwordx4 global stores; the query is staged in LDS and the node
//        table / node bases come through the scalar cache, so the row loop issues no vector loads at all
//        (gfx950 retires vector memory operations in order: a load behind those stores would wait for them to
//        reach HBM).
//      * In-place hazard: inside a step every lane reads Hs[j0-1 .. j0+3] before the step's barrier and writes
//        after it; the first lane of the NEXT step needs the old Hs of this step's last column, which the last
//        lane parks in `edge` before the barrier.
//      * Rows with a predecessor that is not the row directly above (bubble arms, multi-predecessor rows) read
//        that predecessor's value row from HBM; such a row starts with a full __syncthreads() (vmcnt(0)).
// -DPOA_MARKERS puts region markers into the ISA (tests/isa_regions.py counts instructions between them)
#ifdef POA_MARKERS
#define POA_MARK(name) asm volatile("; MARK " name)
#else
#define POA_MARK(name)
#endif
#define POA_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int poa_dpp(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
// wave64 inclusive max-scan (the sequence LLVM's atomic optimizer emits for gfx9).  `old` is the identity of signed
// max so that the DPP combiner folds each stage into a single v_max_i32_dpp.
__device__ __forceinline__ int poa_wave_scan_max(int v)
{
    int t;
    t = poa_dpp<0x111, 0xf>(INT32_MIN, v); v = t > v ? t : v;  // row_shr:1
    t = poa_dpp<0x112, 0xf>(INT32_MIN, v); v = t > v ? t : v;  // row_shr:2
    t = poa_dpp<0x114, 0xf>(INT32_MIN, v); v = t > v ? t : v;  // row_shr:4
    t = poa_dpp<0x118, 0xf>(INT32_MIN, v); v = t > v ? t : v;  // row_shr:8
    t = poa_dpp<0x142, 0xa>(INT32_MIN, v); v = t > v ? t : v;  // row_bcast:15 -> rows 1,3
    t = poa_dpp<0x143, 0xc>(INT32_MIN, v); v = t > v ? t : v;  // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ int poa_wave_shr1(int v) { return poa_dpp<0x138, 0xf>(POA_IDENT, v); }  // wave_shr:1

template <int NT, int CPT, bool STAMP = false>
__global__ __launch_bounds__(NT) void k_poa_dp_lds(
    const poa_prob *__restrict__ probs, const char *__restrict__ queries, const uint4 *__restrict__ node_tab,
    const uint32_t *__restrict__ seq32, const uint32_t *__restrict__ preds, const uint32_t *__restrict__ sink_preds,
    poa_dev_params P, poa_row *rows, uint8_t *pool, unsigned long long *pool_next, uint64_t pool_size,
    poa_out *__restrict__ outs, uint32_t lds_cols, unsigned long long *stamps = nullptr)
{
    static_assert(CPT == 4, "row storage is 4-column aligned");
    constexpr int NW = NT / 64;
    constexpr int STEP = NT * CPT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int32_t *Hs = (int32_t *)smem;                           // [lds_cols]
    uint16_t *Ds = (uint16_t *)(smem + 4ull * lds_cols);     // [lds_cols]
    uint8_t *Qs = smem + 6ull * lds_cols;                    // [lds_cols] the query
    int32_t *scr = (int32_t *)(smem + 7ull * lds_cols);      // lds_cols is a multiple of 16
    int32_t *sW1 = scr;               // [2][NW] inclusive wave maxima of a1
    int32_t *sW2 = sW1 + 2 * NW;      // [2][NW]
    int32_t *sL1 = sW2 + 2 * NW;      // [2][NW] a1 of each wave's last cell
    int32_t *sL2 = sL1 + 2 * NW;      // [2][NW]
    int32_t *sRed = sL2 + 2 * NW;     // [NW][3]
    int32_t *edgeH = sRed + 3 * NW;   // [2]
    unsigned long long *s_alloc = (unsigned long long *)(edgeH + 2);

    const poa_prob pb = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qlen = (int)pb.qlen;
    const char *query = queries + pb.q0;
    const uint4 *ntab = node_tab + pb.node0;  // {first_row, len | npred << 24, remain_last, pred_start}: one scalar load per node
    const uint32_t *plist = preds + pb.pred0;
    // Plain (non-volatile) accesses: a volatile store makes hipcc wait vmcnt(0) first, i.e. for every direction-byte
    // store still in flight.  Cross-wave visibility of these arrays is only needed by "far" rows and by the sink
    // evaluation, both of which sit behind a full __syncthreads().
    poa_row *R = rows + pb.row0;

    const int o1 = P.o1, e1 = P.e1, o2 = P.o2, e2 = P.e2;
    const int bw = (int)pb.w;
    // diagnostic build only (STAMP): cycles per row segment, summed over the rows of this workgroup's wave 0 / last wave
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, n_far = 0, n_multi = 0, n_rows = 0, n_steps = 0;
    auto stamp = [&](int seg) {
        if constexpr (STAMP) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (seg >= 0) tacc[seg] += t - tprev;
            tprev = t;
        }
    };

    uint64_t dcur = 0, dend = 0, vcur = 0, vendp = 0;
    bool failed = false;
    auto take_chunk = [&](uint64_t &cur, uint64_t &end) {
        __syncthreads();
        if (tid == 0) *s_alloc = atomicAdd(pool_next, (unsigned long long)POA_CHUNK);
        __syncthreads();
        uint64_t b = *s_alloc;
        if (b + POA_CHUNK > pool_size) failed = true;
        cur = b;
        end = b + POA_CHUNK;
    };
    auto alloc = [&](uint64_t &cur, uint64_t &end, uint64_t bytes) -> uint64_t {
        bytes = (bytes + 15ull) & ~15ull;
        if (cur + bytes > end) take_chunk(cur, end);
        uint64_t r = cur;
        cur += bytes;
        return r;
    };

    for (int t = tid; t < (int)lds_cols; t += NT) {
        uint8_t code = 4;
        if (t < qlen) {
            const char ch = query[t];
            code = ch == 'A' ? 0 : (ch == 'C' ? 1 : (ch == 'G' ? 2 : (ch == 'T' ? 3 : 4)));
        }
        Qs[t] = code;  // 0..3 = ACGT, 4 = anything else (scores 0 against everything)
    }
    __syncthreads();

    int prev_beg = 0, prev_end = -1, prev_lmax = 0, prev_rmax = 0;
    uint64_t cells = 0, vcells = 0;
    uint32_t seq_word = 0, seq_word_idx = 0xFFFFFFFFu;

    // Rows are generated from the node table: node 0 is the virtual source (one row, no predecessor), node v
    // (v >= 1) contributes len rows whose first has the node's predecessor list and whose others follow the row
    // above.  Everything here is wave-uniform and comes through the scalar cache.
    for (uint32_t v = 0; v < pb.n_nodes && !failed; v++) {
    const uint4 nt = ntab[v];
    const uint32_t nlen = nt.y & 0xFFFFFFu;
    for (uint32_t tn = 0; tn < nlen && !failed; tn++) {
        const uint32_t r = nt.x + tn;
        const bool first = tn == 0 && v > 0;
        const bool last = tn + 1 == nlen;
        const int np = v == 0 ? 0 : (tn == 0 ? (int)(nt.y >> 24) : 1);
        const uint32_t ps = nt.w;
        const int remain = (int)(nt.z & 0x3fffffffu) + (int)(nlen - 1 - tn);
        uint8_t gb = 0;
        if (v > 0) {
            // row r is base r-1 of the problem's node sequences (seq0 is 4-aligned): one scalar dword per 4 rows
            const uint32_t bi = r - 1;
            if ((bi & 3u) == 0 || (bi >> 2) != seq_word_idx) { seq_word_idx = bi >> 2; seq_word = seq32[(pb.seq0 >> 2) + seq_word_idx]; }
            gb = (uint8_t)(seq_word >> (8u * (bi & 3u)));
        }
        // nt.w is the predecessor ROW itself for a node with one predecessor, the slice start in `preds` otherwise
        bool far = false;
        if (first) {
            if (np == 1) far = ps != r - 1;
            else
                for (int t = 0; t < np; t++) far |= plist[ps + t] != r - 1;
        }
        stamp(-1);
        if (far) __syncthreads();  // vmcnt(0) + barrier: the value rows / row arrays of far predecessors have landed
        int mpl, mpr;
        if (r == 0) { mpl = 0; mpr = 0; }
        else if (!first) { mpl = prev_lmax + 1; mpr = prev_rmax + 1; }
        else {
            mpl = INT32_MAX; mpr = 0;
            for (int t = 0; t < np; t++) {
                const uint32_t p = np == 1 ? ps : plist[ps + t];
                int lm, rm;
                if (p == r - 1) { lm = prev_lmax + 1; rm = prev_rmax + 1; }
                else { lm = R[p].lmax + 1; rm = R[p].rmax + 1; }
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
        const int bal = beg & ~3;
        const int W = (end - bal + 1 + 3) & ~3;  // storage width / plane stride
        if (r > 0) cells += (uint64_t)(end - beg + 1);
        if (last) vcells += (uint64_t)(end - beg + 1);
        const uint64_t doff = alloc(dcur, dend, (uint64_t)W * (np > 1 ? 4u : 1u));
        if (failed) break;
        uint64_t voff = 0;
        if (last) { voff = alloc(vcur, vendp, 6ull * (uint64_t)W); if (failed) break; }
        if (tid == 0) {
            R[r].beg = beg; R[r].end = end;
            R[r].doff = doff; R[r].voff = voff;
            R[r].pred = ps; R[r].npred = first ? (uint32_t)np : 0u;
        }
        int32_t *Hrow = (int32_t *)(pool + voff);                       // value row: int32 H[W] then uint16 D[W]
        uint16_t *Drow = (uint16_t *)(pool + voff + 4ull * (uint64_t)W);
        uint8_t *drow = pool + doff;
        // substitution score of this row's base against a query CODE (0..3 = ACGT, 4 = anything else)
        const int gcode = gb == 'A' ? 0 : (gb == 'C' ? 1 : (gb == 'G' ? 2 : (gb == 'T' ? 3 : 4)));
        const int sc_eq = gcode == 4 ? 0 : P.match, sc_ne = gcode == 4 ? 0 : -P.mismatch;
        // single predecessor = the row directly above (in LDS): the branch-free fast path applies
        const bool single_lds = r > 0 && np == 1 && !far;
        stamp(0);  // row prologue (band, allocation, metadata)
        if constexpr (STAMP) { n_far += far; n_multi += np > 1; n_rows++; n_steps += (W + STEP - 1) / STEP; }

        int carry1 = POA_IDENT, carry2 = POA_IDENT, left1 = POA_IDENT, left2 = POA_IDENT;
        int best = INT32_MIN, lpos = beg, rpos = beg;
        int buf = 0;
        for (int c0 = 0; c0 < W; c0 += STEP, buf ^= 1) {
            const int c = c0 + CPT * tid;   // storage index of this lane's first cell (multiple of 4)
            const int j0 = bal + c;         // absolute column of this lane's first cell (multiple of 4)
            const bool lane_act = j0 <= end;  // j0 + CPT - 1 >= beg always holds (bal > beg - 4)
            const int nw_step = (W - c0 + 64 * CPT - 1) / (64 * CPT) < NW ? (W - c0 + 64 * CPT - 1) / (64 * CPT) : NW;
            const bool wave_act = wv < nw_step;
            const bool fast = single_lds && wave_act;  // lean branch-free path (band edges handled by masks)
            int ht[CPT], hts[CPT], ev1[CPT], ev2[CPT], pmv[CPT], p1v[CPT], p2v[CPT], ofl[CPT];
            int a1[CPT], a2[CPT];
            int agg1 = POA_IDENT, agg2 = POA_IDENT;
            if (fast) {
                // ---------------- lean path, phase 1: one predecessor, the row above (LDS); masks instead of branches
                const int4 hv = *(const int4 *)(Hs + j0);
                const uint2 dv = *(const uint2 *)(Ds + j0);
                const uint32_t qw = *(const uint32_t *)(Qs + j0);
                int hm[CPT], hj[CPT], g1[CPT], g2[CPT], qc[CPT];
                hj[0] = hv.x; hj[1] = hv.y; hj[2] = hv.z; hj[3] = hv.w;
                if (tid == NT - 1) edgeH[buf] = hv.w;
                const int jm1 = j0 > 0 ? j0 - 1 : 0;
                hm[0] = (tid == 0 && c0 > 0) ? edgeH[buf ^ 1] : Hs[jm1];
                hm[1] = hv.x; hm[2] = hv.y; hm[3] = hv.z;
                g1[0] = dv.x & 255; g2[0] = (dv.x >> 8) & 255; g1[1] = (dv.x >> 16) & 255; g2[1] = dv.x >> 24;
                g1[2] = dv.y & 255; g2[2] = (dv.y >> 8) & 255; g1[3] = (dv.y >> 16) & 255; g2[3] = dv.y >> 24;
                qc[0] = Qs[jm1]; qc[1] = qw & 255; qc[2] = (qw >> 8) & 255; qc[3] = (qw >> 16) & 255;
                const unsigned pspan = (unsigned)(prev_end - prev_beg), span = (unsigned)(end - beg);
                bool inprev = j0 >= 1 && (unsigned)(j0 - 1 - prev_beg) <= pspan;  // column j-1 inside the predecessor's band
#pragma unroll
                for (int k = 0; k < CPT; k++) {
                    const int j = j0 + k;
                    const bool inj = (unsigned)(j - prev_beg) <= pspan;
                    const bool actk = (unsigned)(j - beg) <= span;
                    const int s = qc[k] == gcode ? sc_eq : (qc[k] == 4 ? 0 : sc_ne);
                    const int m = inprev ? hm[k] + s : POA_NEG;
                    ev1[k] = inj ? hj[k] - g1[k] : POA_NEG;
                    ev2[k] = inj ? hj[k] - g2[k] : POA_NEG;
                    const int me = m > ev1[k] ? m : ev1[k];
                    ht[k] = me > ev2[k] ? me : ev2[k];
                    hts[k] = ev2[k] > me ? 2 : (ev1[k] > m ? 1 : 0);
                    ofl[k] = (g1[k] == o1 + e1 ? 1 : 0) | (g2[k] == o2 + e2 ? 2 : 0);
                    pmv[k] = 0; p1v[k] = 0; p2v[k] = 0;
                    a1[k] = actk ? ht[k] + e1 * j : POA_IDENT;
                    a2[k] = actk ? ht[k] + e2 * j : POA_IDENT;
                    agg1 = a1[k] > agg1 ? a1[k] : agg1;
                    agg2 = a2[k] > agg2 ? a2[k] : agg2;
                    inprev = inj;
                }
            } else if (wave_act) {
                // ---------------- general path, phase 1: band edges, source row, several / far predecessors
                bool act[CPT];
#pragma unroll
                for (int k = 0; k < CPT; k++) {
                    const int j = j0 + k;
                    act[k] = j >= beg && j <= end;
                    ht[k] = POA_NEG; hts[k] = 0; ev1[k] = POA_NEG; ev2[k] = POA_NEG; pmv[k] = 0; p1v[k] = 0; p2v[k] = 0; ofl[k] = 0;
                }
                if (r == 0) {
#pragma unroll
                    for (int k = 0; k < CPT; k++) ht[k] = (j0 + k == 0) ? 0 : POA_NEG;
                } else if (lane_act) {
                    int sub[CPT], m[CPT];
                    {
                        const uint32_t qw = *(const uint32_t *)(Qs + j0);  // codes of query[j0 .. j0+3]
                        const int qm1 = j0 >= 1 ? (int)Qs[j0 - 1] : 4;
#pragma unroll
                        for (int k = 0; k < CPT; k++) {
                            const int qc = k == 0 ? qm1 : (int)((qw >> (8 * (k - 1))) & 255u);
                            sub[k] = qc == gcode ? sc_eq : (qc == 4 ? 0 : sc_ne);
                            m[k] = POA_NEG;
                        }
                    }
                    for (int t = 0; t < np; t++) {
                        const uint32_t p = first ? (np == 1 ? ps : plist[ps + t]) : r - 1;
                        int hj[CPT], dj[CPT], hm0;
                        int bp, ep;
                        if (p == r - 1) {
                            bp = prev_beg; ep = prev_end;
                            const int4 hv = *(const int4 *)(Hs + j0);
                            const uint2 dv = *(const uint2 *)(Ds + j0);
                            hj[0] = hv.x; hj[1] = hv.y; hj[2] = hv.z; hj[3] = hv.w;
                            dj[0] = dv.x & 0xffff; dj[1] = dv.x >> 16; dj[2] = dv.y & 0xffff; dj[3] = dv.y >> 16;
                            if (tid == NT - 1) edgeH[buf] = hv.w;   // old value of this step's last column
                            hm0 = POA_NEG;
                            if (j0 >= 1) hm0 = (tid == 0 && c0 > 0) ? edgeH[buf ^ 1] : Hs[j0 - 1];
                        } else {
                            bp = R[p].beg; ep = R[p].end;
                            const uint64_t pv = R[p].voff;
                            const int balp = bp & ~3;
                            const int Wp = (ep - balp + 1 + 3) & ~3;
                            const int32_t *Hp = (const int32_t *)(pool + pv);
                            const uint16_t *Dp = (const uint16_t *)(pool + pv + 4ull * (uint64_t)Wp);
#pragma unroll
                            for (int k = 0; k < CPT; k++) {
                                const int j = j0 + k;
                                hj[k] = POA_NEG; dj[k] = 0;
                                if (j >= bp && j <= ep) { hj[k] = Hp[j - balp]; dj[k] = (int)Dp[j - balp]; }
                            }
                            hm0 = POA_NEG;
                            if (j0 - 1 >= bp && j0 - 1 <= ep) hm0 = Hp[j0 - 1 - balp];
                        }
#pragma unroll
                        for (int k = 0; k < CPT; k++) {
                            const int j = j0 + k;
                            const int hm = k == 0 ? hm0 : hj[k - 1];
                            if (act[k] && j >= 1 && j - 1 >= bp && j - 1 <= ep) {
                                const int cnd = hm + sub[k];
                                if (cnd > m[k]) { m[k] = cnd; pmv[k] = t; }
                            }
                            if (act[k] && j >= bp && j <= ep) {
                                const int gg1 = dj[k] & 255, gg2 = dj[k] >> 8;  // E + d
                                const int c1 = hj[k] - gg1;
                                if (c1 > ev1[k]) { ev1[k] = c1; p1v[k] = t; ofl[k] = (ofl[k] & 2) | (gg1 == o1 + e1 ? 1 : 0); }
                                const int c2 = hj[k] - gg2;
                                if (c2 > ev2[k]) { ev2[k] = c2; p2v[k] = t; ofl[k] = (ofl[k] & 1) | (gg2 == o2 + e2 ? 2 : 0); }
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < CPT; k++) {
                        ht[k] = m[k];
                        if (ev1[k] > ht[k]) { ht[k] = ev1[k]; hts[k] = 1; }
                        if (ev2[k] > ht[k]) { ht[k] = ev2[k]; hts[k] = 2; }
                    }
                }
#pragma unroll
                for (int k = 0; k < CPT; k++) {
                    a1[k] = act[k] ? ht[k] + e1 * (j0 + k) : POA_IDENT;
                    a2[k] = act[k] ? ht[k] + e2 * (j0 + k) : POA_IDENT;
                    agg1 = a1[k] > agg1 ? a1[k] : agg1;
                    agg2 = a2[k] > agg2 ? a2[k] : agg2;
                }
            } else {
#pragma unroll
                for (int k = 0; k < CPT; k++) { a1[k] = POA_IDENT; a2[k] = POA_IDENT; ht[k] = POA_NEG; hts[k] = 0; ev1[k] = ev2[k] = POA_NEG; pmv[k] = p1v[k] = p2v[k] = ofl[k] = 0; }
            }
            stamp(1);  // phase 1
            // ---- insertion recurrence: serial inside the lane, scan of the lane aggregates across the wave
            int i1 = POA_IDENT, i2 = POA_IDENT;
            if (wave_act) {
                i1 = poa_wave_scan_max(agg1);
                i2 = poa_wave_scan_max(agg2);
            }
            if (lane == 63) {  // inactive waves publish the identity, so readers need no masks
                sW1[buf * NW + wv] = i1; sW2[buf * NW + wv] = i2;
                sL1[buf * NW + wv] = a1[CPT - 1]; sL2[buf * NW + wv] = a2[CPT - 1];
            }
            stamp(2);  // scans
            POA_LDS_BARRIER();
            stamp(3);  // step barrier
            int tw1[NW], tw2[NW];
#pragma unroll
            for (int q = 0; q < NW; q++) { tw1[q] = sW1[buf * NW + q]; tw2[q] = sW2[buf * NW + q]; }
            int all1 = carry1, all2 = carry2;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                all1 = tw1[q] > all1 ? tw1[q] : all1;
                all2 = tw2[q] > all2 ? tw2[q] : all2;
            }
            if (wave_act) {
                int x1 = poa_wave_shr1(i1), x2 = poa_wave_shr1(i2);
                int la1 = poa_wave_shr1(a1[CPT - 1]), la2 = poa_wave_shr1(a2[CPT - 1]);
                if (lane == 0) {
                    la1 = wv == 0 ? left1 : sL1[buf * NW + wv - 1];
                    la2 = wv == 0 ? left2 : sL2[buf * NW + wv - 1];
                }
                int pre1 = carry1, pre2 = carry2;
#pragma unroll
                for (int q = 0; q < NW; q++) {
                    if (q < wv) { pre1 = tw1[q] > pre1 ? tw1[q] : pre1; pre2 = tw2[q] > pre2 ? tw2[q] : pre2; }
                }
                int run1 = pre1 > x1 ? pre1 : x1;   // max of a1 over every column before this lane's first cell
                int run2 = pre2 > x2 ? pre2 : x2;
                int hv[CPT], codev[CPT], dpk[CPT];
                if (fast) {
                    // ---------------- lean path, phase 2
                    const unsigned span = (unsigned)(end - beg);
#pragma unroll
                    for (int k = 0; k < CPT; k++) {
                        const int j = j0 + k;
                        const bool actk = (unsigned)(j - beg) <= span;
                        const int f1 = run1 - (o1 + e1 * j), f2 = run2 - (o2 + e2 * j);  // run = POA_IDENT at the first column
                        const int fo = (run1 == la1 ? 64 : 0) | (run2 == la2 ? 128 : 0);
                        const int hf = ht[k] > f1 ? ht[k] : f1;
                        const int h = hf > f2 ? hf : f2;
                        const int fsel = f2 > hf ? 32 : (f1 > ht[k] ? 16 : 0);
                        codev[k] = hts[k] | (ofl[k] << 2) | fsel | fo;
                        int dd1 = h - ev1[k]; dd1 = (dd1 < o1 ? dd1 : o1) + e1;
                        int dd2 = h - ev2[k]; dd2 = (dd2 < o2 ? dd2 : o2) + e2;
                        hv[k] = h;
                        dpk[k] = dd1 | (dd2 << 8);
                        const int hb = actk ? h : INT32_MIN;  // inactive cells never take part in the row maximum
                        if (hb > best) { best = hb; lpos = j; rpos = j; }
                        else if (actk && hb == best) rpos = j;
                        run1 = a1[k] > run1 ? a1[k] : run1;   // a1 is IDENT on inactive cells
                        run2 = a2[k] > run2 ? a2[k] : run2;
                        la1 = actk ? a1[k] : la1; la2 = actk ? a2[k] : la2;
                    }
                } else if (lane_act) {
                    // ---------------- general path, phase 2
#pragma unroll
                    for (int k = 0; k < CPT; k++) {
                        const int j = j0 + k;
                        const bool actk = j >= beg && j <= end;
                        const int f1 = run1 - o1 - e1 * j, f2 = run2 - o2 - e2 * j;  // run = POA_IDENT at the first column
                        const int fo1 = run1 == la1, fo2 = run2 == la2;
                        int h = ht[k], hs = hts[k];
                        if (f1 > h) { h = f1; hs = 3; }
                        if (f2 > h) { h = f2; hs = 4; }
                        codev[k] = hts[k] | (ofl[k] << 2) | (hs >= 3 ? (hs - 2) << 4 : 0) | (fo1 << 6) | (fo2 << 7);
                        int dd1 = h - ev1[k]; dd1 = (dd1 < o1 ? dd1 : o1) + e1;
                        int dd2 = h - ev2[k]; dd2 = (dd2 < o2 ? dd2 : o2) + e2;
                        hv[k] = h;
                        dpk[k] = dd1 | (dd2 << 8);
                        if (actk) {
                            if (h > best) { best = h; lpos = j; rpos = j; }
                            else if (h == best) rpos = j;
                            run1 = a1[k] > run1 ? a1[k] : run1;
                            run2 = a2[k] > run2 ? a2[k] : run2;
                            la1 = a1[k]; la2 = a2[k];
                        }
                    }
                }
                if (lane_act) {
                    const uint2 dq = make_uint2((uint32_t)dpk[0] | ((uint32_t)dpk[1] << 16), (uint32_t)dpk[2] | ((uint32_t)dpk[3] << 16));
                    *(int4 *)(Hs + j0) = make_int4(hv[0], hv[1], hv[2], hv[3]);
                    *(uint2 *)(Ds + j0) = dq;
                    *(uint32_t *)(drow + c) = (uint32_t)codev[0] | ((uint32_t)codev[1] << 8) | ((uint32_t)codev[2] << 16) | ((uint32_t)codev[3] << 24);
                    if (last) {
                        *(int4 *)(Hrow + c) = make_int4(hv[0], hv[1], hv[2], hv[3]);
                        *(uint2 *)(Drow + c) = dq;
                    }
                    if (np > 1) {
                        *(uint32_t *)(drow + (uint64_t)W + c) = (uint32_t)pmv[0] | ((uint32_t)pmv[1] << 8) | ((uint32_t)pmv[2] << 16) | ((uint32_t)pmv[3] << 24);
                        *(uint32_t *)(drow + 2ull * W + c) = (uint32_t)p1v[0] | ((uint32_t)p1v[1] << 8) | ((uint32_t)p1v[2] << 16) | ((uint32_t)p1v[3] << 24);
                        *(uint32_t *)(drow + 3ull * W + c) = (uint32_t)p2v[0] | ((uint32_t)p2v[1] << 8) | ((uint32_t)p2v[2] << 16) | ((uint32_t)p2v[3] << 24);
                    }
                }
            }
            carry1 = all1; carry2 = all2;
            left1 = sL1[buf * NW + nw_step - 1]; left2 = sL2[buf * NW + nw_step - 1];
        }
        stamp(4);  // phase 2 + stores
        {
            // wave-level: maximum of best, then the leftmost / rightmost column among the lanes that hold it
            int wb = poa_wave_scan_max(best);                       // lane 63 holds the wave maximum
            wb = __builtin_amdgcn_readlane(wb, 63);
            int lm = best == wb ? -lpos : POA_IDENT;                 // min(lpos) = -max(-lpos)
            int rm = best == wb ? rpos : POA_IDENT;
            lm = poa_wave_scan_max(lm);
            rm = poa_wave_scan_max(rm);
            if (lane == 63) { sRed[wv * 3 + 0] = wb; sRed[wv * 3 + 1] = -lm; sRed[wv * 3 + 2] = rm; }
        }
        POA_LDS_BARRIER();  // row complete in LDS; also fences the scratch buffers between rows
        {
            int rb[NW], rl[NW], rr[NW];
#pragma unroll
            for (int q = 0; q < NW; q++) { rb[q] = sRed[q * 3]; rl[q] = sRed[q * 3 + 1]; rr[q] = sRed[q * 3 + 2]; }
            best = rb[0]; lpos = rl[0]; rpos = rr[0];
#pragma unroll
            for (int q = 1; q < NW; q++) {
                if (rb[q] > best) { best = rb[q]; lpos = rl[q]; rpos = rr[q]; }
                else if (rb[q] == best) { lpos = rl[q] < lpos ? rl[q] : lpos; rpos = rr[q] > rpos ? rr[q] : rpos; }
            }
        }
        // uniform values: keep them in scalar registers so the next row's band arithmetic runs on the scalar unit
        lpos = __builtin_amdgcn_readfirstlane(lpos);
        rpos = __builtin_amdgcn_readfirstlane(rpos);
        if (tid == 0) { R[r].lmax = lpos; R[r].rmax = rpos; }
        prev_beg = beg; prev_end = end; prev_lmax = lpos; prev_rmax = rpos;
        stamp(5);  // row reduce + row barrier
    }
    }
    __syncthreads();
    if (tid == 0) {
        poa_out &O = outs[blockIdx.x];
        O.cells = cells;
        O.vcells = vcells;
        O.maxw = 0;
        if constexpr (STAMP) {
            if (stamps && blockIdx.x < 64)
                for (int s = 0; s < 6; s++) stamps[blockIdx.x * 6 + s] = tacc[s];
            if (stamps && blockIdx.x == 0) { stamps[384] = n_far; stamps[385] = n_multi; stamps[386] = n_rows; stamps[387] = n_steps; }
        }
        if (failed) {
            O.status = POA_ST_POOL;
            O.score = POA_NEG;
            O.row = 0;
        } else {
            int bestv = INT32_MIN;
            uint32_t brow = 0;
            bool have = false;
            for (uint32_t t = 0; t < pb.n_sink; t++) {
                const uint32_t p = sink_preds[pb.sink0 + t];
                const int bp = R[p].beg, ep = R[p].end;
                int val = POA_NEG;
                if (qlen >= bp && qlen <= ep) val = ((const int32_t *)(pool + R[p].voff))[qlen - (bp & ~3)];
                if (!have || val > bestv) { bestv = val; brow = p; have = true; }
            }
            O.score = bestv;
            O.row = brow;
            O.status = (have && bestv > POA_NEG / 2) ? POA_ST_OK : POA_ST_NOALN;
        }
    }
}

// K4b: one lane per problem.  ops are written in reverse (sink -> source) order.
// ENC 0: direction bytes of k_poa_dp_lds / k_poa_dp_pk; ENC 1: of k_poa_dp_t4 (vga_poa_t4.hpp)
struct tb_code { int hts, fsel, eo1, eo2, fo1, fo2; };
template <int ENC>
__device__ __forceinline__ tb_code tb_decode(int code)
{
    tb_code c;
    if constexpr (ENC == 0) {
        // [1:0] source of Ht (M, E1, E2), [3:2] E1/E2 opened here, [4] F1 > Ht, [5] F2 > max(Ht, F1), [7:6] F1/F2 opened here
        c.hts = code & 3;
        c.fsel = (code & 32) ? 2 : ((code >> 4) & 1);
        c.eo1 = (code >> 2) & 1; c.eo2 = (code >> 3) & 1;
        c.fo1 = (code >> 6) & 1; c.fo2 = (code >> 7) & 1;
    } else {
        // [7:6] tag of H (3 Ht, 1 F1, 0 F2), [5:4] tag of Ht (2 M, 1 E1, 0 E2), [3] / [2] E1 / E2 of a successor opens from
        // this cell, [1] / [0] F1 / F2 of this cell did not open
        const int th = (code >> 6) & 3;
        c.hts = 2 - ((code >> 4) & 3);
        c.fsel = th == 3 ? 0 : (th == 1 ? 1 : 2);
        c.eo1 = (code >> 3) & 1; c.eo2 = (code >> 2) & 1;
        c.fo1 = ((code >> 1) & 1) ^ 1; c.fo2 = (code & 1) ^ 1;
    }
    return c;
}

#ifdef VGA_VARIANTS  // the one-lane walk (VGA_POA_TB=lane), superseded by poa_traceback_wave
template <int ENC>
__global__ __launch_bounds__(64) void k_poa_traceback(
    uint32_t n, const poa_prob *__restrict__ probs, const poa_row *__restrict__ rows,
    const uint32_t *__restrict__ preds, const uint8_t *__restrict__ pool, poa_out *__restrict__ outs,
    uint8_t *__restrict__ ops, uint32_t *__restrict__ orow, int code_xor)
{
    const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= n) return;
    outs[pi].nops = 0;
    if (outs[pi].status != POA_ST_OK) return;
    const poa_prob pb = probs[pi];
    const uint64_t cap = (uint64_t)pb.N + pb.qlen + 2;
    uint8_t *po = ops + pb.ops0;
    uint32_t *pr = orow + pb.ops0;
    uint32_t i = outs[pi].row;
    int j = (int)pb.qlen;
    int st = 0;  // 0 H, 1 E1, 2 E2, 3 F1, 4 F2, 5 Ht
    int pend_e = 0;
    uint64_t nops = 0;
    bool bad = false;
    while (i > 0 && !bad) {
        const uint64_t ri = pb.row0 + i;
        const poa_row rw = rows[ri];
        const uint2 inf = make_uint2(rw.pred, rw.npred);  // {pred_start, npred if this row starts a node else 0}
        const bool first = inf.y != 0;
        const int np = first ? (int)inf.y : 1;
        const int beg = rw.beg, end = rw.end;
        const int bal = beg & ~3;
        const uint64_t W = (uint64_t)((end - bal + 1 + 3) & ~3);
        const uint64_t doff = rw.doff;
        if (j < beg || j > end) { bad = true; break; }
        const uint64_t c = (uint64_t)(j - bal);
        const int code = pool[doff + c] ^ code_xor;  // the 16-bit DP kernel stores the F-open bits inverted
        const tb_code dc = tb_decode<ENC>(code);
        if (ENC == 1 && pend_e) {  // arrived through a deletion: this cell says whether that gap opened from it
            if (pend_e == 1 ? dc.eo1 : dc.eo2) st = 0;
            pend_e = 0;
        }
        const int hts = dc.hts;
        const int fsel = dc.fsel;
        const int hs = fsel ? 2 + fsel : hts;
        const int src = st == 0 ? hs : (st == 5 ? hts : st);
        if (nops + 1 >= cap) { bad = true; break; }
        if (src == 0) {
            const int t = np > 1 ? pool[doff + W + c] : 0;
            const uint32_t p = first ? (np == 1 ? inf.x : preds[pb.pred0 + inf.x + t]) : i - 1;
            if (j < 1) { bad = true; break; }
            po[nops] = 0; pr[nops] = i; nops++;
            i = p; j -= 1; st = 0;
        } else if (src == 1 || src == 2) {
            const int t = np > 1 ? pool[doff + (src == 1 ? 2 : 3) * W + c] : 0;
            const uint32_t p = first ? (np == 1 ? inf.x : preds[pb.pred0 + inf.x + t]) : i - 1;
            const int open = src == 1 ? dc.eo1 : dc.eo2;
            po[nops] = 2; pr[nops] = i; nops++;
            if (ENC == 1) { st = src; pend_e = src; }
            else st = open ? 0 : src;
            i = p;
        } else {
            const int open = src == 3 ? dc.fo1 : dc.fo2;
            if (j - 1 < beg) { bad = true; break; }
            po[nops] = 1; pr[nops] = 0; nops++;
            st = open ? 5 : src;
            j -= 1;
        }
    }
    while (!bad && j > 0) {
        if (nops + 1 >= cap) { bad = true; break; }
        po[nops] = 1; pr[nops] = 0; nops++;
        j--;
    }
    if (bad) { outs[pi].status = POA_ST_TRACE; nops = 0; }
    outs[pi].nops = (uint32_t)nops;
}

#endif  // VGA_VARIANTS

// K4b, cooperative form (the default): one wave per problem.  The walk itself is a chain of dependent reads (row
// record -> direction byte -> predecessor), two HBM round trips per operation when one lane does it alone.  Here the
// 64 lanes stage, in two round trips, what the next stretch of the walk can need -- the records of rows i .. i-63
// and, for each of them, a TB_WIN-byte window of its direction row around the column the path would reach it at if
// every row in between lay on the path (fewer columns are consumed when rows are skipped, so the window starts just
// below that column and extends above it); rows with several predecessors also stage the window of their predecessor
// choice plane and their first four predecessors -- and then all lanes walk in lock step out of LDS until the path
// leaves the staged rows or a window (30-60 operations on the HLA graphs).  Same outputs as k_poa_traceback.
#define TB_WIN 32
struct tb_lds {  // 6 912 B: what one wave stages per stretch
    int beg[64], end[64], ws[64];
    uint64_t doff[64];
    uint32_t pred[64], np[64];
    uint32_t dir[64][TB_WIN / 4], pl1[64][TB_WIN / 4], pr4[64][4];
};
// LDS traffic of one wave is processed in program order: between the staging stores and the walk's loads (other
// lanes' data) the wave only has to wait for its own stores to be issued -- no s_barrier, so the function can run in
// one wave of a larger workgroup whose other waves have finished
__device__ __forceinline__ void tb_wave_sync()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// every lane of the wave calls this with the same arguments (lane = its index); status / start_row are the DP's result
template <int ENC = 0>
__device__ __forceinline__ void poa_traceback_wave(
    tb_lds &T, const int lane, const poa_prob &pb, const poa_row *__restrict__ rows, const uint32_t *__restrict__ preds,
    const uint8_t *__restrict__ pool, poa_out &O, uint8_t *__restrict__ ops, uint32_t *__restrict__ orow, const int code_xor,
    const int status, const uint32_t start_row)
{
    if (status != POA_ST_OK) {
        if (lane == 0) O.nops = 0;
        return;
    }
    const uint32_t cap = (uint32_t)std::min<uint64_t>((uint64_t)pb.N + pb.qlen + 2, 0xffffffffu);
    uint8_t *po = ops + pb.ops0;
    uint32_t *pr = orow + pb.ops0;
    uint32_t i = start_row;
    int j = (int)pb.qlen;
    int st = 0;  // 0 H, 1 E1, 2 E2, 3 F1, 4 F2, 5 Ht
    int pend_e = 0;
    uint32_t nops = 0;
    bool bad = false;
    uint32_t my_op = 0, my_row = 0;  // lane l keeps operation number (64 m + l) until the wave stores 64 of them together
    auto emit = [&](uint32_t op, uint32_t row) {
        if ((uint32_t)lane == (nops & 63u)) { my_op = op; my_row = row; }
        nops++;
        if ((nops & 63u) == 0) {
            po[nops - 64 + lane] = (uint8_t)my_op;
            pr[nops - 64 + lane] = my_row;
        }
    };
    while (i > 0 && !bad) {
        // ---- stage rows i0 .. i0 - 63
        const uint32_t i0 = i;
        tb_wave_sync();  // the previous stretch is done with the LDS arrays
        if ((uint32_t)lane < i0) {
            const uint32_t r = i0 - (uint32_t)lane;
            const poa_row rw = rows[pb.row0 + r];
            const int bal = rw.beg & ~3;
            const int W = (rw.end - bal + 1 + 3) & ~3;
            int ws = ((j - lane - 8) - bal) & ~3;  // window start relative to bal
            if (ws > W - TB_WIN) ws = W - TB_WIN;
            if (ws < 0) ws = 0;
            const uint8_t *drow = pool + rw.doff;
#pragma unroll
            for (int d = 0; d < TB_WIN / 4; d++) {
                const int cc = ws + 4 * d;
                T.dir[lane][d] = cc < W ? *(const uint32_t *)(drow + cc) : 0u;
            }
            if (rw.npred > 1) {
#pragma unroll
                for (int d = 0; d < TB_WIN / 4; d++) {
                    const int cc = ws + 4 * d;
                    T.pl1[lane][d] = cc < W ? *(const uint32_t *)(drow + (uint64_t)W + cc) : 0u;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) T.pr4[lane][q] = (uint32_t)q < rw.npred ? preds[pb.pred0 + rw.pred + q] : 0u;
            }
            T.beg[lane] = rw.beg;
            T.end[lane] = rw.end;
            T.ws[lane] = bal + ws;
            T.doff[lane] = rw.doff;
            T.pred[lane] = rw.pred;
            T.np[lane] = rw.npred;
        }
        tb_wave_sync();
        // ---- walk out of LDS, every lane the same steps
        while (i > 0) {
            const uint32_t t = i0 - i;
            if (t >= 64) break;
            const int beg = T.beg[t], end = T.end[t];
            if (j < beg || j > end) { bad = true; break; }
            const int off = j - T.ws[t];
            if (off < 0 || off >= TB_WIN) break;  // t > 0 here: the window of row i0 was placed around j
            const uint32_t npred = T.np[t];
            const bool first = npred != 0;
            const int np = first ? (int)npred : 1;
            const int code = (int)((T.dir[t][off >> 2] >> (8 * (off & 3))) & 0xffu) ^ code_xor;
            const tb_code dc = tb_decode<ENC>(code);
            if (ENC == 1 && pend_e) {  // arrived through a deletion: this cell says whether that gap opened from it
                if (pend_e == 1 ? dc.eo1 : dc.eo2) st = 0;
                pend_e = 0;
            }
            const int hts = dc.hts;
            const int fsel = dc.fsel;
            const int hs = fsel ? 2 + fsel : hts;
            const int src = st == 0 ? hs : (st == 5 ? hts : st);
            if (nops + 1 >= cap) { bad = true; break; }
            if (src <= 2) {
                uint32_t p = i - 1;
                if (first) {
                    p = T.pred[t];
                    if (np > 1) {
                        int tt;
                        if (src == 0) tt = (int)((T.pl1[t][off >> 2] >> (8 * (off & 3))) & 0xffu);
                        else {
                            const int bal = beg & ~3;
                            const uint64_t W = (uint64_t)((end - bal + 1 + 3) & ~3);
                            tt = pool[T.doff[t] + (src == 1 ? 2 : 3) * W + (uint64_t)(j - bal)];
                        }
                        p = tt < 4 ? T.pr4[t][tt] : preds[pb.pred0 + p + tt];
                    }
                }
                p = (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
                if (src == 0) {
                    if (j < 1) { bad = true; break; }
                    emit(0, i);
                    i = p; j -= 1; st = 0;
                } else {
                    const int open = src == 1 ? dc.eo1 : dc.eo2;
                    emit(2, i);
                    if (ENC == 1) { st = src; pend_e = src; }
                    else st = open ? 0 : src;
                    i = p;
                }
            } else {
                const int open = src == 3 ? dc.fo1 : dc.fo2;
                if (j - 1 < beg) { bad = true; break; }
                emit(1, 0);
                st = open ? 5 : src;
                j -= 1;
            }
        }
    }
    // the rest of the query is an insertion before the first aligned row
    if (!bad && j > 0 && (uint64_t)nops + (uint64_t)j >= cap) bad = true;
    if (bad) {
        if (lane == 0) { O.status = POA_ST_TRACE; O.nops = 0; }
        return;
    }
    const uint32_t base = nops & ~63u;
    if ((uint32_t)lane < (nops & 63u)) { po[base + lane] = (uint8_t)my_op; pr[base + lane] = my_row; }
    for (uint32_t x = (uint32_t)lane; x < (uint32_t)j; x += 64) { po[nops + x] = 1; pr[nops + x] = 0; }
    if (lane == 0) O.nops = nops + (uint32_t)j;
}

template <int ENC>
__global__ __launch_bounds__(64) void k_poa_traceback_wave(
    uint32_t n, const poa_prob *__restrict__ probs, const poa_row *__restrict__ rows,
    const uint32_t *__restrict__ preds, const uint8_t *__restrict__ pool, poa_out *__restrict__ outs,
    uint8_t *__restrict__ ops, uint32_t *__restrict__ orow, int code_xor)
{
    __shared__ tb_lds T;
    const uint32_t pi = blockIdx.x;
    if (pi >= n) return;
    const poa_prob pb = probs[pi];
    poa_traceback_wave<ENC>(T, (int)threadIdx.x, pb, rows, preds, pool, outs[pi], ops, orow, code_xor, outs[pi].status, outs[pi].row);
}

#ifdef VGA_VARIANTS  // k_poa_dp_pk (round 1's kernel, incl. its 16-bit and stamp builds), superseded by k_poa_dp_t4: `make variants`
// ---------------------------------------------------------------------------------------------------------
// K4, packed form (round 1's default).  Same algorithm and outputs as k_poa_dp_lds; the differences are about
// residency:
//   * one 32-bit LDS word per column:  (H << 8) | g1 | g2 << G1B   with gk = Ek + min(H - E_k, Ok) (G1B + G2B <= 8
//     bits for the usual penalties; H needs 23 signed bits), and the query as 4-bit codes indexed by COLUMN
//     (nibble j = code of query[j-1]).  4.5 B per column instead of 7 => 45 KB at 10 kbp => three workgroups per CU;
//   * between the two phases of a step a cell is carried as  ht  plus one packed word (source of Ht, open flags,
//     u_k = min(Ht - E_k, O_k)); the open/extend deltas of H are rebuilt from u_k in phase 2
//     (min(H - E_k, O_k) = min(u_k + (H - Ht), O_k) because H >= Ht), so the kernel fits 80 VGPRs = 6 waves/SIMD;
//   * node-end value rows in HBM are the same packed words (4 B per cell);
//   * single-predecessor rows whose predecessor is not the row above ("far") use the lean path as well, with
//     their five words coming from HBM instead of LDS.
// ---- packed 16-bit helpers (two cells per 32-bit register; v_pk_* instructions)
typedef short pk16 __attribute__((ext_vector_type(2)));
typedef unsigned short pku16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 pk(int x) { return __builtin_bit_cast(pk16, x); }
__device__ __forceinline__ pku16 pku(int x) { return __builtin_bit_cast(pku16, x); }
__device__ __forceinline__ int ipk(pk16 x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int ipk(pku16 x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int pk_bcast(int v) { return (int)(((uint32_t)v & 0xffffu) | ((uint32_t)v << 16)); }
__device__ __forceinline__ int pk_make(int lo, int hi) { return (int)(((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16)); }
__device__ __forceinline__ int pk_adds(int a, int b) { return ipk(__builtin_elementwise_add_sat(pk(a), pk(b))); }  // saturating
__device__ __forceinline__ int pk_subs(int a, int b) { return ipk(__builtin_elementwise_sub_sat(pk(a), pk(b))); }
__device__ __forceinline__ int pk_add(int a, int b) { return ipk(pku(a) + pku(b)); }                               // wrapping
__device__ __forceinline__ int pk_sub(int a, int b) { return ipk(pku(a) - pku(b)); }
__device__ __forceinline__ int pk_max(int a, int b) { return ipk(__builtin_elementwise_max(pk(a), pk(b))); }
__device__ __forceinline__ int pk_min(int a, int b) { return ipk(__builtin_elementwise_min(pk(a), pk(b))); }
__device__ __forceinline__ int pk_minu(int a, int b) { return ipk(__builtin_elementwise_min(pku(a), pku(b))); }
__device__ __forceinline__ int pk_mad(int a, int b, int c) { return ipk(pku(a) * pku(b) + pku(c)); }
// Written as instructions: left to itself the compiler recognises min(x, 1) / small shifts of packed values as per-half
// compares and 16-bit scalar operations and un-packs them (v_cmp_ne_u16 + v_cndmask + v_perm per half).
template <int C>
__device__ __forceinline__ int pk_minu_c(int a)  // per half: min(a, C) unsigned, C an inline constant
{
    int r;
    asm("v_pk_min_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "n"(C));
    return r;
}
template <int C>
__device__ __forceinline__ int pk_shr_c(int a)  // per half: a >> C, logical
{
    int r;
    asm("v_pk_lshrrev_b16 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "n"(C), "v"(a));
    return r;
}
template <int C>
__device__ __forceinline__ int pk_addu_c(int a)  // per half: a + C, wrapping
{
    int r;
    asm("v_pk_add_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "n"(C));
    return r;
}
__device__ __forceinline__ int pk_mad_v(int a, int b, int c)  // per half: a * b + c, wrapping
{
    int r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int C>
__device__ __forceinline__ int pk_mad_c(int a, int c)  // per half: a * C + c, C an inline constant
{
    int r;
    asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "n"(C), "v"(c));
    return r;
}
__device__ __forceinline__ int pk_lo(int a) { return (int)(short)(a & 0xffff); }  // sign-extended halves
__device__ __forceinline__ int pk_hi(int a) { return a >> 16; }

// DEF: the gap penalties are abPOA's defaults (4/2, 24/1 => 3 + 5 bit deltas), known at compile time.
// CPT: columns per lane and step (4 or 8).  Eight halve the per-lane-step overhead (wave scan, cross-wave exchange, address
// arithmetic) per cell but need ~22 more vector registers (five waves per SIMD instead of seven); measured on config 3 the
// instruction count per cell does not drop and the step is 3 % slower, so only CPT = 4 is instantiated.
// H16: the row state (LDS window and HBM value rows) is an int16 plane of H relative to a per-row base (the
// predecessors' maximum) plus a byte plane of the gap deltas: 3 B per column.  Scores that come within 2 768 of the int16
// range stop the problem with POA_ST_RANGE and the host re-runs it with 32-bit words.
template <int NT, bool STAMP = false, bool DEF = false, int CPT = 4, bool H16 = false>
// Six waves per SIMD (80 VGPRs), not the seven that 21.6 KB of LDS per workgroup would allow: measured same-box
// (tests/prof_lib_ab.sh), 7 -> 6 / 5 / 4 waves per SIMD are all +1.5 % on config 3 (+2 % on configs 4 and 5) and equal among
// themselves -- the kernel is bound by VALU issue, and the allocator does better with eight more registers.
__global__ __launch_bounds__(NT, (CPT == 8 ? 5 : 6)) void k_poa_dp_pk(
    const poa_prob *__restrict__ probs, const char *__restrict__ queries, const uint4 *__restrict__ node_tab,
    const uint32_t *__restrict__ seq32, const uint32_t *__restrict__ preds, const uint32_t *__restrict__ sink_preds,
    poa_dev_params P, poa_row *rows, uint8_t *pool_arg, unsigned long long *pool_next_arg, uint64_t pool_size_arg,
    poa_out *__restrict__ outs, uint32_t lds_cols, uint32_t hg_cols, uint32_t win_mask, int g1bits_arg,
    uint8_t *__restrict__ tb_ops, uint32_t *__restrict__ tb_orow, uint32_t n_arenas, uint64_t arena_size,
    unsigned long long *arena_ctr, uint32_t *arena_flag, unsigned long long *stamps = nullptr)
{
    constexpr int QPT = CPT / 4;  // quads (one LDS int4 / one direction dword each) per lane and step
    constexpr int NW = NT / 64;
    constexpr int STEP = NT * CPT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // HG is a window of hg_cols columns addressed by (column & win_mask): a power of two smaller than the query when
    // the launch was sized for narrow bands (more workgroups per CU), or every column (win_mask = ~0).
    // The small exchange structures come first so that their addresses, and HG's, are compile-time constants.
    int4 *sX = (int4 *)smem;                  // [2][NW] {scan1, scan2, last1, last2} per wave
    int4 *sRed = sX + 2 * NW;                 // [NW] {row max, -leftmost, rightmost, 0} per wave
    int32_t *edgeW = (int32_t *)(sRed + NW);  // [2] (+2 pad)
    int4 *sRow = (int4 *)(edgeW + 4);         // [4] the row's parameters, written by wave 0 (see the row loop)
    int4 *sLead = sRow + 4;                   // [6] wave 0's allocator state and counters
    int32_t *sSink = (int32_t *)(sLead + 6);  // [1] (+3 pad) column qlen of a row that feeds the sink, parked by the lane that owns it
    constexpr int HDR = (3 * NW + 1 + 4 + 6 + 1) * 16;
    int32_t *HG = (int32_t *)(smem + HDR);                       // [hg_cols] (H << 8) | g            (32-bit storage)
    int16_t *H16a = (int16_t *)(smem + HDR);                     // [hg_cols] H - row base            (16-bit storage)
    uint8_t *G8a = smem + HDR + 2ull * hg_cols;                  // [hg_cols] g
    uint16_t *Qn = (uint16_t *)(smem + HDR + (H16 ? 3ull : 4ull) * hg_cols);  // [lds_cols / 4] four column codes per halfword
    const int edge_idx = (int)(edgeW - HG);  // edgeW addressed through HG, see phase 1
    constexpr int VB = H16 ? 3 : 4;  // bytes per cell of a value row in HBM
    // ---- accessors of the row state.  A "word" is (H << 8) | g whatever the storage; with 16-bit storage H is relative to
    // the base of the row the word belongs to (the consumer adds base differences).
    auto mk = [](int h, uint32_t g) -> int { return (int)(((uint32_t)h << 8) | (g & 255u)); };
    auto lds_quad = [&](int j) -> int4 {
        if constexpr (!H16) return *(const int4 *)(HG + (j & win_mask));
        else {
            const uint2 hh = *(const uint2 *)(H16a + (j & win_mask));
            const uint32_t gg = *(const uint32_t *)(G8a + (j & win_mask));
            return make_int4(mk((int)(hh.x << 16) >> 16, gg), mk((int)hh.x >> 16, gg >> 8), mk((int)(hh.y << 16) >> 16, gg >> 16),
                             mk((int)hh.y >> 16, gg >> 24));
        }
    };
    // column j of the row above, or the word the previous step parked in edgeW (edge != 0)
    auto lds_word = [&](int j, bool edge, int ebuf) -> int {
        if constexpr (!H16) {
            // one LDS read through an index (a pointer select would turn into a flat load, which also waits for the
            // outstanding global stores)
            int w = HG[edge ? edge_idx + ebuf : (j & win_mask)];
            asm volatile("" : "+v"(w));
            return w;
        } else {
            const int we = edgeW[ebuf];
            const int wl = mk((int)H16a[j & win_mask], (uint32_t)G8a[j & win_mask]);
            return edge ? we : wl;
        }
    };
    auto lds_store = [&](int j, int4 w) {
        if constexpr (!H16) *(int4 *)(HG + (j & win_mask)) = w;
        else {
            auto c16 = [](int wv) -> uint32_t { int h = wv >> 8; h = h < -32768 ? -32768 : h; return (uint32_t)h & 0xffffu; };
            *(uint2 *)(H16a + (j & win_mask)) = make_uint2(c16(w.x) | (c16(w.y) << 16), c16(w.z) | (c16(w.w) << 16));
            *(uint32_t *)(G8a + (j & win_mask)) = ((uint32_t)w.x & 255u) | (((uint32_t)w.y & 255u) << 8) | (((uint32_t)w.z & 255u) << 16) | ((uint32_t)w.w << 24);
        }
    };
    // value rows in HBM: `rowp` points at the row, Wr is its storage width, idx the column offset inside it
    auto hbm_quad = [&](const uint8_t *rowp, int Wr, int idx) -> int4 {
        if constexpr (!H16) return *(const int4 *)((const int32_t *)rowp + idx);
        else {
            const uint2 hh = *(const uint2 *)((const int16_t *)rowp + idx);
            const uint32_t gg = *(const uint32_t *)(rowp + 2 * (int64_t)Wr + idx);
            return make_int4(mk((int)(hh.x << 16) >> 16, gg), mk((int)hh.x >> 16, gg >> 8), mk((int)(hh.y << 16) >> 16, gg >> 16),
                             mk((int)hh.y >> 16, gg >> 24));
        }
    };
    auto hbm_word = [&](const uint8_t *rowp, int Wr, int idx) -> int {
        if constexpr (!H16) return ((const int32_t *)rowp)[idx];
        else return mk((int)((const int16_t *)rowp)[idx], (uint32_t)rowp[2 * (int64_t)Wr + idx]);
    };
    auto hbm_store = [&](uint8_t *rowp, int Wr, int idx, int4 w) {
        if constexpr (!H16) *(int4 *)((int32_t *)rowp + idx) = w;
        else {
            auto c16 = [](int wv) -> uint32_t { int h = wv >> 8; h = h < -32768 ? -32768 : h; return (uint32_t)h & 0xffffu; };
            *(uint2 *)((int16_t *)rowp + idx) = make_uint2(c16(w.x) | (c16(w.y) << 16), c16(w.z) | (c16(w.w) << 16));
            *(uint32_t *)(rowp + 2 * (int64_t)Wr + idx) = ((uint32_t)w.x & 255u) | (((uint32_t)w.y & 255u) << 8) | (((uint32_t)w.z & 255u) << 16) | ((uint32_t)w.w << 24);
        }
    };

    const uint64_t t_begin = __builtin_amdgcn_s_memrealtime();
    const poa_prob pb = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qlen = (int)pb.qlen;
    const char *query = queries + pb.q0;
    const uint4 *ntab = node_tab + pb.node0;
    const uint32_t *plist = preds + pb.pred0;
    poa_row *R = rows + pb.row0;

    // ---- where this problem's direction and value rows go.  Classic mode (n_arenas == 0): chunks of the launch's pool
    // segment, handed out by one atomic counter; the segment is recycled when the whole launch has been traced back.
    // Arena mode: the pool is cut into n_arenas equal arenas; a workgroup takes a free one, is the only user of its bump
    // counter, traces its problem back itself (tb_ops) and gives the arena back -- so launches of any size can follow
    // each other on several streams and the GPU stays full across them.
    uint8_t *pool = pool_arg;
    unsigned long long *pool_next = pool_next_arg;
    uint64_t pool_size = pool_size_arg;
    uint32_t arena = 0;
    if (n_arenas) {
        int got = -1;
        if (!(pb.flags & 1u)) {
            if (tid == 0) {
                uint32_t a = (uint32_t)(((uint64_t)blockIdx.x * 2654435761ull) % n_arenas);
                // every arena is held by a running workgroup that will give it back: wait (bounded, as a fail-safe)
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

    const int g1bits = DEF ? 3 : g1bits_arg;
    const int o1 = DEF ? 4 : P.o1, e1 = DEF ? 2 : P.e1, o2 = DEF ? 24 : P.o2, e2 = DEF ? 1 : P.e2;
    const int oe1 = o1 + e1, oe2 = o2 + e2;
    const int g1mask = (1 << g1bits) - 1;
    const int g2w = 8 - g1bits;
    const int bw = (int)pb.w;
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int seg) {
        if constexpr (STAMP) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (seg >= 0) tacc[seg] += t - tprev;
            tprev = t;
        }
    };

    // Wave 0 owns everything that is uniform per row and only needed to set the row up: the band pull, the pool
    // allocator, the row record, the counters.  It publishes the row's parameters in sRow; the other waves pick them up
    // after one LDS barrier instead of recomputing ~350 scalar instructions each.  Its own state (allocator cursors,
    // counters) lives in LDS (sLead) and is handled in vector registers inside the leader block, so that it does not
    // occupy scalar registers across the row loop (the kernel is short of them: every spilled SGPR costs v_writelane /
    // v_readlane instructions in all waves).
    const bool leader = wv == 0;
    struct lead_t {
        uint64_t dcur, dend, vcur, vendp, wide_scratch, cells, vcells;
        int maxw, failed;
        // value rows of node-end rows: a ring of ring_size bytes at ring_base (see the leader block)
        uint64_t ring_base;
        uint32_t ring_head, ring_size;
        // best sink candidate so far (rows are visited in the order of the sink list; the first maximum wins)
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
    auto take_chunk = [&](lead_t &L, uint64_t &cur, uint64_t &end) {  // wave 0 only
        unsigned long long bv = 0;
        if (lane == 0) bv = atomicAdd(pool_next, (unsigned long long)POA_CHUNK);
        const uint64_t b = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bv);
        if (b + POA_CHUNK > pool_size) L.failed = 1;
        cur = b;
        end = b + POA_CHUNK;
    };
    auto alloc = [&](lead_t &L, uint64_t &cur, uint64_t &end, uint64_t bytes) -> uint64_t {
        bytes = (bytes + 15ull) & ~15ull;
        if (cur + bytes > end) take_chunk(L, cur, end);
        uint64_t r = cur;
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
    // the branch-free interior path assumes every query base scores match or mismatch
    const bool q_plain = __syncthreads_or(non_acgt) == 0;

    if (leader) {
        lead_t L = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, POA_NEG, 0, 0};
        // two scratch value rows for rows wider than the LDS window (they only feed the row directly below)
        if (win_mask != 0xFFFFFFFFu) L.wide_scratch = alloc(L, L.vcur, L.vendp, 2ull * VB * lds_cols);
        // The value row of a node's last base is read by the first rows of that node's successors only.  No edge spans
        // more than ring_rows - 1 nodes except the ones the host marked as long-lived (those rows are kept for good), so a
        // ring of ring_rows slots of one worst-case row each never overwrites a row that is still needed -- and a problem holds ~1 MB of
        // value rows instead of ~28 MB, which is what lets twice as many problems share the pool.
        {
            const uint64_t maxrow = ((uint64_t)VB * (uint64_t)((qlen + 8) & ~3) + 15ull) & ~15ull;
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
    int prev_beg = 0, prev_end = -1, prev_lmax = 0, prev_rmax = 0;
    int prev_base = 0, prev_hmax = 0;  // (16-bit storage) base and absolute maximum of the row just computed
    int danger = 0;                    // (16-bit storage) this lane saw a score near the int16 range
    bool prev_lds = true;  // the row just computed is resident in the LDS window (false after a row wider than the window)
    uint32_t seq_word = 0, seq_word_idx = 0xFFFFFFFFu;
    bool stop = false, range_stop = false;

    for (uint32_t v = 0; v < pb.n_nodes && !stop; v++) {
    const uint4 nt = ntab[v];
    const uint32_t nlen = nt.y & 0xFFFFFFu;
    for (uint32_t tn = 0; tn < nlen && !stop; tn++) {
        POA_MARK("row_topo");
        // ---- every wave: the row's place in the graph
        const uint32_t r = nt.x + tn;
        const bool first = tn == 0 && v > 0;
        const bool last = tn + 1 == nlen;
        const bool is_sink = last && (nt.z >> 31) != 0;  // this row is a predecessor of the sink
        const int np = v == 0 ? 0 : (tn == 0 ? (int)(nt.y >> 24) : 1);
        const uint32_t ps = nt.w;
        uint8_t gb = 0;
        if (v > 0) {
            const uint32_t bi = r - 1;
            if ((bi & 3u) == 0 || (bi >> 2) != seq_word_idx) { seq_word_idx = bi >> 2; seq_word = seq32[(pb.seq0 >> 2) + seq_word_idx]; }
            gb = (uint8_t)(seq_word >> (8u * (bi & 3u)));
        }
        stamp(-1);
        // a predecessor is "near" when it is the row directly above AND that row is still in the LDS window
        bool far = r > 0 && !prev_lds;
        if (first) {
            if (np == 1) far |= ps != r - 1;
            else
                for (int t = 0; t < np; t++) far |= plist[ps + t] != r - 1;
        }
        if (__builtin_expect(far, 0)) __syncthreads();  // vmcnt(0) + barrier: the value rows / row records of far predecessors have landed
        const bool single = r > 0 && np == 1;   // one predecessor (the row above in LDS, or a far row in HBM)
        const uint32_t sp = first ? ps : r - 1;  // that predecessor
        const bool sp_near = sp == r - 1 && prev_lds;
        POA_MARK("row_leader");
        // ---- wave 0: band, pool space, row record
        if (leader) {
            lead_t L = lead_load();
            const int remain = (int)(nt.z & 0x3fffffffu) + (int)(nlen - 1 - tn);
            if (lane == 0) sSink[0] = 0;  // (rewritten in phase 2 by the lane that owns column qlen of a sink row)
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
                // kept for good: the source row (every root reads it) and rows that are read far ahead
                if (r == 0 || (nt.z & 0x40000000u)) voff = alloc(L, L.vcur, L.vendp, (uint64_t)VB * (uint64_t)W);
                else {
                    // fixed slots of one worst-case row (see k_poa_dp_t4)
                    voff = L.ring_base + (uint64_t)L.ring_head * L.ring_size;
                    L.ring_head = L.ring_head + 1 == pb.ring_rows ? 0 : L.ring_head + 1;
                }
            } else if (wide) voff = L.wide_scratch + (r & 1u) * (uint64_t)VB * lds_cols;
            int pbeg = prev_beg, pend = prev_end;
            uint64_t vpo = 0;
            // (16-bit storage) this row's base = the largest maximum among its predecessors; dlt = what to add to the
            // single predecessor's stored values to bring them into this row's frame
            int base = 0, dlt = 0;
            if constexpr (H16) {
                if (r > 0) {
                    if (!first) { base = prev_hmax; dlt = prev_base - base; }
                    else {
                        base = INT32_MIN;
                        const uint32_t *pl2 = plist;
                        int pb1 = 0;
                        for (int t = 0; t < np; t++) {
                            const uint32_t p = np == 1 ? ps : pl2[ps + t];
                            int hm, bs;
                            if (p == r - 1) { hm = prev_hmax; bs = prev_base; }
                            else {
                                hm = __builtin_amdgcn_readfirstlane(R[p].hmax);
                                bs = __builtin_amdgcn_readfirstlane(R[p].base);
                            }
                            base = hm > base ? hm : base;
                            pb1 = bs;
                        }
                        dlt = pb1 - base;  // meaningful for np == 1 only
                    }
                }
            }
            if (single && !sp_near) {
                // uniform values of a far row: read them into scalar registers right here, so that their s_waitcnt vmcnt
                // stays inside this branch
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
                    R[r].base = base;
                }
                sRow[0] = make_int4(beg, end, (int)(uint32_t)doff, (int)(uint32_t)(doff >> 32));
                sRow[1] = make_int4((int)(uint32_t)voff, (int)(uint32_t)(voff >> 32), pbeg, pend);
                sRow[2] = make_int4((int)(uint32_t)vpo, (int)(uint32_t)(vpo >> 32), L.failed, 0);
                sRow[3] = make_int4(base, dlt, 0, 0);
            }
        }
        POA_LDS_BARRIER();
        POA_MARK("row_pickup");
        // ---- every wave: pick the parameters up
        const int4 rw0 = sRow[0], rw1 = sRow[1], rw2 = sRow[2];
        const int beg = __builtin_amdgcn_readfirstlane(rw0.x), end = __builtin_amdgcn_readfirstlane(rw0.y);
        const uint64_t doff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rw0.w) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(rw0.z);
        const uint64_t voff = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rw1.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(rw1.x);
        const int pbeg = __builtin_amdgcn_readfirstlane(rw1.z), pend = __builtin_amdgcn_readfirstlane(rw1.w);
        const uint64_t vpo = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rw2.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(rw2.x);
        if (__builtin_amdgcn_readfirstlane(rw2.z)) { stop = true; break; }
        int base = 0, dlt8 = 0;  // (16-bit storage) row base; single-predecessor frame shift, in word units (<< 8)
        if constexpr (H16) {
            const int4 rw3 = sRow[3];
            base = __builtin_amdgcn_readfirstlane(rw3.x);
            dlt8 = __builtin_amdgcn_readfirstlane(rw3.y) * 256;
        }
        const int bal = beg & ~3;
        const int W = (end - bal + 1 + 3) & ~3;  // storage width / plane stride
        // A row wider than the LDS window (columns would alias) is not written to LDS: it keeps a value row in HBM like
        // a node-end row does (in one of two scratch rows unless it ends a node), and the row below reads it from there.
        const bool wide = (uint32_t)W + 8u > hg_cols;
        const bool keep = last || wide;
        uint8_t *Vrow = pool + voff;  // value row
        uint8_t *drow = pool + doff;
        const int gcode = gb == 'A' ? 0 : (gb == 'C' ? 1 : (gb == 'G' ? 2 : (gb == 'T' ? 3 : 4)));
        const int sc_eq = gcode == 4 ? 0 : P.match, sc_ne = gcode == 4 ? 0 : -P.mismatch;
        const int gsh = gcode & 3;            // bit of the one-hot column code that means "equal to this row's base"
        const int sc_mm = sc_eq - sc_ne;
        const uint8_t *Vp = pool + vpo;  // value row of a far single predecessor
        const int balp = pbeg & ~3;
        stamp(0);

        POA_MARK("row_steps");
        int carry1 = POA_IDENT, carry2 = POA_IDENT, left1 = POA_IDENT, left2 = POA_IDENT;
        int best = INT32_MIN, lpos = beg, rpos = beg;
        int zacc = -1;  // (16-bit storage) per half: min over the row's active cells of (H + 32000) as u16
        int buf = 0;
        for (int c0 = 0; c0 < W; c0 += STEP, buf ^= 1) {
            const int c = c0 + CPT * tid;
            const int j0 = bal + c;
            const bool lane_act = j0 <= end;
            const int nw_step = (W - c0 + 64 * CPT - 1) / (64 * CPT) < NW ? (W - c0 + 64 * CPT - 1) / (64 * CPT) : NW;
            const bool wave_act = wv < nw_step;
            // carried from phase 1 to phase 2, per cell: ht and  meta = hts | ofl << 2 | u1 << 8 | u2 << 16
            int ht[CPT], meta[CPT], pmeta[CPT];
            int agg1 = POA_IDENT, agg2 = POA_IDENT, alast1 = POA_IDENT, alast2 = POA_IDENT;
#pragma unroll
            for (int k = 0; k < CPT; k++) { ht[k] = POA_NEG; meta[k] = (o1 << 8) | (o2 << 16); pmeta[k] = 0; }
            // (16-bit storage) the same, two cells per register: Ht, flags (hts | of1 << 2 | of2 << 3), u1, u2
            int hq[2] = {(int)0x80008000, (int)0x80008000}, fq[2] = {0, 0};
            int u1q[2] = {0x00040004, 0x00040004}, u2q[2] = {0x00180018, 0x00180018};
            uint32_t qn = 0u;  // one-hot code nibbles of this lane's columns
            if (wave_act) {
#pragma unroll
                for (int q = 0; q < QPT; q++) qn |= (uint32_t)Qn[(j0 >> 2) + q] << (16 * q);
            }
            // Fast path of a single-predecessor row (every query base A/C/G/T): no per-cell band masks.  It needs every
            // ACTIVE cell of the wave, and the column to its left, inside the predecessor's band.  Two cheap patches let
            // the band's edge waves take it too:
            //  lp  the wave starts left of `beg` (only lane 0's first cells): those cells are computed from whatever
            //      the LDS holds and their Ht is then replaced by POA_IDENT, so they stay out of the max-plus scan;
            //  rp  the wave reaches past `end` or `pend`: words right of `pend` are replaced by (NEG, g = 0), which
            //      gives E = NEG exactly; allowed when at most column pend+1 is active (its M comes from pend), and
            //      only for LDS predecessors (the HBM row ends at pend).  Cells right of `end` are kept out of the
            //      row maximum and are not stored.
            const int jw0 = bal + c0 + 64 * CPT * wv, jw1 = jw0 + 64 * CPT - 1;
            const bool lp = jw0 < beg;
            const bool rp = jw1 > end || jw1 > pend;
            //  lq  the wave's first active column is not right of `pbeg` (the band did not move right, or moved left):
            //      words left of `pbeg` are replaced by (NEG, g = 0) and M is forced to NEG where column j-1 lies left of
            //      `pbeg`, exactly what the lean path computes for such cells.  LDS predecessors only (the HBM row starts
            //      at pbeg), 32-bit row state only.
            const bool lq = (lp ? beg : jw0) <= pbeg;
            const bool fastw = single && q_plain && wave_act && (!lq || (sp_near && !H16)) && (!rp || (sp_near && end <= pend + 1));
            const int base1 = e1 * j0, base2 = e2 * j0;  // the max-plus scan runs on lane-relative values in the fast path
            if constexpr (STAMP) tacc[7] += (wave_act ? (1ull << 42) : 0ull) + (fastw ? 1ull : 0ull) + ((fastw && (lp || rp)) ? (1ull << 21) : 0ull);
        POA_MARK("p1_fast");
            if (__builtin_expect(fastw, 1)) {
              if constexpr (H16) {
                // ---------------- interior path, phase 1, two cells per instruction
                static_assert(!H16 || (CPT == 4 && DEF), "the packed 16-bit path is written for 4 columns per lane and the default penalties");
                int hp[2];      // the predecessor's H pairs (cells 0,1 / 2,3), in ITS frame
                uint32_t gg;    // its four g bytes
                int hprev;      // its H at column j0 - 1
                if (__builtin_expect(sp_near, 1)) {
                    const uint2 hh = *(const uint2 *)(H16a + (j0 & win_mask));
                    gg = *(const uint32_t *)(G8a + (j0 & win_mask));
                    hp[0] = (int)hh.x; hp[1] = (int)hh.y;
                    if (tid == NT - 1) edgeW[buf] = ((int)hh.y >> 16) << 8;
                    const int he = edgeW[buf ^ 1] >> 8;
                    const int hl = (int)H16a[(j0 > 0 ? j0 - 1 : 0) & win_mask];
                    hprev = (tid == 0 && c0 > 0) ? he : hl;
                } else {
                    const int Wp = (pend - balp + 1 + 3) & ~3;
                    const int idx = j0 - balp;
                    const uint2 hh = *(const uint2 *)((const int16_t *)Vp + idx);
                    gg = *(const uint32_t *)(Vp + 2 * (int64_t)Wp + idx);
                    hprev = (int)((const int16_t *)Vp)[idx > 0 ? idx - 1 : 0];
                    hp[0] = (int)hh.x; hp[1] = (int)hh.y;
                    // consume the loads inside this branch (see the 32-bit path)
                    asm volatile("" : "+v"(hp[0]), "+v"(hp[1]), "+v"(gg), "+v"(hprev));
                }
                auto phase1h = [&](auto edge_c) {
                    constexpr bool EDGE = decltype(edge_c)::value;
                    constexpr int SENT = (int)0x80008000;
                    int gp[2] = {(int)__builtin_amdgcn_perm(0u, gg, 0x0c010c00u), (int)__builtin_amdgcn_perm(0u, gg, 0x0c030c02u)};
                    if constexpr (EDGE) {
                        if (rp) {  // columns right of pend: H = sentinel, g = 0
                            int d = j0 - pend - 1;
                            d = d < -64 ? -64 : (d > 64 ? 64 : d);
                            const int db = pk_bcast(d);
                            const int in0 = ipk(pk(pk_add(db, 0x00010000)) >> 15), in1 = ipk(pk(pk_add(db, 0x00030002)) >> 15);  // 0xffff = inside
                            hp[0] = (hp[0] & in0) | (SENT & ~in0); hp[1] = (hp[1] & in1) | (SENT & ~in1);
                            gp[0] &= in0; gp[1] &= in1;
                        }
                    }
                    const int dltp = pk_bcast(dlt8 >> 8), ned = pk_bcast(sc_ne + (dlt8 >> 8)), mmp = pk_bcast(sc_mm);
                    // M: the predecessor's columns j-1, plus match / mismatch (and the frame shift)
                    const int sp0 = (int)__builtin_amdgcn_perm((uint32_t)hp[0], (uint32_t)hprev, 0x05040100u);  // (h[-1], h0)
                    const int sp1 = (int)__builtin_amdgcn_perm((uint32_t)hp[1], (uint32_t)hp[0], 0x05040302u);  // (h1, h2)
                    const int xb = pk_bcast((int)(qn >> gsh));  // bit 4k of either half: cell k matches
                    const int eq0 = ipk(pku(xb) >> pku(0x00040000)) & 0x00010001, eq1 = ipk(pku(xb) >> pku(0x000c0008)) & 0x00010001;
                    const int m0 = pk_adds(sp0, pk_mad(eq0, mmp, ned)), m1 = pk_adds(sp1, pk_mad(eq1, mmp, ned));
                    const int hd[2] = {pk_adds(hp[0], dltp), pk_adds(hp[1], dltp)};
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int g1 = gp[q] & 0x00070007, g2 = pk_shr_c<3>(gp[q]);
                        const int ev1 = pk_subs(hd[q], g1), ev2 = pk_subs(hd[q], g2);
                        const int m = q ? m1 : m0;
                        const int h = pk_max(pk_max(m, ev1), ev2);
                        const int u1r = pk_subs(h, ev1);
                        const int t = pk_minu_c<1>(pk_subs(h, m)), e = pk_minu_c<1>(u1r);
                        const int hts = pk_mad_v(t, e, t);  // 0 = M, 1 = E1, 2 = E2 (first maximum wins)
                        u1q[q] = pk_min(u1r, 0x00040004);
                        u2q[q] = pk_min(pk_subs(h, ev2), 0x00180018);
                        const int of1 = pk_shr_c<3>(pk_addu_c<2>(g1)), of2 = pk_shr_c<5>(pk_addu_c<7>(g2));
                        fq[q] = pk_mad_c<8>(of2, pk_mad_c<4>(of1, hts));
                        hq[q] = h;
                    }
                    if constexpr (EDGE) {
                        if (lp) {  // lane 0's cells left of beg stay out of the scan
                            int d = j0 - beg;
                            d = d < -64 ? -64 : (d > 64 ? 64 : d);
                            const int db = pk_bcast(d);
                            const int out0 = ipk(pk(pk_add(db, 0x00010000)) >> 15), out1 = ipk(pk(pk_add(db, 0x00030002)) >> 15);  // 0xffff = left of beg
                            hq[0] = (SENT & out0) | (hq[0] & ~out0); hq[1] = (SENT & out1) | (hq[1] & ~out1);
                        }
                    }
                    // lane-relative a_k = Ht_k + e k  (16 bit), the lane's maximum and last value (32 bit, row-relative)
                    const int r1a = pk_add(hq[0], 0x00020000), r1b = pk_add(hq[1], 0x00060004);
                    const int r2a = pk_add(hq[0], 0x00010000), r2b = pk_add(hq[1], 0x00030002);
                    const int x1 = pk_max(r1a, r1b), x2 = pk_max(r2a, r2b);
                    const int ag1 = pk_lo(x1) > pk_hi(x1) ? pk_lo(x1) : pk_hi(x1);
                    const int ag2 = pk_lo(x2) > pk_hi(x2) ? pk_lo(x2) : pk_hi(x2);
                    agg1 = ag1 + base1; agg2 = ag2 + base2;
                    alast1 = pk_hi(r1b) + base1; alast2 = pk_hi(r2b) + base2;
                };
                if (__builtin_expect(lp || rp, 0)) phase1h(std::true_type{});
                else phase1h(std::false_type{});
              } else {
                // ---------------- interior path, phase 1
                int4 hv[QPT];
                int hprev;
                if (__builtin_expect(sp_near, 1)) {
#pragma unroll
                    for (int q = 0; q < QPT; q++) hv[q] = lds_quad(j0 + 4 * q);
                    if (tid == NT - 1) edgeW[buf] = hv[QPT - 1].w;
                    hprev = lds_word(j0 > 0 ? j0 - 1 : 0, tid == 0 && c0 > 0, buf ^ 1);
                    if constexpr (H16) {
#pragma unroll
                        for (int q = 0; q < QPT; q++) { hv[q].x += dlt8; hv[q].y += dlt8; hv[q].z += dlt8; hv[q].w += dlt8; }
                        hprev += dlt8;
                    }
                    hprev >>= 8;
                } else {
                    const int Wp = (pend - balp + 1 + 3) & ~3;
#pragma unroll
                    for (int q = 0; q < QPT; q++) hv[q] = hbm_quad(Vp, Wp, (j0 - balp) + 4 * q);
                    hprev = hbm_word(Vp, Wp, j0 - balp > 0 ? j0 - balp - 1 : 0);
                    if constexpr (H16) {
#pragma unroll
                        for (int q = 0; q < QPT; q++) { hv[q].x += dlt8; hv[q].y += dlt8; hv[q].z += dlt8; hv[q].w += dlt8; }
                        hprev += dlt8;
                    }
                    // consume the loads inside this branch: otherwise their s_waitcnt vmcnt lands in the code shared
                    // with near rows, where it would also wait for every outstanding direction / value store
#pragma unroll
                    for (int q = 0; q < QPT; q++) asm volatile("" : "+v"(hv[q].x), "+v"(hv[q].y), "+v"(hv[q].z), "+v"(hv[q].w));
                    asm volatile("" : "+v"(hprev));
                    hprev >>= 8;
                }
                stamp(6);
                // the pure interior variant and the edge variant (with the lp / rp patches) are separate instantiations
                auto phase1 = [&](auto edge_c) {
                    constexpr bool EDGE = decltype(edge_c)::value;
                    int wj[CPT];
#pragma unroll
                    for (int q = 0; q < QPT; q++) { wj[4 * q] = hv[q].x; wj[4 * q + 1] = hv[q].y; wj[4 * q + 2] = hv[q].z; wj[4 * q + 3] = hv[q].w; }
                    if constexpr (EDGE) {
                        if (rp) {
#pragma unroll
                            for (int k = 0; k < CPT; k++) wj[k] = j0 + k > pend ? (int)((uint32_t)POA_NEG << 8) : wj[k];
                        }
                        if (lq) {
#pragma unroll
                            for (int k = 0; k < CPT; k++) wj[k] = j0 + k < pbeg ? (int)((uint32_t)POA_NEG << 8) : wj[k];
                        }
                    }
                    const uint32_t eqb = qn >> gsh;
#pragma unroll
                    for (int k = 0; k < CPT; k++) {
                        const int hj = wj[k] >> 8;
                        const int g1 = wj[k] & g1mask, g2 = (int)__builtin_amdgcn_ubfe((uint32_t)wj[k], (uint32_t)g1bits, (uint32_t)g2w);
                        int m = (hprev + sc_ne) + (int)((eqb >> (4 * k)) & 1u) * sc_mm;
                        if constexpr (EDGE) {
                            if (lq) m = j0 + k - 1 < pbeg ? POA_NEG : m;
                        }
                        const int ev1 = hj - g1, ev2 = hj - g2;
                        const int me = m > ev1 ? m : ev1;
                        const int h = me > ev2 ? me : ev2;
                        const int hts = ev2 > me ? 2 : (ev1 > m ? 1 : 0);
                        const int ofl = (g1 == oe1 ? 4 : 0) | (g2 == oe2 ? 8 : 0);
                        int u1 = h - ev1; u1 = u1 < o1 ? u1 : o1;
                        int u2 = h - ev2; u2 = u2 < o2 ? u2 : o2;
                        ht[k] = h;
                        meta[k] = hts | ofl | (u1 << 8) | (u2 << 16);
                        hprev = hj;
                    }
                    if constexpr (EDGE) {
                        if (lp) {
#pragma unroll
                            for (int k = 0; k < CPT; k++) ht[k] = j0 + k < beg ? POA_IDENT : ht[k];
                        }
                    }
                    int ag1 = POA_IDENT, ag2 = POA_IDENT;
#pragma unroll
                    for (int k = 0; k < CPT; k++) {
                        const int r1 = ht[k] + e1 * k, r2 = ht[k] + e2 * k;
                        ag1 = r1 > ag1 ? r1 : ag1;
                        ag2 = r2 > ag2 ? r2 : ag2;
                        if (k == CPT - 1) { alast1 = r1 + base1; alast2 = r2 + base2; }
                    }
                    agg1 = ag1 + base1;
                    agg2 = ag2 + base2;
                };
                if (__builtin_expect(lp || rp || lq, 0)) phase1(std::true_type{});
                else phase1(std::false_type{});
              }
        POA_MARK("p1_lean");
            } else if (wave_act && single) {
                // ---------------- lean path, phase 1
                // Cold path.  Its band limits go through an empty asm so that everything derived from them (spans, masks,
                // predecessor widths) is computed in here: hoisted out of the step loop it would sit in scalar registers
                // the hot path is short of, i.e. in v_writelane / v_readlane pairs executed by every wave and row.
                int pbeg_ = pbeg, pend_ = pend, beg_ = beg, end_ = end, balp_ = balp, gsh_ = gsh;
                asm volatile("" : "+s"(pbeg_), "+s"(pend_), "+s"(beg_), "+s"(end_), "+s"(balp_), "+s"(gsh_));
                const int pbeg = pbeg_, pend = pend_, beg = beg_, end = end_, balp = balp_, gsh = gsh_;
                const unsigned span = (unsigned)(end - beg);
                int wj[CPT], wm0;
                const unsigned pspan = (unsigned)(pend - pbeg);
                if (sp_near) {
#pragma unroll
                    for (int q = 0; q < QPT; q++) {
                        const int4 hv = lds_quad(j0 + 4 * q);
                        wj[4 * q] = hv.x; wj[4 * q + 1] = hv.y; wj[4 * q + 2] = hv.z; wj[4 * q + 3] = hv.w;
                    }
                    if (tid == NT - 1) edgeW[buf] = wj[CPT - 1];
                    wm0 = lds_word(j0 > 0 ? j0 - 1 : 0, tid == 0 && c0 > 0, buf ^ 1);
                } else {
                    const int idx = j0 - balp;
                    const int Wp = (pend - balp + 1 + 3) & ~3;
#pragma unroll
                    for (int q = 0; q < QPT; q++) {
                        int4 hv = make_int4(0, 0, 0, 0);
                        if (idx + 4 * q >= 0 && idx + 4 * q < Wp) hv = hbm_quad(Vp, Wp, idx + 4 * q);
                        wj[4 * q] = hv.x; wj[4 * q + 1] = hv.y; wj[4 * q + 2] = hv.z; wj[4 * q + 3] = hv.w;
                    }
                    wm0 = (idx >= 1 && idx - 1 < Wp) ? hbm_word(Vp, Wp, idx - 1) : 0;
                    // wait here, not in shared code
#pragma unroll
                    for (int k = 0; k < CPT; k++) asm volatile("" : "+v"(wj[k]));
                    asm volatile("" : "+v"(wm0));
                }
                if constexpr (H16) {
#pragma unroll
                    for (int k = 0; k < CPT; k++) wj[k] += dlt8;
                    wm0 += dlt8;
                }
                bool inprev = j0 >= 1 && (unsigned)(j0 - 1 - pbeg) <= pspan;
#pragma unroll
                for (int k = 0; k < CPT; k++) {
                    const int j = j0 + k;
                    const bool inj = (unsigned)(j - pbeg) <= pspan;
                    const bool actk = (unsigned)(j - beg) <= span;
                    const int qc = (int)((qn >> (4 * k)) & 15u);
                    const int s = ((qc >> gsh) & 1) ? sc_eq : (qc == 0 ? 0 : sc_ne);
                    const int wm = k == 0 ? wm0 : wj[k - 1];
                    const int hj = wj[k] >> 8, g = wj[k] & 255;
                    const int g1 = g & g1mask, g2 = g >> g1bits;
                    const int m = inprev ? (wm >> 8) + s : POA_NEG;
                    const int ev1 = inj ? hj - g1 : POA_NEG;
                    const int ev2 = inj ? hj - g2 : POA_NEG;
                    const int me = m > ev1 ? m : ev1;
                    const int h = me > ev2 ? me : ev2;
                    const int hts = ev2 > me ? 2 : (ev1 > m ? 1 : 0);
                    const int ofl = (g1 == oe1 ? 1 : 0) | (g2 == oe2 ? 2 : 0);
                    int u1 = h - ev1; u1 = u1 < o1 ? u1 : o1;
                    int u2 = h - ev2; u2 = u2 < o2 ? u2 : o2;
                    ht[k] = h;
                    meta[k] = hts | (ofl << 2) | (u1 << 8) | (u2 << 16);
                    const int a1 = actk ? h + e1 * j : POA_IDENT, a2 = actk ? h + e2 * j : POA_IDENT;
                    agg1 = a1 > agg1 ? a1 : agg1;
                    agg2 = a2 > agg2 ? a2 : agg2;
                    if (k == CPT - 1) { alast1 = a1; alast2 = a2; }
                    inprev = inj;
                }
        POA_MARK("p1_general");
            } else if (wave_act) {
                // ---------------- general path, phase 1: the source row and rows with several predecessors
                // (cold path: see the note in the lean path)
                int beg_ = beg, end_ = end, gsh_ = gsh;
                asm volatile("" : "+s"(beg_), "+s"(end_), "+s"(gsh_));
                const int beg = beg_, end = end_, gsh = gsh_;
                const unsigned span = (unsigned)(end - beg);
                if (r == 0) {
#pragma unroll
                    for (int k = 0; k < CPT; k++) ht[k] = (j0 + k == 0) ? 0 : POA_NEG;
                } else if (lane_act) {
                    int m[CPT], ev1[CPT], ev2[CPT], hts[CPT], ofl[CPT];
#pragma unroll
                    for (int k = 0; k < CPT; k++) { m[k] = POA_NEG; ev1[k] = POA_NEG; ev2[k] = POA_NEG; hts[k] = 0; ofl[k] = 0; }
                    for (int t = 0; t < np; t++) {
                        const uint32_t p = plist[ps + t];
                        int wj[CPT], wm0 = 0;
                        int bp, ep;
                        int pd8 = 0;  // (16-bit storage) frame shift of this predecessor, in word units
                        if (p == r - 1 && prev_lds) {
                            bp = prev_beg; ep = prev_end;
                            if constexpr (H16) pd8 = (prev_base - base) * 256;
#pragma unroll
                            for (int q = 0; q < QPT; q++) {
                                const int4 hv = lds_quad(j0 + 4 * q);
                                wj[4 * q] = hv.x; wj[4 * q + 1] = hv.y; wj[4 * q + 2] = hv.z; wj[4 * q + 3] = hv.w;
                            }
                            if (tid == NT - 1) edgeW[buf] = wj[CPT - 1];
                            wm0 = lds_word(j0 > 0 ? j0 - 1 : 0, tid == 0 && c0 > 0, buf ^ 1);
                        } else {
                            bp = __builtin_amdgcn_readfirstlane(R[p].beg);
                            ep = __builtin_amdgcn_readfirstlane(R[p].end);
                            const uint64_t vo = R[p].voff;
                            const uint64_t vos = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(vo >> 32)) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vo);
                            const uint8_t *Vq = pool + vos;
                            if constexpr (H16) pd8 = (__builtin_amdgcn_readfirstlane(R[p].base) - base) * 256;
                            const int balq = bp & ~3;
                            const int Wq = (ep - balq + 1 + 3) & ~3;
                            const int idx = j0 - balq;
#pragma unroll
                            for (int q = 0; q < QPT; q++) {
                                int4 hv = make_int4(0, 0, 0, 0);
                                if (idx + 4 * q >= 0 && idx + 4 * q < Wq) hv = hbm_quad(Vq, Wq, idx + 4 * q);
                                wj[4 * q] = hv.x; wj[4 * q + 1] = hv.y; wj[4 * q + 2] = hv.z; wj[4 * q + 3] = hv.w;
                            }
                            wm0 = (idx >= 1 && idx - 1 < Wq) ? hbm_word(Vq, Wq, idx - 1) : 0;
                            // wait here, not in shared code
#pragma unroll
                            for (int k = 0; k < CPT; k++) asm volatile("" : "+v"(wj[k]));
                            asm volatile("" : "+v"(wm0));
                        }
                        if constexpr (H16) {
#pragma unroll
                            for (int k = 0; k < CPT; k++) wj[k] += pd8;
                            wm0 += pd8;
                        }
                        const unsigned pspan = (unsigned)(ep - bp);
#pragma unroll
                        for (int k = 0; k < CPT; k++) {
                            const int j = j0 + k;
                            const bool actk = (unsigned)(j - beg) <= span;
                            const int qc = (int)((qn >> (4 * k)) & 15u);
                            const int s = ((qc >> gsh) & 1) ? sc_eq : (qc == 0 ? 0 : sc_ne);
                            const int wm = k == 0 ? wm0 : wj[k - 1];
                            if (actk && j >= 1 && (unsigned)(j - 1 - bp) <= pspan) {
                                const int cnd = (wm >> 8) + s;
                                if (cnd > m[k]) { m[k] = cnd; pmeta[k] = (pmeta[k] & ~255) | t; }
                            }
                            if (actk && (unsigned)(j - bp) <= pspan) {
                                const int hj = wj[k] >> 8, g = wj[k] & 255;
                                const int g1 = g & g1mask, g2 = g >> g1bits;
                                const int c1 = hj - g1;
                                if (c1 > ev1[k]) { ev1[k] = c1; pmeta[k] = (pmeta[k] & ~0xff00) | (t << 8); ofl[k] = (ofl[k] & 2) | (g1 == oe1 ? 1 : 0); }
                                const int c2 = hj - g2;
                                if (c2 > ev2[k]) { ev2[k] = c2; pmeta[k] = (pmeta[k] & ~0xff0000) | (t << 16); ofl[k] = (ofl[k] & 1) | (g2 == oe2 ? 2 : 0); }
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < CPT; k++) {
                        int h = m[k];
                        if (ev1[k] > h) { h = ev1[k]; hts[k] = 1; }
                        if (ev2[k] > h) { h = ev2[k]; hts[k] = 2; }
                        int u1 = h - ev1[k]; u1 = u1 < o1 ? u1 : o1;
                        int u2 = h - ev2[k]; u2 = u2 < o2 ? u2 : o2;
                        ht[k] = h;
                        meta[k] = hts[k] | (ofl[k] << 2) | (u1 << 8) | (u2 << 16);
                    }
                }
#pragma unroll
                for (int k = 0; k < CPT; k++) {
                    const int j = j0 + k;
                    const bool actk = (unsigned)(j - beg) <= span;
                    const int a1 = actk ? ht[k] + e1 * j : POA_IDENT, a2 = actk ? ht[k] + e2 * j : POA_IDENT;
                    agg1 = a1 > agg1 ? a1 : agg1;
                    agg2 = a2 > agg2 ? a2 : agg2;
                    if (k == CPT - 1) { alast1 = a1; alast2 = a2; }
                }
            }
            if constexpr (H16) {
                // the cold paths worked on 32-bit cells: hand their result over in the packed form, so that only that is
                // alive across the barrier
                if (!fastw && wave_act) {
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        auto c16 = [](int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); };
                        hq[q] = pk_make(c16(ht[2 * q]), c16(ht[2 * q + 1]));
                        fq[q] = pk_make(meta[2 * q] & 15, meta[2 * q + 1] & 15);
                        u1q[q] = pk_make((meta[2 * q] >> 8) & 255, (meta[2 * q + 1] >> 8) & 255);
                        u2q[q] = pk_make((meta[2 * q] >> 16) & 255, (meta[2 * q + 1] >> 16) & 255);
                    }
                }
            }
        POA_MARK("scan");
            stamp(1);
            int i1 = POA_IDENT, i2 = POA_IDENT;
            if (wave_act) {
                i1 = poa_wave_scan_max(agg1);
                i2 = poa_wave_scan_max(agg2);
            }
            if (lane == 63) sX[buf * NW + wv] = make_int4(i1, i2, alast1, alast2);
            stamp(2);
            POA_LDS_BARRIER();
            stamp(3);
        POA_MARK("exchange");
            // lane q < NW picks up wave q's totals; an 8-lane DPP scan gives every wave its prefix and the step total
            int4 xw = make_int4(INT32_MIN, INT32_MIN, POA_IDENT, POA_IDENT);
            if (lane < NW) xw = sX[buf * NW + lane];
            int t1 = xw.x, t2 = xw.y, t;
            t = poa_dpp<0x111, 0xf>(INT32_MIN, t1); t1 = t > t1 ? t : t1;
            t = poa_dpp<0x111, 0xf>(INT32_MIN, t2); t2 = t > t2 ? t : t2;
            if (NW > 2) {
                t = poa_dpp<0x112, 0xf>(INT32_MIN, t1); t1 = t > t1 ? t : t1;
                t = poa_dpp<0x112, 0xf>(INT32_MIN, t2); t2 = t > t2 ? t : t2;
            }
            if (NW > 4) {
                t = poa_dpp<0x114, 0xf>(INT32_MIN, t1); t1 = t > t1 ? t : t1;
                t = poa_dpp<0x114, 0xf>(INT32_MIN, t2); t2 = t > t2 ? t : t2;
            }
            const int tot1 = __builtin_amdgcn_readlane(t1, NW - 1), tot2 = __builtin_amdgcn_readlane(t2, NW - 1);
            const int all1 = tot1 > carry1 ? tot1 : carry1, all2 = tot2 > carry2 ? tot2 : carry2;
            const int nleft1 = __builtin_amdgcn_readlane(xw.z, nw_step - 1), nleft2 = __builtin_amdgcn_readlane(xw.w, nw_step - 1);
            if (wave_act) {
                int x1 = poa_wave_shr1(i1), x2 = poa_wave_shr1(i2);
                int la1 = poa_wave_shr1(alast1), la2 = poa_wave_shr1(alast2);
                int pre1 = carry1, pre2 = carry2, pl1 = left1, pl2 = left2;
                if (wv > 0) {
                    const int p1 = __builtin_amdgcn_readlane(t1, wv - 1), p2 = __builtin_amdgcn_readlane(t2, wv - 1);
                    pre1 = p1 > pre1 ? p1 : pre1;
                    pre2 = p2 > pre2 ? p2 : pre2;
                    pl1 = __builtin_amdgcn_readlane(xw.z, wv - 1);
                    pl2 = __builtin_amdgcn_readlane(xw.w, wv - 1);
                }
                if (lane == 0) { la1 = pl1; la2 = pl2; }
                int run1 = pre1 > x1 ? pre1 : x1;
                int run2 = pre2 > x2 ? pre2 : x2;
        POA_MARK("p2_fast");
                if (__builtin_expect(fastw, 1)) {
                  if constexpr (H16) {
                    // ---------------- interior path, phase 2, two cells per instruction
                    auto phase2h = [&](auto edge_c) {
                        constexpr bool EDGE = decltype(edge_c)::value;
                        constexpr int SENT = (int)0x80008000;
                        auto c16 = [](int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); };
                        // lane-relative a_k again (cheaper to redo than to keep), and the in-lane part of the max-plus scan:
                        // R_k = max(Rin, a_0 .. a_{k-1}) for the four cells, as two pairs
                        int f[2][2], t[2][2];  // [gap][pair]: F candidates; 0 where F would be opened from the previous column
#pragma unroll
                        for (int g = 0; g < 2; g++) {
                            const int A0 = pk_add(hq[0], g ? 0x00010000 : 0x00020000), A1 = pk_add(hq[1], g ? 0x00030002 : 0x00060004);
                            const int Rin = c16((g ? run2 : run1) - (g ? base2 : base1)), Lin = c16((g ? la2 : la1) - (g ? base2 : base1));
                            const int R01 = pk_max(pk_bcast(Rin), (int)(((uint32_t)A0 << 16) | 0x8000u));           // (Rin, max(Rin, a0))
                            const int M01 = pk_max(R01, A0);                                                      // (.., max(Rin, a0, a1))
                            const int X2 = (int)__builtin_amdgcn_perm((uint32_t)M01, (uint32_t)M01, 0x07060706u);  // both = max(Rin, a0, a1)
                            const int R23 = pk_max(X2, (int)(((uint32_t)A1 << 16) | 0x8000u));                      // (X, max(X, a2))
                            f[g][0] = pk_subs(R01, g ? 0x00190018 : 0x00060004);  // - (o + e k)
                            f[g][1] = pk_subs(R23, g ? 0x001b001a : 0x000a0008);
                            const int L01 = (int)__builtin_amdgcn_perm((uint32_t)A0, (uint32_t)Lin, 0x05040100u);  // (Lin, a0)
                            const int L23 = (int)__builtin_amdgcn_perm((uint32_t)A1, (uint32_t)A0, 0x05040302u);   // (a1, a2)
                            t[g][0] = pk_minu_c<1>(pk_sub(R01, L01));
                            t[g][1] = pk_minu_c<1>(pk_sub(R23, L23));
                        }
                        int hh[2], gq[2], cq[2];
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            const int hf = pk_max(hq[q], f[0][q]);
                            const int h = pk_max(hf, f[1][q]);
                            const int d1 = pk_minu_c<1>(pk_sub(hf, hq[q])), d2 = pk_minu_c<1>(pk_sub(h, hf));
                            // direction byte: flags | F1 chosen << 4 | F2 chosen << 5 | F1 opened << 6 | F2 opened << 7
                            // (bits 6,7 are written inverted by this kernel -- 1 = NOT opened -- the traceback is told so)
                            int c = pk_mad_c<16>(pk_mad_c<2>(d2, d1), fq[q]);
                            c = pk_mad_c<64>(pk_mad_c<2>(t[1][q], t[0][q]), c);
                            cq[q] = c;
                            const int dh = pk_subs(h, hq[q]);
                            const int b1 = pk_min(pk_adds(u1q[q], dh), 0x00040004), b2 = pk_min(pk_adds(u2q[q], dh), 0x00180018);
                            gq[q] = pk_addu_c<10>(pk_mad_c<8>(b2, b1));  // (b1 + e1) | (b2 + e2) << 3
                            hh[q] = h;
                        }
                        // row maximum and the range check run on the active cells only
                        int hb[2] = {hh[0], hh[1]};
                        if constexpr (EDGE) {
                            if (rp) {
                                int d = j0 - end - 1;
                                d = d < -64 ? -64 : (d > 64 ? 64 : d);
                                const int db = pk_bcast(d);
                                const int in0 = ipk(pk(pk_add(db, 0x00010000)) >> 15), in1 = ipk(pk(pk_add(db, 0x00030002)) >> 15);
                                hb[0] = (hb[0] & in0) | (SENT & ~in0); hb[1] = (hb[1] & in1) | (SENT & ~in1);
                            }
                        }
                        zacc = pk_minu(zacc, pk_minu(pk_add(hb[0], 0x7d007d00), pk_add(hb[1], 0x7d007d00)));  // + 32000
                        {
                            const int pm = pk_max(hb[0], hb[1]);
                            const int m4p = pk_max(pm, (int)__builtin_amdgcn_perm((uint32_t)pm, (uint32_t)pm, 0x05040706u));  // both halves = max of 4
                            const int m4 = pk_lo(m4p);
                            if (m4 >= best) {
                                // which of the four cells hold it: bit k of nm
                                const int z0 = pk_minu_c<1>(pk_sub(m4p, hb[0])), z1 = pk_minu_c<1>(pk_sub(m4p, hb[1]));
                                const uint32_t mb = (uint32_t)z0 | ((uint32_t)z1 << 2);
                                const uint32_t nm = ((mb | (mb >> 15)) & 15u) ^ 15u;
                                const int kf = __builtin_ctz(nm), kl = 31 - __builtin_clz(nm);
                                if (m4 > best) { best = m4; lpos = j0 + kf; }
                                rpos = j0 + kl;
                            }
                        }
                        if (__builtin_expect(is_sink, 0)) {
                            const int kq = qlen - j0;
                            if (kq >= 0 && kq < 4) {
                                const int hv_ = kq < 2 ? hh[0] : hh[1];
                                sSink[0] = ((kq & 1) ? (hv_ >> 16) : pk_lo(hv_)) << 8;
                            }
                        }
                        if (!EDGE || lane_act) {
                            const uint32_t gb4 = __builtin_amdgcn_perm((uint32_t)gq[1], (uint32_t)gq[0], 0x06040200u);
                            const uint32_t cb4 = __builtin_amdgcn_perm((uint32_t)cq[1], (uint32_t)cq[0], 0x06040200u);
                            if (!wide) {
                                *(uint2 *)(H16a + (j0 & win_mask)) = make_uint2((uint32_t)hh[0], (uint32_t)hh[1]);
                                *(uint32_t *)(G8a + (j0 & win_mask)) = gb4;
                            }
                            *(uint32_t *)(drow + c) = cb4;
                            if (keep) {
                                *(uint2 *)((int16_t *)Vrow + c) = make_uint2((uint32_t)hh[0], (uint32_t)hh[1]);
                                *(uint32_t *)(Vrow + 2 * (int64_t)W + c) = gb4;
                            }
                        }
                    };
                    if (__builtin_expect(lp || rp, 0)) phase2h(std::true_type{});
                    else phase2h(std::false_type{});
                  } else {
                    // ---------------- interior path, phase 2
                    auto phase2 = [&](auto edge_c) {
                        constexpr bool EDGE = decltype(edge_c)::value;
                        int wv4[CPT], codev[CPT];
                        int R1 = run1 - base1, R2 = run2 - base2, L1 = la1 - base1, L2 = la2 - base2;
#pragma unroll
                        for (int k = 0; k < CPT; k++) {
                            const int j = j0 + k;
                            const int f1 = R1 - (o1 + e1 * k), f2 = R2 - (o2 + e2 * k);
                            const int fo = (R1 == L1 ? 64 : 0) | (R2 == L2 ? 128 : 0);
                            const int hf = ht[k] > f1 ? ht[k] : f1;
                            const int h = hf > f2 ? hf : f2;
                            const int fsel = f2 > hf ? 32 : (f1 > ht[k] ? 16 : 0);
                            codev[k] = (meta[k] & 15) | fsel | fo;
                            const int dh = h - ht[k];
                            int dd1 = ((meta[k] >> 8) & 255) + dh; dd1 = (dd1 < o1 ? dd1 : o1) + e1;
                            int dd2 = ((meta[k] >> 16) & 255) + dh; dd2 = (dd2 < o2 ? dd2 : o2) + e2;
                            wv4[k] = (int)(((uint32_t)h << 8) | (uint32_t)(dd1 | (dd2 << g1bits)));
                            int hb = h;
                            if constexpr (EDGE) hb = j > end ? INT32_MIN : h;
                            if constexpr (H16) {
                                // a real score must stay above -30 000; the band-edge sentinels sit at or below -32 000
                                const bool zone = (unsigned)(h + 32000) < 2000u;
                                if constexpr (EDGE) danger |= (zone && j >= beg && j <= end) ? 1 : 0;
                                else danger |= zone ? 1 : 0;
                            }
                            if (hb > best) { best = hb; lpos = j; rpos = j; }
                            else if (hb == best) rpos = j;
                            L1 = ht[k] + e1 * k; L2 = ht[k] + e2 * k;
                            R1 = L1 > R1 ? L1 : R1;
                            R2 = L2 > R2 ? L2 : R2;
                        }
                        if (__builtin_expect(is_sink, 0)) {
                            const int kq = qlen - j0;
#pragma unroll
                            for (int k = 0; k < CPT; k++)
                                if (kq == k) sSink[0] = wv4[k];
                        }
#pragma unroll
                        for (int q = 0; q < QPT; q++) {
                            if (!EDGE || j0 + 4 * q <= end) {
                                const int4 wq = make_int4(wv4[4 * q], wv4[4 * q + 1], wv4[4 * q + 2], wv4[4 * q + 3]);
                                if (!wide) lds_store(j0 + 4 * q, wq);
                                *(uint32_t *)(drow + c + 4 * q) = (uint32_t)codev[4 * q] | ((uint32_t)codev[4 * q + 1] << 8) |
                                                                 ((uint32_t)codev[4 * q + 2] << 16) | ((uint32_t)codev[4 * q + 3] << 24);
                                if (keep) hbm_store(Vrow, W, c + 4 * q, wq);
                            }
                        }
                    };
                    if (__builtin_expect(lp || rp, 0)) phase2(std::true_type{});
                    else phase2(std::false_type{});
                  }
        POA_MARK("p2_slow");
                } else if (lane_act) {
                    int beg_ = beg, end_ = end;  // (cold path: see the note in phase 1)
                    asm volatile("" : "+s"(beg_), "+s"(end_));
                    const int beg = beg_;
                    const unsigned span = (unsigned)(end_ - beg_);
                    int wv4[CPT], codev[CPT];
                    if constexpr (H16) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int q = k >> 1;
                            auto half = [&](int v) { return (k & 1) ? (v >> 16) : pk_lo(v); };
                            ht[k] = half(hq[q]);
                            meta[k] = (half(fq[q]) & 15) | ((half(u1q[q]) & 255) << 8) | ((half(u2q[q]) & 255) << 16);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < CPT; k++) {
                        const int j = j0 + k;
                        const bool actk = (unsigned)(j - beg) <= span;
                        // at the first column run1/run2 are still POA_IDENT, which keeps F below everything (no special case)
                        const int f1 = run1 - (o1 + e1 * j), f2 = run2 - (o2 + e2 * j);
                        const int fo = (run1 == la1 ? 64 : 0) | (run2 == la2 ? 128 : 0);
                        const int hf = ht[k] > f1 ? ht[k] : f1;
                        const int h = hf > f2 ? hf : f2;
                        const int fsel = f2 > hf ? 32 : (f1 > ht[k] ? 16 : 0);
                        codev[k] = ((meta[k] & 15) | fsel | fo) ^ (H16 ? 0xC0 : 0);
                        const int dh = h - ht[k];
                        int dd1 = ((meta[k] >> 8) & 255) + dh; dd1 = (dd1 < o1 ? dd1 : o1) + e1;
                        int dd2 = ((meta[k] >> 16) & 255) + dh; dd2 = (dd2 < o2 ? dd2 : o2) + e2;
                        wv4[k] = (int)(((uint32_t)h << 8) | (uint32_t)(dd1 | (dd2 << g1bits)));
                        const int hb = actk ? h : INT32_MIN;
                        if (hb > best) { best = hb; lpos = j; rpos = j; }
                        else if (actk && hb == best) rpos = j;
                        if constexpr (H16) danger |= (actk && (unsigned)(h + 32000) < 2000u) ? 1 : 0;
                        const int a1 = actk ? ht[k] + e1 * j : POA_IDENT, a2 = actk ? ht[k] + e2 * j : POA_IDENT;
                        run1 = a1 > run1 ? a1 : run1;
                        run2 = a2 > run2 ? a2 : run2;
                        la1 = actk ? a1 : la1; la2 = actk ? a2 : la2;
                    }
                    if (__builtin_expect(is_sink, 0)) {
                        const int kq = qlen - j0;
#pragma unroll
                        for (int k = 0; k < CPT; k++)
                            if (kq == k) sSink[0] = wv4[k];
                    }
                    int Wl = (end_ - bal + 1 + 3) & ~3;  // (recomputed behind the barrier above so that the plane addresses are
                                                        //  not hoisted out of the row's step loop into scalar registers)
#pragma unroll
                    for (int q = 0; q < QPT; q++) {
                        if (j0 + 4 * q > end_) continue;  // beyond the row's storage
                        const int4 wq = make_int4(wv4[4 * q], wv4[4 * q + 1], wv4[4 * q + 2], wv4[4 * q + 3]);
                        const int cq = c + 4 * q;
                        if (!wide) lds_store(j0 + 4 * q, wq);
                        *(uint32_t *)(drow + cq) = (uint32_t)codev[4 * q] | ((uint32_t)codev[4 * q + 1] << 8) |
                                                   ((uint32_t)codev[4 * q + 2] << 16) | ((uint32_t)codev[4 * q + 3] << 24);
                        if (keep) hbm_store(Vrow, Wl, cq, wq);
                        if (__builtin_expect(np > 1, 0)) {
                            const int *pm = pmeta + 4 * q;
                            *(uint32_t *)(drow + (uint64_t)Wl + cq) = (uint32_t)(pm[0] & 255) | ((uint32_t)(pm[1] & 255) << 8) | ((uint32_t)(pm[2] & 255) << 16) | ((uint32_t)(pm[3] & 255) << 24);
                            *(uint32_t *)(drow + 2ull * Wl + cq) = (uint32_t)((pm[0] >> 8) & 255) | ((uint32_t)((pm[1] >> 8) & 255) << 8) | ((uint32_t)((pm[2] >> 8) & 255) << 16) | ((uint32_t)((pm[3] >> 8) & 255) << 24);
                            *(uint32_t *)(drow + 3ull * Wl + cq) = (uint32_t)((pm[0] >> 16) & 255) | ((uint32_t)((pm[1] >> 16) & 255) << 8) | ((uint32_t)((pm[2] >> 16) & 255) << 16) | ((uint32_t)((pm[3] >> 16) & 255) << 24);
                        }
                    }
                }
            }
        POA_MARK("step_end");
            carry1 = all1; carry2 = all2;
            left1 = nleft1; left2 = nleft2;
        }
        POA_MARK("row_reduce");
        stamp(4);
        {
            int wb = poa_wave_scan_max(best);
            wb = __builtin_amdgcn_readlane(wb, 63);
            int lm = best == wb ? -lpos : INT32_MIN;
            int rm = best == wb ? rpos : INT32_MIN;
            lm = poa_wave_scan_max(lm);
            rm = poa_wave_scan_max(rm);
            int dz = 0;
            if constexpr (H16) {
                danger |= (((uint32_t)zacc & 0xffffu) < 2000u || ((uint32_t)zacc >> 16) < 2000u) ? 1 : 0;
                dz = __builtin_amdgcn_ballot_w64(danger != 0) != 0ull ? 1 : 0;
            }
            if (lane == 63) sRed[wv] = make_int4(wb, lm, rm, dz);
        }
        POA_LDS_BARRIER();
        {
            int4 rw = make_int4(INT32_MIN, INT32_MIN, INT32_MIN, 0);
            if (lane < NW) rw = sRed[lane];
            int b = rw.x, t;
            t = poa_dpp<0x111, 0xf>(INT32_MIN, b); b = t > b ? t : b;
            if (NW > 2) { t = poa_dpp<0x112, 0xf>(INT32_MIN, b); b = t > b ? t : b; }
            if (NW > 4) { t = poa_dpp<0x114, 0xf>(INT32_MIN, b); b = t > b ? t : b; }
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
            lpos = -__builtin_amdgcn_readlane(lm, NW - 1);
            rpos = __builtin_amdgcn_readlane(rm, NW - 1);
            if constexpr (H16) {
                prev_hmax = rbest + base;
                prev_base = base;
                if (__builtin_amdgcn_ballot_w64(rw.w != 0) != 0ull) { range_stop = true; }
            }
        }
        POA_MARK("row_end");
        if (__builtin_expect(is_sink, 0) && leader) {
            // the sink takes the first largest H[qlen] among its predecessors (rows come in the order of the sink list)
            lead_t L = lead_load();
            int val = POA_NEG;
            if (qlen >= beg && qlen <= end) {
                const int wq_ = sSink[0] >> 8;
                val = H16 ? (wq_ <= -30000 ? POA_NEG : wq_ + base) : wq_;
            }
            if (!L.sink_have || val > L.sink_best) { L.sink_best = val; L.sink_row = r; L.sink_have = 1; }
            lead_store(L);
        }
        if (tid == 0) {
            R[r].lmax = lpos; R[r].rmax = rpos;
            if constexpr (H16) R[r].hmax = prev_hmax;
        }
        if (range_stop) { stop = true; break; }
        prev_beg = beg; prev_end = end; prev_lmax = lpos; prev_rmax = rpos;
        prev_lds = !wide;
        stamp(5);
    }
    }
    __syncthreads();
    if constexpr (STAMP) {
        // wave 0 holds the band's left edge; wave 2 is an interior wave on 10 kbp reads
        if (stamps && blockIdx.x < 64 && (tid == 0 || tid == 128))
            for (int s = 0; s < 8; s++) stamps[(tid ? 64 * 8 : 0) + blockIdx.x * 8 + s] = tacc[s];
    }
    if (tid >= 64) return;
    // ---- wave 0: the result record, then (tb_ops != nullptr) the traceback of this problem, out of the same LDS -- the
    // direction rows were written by this workgroup, through this CU's L1, and the barrier above ordered them
    int status = POA_ST_OK;
    uint32_t start_row = 0;
    {
        const lead_t L = lead_load();
        const bool failed = L.failed != 0;
        poa_out &O = outs[blockIdx.x];
        if (range_stop) status = POA_ST_RANGE;
        else if (failed) status = POA_ST_POOL;
        else {
            start_row = L.sink_row;
            status = (L.sink_have != 0 && L.sink_best > POA_NEG / 2) ? POA_ST_OK : POA_ST_NOALN;
        }
        if (tid == 0) {
            O.t_begin = t_begin;
            O.cells = L.cells;
            O.vcells = L.vcells;
            O.maxw = (uint32_t)L.maxw;
            O.score = (range_stop || failed) ? POA_NEG : L.sink_best;
            O.row = start_row;
            O.status = status;
        }
    }
    if (tb_ops) {
        status = __builtin_amdgcn_readfirstlane(status);
        start_row = (uint32_t)__builtin_amdgcn_readfirstlane((int)start_row);
        poa_traceback_wave(*(tb_lds *)(smem + HDR), tid, pb, rows, preds, pool, outs[blockIdx.x], tb_ops, tb_orow, H16 ? 0xC0 : 0,
                           status, start_row);
    }
    if (tid == 0) {
        outs[blockIdx.x].t_end = __builtin_amdgcn_s_memrealtime();
        if (n_arenas) {
            // bytes this problem took (statistics of the launch), then hand the arena on
            const unsigned long long used = atomicAdd(pool_next, 0ull);
            (void)atomicAdd(pool_next_arg, used < arena_size ? used : (unsigned long long)arena_size);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            (void)atomicExch(&arena_flag[arena], 0u);
        }
    }
}

#endif  // VGA_VARIANTS

static inline size_t poa_pk_lds_bytes(uint32_t hg_cols, uint32_t lds_cols, int nt, bool h16 = false)
{
    const int nw = nt / 64;
    // (the fused traceback reuses the row-state area: at least sizeof(tb_lds) behind the header)
    return std::max<size_t>((h16 ? 3ull : 4ull) * hg_cols + ((lds_cols / 2 + 15u) & ~15u), sizeof(tb_lds)) + (size_t)(3 * nw + 1 + 4 + 6 + 1) * 16 + 16;
}

static inline uint32_t poa_lds_cols(uint32_t max_q) { return ((max_q + 1 + 15u) & ~15u) + 16u; }

static inline size_t poa_lds_bytes(uint32_t lds_cols, int nt)
{
    const int nw = nt / 64;
    return 7ull * lds_cols + (size_t)(8 * nw + 3 * nw + 2) * 4 + 16;
}

#include "vga_poa_t4.hpp"
#include "vga_poa_w1.hpp"

// One workgroup per staged problem: copies its node table, predecessor rows, sink rows, bases and query from the device
// store of vga_subgraph.hip (and the batch's reads) to where this sub-batch's poa_prob says they are.
// ids: per staged problem its index in the store and its number of predecessor entries.  The store has two parts
// (problems below / from `split`); a re-run launch may hold problems of both.
struct sg_gather_src {
    const uint4 *ntab;
    const uint32_t *preds, *sinks;
    const char *seq;
    uint32_t p0;
};
__global__ __launch_bounds__(256) void k_sg_gather(const uint32_t *__restrict__ ids, const poa_prob *__restrict__ probs, const sg_off *__restrict__ offs,
                                                   uint32_t split, sg_gather_src s0, sg_gather_src s1, const char *__restrict__ reads,
                                                   uint4 *ntab, uint32_t *preds, uint32_t *sinks, char *seq, char *q)
{
    const uint32_t p = ids[2 * blockIdx.x], n_preds = ids[2 * blockIdx.x + 1];
    const poa_prob pb = probs[blockIdx.x];
    const sg_off of = offs[p];
    const sg_gather_src S = p >= split ? s1 : s0;
    const int tid = threadIdx.x;
    const uint4 *a = S.ntab + of.node0 + (p - S.p0);  // (a part's node tables carry one source entry per problem of the part)
    for (uint32_t i = (uint32_t)tid; i < pb.n_nodes; i += 256) ntab[pb.node0 + i] = a[i];
    for (uint32_t i = (uint32_t)tid; i < n_preds; i += 256) preds[pb.pred0 + i] = S.preds[of.pred0 + i];
    for (uint32_t i = (uint32_t)tid; i < pb.n_sink; i += 256) sinks[pb.sink0 + i] = S.sinks[of.sink0 + i];
    const uint32_t *sw = (const uint32_t *)(S.seq + of.seq0);  // (seq0 is a multiple of 4 on both sides)
    uint32_t *dw = (uint32_t *)(seq + pb.seq0);
    for (uint32_t i = (uint32_t)tid; i < (pb.N + 3) / 4; i += 256) dw[i] = sw[i];
    for (uint32_t i = (uint32_t)tid; i < pb.qlen; i += 256) q[pb.q0 + i] = reads[of.q_src + i];
}

// ============================================================================================ host
namespace {

struct poa_prep {
    bool ok = false;
    uint32_t N = 0, qlen = 0;
    int32_t longest = 0;                // graph bases on the longest source-sink path
    uint32_t life = 1;                  // largest dst - src over the edges: how many node-end rows a value row must outlive
    std::vector<uint4> ntab;            // node table incl. the source entry
    std::vector<uint32_t> preds, sinks; // row ids
    std::vector<uint32_t> first_row;    // per input node
    uint32_t n_ntab = 0, n_preds = 0, n_sinks = 0;  // entries of the three lists (with a device store the vectors stay empty)
    const uint32_t *first_row_p = nullptr;          // first rows, n_ntab - 1 entries (host graphs: first_row.data())
};

// Node-level graph description: first rows, predecessor rows in edge-list order, remain of the last base of each
// node (longest-path DP over the nodes), sink predecessors.  Mirrors the row construction of oracle/og_poa.c.
void poa_prepare(const poa_view &v, poa_prep &g)
{
    g.ok = false;
    const uint64_t nv = v.n_nodes;
    if (nv == 0 || v.qlen >= (1u << 24)) return;
    g.first_row.resize(nv);
    std::vector<uint32_t> last_row(nv);
    uint64_t N = 0;
    for (uint64_t i = 0; i < nv; i++) {
        const uint64_t len = v.node_off[i + 1] - v.node_off[i];
        if (len == 0 || len >= (1u << 24)) return;
        g.first_row[i] = (uint32_t)(N + 1);
        N += len;
        last_row[i] = (uint32_t)N;
    }
    if (N >= (1ull << 31)) return;
    g.N = (uint32_t)N;
    g.qlen = v.qlen;
    std::vector<uint32_t> in_off(nv + 1, 0), out_off(nv + 1, 0);
    g.life = 1;
    std::vector<uint32_t> reach(nv, 0);  // per node: how far (in nodes) its farthest successor is
    for (uint64_t e = 0; e < v.n_edges; e++) {
        if (v.esrc[e] >= v.edst[e] || v.edst[e] >= nv) return;
        in_off[v.edst[e] + 1]++;
        out_off[v.esrc[e] + 1]++;
        reach[v.esrc[e]] = std::max(reach[v.esrc[e]], v.edst[e] - v.esrc[e]);
    }
    // nodes whose value row is read far ahead keep it for good; the rest share a ring of POA_RING_SPAN + 1 rows
    for (uint64_t i = 0; i < nv; i++)
        if (reach[i] <= POA_RING_SPAN) g.life = std::max(g.life, reach[i]);
    for (uint64_t i = 0; i < nv; i++) { in_off[i + 1] += in_off[i]; out_off[i + 1] += out_off[i]; }
    std::vector<uint32_t> in_adj(v.n_edges ? v.n_edges : 1), out_adj(v.n_edges ? v.n_edges : 1), fi(nv, 0), fo(nv, 0);
    for (uint64_t e = 0; e < v.n_edges; e++) {
        in_adj[in_off[v.edst[e]] + fi[v.edst[e]]++] = v.esrc[e];
        out_adj[out_off[v.esrc[e]] + fo[v.esrc[e]]++] = v.edst[e];
    }
    // remain of the LAST base of each node; interior bases add their distance to it on the device
    std::vector<int32_t> remain_last(nv, 0), remain_first(nv, 0);
    for (uint64_t i = nv; i-- > 0;) {
        int32_t rl = 0;
        for (uint32_t t = out_off[i]; t < out_off[i + 1]; t++) rl = std::max(rl, 1 + remain_first[out_adj[t]]);
        remain_last[i] = rl;
        remain_first[i] = rl + (int32_t)(last_row[i] - g.first_row[i]);
    }
    int32_t longest = 0;
    g.ntab.clear();
    g.preds.clear();
    g.sinks.clear();
    g.ntab.resize(nv + 1);
    for (uint64_t i = 0; i < nv; i++) {
        const uint32_t deg = in_off[i + 1] - in_off[i];
        if (deg > 255) return;
        const uint32_t pstart = (uint32_t)g.preds.size();
        if (deg == 0) {
            g.preds.push_back(0);
            longest = std::max(longest, 1 + remain_first[i]);
        } else {
            for (uint32_t t = in_off[i]; t < in_off[i + 1]; t++) g.preds.push_back(last_row[in_adj[t]]);
        }
        const uint32_t len = last_row[i] - g.first_row[i] + 1;
        const bool is_sink = out_off[i + 1] == out_off[i];
        // .z: remain of the node's last base; bit 31 marks a node without successors (its last row feeds the sink),
        // bit 30 a node whose value row is read more than POA_RING_SPAN nodes ahead
        g.ntab[i + 1] = make_uint4(g.first_row[i], len | ((deg ? deg : 1u) << 24),
                                   (uint32_t)remain_last[i] | (is_sink ? 0x80000000u : 0u) | (reach[i] > POA_RING_SPAN ? 0x40000000u : 0u),
                                   deg <= 1 ? g.preds[pstart] : pstart);
        if (is_sink) g.sinks.push_back(last_row[i]);
    }
    g.longest = longest;
    g.ntab[0] = make_uint4(0u, 1u, (uint32_t)longest, 0u);  // the virtual source: row 0, remain = longest path
    g.n_ntab = (uint32_t)g.ntab.size(); g.n_preds = (uint32_t)g.preds.size(); g.n_sinks = (uint32_t)g.sinks.size();
    g.first_row_p = g.first_row.data();
    g.ok = true;
}

// device + pinned staging of one sub-batch; sub-batches alternate between two of these (and two streams)
struct poa_slot {
    vga_dbuf<poa_prob> d_probs;
    vga_dbuf<uint4> d_ntab;
    vga_dbuf<uint32_t> d_seq32, d_preds, d_sink, d_orow, d_ids;
    vga_dbuf<uint8_t> d_ops;
    vga_dbuf<poa_row> d_rows;
    vga_dbuf<poa_out> d_outs;
    vga_dbuf<char> d_q;
    vga_hbuf<poa_prob> h_probs;
    vga_hbuf<uint4> h_ntab;
    vga_hbuf<uint32_t> h_seq32, h_preds, h_sink, h_ids;
    vga_hbuf<char> h_q;
    // results come back into one of two sets, alternating per use of the slot: the host is still reading set A of the
    // sub-batch that just finished when the next sub-batch on this slot is enqueued (it will write set B)
    struct out_set {
        vga_hbuf<uint32_t> h_orow;
        vga_hbuf<uint8_t> h_ops;
        vga_hbuf<poa_out> h_outs;
    } outs[2];
    uint32_t uses = 0;
};

struct poa_ws {
    poa_slot slot[POA_SLOTS];
    vga_dbuf<unsigned long long> d_next;
    vga_hbuf<unsigned long long> h_next;
    vga_dbuf<unsigned long long> d_arena_ctr;  // arena mode: one bump counter and one busy flag per arena
    vga_dbuf<uint32_t> d_arena_flag;
    uint8_t *pool = nullptr;
    uint64_t pool_size = 0;
    hipStream_t extra[POA_SLOTS] = {};  // streams of slots 1.. (slot 0 runs on the context's stream)
    double pool_scale = 1.0;   // measured pool bytes / estimated bytes, adapted after every sub-batch
    ~poa_ws()
    {
        if (pool) (void)hipFree(pool);
        for (int i = 0; i < POA_SLOTS; i++)
            if (extra[i]) (void)hipStreamDestroy(extra[i]);
    }
};

template <typename T>
T *pmalloc(size_t n)
{
    return (T *)malloc((n ? n : 1) * sizeof(T));
}

void append_u(std::string &s, uint64_t v)
{
    char t[24];
    int n = snprintf(t, sizeof t, "%llu", (unsigned long long)v);
    s.append(t, (size_t)n);
}

inline char lower(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c + 32) : c; }

template <typename F>
void parallel_for(uint64_t n, F f) { vga_parallel_for(n, f); }

}  // namespace

int poa_run(vga_ctx *ctx, poa_feed &feed, const vga_poa_params *params, std::vector<poa_item> &out, poa_timing &tm)
{
    const uint64_t n = feed.views.size();
    std::vector<poa_view> &views = feed.views;
    out.assign(n, poa_item());
    tm = poa_timing();
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    auto t_host0 = std::chrono::steady_clock::now();
    vga_trace tr("poa");
    if (params->gap_open1 < 0 || params->gap_open1 > 255 || params->gap_open2 < 0 || params->gap_open2 > 255 ||
        params->gap_ext1 < 0 || params->gap_ext2 < 0 || params->gap_open1 + params->gap_ext1 > 255 ||
        params->gap_open2 + params->gap_ext2 > 255)
        return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "gap penalties: open + extend must be in 0..255 (one byte per gap state)");
    if (!feed.keep_timers) vga_timers_reset(ctx);
    if (n == 0) return VGA_OK;
    uint32_t max_q = 0;
    for (uint64_t p = 0; p < n; p++) {
        if (views[p].qlen >= (1u << 24)) return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "query %llu too long", (unsigned long long)p);
        max_q = std::max(max_q, views[p].qlen);
    }
    {
        const uint32_t lds_cols_all = poa_lds_cols(max_q);
        int g1b = 0, g2b = 0;
        while ((1 << g1b) <= params->gap_open1 + params->gap_ext1) g1b++;
        while ((1 << g2b) <= params->gap_open2 + params->gap_ext2) g2b++;
        const char *force = getenv("VGA_POA_KERNEL");
        const bool unpacked = g1b + g2b > 8 || (force && strstr(force, "unpacked"));
        // the packed kernel keeps a 4096-column window of the row state; the unpacked one every column
        const uint32_t hg_need = (force && strstr(force, "full")) ? lds_cols_all : std::min<uint32_t>(lds_cols_all, 4096);
        const size_t need = unpacked ? poa_lds_bytes(lds_cols_all, 128) : poa_pk_lds_bytes(hg_need, lds_cols_all, 128);
        if (need > 160 * 1024 - 256)
            return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "query of %u bases does not fit the LDS-resident POA kernel (limit ~280 kbp, ~22 kbp with large gap penalties)", max_q);
    }

    // ---- launch order and lazy preparation.  The caller may hand the problems over lazily (feed.prepare fills the graph
    // part of a view on request): then the order is fixed up front from a cheap size proxy and a sub-batch's subgraphs
    // and node tables are built by the host threads while earlier sub-batches are on the GPU.  Without a proxy every
    // problem is prepared first and the order is by the footprint estimate (longest first).
    std::vector<poa_prep> G(n);
    std::vector<poa_prob> probs(n);
    std::vector<double> est(n, 0.0), estw(n, 0.0);
    std::vector<uint8_t> ready(n, 0);
    std::vector<uint32_t> order(n);
    for (uint64_t p = 0; p < n; p++) order[p] = (uint32_t)p;
    // Mean band width of a problem.  The band of a row spans from the row maxima to the diagonal qlen - remain, so it
    // grows with the excess of the longest source-sink path over the query; on 10 kbp reads against DRB1-3123 the mean is
    // 2w + 1 + 430 + 0.27 * excess (rms error ~25 %).  Only the pool budget and the launch order depend on it, and the
    // budget scale adapts to the measured footprint after every sub-batch.
    auto est_width = [&](uint64_t p) -> double {
        const poa_prep &g = G[p];
        const double w = params->wb < 0 ? (double)g.qlen : (double)params->wb + (double)(uint64_t)(params->wf * (double)g.qlen);
        double excess = (double)g.longest - (double)g.qlen;
        if (excess < 0) excess = -excess;
        return std::min((double)g.qlen + 1.0, 2.0 * w + 1.0 + 430.0 + 0.3 * excess);
    };
    bool malformed = false, dev_failed = false;
    int dev_rc = VGA_OK;
    // prepares launch positions [a, b): the caller's part (subgraphs), then node tables and estimates
    std::vector<uint32_t> ids;
    auto ensure = [&](uint64_t a, uint64_t b) {
        ids.clear();
        for (uint64_t i = a; i < b && i < order.size(); i++)
            if (!ready[order[i]]) ids.push_back(order[i]);
        if (ids.empty()) return;
        if (feed.prepare) feed.prepare(ids.data(), ids.size());
        if (feed.dev && !feed.dev->part[1].ready) {
            // the second part of the device store is built when a problem of it is first needed -- by then the first DP
            // launch is on the GPU and the subgraph kernels run beside it
            bool need = false;
            for (uint32_t p : ids) need |= p >= feed.dev->split;
            if (need) {
                if ((dev_rc = feed.dev_rest()) != VGA_OK) { dev_failed = true; return; }
                // the caller may have re-ordered the second part (none of it has been staged): take its order over, and
                // prepare what now stands at the positions asked for
                if (feed.order) {
                    for (uint64_t i = feed.dev->split; i < n; i++) order[i] = feed.order[i];
                    ids.clear();
                    for (uint64_t i = a; i < b && i < order.size(); i++)
                        if (!ready[order[i]]) ids.push_back(order[i]);
                }
            }
        }
        parallel_for(ids.size(), [&](uint64_t t) {
            const uint32_t p = ids[t];
            if (feed.dev) {
                // the device store holds the graph: only its sizes come to the host
                const sg_sum &sm = feed.dev->sum[p];
                poa_prep &g = G[p];
                g.ok = !(sm.flags & 1u) && sm.n_nodes > 0 && views[p].qlen < (1u << 24);
                g.N = sm.N; g.qlen = views[p].qlen; g.longest = (int32_t)sm.longest; g.life = sm.life;
                g.n_ntab = sm.n_nodes + 1; g.n_preds = sm.n_preds; g.n_sinks = sm.n_sinks;
                g.first_row_p = feed.dev->of(p).h_first_row + feed.dev->off[p].node0;
            } else
                poa_prepare(views[p], G[p]);
            // footprint in the pool: a direction byte per cell plus the value-row ring (packed kernel)
            if (G[p].ok) {
                estw[p] = est_width(p);
                est[p] = (double)G[p].N * estw[p] * 1.15 + (double)(G[p].life + 1) * 6.0 * ((double)G[p].qlen + 8.0) + 2.0 * (double)POA_CHUNK;
            }
            ready[p] = 1;
        });
        for (uint32_t p : ids)
            if (!G[p].ok) malformed = true;
    };
    if (feed.order) {
        for (uint64_t p = 0; p < n; p++) order[p] = feed.order[p];
    } else if (feed.proxy) {
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return feed.proxy[x] > feed.proxy[y]; });
    } else {
        ensure(0, n);
        if (!malformed) std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return est[x] > est[y]; });
    }
    auto malformed_error = [&]() {
        return vga_set_error(ctx, VGA_ERR_ARG,
                             "a POA problem is malformed (no node, empty node, edge with src >= dst, in-degree > 255, or sequence too long)");
    };
    if (malformed) return malformed_error();
    tr.mark("order (+ node tables when not lazy)");

    if (!ctx->poa_ws) {
        ctx->poa_ws = new poa_ws();
        ctx->poa_ws_free = [](void *q) { delete (poa_ws *)q; };
    }
    poa_ws &W = *(poa_ws *)ctx->poa_ws;
#define POA_CHECK(call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return vga_set_error(ctx, VGA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                                 __LINE__);                                                          \
    } while (0)
    POA_CHECK(W.h_next.reserve(POA_SLOTS));
    POA_CHECK(W.d_next.reserve(POA_SLOTS));
    // ---- the pool: sized from the first prepared problems, grown generously (a hipMalloc of this size costs seconds)
    const uint64_t n_probe = std::min<uint64_t>(n, 512);
    ensure(0, n_probe);
    if (malformed) return malformed_error();
    {
        double probe = 0;
        for (uint64_t i = 0; i < n_probe; i++) probe += est[order[i]];
        const double want_d = probe / (double)n_probe * (double)n * W.pool_scale * 1.3 + (double)n * 3.0 * (double)POA_CHUNK;
        uint64_t want = (uint64_t)want_d + 64 * POA_CHUNK;
        const char *env_pool = getenv("VGA_POOL_BYTES");
        if (env_pool) want = std::min<uint64_t>(want, strtoull(env_pool, nullptr, 10));
        if (W.pool_size < want) {
            size_t free_b = 0, total_b = 0;
            POA_CHECK(hipMemGetInfo(&free_b, &total_b));
            // leave room for everything else this context allocates (staging of three sub-batches, the subgraph store, the
            // map workspace): 15 % of what is free, at least 16 GB -- two processes sharing a GPU otherwise starve each other
            const uint64_t have = free_b + W.pool_size;
            const uint64_t reserve = std::max<uint64_t>((uint64_t)((double)have * 0.15), 16ull << 30);
            uint64_t avail = have > reserve ? have - reserve : have / 4;
            // several contexts on one GPU (vgaligner map --devices 0,0: the driver sets this to 1 / their number) share it
            if (const char *fr = getenv("VGA_POOL_FRACTION")) {
                const double f = atof(fr);
                if (f > 0.0 && f < 1.0) avail = (uint64_t)((double)avail * f);
            }
            const uint64_t target = std::min(std::max<uint64_t>(2 * want, 8ull << 30), avail) & ~(POA_CHUNK - 1);
            if (target > W.pool_size) {
                if (W.pool) { (void)hipFree(W.pool); W.pool = nullptr; W.pool_size = 0; }
                if (target < 64 * POA_CHUNK)
                    return vga_set_error(ctx, VGA_ERR_NOMEM, "only %llu bytes of HBM free for the traceback pool", (unsigned long long)free_b);
                POA_CHECK(hipMalloc((void **)&W.pool, target));
                W.pool_size = target;
            }
        }
    }
    tr.mark("pool");
    // Two sub-batches are in flight at any time, one per stream, each carving from its own half of the pool: while one
    // drains (its last workgroups, then the latency-bound traceback and the copies back) the other one's
    // workgroups fill the CUs.
    const char *force_k = getenv("VGA_POA_KERNEL");
    int g1b_ = 0, g2b_ = 0;
    while ((1 << g1b_) <= params->gap_open1 + params->gap_ext1) g1b_++;
    while ((1 << g2b_) <= params->gap_open2 + params->gap_ext2) g2b_++;
    const bool tb_fused_early = !getenv("VGA_POA_TB") || strstr(getenv("VGA_POA_TB"), "fused");
#ifdef VGA_VARIANTS
    const bool pk_built = true;
#else
    const bool pk_built = false;  // k_poa_dp_pk / the 16-bit build / the lane traceback only exist in the variants build
#endif
    const bool packed_fit = g1b_ + g2b_ <= 8 && !(force_k && strstr(force_k, "unpacked"));
    // k_poa_dp_t4 (vga_poa_t4.hpp), the default: scores scaled by 4 with argmax tags, G bytes 4 g - 1 / 4 g
    const bool t4_k = !(force_k && strstr(force_k, "unpacked")) && 4 * (params->gap_open1 + params->gap_ext1) - 1 <= 255 && 4 * (params->gap_open2 + params->gap_ext2) <= 255 &&
                      params->gap_ext1 >= 1 && params->match + params->mismatch >= 0 && params->match + params->mismatch < (1 << 20) &&
                      !(pk_built && ((force_k && (strstr(force_k, "pk") || strstr(force_k, "full"))) || (getenv("VGA_POA_H16") && atoi(getenv("VGA_POA_H16")) != 0) ||
                                     getenv("VGA_POA_STAMPS")));
    const bool packed_k = t4_k || (pk_built && packed_fit);  // a kernel with the LDS column window and the fused traceback
    // k_poa_dp_w1 (vga_poa_w1.hpp): one wave per problem, row state in registers; default penalties, queries whose codes fit its LDS
    const bool def_pen_k = params->gap_open1 == 4 && params->gap_ext1 == 2 && params->gap_open2 == 24 && params->gap_ext2 == 1;
    const bool w1_k = t4_k && def_pen_k && tb_fused_early && max_q <= 16000 && !(force_k && strstr(force_k, "t4")) &&
                      (force_k ? strstr(force_k, "w1") != nullptr : getenv("VGA_POA_W1") != nullptr);
    // k_poa_dp_pk / k_poa_dp_lds hand pool space out in 1 MiB chunks and assume that a request fits one (k_poa_dp_t4 takes
    // whole chunks for a larger one): their two wide-row scratch rows (8 B per column) and an unbanded direction row with its
    // three predecessor planes (4 B per column) must stay below that
    if (!t4_k && 8ull * (uint64_t)poa_lds_cols(max_q) > POA_CHUNK)
        return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "query of %u bases: only k_poa_dp_t4 (default penalties range) handles queries beyond ~131 kbp", max_q);
    // traceback: fused into the packed DP kernel (default), or VGA_POA_TB=wave / lane: a kernel of its own after the DP
    const bool tb_lane = pk_built && getenv("VGA_POA_TB") && strstr(getenv("VGA_POA_TB"), "lane");
    (void)tb_lane;
    const bool tb_fused = !getenv("VGA_POA_TB") || strstr(getenv("VGA_POA_TB"), "fused");
    const bool arena_wanted = packed_k && tb_fused && !getenv("VGA_POA_STAMPS") && !(getenv("VGA_POA_ARENAS") && atoi(getenv("VGA_POA_ARENAS")) == 0);
    // classic mode: two sub-batches in flight (three are no faster, four overflow their pool quarters).  Arena mode: the
    // pool is not split and a third slot only costs staging buffers (round 1 ran three throughout: +1.4 % on config 3 with
    // first-in-first-out completion; with launches handled in the order they finish, round 2, that reversed).
    // Two launches in flight keep the GPU full when the problems of a call are of one kind (config 3: 8 410-8 460 reads/s
    // with two, 8 100-8 370 with three, same-box); when the call holds very long problems (poa_feed::klass: config 4's
    // 100 000-row chains) their launch occupies a slot for a second, and a third slot keeps two for everything else
    // (config 4: 7 900 reads/s with three, 5 900 with two)
    bool any_long = false;
    if (feed.klass)
        for (uint64_t p = 0; p < n && !any_long; p++) any_long = feed.klass[p] != 0;
    int n_slots = arena_wanted ? (any_long || !feed.klass ? 3 : 2) : 2;
    if (const char *e = getenv("VGA_POA_SLOTS")) n_slots = std::max(1, std::min(POA_SLOTS, atoi(e)));
    hipStream_t sarr[POA_SLOTS];
    sarr[0] = st;
    for (int i = 1; i < n_slots; i++) {
        if (!W.extra[i]) POA_CHECK(hipStreamCreateWithFlags(&W.extra[i], hipStreamNonBlocking));
        sarr[i] = W.extra[i];
    }
    const uint64_t half_pool = (W.pool_size / (uint64_t)n_slots) & ~(POA_CHUNK - 1);  // one slot's segment of the pool

    // ---- arena mode (the default with the packed kernel and its fused traceback): the whole pool is cut into arenas, a
    // workgroup holds one from its first row to the end of its traceback.  No launch has to wait for another one's
    // pool segment, so sub-batches are cut for the host pipeline only and workgroups of consecutive launches fill the
    // CUs back to back.  Problems that would not fit an arena (and the ones that turn out not to) are collected and run
    // at the end in classic mode, which gives each of them as much of the pool as it needs.
    uint32_t n_arenas = 0;
    uint64_t arena_size = 0;
    if (arena_wanted) {
        // an arena should hold the largest of the probed problems with a margin; more arenas than workgroups can be
        // resident (16 per CU at most) are of no use, fewer than 4 per CU would leave most of the GPU waiting for one
        double big = 0;
        for (uint64_t i = 0; i < n_probe; i++) big = std::max(big, est[order[i]]);
        const double want_arena = big * W.pool_scale * 1.3 + 4.0 * (double)POA_CHUNK;
        uint64_t na = (uint64_t)((double)W.pool_size / want_arena);
        na = std::min<uint64_t>(na, 16ull * (uint64_t)ctx->n_cu);
        if (const char *e = getenv("VGA_POA_ARENAS")) na = std::min<uint64_t>(na, strtoull(e, nullptr, 10));
        if (na >= 4ull * (uint64_t)ctx->n_cu || na >= n) {
            n_arenas = (uint32_t)na;
            arena_size = (W.pool_size / na) & ~(POA_CHUNK - 1);
            POA_CHECK(W.d_arena_ctr.reserve(n_arenas));
            POA_CHECK(W.d_arena_flag.reserve(n_arenas));
            POA_CHECK(hipMemset(W.d_arena_flag.p, 0, n_arenas * sizeof(uint32_t)));
        }
    }
    if (tr.on) fprintf(stderr, "[vga-trace] poa: pool %.1f GB, %u arenas of %.1f MB\n", (double)W.pool_size / 1e9, n_arenas, (double)arena_size / 1e6);

    poa_dev_params P;
    P.match = params->match; P.mismatch = params->mismatch; P.o1 = params->gap_open1; P.e1 = params->gap_ext1;
    P.o2 = params->gap_open2; P.e2 = params->gap_ext2; P.banded = params->wb >= 0;

    bool packed_all = true;  // value rows are 4 B per cell with the packed kernel, 6 B otherwise (byte model)
    bool t4_any = false;     // ... and 6 B with k_poa_dp_t4
    bool h16_all = true;     // ... and 3 B with 16-bit row state
    bool any_fused = false;  // the DP kernel walked the alignments back itself
    int t_total = vga_timer_begin(ctx, "poa_total", 0);
    struct sub_t { uint64_t i0, i1; double raw_est; int slot; int oset; bool use32 = false; bool arena = false; };
    hipError_t launch_err = hipSuccess;
    // a sub-batch is closed once it holds this many problems and this many estimated DP cells (or its pool half is full)
    // measured on configs 3-5 (tests/prof_sub_sweep.sh, tests/prof_ab.sh).  Classic mode: 3072..5120 is flat, uncapped
    // loses 40 % on config 5.  Arena mode: launches share the GPU seamlessly, so shorter ones only cost when two of them
    // cannot fill it (1024: -12 % on config 3); 2048 is best on all three.
    uint64_t sub_problems = n_arenas ? 2048 : 4096;
    double sub_cells = 2e9;
    if (const char *e = getenv("VGA_POA_SUB")) sub_problems = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    uint64_t in_flight_other = 0;  // problems of the sub-batch on the other stream (they share the GPU with this launch)
    // stage, upload and enqueue DP + traceback + result copies of a sub-batch that starts at launch position i0 and ends
    // at cap at the latest
    auto launch = [&](uint64_t i0, uint64_t cap, int slot, bool use32, bool arena) -> sub_t {
        hipStream_t st = sarr[slot];  // shadows the context's stream inside this lambda
        poa_slot &S = W.slot[slot];
        uint8_t *pool_base = arena ? W.pool : W.pool + (uint64_t)slot * half_pool;
        const double budget = (double)half_pool * 0.92;
        double used_est = 0, raw_est = 0, cells_est = 0;
        uint64_t i1 = i0;
        // (device store: a launch gathers from one part of it)
        if (feed.dev && feed.dev->split > i0 && feed.dev->split < cap) cap = feed.dev->split;
        while (i1 < cap) {
            if (!ready[order[i1]]) ensure(i1, std::min<uint64_t>(cap, i1 + 256));
            if (malformed || dev_failed) break;
            const double e = est[order[i1]] * W.pool_scale + 3.0 * (double)POA_CHUNK;
            if (!arena && i1 > i0 && used_est + e > budget) break;
            // the pool is not the only reason to cut: the host work either side of a sub-batch (subgraphs and node
            // tables before, CIGAR / cs strings after) only overlaps with the GPU when there are several sub-batches
            if (i1 - i0 >= sub_problems && cells_est >= sub_cells) break;
            // very long problems (a chain that spans 100 kbp of the linearisation: 100 000 sequential rows) are a launch of
            // their own: they decide how long the whole call takes, so they get the largest workgroup and window (below)
            if (feed.klass && i1 > i0 && feed.klass[order[i1]] != feed.klass[order[i0]]) break;
            used_est += e;
            // (arena mode: problems that are sent on to the classic pass take no arena and do not count)
            if (!arena || e * 1.1 <= (double)arena_size) raw_est += est[order[i1]];
            cells_est += (double)G[order[i1]].N * estw[order[i1]];
            i1++;
        }
        auto chk = [&](hipError_t e) { if (e != hipSuccess && launch_err == hipSuccess) launch_err = e; };
        bool sub_h16 = false;  // this sub-batch runs the 16-bit DP kernel
        bool sub_t4 = false;   // ... k_poa_dp_t4 (its own direction-byte encoding)
        bool sub_fused = false;  // ... and its DP kernel does the traceback as well
        if (malformed || dev_failed || i1 == i0) return {i0, i0, 0.0, slot, 0};
        const int oset = (int)(S.uses++ & 1u);
        poa_slot::out_set &O = S.outs[oset];
        const uint32_t nb = (uint32_t)(i1 - i0);
        // offsets of the sub-batch's problems inside this slot's buffers
        uint64_t tot_nodes = 0, tot_preds = 0, tot_sink = 0, tot_q = 0, tot_ops = 0, tot_rows = 0, tot_seq = 0;
        for (uint64_t i = i0; i < i1; i++) {
            const uint32_t p = order[i];
            poa_prob &pb = probs[p];
            const poa_prep &g = G[p];
            pb.node0 = tot_nodes; pb.pred0 = tot_preds; pb.sink0 = tot_sink; pb.q0 = tot_q; pb.ops0 = tot_ops; pb.row0 = tot_rows;
            pb.seq0 = tot_seq;
            pb.n_sink = g.n_sinks; pb.qlen = g.qlen; pb.N = g.N; pb.n_nodes = g.n_ntab; pb.ring_rows = g.life + 1;
            pb.flags = arena && (est[p] * W.pool_scale + 3.0 * (double)POA_CHUNK) * 1.1 > (double)arena_size ? 1u : 0u;
            pb.pad = 0;
            pb.w = params->wb < 0 ? g.qlen : (uint32_t)((int64_t)params->wb + (int64_t)(params->wf * (double)g.qlen));
            tot_nodes += g.n_ntab;
            tot_preds += g.n_preds;
            tot_sink += g.n_sinks;
            tot_q += g.qlen;
            tot_ops += (uint64_t)g.N + g.qlen + 2;
            tot_rows += (uint64_t)g.N + 1;
            tot_seq += ((uint64_t)g.N + 3) & ~3ull;
            out[p].n_rows = g.N;
        }
        const bool dev = feed.dev != nullptr;
        chk(S.h_probs.reserve(nb));
        if (dev) chk(S.h_ids.reserve(2 * (size_t)nb));
        else {
            chk(S.h_ntab.reserve(tot_nodes)); chk(S.h_seq32.reserve(tot_seq / 4 + 1));
            chk(S.h_preds.reserve(tot_preds + 1)); chk(S.h_sink.reserve(tot_sink + 1)); chk(S.h_q.reserve(tot_q + 1));
        }
        chk(O.h_ops.reserve(tot_ops)); chk(O.h_orow.reserve(tot_ops)); chk(O.h_outs.reserve(nb));
        chk(S.d_probs.reserve(nb)); chk(S.d_ntab.reserve(tot_nodes)); chk(S.d_seq32.reserve(tot_seq / 4 + 1));
        chk(S.d_preds.reserve(tot_preds + 1)); chk(S.d_sink.reserve(tot_sink + 1)); chk(S.d_q.reserve(tot_q + 1));
        chk(S.d_rows.reserve(tot_rows)); chk(S.d_outs.reserve(nb)); chk(S.d_ops.reserve(tot_ops)); chk(S.d_orow.reserve(tot_ops));
        if (dev) chk(S.d_ids.reserve(2 * (size_t)nb));
        if (launch_err != hipSuccess) return {i0, i0, 0.0, slot, 0};
        if (dev) {
            // the graphs are in the device store: one workgroup per problem copies its pieces into this slot's buffers
            for (uint32_t t = 0; t < nb; t++) {
                const uint32_t p = order[i0 + t];
                S.h_probs.p[t] = probs[p]; S.h_ids.p[2 * t] = p; S.h_ids.p[2 * t + 1] = G[p].n_preds;
            }
            chk(hipMemcpyAsync(S.d_probs.p, S.h_probs.p, nb * sizeof(poa_prob), hipMemcpyHostToDevice, st));
            chk(hipMemcpyAsync(S.d_ids.p, S.h_ids.p, 2 * (size_t)nb * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            const sg_store &D = *feed.dev;
            const sg_gather_src g0 = {D.part[0].d_ntab, D.part[0].d_preds, D.part[0].d_sinks, D.part[0].d_seq, (uint32_t)D.part[0].p0};
            const sg_gather_src g1 = {D.part[1].d_ntab, D.part[1].d_preds, D.part[1].d_sinks, D.part[1].d_seq, (uint32_t)D.part[1].p0};
            hipLaunchKernelGGL(k_sg_gather, dim3(nb), dim3(256), 0, st, S.d_ids.p, S.d_probs.p, D.d_off, (uint32_t)D.split, g0, g1, D.d_reads,
                               S.d_ntab.p, S.d_preds.p, S.d_sink.p, (char *)S.d_seq32.p, S.d_q.p);
        } else {
            parallel_for(nb, [&](uint64_t t) {
                const uint32_t p = order[i0 + t];
                const poa_prob &pb = probs[p];
                const poa_prep &g = G[p];
                S.h_probs.p[t] = pb;
                memcpy(S.h_ntab.p + pb.node0, g.ntab.data(), g.ntab.size() * sizeof(uint4));
                if (!g.preds.empty()) memcpy(S.h_preds.p + pb.pred0, g.preds.data(), g.preds.size() * 4);
                if (!g.sinks.empty()) memcpy(S.h_sink.p + pb.sink0, g.sinks.data(), g.sinks.size() * 4);
                // node strings of one problem are contiguous in the view
                memcpy((char *)S.h_seq32.p + pb.seq0, views[p].nodes + views[p].node_off[0], g.N);
                if (g.qlen) memcpy(S.h_q.p + pb.q0, views[p].query, g.qlen);
            });
            chk(hipMemcpyAsync(S.d_probs.p, S.h_probs.p, nb * sizeof(poa_prob), hipMemcpyHostToDevice, st));
            chk(hipMemcpyAsync(S.d_ntab.p, S.h_ntab.p, tot_nodes * sizeof(uint4), hipMemcpyHostToDevice, st));
            chk(hipMemcpyAsync(S.d_seq32.p, S.h_seq32.p, tot_seq, hipMemcpyHostToDevice, st));
            chk(hipMemcpyAsync(S.d_preds.p, S.h_preds.p, tot_preds * 4, hipMemcpyHostToDevice, st));
            chk(hipMemcpyAsync(S.d_sink.p, S.h_sink.p, tot_sink * 4, hipMemcpyHostToDevice, st));
            chk(hipMemcpyAsync(S.d_q.p, S.h_q.p, tot_q, hipMemcpyHostToDevice, st));
        }
        chk(hipMemsetAsync(W.d_next.p + slot, 0, sizeof(unsigned long long), st));
        int t_dp = vga_timer_begin(ctx, "poa_band_dp", 0, st);
        {
            uint32_t mq = 0;
            double mw = 0;
            double sum_w = 0;
            for (uint64_t i = i0; i < i1; i++) { mq = std::max(mq, G[order[i]].qlen); mw = std::max(mw, estw[order[i]]); sum_w += estw[order[i]]; }
            const double mean_w = sum_w / (double)nb;
            const uint32_t lds_cols = poa_lds_cols(mq);
            // VGA_POA_KERNEL (testing): "unpacked" selects k_poa_dp_lds, "128" / "256" / "512" pin the workgroup size,
            // "full" keeps every column in LDS; VGA_POA_WINDOW=<power of two> pins the LDS column window
            const char *force = getenv("VGA_POA_KERNEL");
            int g1bits = 0, g2bits = 0;
            while ((1 << g1bits) <= P.o1 + P.e1) g1bits++;
            while ((1 << g2bits) <= P.o2 + P.e2) g2bits++;
            const bool t4 = t4_k;
            const bool packed = t4 || (pk_built && g1bits + g2bits <= 8 && !(force && strstr(force, "unpacked")));
            packed_all = packed_all && packed && !t4;
            t4_any = t4_any || t4;
            // 16-bit row state (3 B per column): default penalties only; problems it gives up on come back with use32
            const bool def_pen = P.o1 == 4 && P.e1 == 2 && P.o2 == 24 && P.e2 == 1 && !(force && strstr(force, "generic"));
            const bool h16 = packed && def_pen && !use32 && getenv("VGA_POA_H16") && atoi(getenv("VGA_POA_H16")) != 0;
            h16_all = h16_all && h16;
            sub_h16 = h16;
            sub_fused = packed && tb_fused && !getenv("VGA_POA_STAMPS");
            any_fused = any_fused || sub_fused;
            // LDS column window (packed kernel): 4096 columns keep almost every row of a 10 kbp read resident (its widest
            // rows, a few per cent, take the HBM detour described in the kernel) and let seven workgroups share a CU
            // instead of three.  Queries that fit a smaller array anyway keep every column.
            uint32_t hg_cols = lds_cols, win_mask = 0xFFFFFFFFu;
            if (packed && !(force && strstr(force, "full"))) {
                uint32_t want = 4096;
                // narrow bands: a window that just covers the launch's widest estimated row (rows that turn out wider take
                // the HBM detour) leaves room for 12 two-wave workgroups per CU instead of 7 -- such launches are bound by
                // the latency of the per-row chain, not by instruction issue (config 5: +20 %)
                if (mean_w <= 800.0) {
                    uint32_t w2 = 512;
                    while (w2 < 4096 && (double)w2 < mw * 1.25 + 16.0) w2 <<= 1;
                    want = w2;
                }
                if (feed.klass && feed.klass[order[i0]] && !getenv("VGA_POA_NO_GIANTS")) want = 8192;
                const char *ew = getenv("VGA_POA_WINDOW");
                if (ew) want = (uint32_t)strtoul(ew, nullptr, 10);
                if (want >= 16 && (want & (want - 1)) == 0 && want < lds_cols) { hg_cols = want; win_mask = want - 1; }
            }
            // workgroup size: a row of the widest band should take about two steps, and the launch should still fill
            // the GPU (blocks per CU: LDS and 28 waves)
            int nt = mq >= 3072 ? 512 : (mq >= 768 ? 256 : 128);
            int cpt = 4;  // columns per lane and step of the packed kernel
            if (packed) {
                // workgroup size: the one that keeps the most waves resident (LDS and the 28 wave slots of a CU bound the
                // workgroups per CU; the problems still to be run -- this sub-batch and the ones that will overlap it --
                // bound how many there are); ties go to the smaller workgroup, whose barriers are cheaper
                auto by_lds = [&](int t) { return std::max<size_t>(1, (160 * 1024) / ((t4 ? poa_t4_lds_bytes(hg_cols, lds_cols, t) : poa_pk_lds_bytes(hg_cols, lds_cols, t, h16)) + 256)); };
                size_t best_waves = 0;
                for (int t = 128; t <= 512; t += 64) {
                    const size_t per_cu = std::min<size_t>(by_lds(t), (size_t)((t4 ? 16 : (cpt == 8 ? 20 : 24)) / (t / 64)));
                    const size_t waves = std::min<size_t>(order.size() - i0 + in_flight_other, per_cu * (size_t)ctx->n_cu) * (size_t)(t / 64);
                    if (waves > best_waves) { best_waves = waves; nt = t; }
                }
                // narrow bands (one step of a 128-thread workgroup covers a typical row): the per-row set-up and the
                // barriers dominate, and they are per wave -- config 5 (mean width 340): +6 % with 128 threads
                if (mean_w <= 800.0) nt = 128;  // (the estimate is of a problem's widest rows: about twice its mean band)
                if (feed.klass && feed.klass[order[i0]] && !getenv("VGA_POA_NO_GIANTS")) nt = getenv("VGA_POA_GIANT_NT") ? atoi(getenv("VGA_POA_GIANT_NT")) : 1024;  // (config 4: +5 % over 512, same-box)
                const char *ent = getenv("VGA_POA_NT");
                if (ent) nt = atoi(ent);
                if (nt < 128 || (nt > 512 && nt != 768 && nt != 1024) || nt % 64) nt = 512;
            }
            if (force) {
                if (strstr(force, "128")) nt = 128;
                else if (strstr(force, "256")) nt = 256;
                else if (strstr(force, "512")) nt = 512;
            }
            auto lds_of = [&](int t) { return t4 ? poa_t4_lds_bytes(hg_cols, lds_cols, t) : (packed ? poa_pk_lds_bytes(hg_cols, lds_cols, t, h16) : poa_lds_bytes(lds_cols, t)); };
            while (nt > 128 && lds_of(nt) > 160 * 1024 - 256) nt = packed ? nt - 64 : nt / 2;
            const size_t lds = lds_of(nt);
            if (tr.on)
                fprintf(stderr, "[vga-trace] poa: launch %u problems, NT %d, %s, window %u of %u columns, width estimate mean %.0f max %.0f, LDS %zu B\n",
                        nb, nt, t4 ? "t4 rows" : (h16 ? "16-bit rows" : "32-bit rows"), hg_cols, lds_cols, mean_w, mw, lds);
            (void)hipGetLastError();  // a launch failure below must be this launch's, not an older ignored status
#define POA_ARGS S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, S.d_sink.p, P, S.d_rows.p, pool_base,          \
                 W.d_next.p + slot, half_pool, S.d_outs.p, lds_cols
#define POA_PK_ARGS POA_ARGS, hg_cols, win_mask, g1bits, (sub_fused ? S.d_ops.p : nullptr), (sub_fused ? S.d_orow.p : nullptr), \
                    (arena ? n_arenas : 0u), arena_size, W.d_arena_ctr.p, W.d_arena_flag.p
            const bool w1 = w1_k && t4 && !use32 && arena;  // (problems it hands back -- POA_ST_WIDE -- come again with use32 set)
            sub_t4 = t4;
            if (w1) {
                const size_t lds1 = poa_w1_lds_bytes(mq);
                if (tr.on) fprintf(stderr, "[vga-trace] poa: launch %u problems, k_poa_dp_w1 (one wave per problem), LDS %zu B\n", nb, lds1);
                hipLaunchKernelGGL(k_poa_rowprep, dim3(nb), dim3(256), 0, st, S.d_probs.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, S.d_rows.p);
                chk(hipFuncSetAttribute((const void *)k_poa_dp_w1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
                hipLaunchKernelGGL(k_poa_dp_w1, dim3(nb), dim3(64), lds1, st, S.d_probs.p, S.d_q.p, S.d_preds.p, P, S.d_rows.p, pool_base,
                                   W.d_next.p + slot, half_pool, S.d_outs.p, (sub_fused ? S.d_ops.p : nullptr), (sub_fused ? S.d_orow.p : nullptr),
                                   (arena ? n_arenas : 0u), arena_size, W.d_arena_ctr.p, W.d_arena_flag.p, (const uint4 *)S.d_rows.p);
            } else if (t4) {
#define POA_T4_ARGS S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, P, S.d_rows.p, pool_base, W.d_next.p + slot, half_pool,   \
                    S.d_outs.p, lds_cols, hg_cols, win_mask, (sub_fused ? S.d_ops.p : nullptr), (sub_fused ? S.d_orow.p : nullptr),       \
                    (arena ? n_arenas : 0u), arena_size, W.d_arena_ctr.p, W.d_arena_flag.p
#define POA_T4_LAUNCH(T)                                                                                                     \
    case T:                                                                                                                  \
        if (def_pen) {                                                                                                       \
            chk(hipFuncSetAttribute((const void *)k_poa_dp_t4<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((k_poa_dp_t4<T, true>), dim3(nb), dim3(T), lds, st, POA_T4_ARGS);                            \
        } else {                                                                                                             \
            chk(hipFuncSetAttribute((const void *)k_poa_dp_t4<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((k_poa_dp_t4<T, false>), dim3(nb), dim3(T), lds, st, POA_T4_ARGS);                           \
        }                                                                                                                    \
        break;
                switch (nt) {
                    POA_T4_LAUNCH(128) POA_T4_LAUNCH(192) POA_T4_LAUNCH(256) POA_T4_LAUNCH(320)
                    POA_T4_LAUNCH(384) POA_T4_LAUNCH(448) POA_T4_LAUNCH(512) POA_T4_LAUNCH(768) POA_T4_LAUNCH(1024)
                default: chk(hipErrorInvalidValue);
                }
#undef POA_T4_LAUNCH
#undef POA_T4_ARGS
            }
#ifdef VGA_VARIANTS
            else if (packed) {
                if (getenv("VGA_POA_STAMPS") && nt == 512) {
                    // diagnostic: per-segment cycle shares of the first 64 workgroups (tid 0's wave), printed to stderr
                    static unsigned long long *d_st = nullptr;
                    if (!d_st) chk(hipMalloc((void **)&d_st, 2 * 64 * 8 * 8));
                    chk(hipMemsetAsync(d_st, 0, 2 * 64 * 8 * 8, st));
                    chk(hipFuncSetAttribute((const void *)k_poa_dp_pk<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((k_poa_dp_pk<512, true>), dim3(nb), dim3(512), lds, st, POA_PK_ARGS, d_st);
                    unsigned long long h_st[2 * 64 * 8];
                    chk(hipMemcpyAsync(h_st, d_st, sizeof h_st, hipMemcpyDeviceToHost, st));
                    chk(hipStreamSynchronize(st));
                    for (int wsel = 0; wsel < 2; wsel++) {
                        unsigned long long sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot = 0;
                        for (int b2 = 0; b2 < 64 && b2 < (int)nb; b2++)
                            for (int s = 0; s < 8; s++) { sum[s] += h_st[wsel * 512 + b2 * 8 + s]; tot += h_st[wsel * 512 + b2 * 8 + s]; }
                        unsigned long long n_act = 0, n_fast = 0, n_edge = 0;
                        for (int b2 = 0; b2 < 64 && b2 < (int)nb; b2++) {
                            const unsigned long long x = h_st[wsel * 512 + b2 * 8 + 7];
                            n_act += x >> 42; n_fast += x & 0x1fffffull; n_edge += (x >> 21) & 0x1fffffull;
                        }
                        fprintf(stderr, "[vga-stamps] wave %d: active wave-steps %llu, fast %llu (of which edge-patched %llu)\n", wsel * 2,
                                n_act, n_fast, n_edge);
                        fprintf(stderr, "[vga-stamps] wave %d cycles: prologue %llu phase1 %llu (interior: loads %llu) scans %llu step-barrier %llu phase2 %llu row-reduce+barrier %llu (total %llu)\n",
                                wsel * 2, sum[0], sum[1], sum[6], sum[2], sum[3], sum[4], sum[5], tot);
                    }
                } else {
#define POA_PK_LAUNCH2(T, D, C, H)                                                                                          \
    {                                                                                                                       \
        chk(hipFuncSetAttribute((const void *)k_poa_dp_pk<T, false, D, C, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_poa_dp_pk<T, false, D, C, H>), dim3(nb), dim3(T), lds, st, POA_PK_ARGS);                     \
    }
#define POA_PK_LAUNCH(T)                                                                                                    \
    case T:                                                                                                                 \
        if (h16) POA_PK_LAUNCH2(T, true, 4, true)                                                                           \
        else if (def_pen) POA_PK_LAUNCH2(T, true, 4, false)                                                                 \
        else POA_PK_LAUNCH2(T, false, 4, false)                                                                             \
        break;
                    switch (nt) {
                        POA_PK_LAUNCH(128) POA_PK_LAUNCH(192) POA_PK_LAUNCH(256) POA_PK_LAUNCH(320)
                        POA_PK_LAUNCH(384) POA_PK_LAUNCH(448) POA_PK_LAUNCH(512)
                    default: chk(hipErrorInvalidValue);
                    }
#undef POA_PK_LAUNCH2
#undef POA_PK_LAUNCH
                }
            }
#endif  // VGA_VARIANTS
            else if (nt == 128) {
                chk(hipFuncSetAttribute((const void *)k_poa_dp_lds<128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_poa_dp_lds<128, 4>), dim3(nb), dim3(128), lds, st, POA_ARGS);
            } else if (nt == 256) {
                chk(hipFuncSetAttribute((const void *)k_poa_dp_lds<256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_poa_dp_lds<256, 4>), dim3(nb), dim3(256), lds, st, POA_ARGS);
            } else {
                chk(hipFuncSetAttribute((const void *)k_poa_dp_lds<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_poa_dp_lds<512, 4>), dim3(nb), dim3(512), lds, st, POA_ARGS);
            }
#undef POA_PK_ARGS
#undef POA_ARGS
            chk(hipGetLastError());
        }
        vga_timer_end(ctx, t_dp);
        int t_tb = vga_timer_begin(ctx, "poa_traceback", 0, st);
        if (sub_fused) {
            // the DP kernel's first wave already walked each problem back
        }
#ifdef VGA_VARIANTS
        else if (tb_lane) {  // VGA_POA_TB=lane: the one-lane-per-problem walk (diagnostic / cross-check)
            if (sub_t4)
                hipLaunchKernelGGL(k_poa_traceback<1>, dim3((nb + 63) / 64), dim3(64), 0, st, nb, S.d_probs.p, S.d_rows.p, S.d_preds.p,
                                   pool_base, S.d_outs.p, S.d_ops.p, S.d_orow.p, 0);
            else
                hipLaunchKernelGGL(k_poa_traceback<0>, dim3((nb + 63) / 64), dim3(64), 0, st, nb, S.d_probs.p, S.d_rows.p, S.d_preds.p,
                                   pool_base, S.d_outs.p, S.d_ops.p, S.d_orow.p, sub_h16 ? 0xC0 : 0);
        }
#endif
        else if (sub_t4)
            hipLaunchKernelGGL(k_poa_traceback_wave<1>, dim3(nb), dim3(64), 0, st, nb, S.d_probs.p, S.d_rows.p, S.d_preds.p,
                               pool_base, S.d_outs.p, S.d_ops.p, S.d_orow.p, 0);
        else
            hipLaunchKernelGGL(k_poa_traceback_wave<0>, dim3(nb), dim3(64), 0, st, nb, S.d_probs.p, S.d_rows.p, S.d_preds.p,
                               pool_base, S.d_outs.p, S.d_ops.p, S.d_orow.p, sub_h16 ? 0xC0 : 0);
        vga_timer_end(ctx, t_tb);
        chk(hipMemcpyAsync(O.h_outs.p, S.d_outs.p, nb * sizeof(poa_out), hipMemcpyDeviceToHost, st));
        chk(hipMemcpyAsync(W.h_next.p + slot, W.d_next.p + slot, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        chk(hipMemcpyAsync(O.h_ops.p, S.d_ops.p, tot_ops, hipMemcpyDeviceToHost, st));
        chk(hipMemcpyAsync(O.h_orow.p, S.d_orow.p, tot_ops * 4, hipMemcpyDeviceToHost, st));
        return {i0, i1, raw_est, slot, oset, false, arena};
    };
    // host: CIGAR / cs / node path of one problem from the raw op stream (reverse order on the device)
    auto post_one = [&](const poa_slot::out_set &S, uint64_t i0, uint64_t i) {
        const uint32_t p = order[i];
        poa_item &it = out[p];
        const poa_out &ho = S.h_outs.p[i - i0];
        if (ho.status == POA_ST_RANGE || ho.status == POA_ST_POOL || ho.status == POA_ST_WIDE) return;  // re-run later
        it.ok = ho.status == POA_ST_OK ? 1 : 0;
        it.score = ho.score;
        it.n_cells = ho.cells;
        it.n_vcells = ho.vcells;
        if (!it.ok) return;
        const poa_prep &g = G[p];
        const poa_prob &pb = probs[p];
        const uint8_t *po = S.h_ops.p + pb.ops0;
        const uint32_t *pr = S.h_orow.p + pb.ops0;
        const char *q = views[p].query;
        // base of graph row r (rows ascend along the path): bases[r - 1] of the host graph, or -- device store -- the
        // index sequence via the node the row belongs to
        const char *bases = feed.dev ? nullptr : views[p].nodes + views[p].node_off[0];
        const uint32_t *frow = g.first_row_p;
        const size_t nv = (size_t)g.n_ntab - 1;
        size_t vcur = 0;
        auto base_of = [&](uint32_t r) -> char {
            if (bases) return bases[r - 1];
            while (vcur + 1 < nv && frow[vcur + 1] <= r) vcur++;
            return feed.row_base(p, (uint32_t)vcur, r - frow[vcur]);
        };
        const uint32_t nops = ho.nops;
        std::string &cg = it.cigar, &cs = it.cs;
        cs = "cs:Z:";
        cg.reserve(nops / 2 + 16);
        cs.reserve(nops / 2 + 16);
        it.rows.reserve(g.N < nops ? g.N : nops);
        uint64_t eq_run = 0, aligned = 0;
        uint32_t qi = 0;
        uint32_t t2 = nops;
        while (t2 > 0) {
            const uint8_t op = po[t2 - 1];
            uint32_t u = t2, run = 0;
            while (u > 0 && po[u - 1] == op) { u--; run++; }
            append_u(cg, run);
            cg.push_back(op == 0 ? 'M' : (op == 1 ? 'I' : 'D'));
            if (op != 0 && eq_run) { cs.push_back(':'); append_u(cs, eq_run); eq_run = 0; }
            if (op == 1) cs.push_back('+');
            if (op == 2) cs.push_back('-');
            for (uint32_t x = t2; x > u; x--) {
                const uint32_t idx = x - 1;
                if (op == 0) {
                    const char gb = base_of(pr[idx]), qb = q[qi++];
                    aligned++;
                    if (gb == qb) eq_run++;
                    else {
                        if (eq_run) { cs.push_back(':'); append_u(cs, eq_run); eq_run = 0; }
                        cs.push_back('*'); cs.push_back(lower(gb)); cs.push_back(lower(qb));
                    }
                    it.rows.push_back(pr[idx]);
                } else if (op == 1) {
                    cs.push_back(lower(q[qi++]));
                } else {
                    cs.push_back(lower(base_of(pr[idx])));
                    it.rows.push_back(pr[idx]);
                }
            }
            t2 = u;
        }
        if (eq_run) { cs.push_back(':'); append_u(cs, eq_run); }
        it.aligned = (uint32_t)aligned;
        // rows ascend along the path: merge-walk the node table to label them
        it.gnodes.resize(it.rows.size());
        size_t v = 0;
        for (size_t t = 0; t < it.rows.size(); t++) {
            while (v + 1 < nv && frow[v + 1] <= it.rows[t]) v++;
            it.gnodes[t] = (uint32_t)v;
        }
        if (!it.rows.empty()) {
            it.start_off = it.rows.front() - frow[it.gnodes.front()];
            it.end_off = it.rows.back() - frow[it.gnodes.back()] + 1;
        }
    };
    // Software pipeline.  `todo` holds the launch-order ranges still to be enqueued (a sub-batch that overflowed its pool
    // half goes back to the front); up to two sub-batches are in flight, one per stream.  While the GPU works on them the
    // host threads prepare the problems of the next sub-batch (the caller's subgraphs, node tables) and turn the op
    // streams of the sub-batch that just finished into CIGAR / cs strings.
    int rc_final = VGA_OK;
    struct seg_t { uint64_t first, second; bool use32; bool arena; };
    std::vector<seg_t> todo;  // used as a stack of [begin, end) ranges of launch positions, front = back()
    todo.push_back({0, n, false, n_arenas != 0});
    std::vector<uint32_t> retry32;  // problems the 16-bit kernel gave up on (POA_ST_RANGE)
    std::vector<uint32_t> too_big;  // problems that did not fit an arena: classic mode once the arena launches are done
    std::vector<sub_t> inflight;
    bool slot_busy[POA_SLOTS] = {};
    uint64_t all_cells = 0, all_vcells = 0, all_rows = 0, all_q = 0, all_ops = 0;
    auto fill = [&]() {
        while ((int)inflight.size() < n_slots && !todo.empty() && !malformed && !dev_failed && launch_err == hipSuccess) {
            int slot = 0;
            while (slot_busy[slot]) slot++;
            auto &seg = todo.back();
            // arena launches use the whole pool, classic ones its per-slot segments: never both at a time
            if (!inflight.empty() && inflight.front().arena != seg.arena) break;
            in_flight_other = 0;
            for (const sub_t &o : inflight) in_flight_other += o.i1 - o.i0;
            sub_t sb = launch(seg.first, seg.second, slot, seg.use32, seg.arena);
            sb.use32 = seg.use32;
            if (sb.i1 == sb.i0) break;
            if (sb.i1 >= seg.second) todo.pop_back();
            else seg.first = sb.i1;
            slot_busy[slot] = true;
            inflight.push_back(sb);
        }
    };
    fill();
    while (!inflight.empty()) {
        // look ahead: prepare the problems the next launch will start with while the GPU is busy
        if (!todo.empty()) ensure(todo.back().first, std::min<uint64_t>(todo.back().second, todo.back().first + 1536));
        // the launch that finishes first is handled first: a launch of long problems (they come first in the order) must
        // not keep the slots of the shorter ones behind it from being refilled
        size_t pick = 0;
        if (inflight.size() > 1) {
            for (bool found = false; !found;) {
                for (size_t q = 0; q < inflight.size() && !found; q++) {
                    const hipError_t qe = hipStreamQuery(sarr[inflight[q].slot]);
                    if (qe != hipErrorNotReady) { pick = q; found = true; }  // finished (or failed: the synchronize below reports it)
                }
                if (!found) std::this_thread::sleep_for(std::chrono::microseconds(100));
            }
        }
        const sub_t cur = inflight[pick];
        inflight.erase(inflight.begin() + (long)pick);
        {
            const hipError_t se = hipStreamSynchronize(sarr[cur.slot]);
            if (se != hipSuccess) { launch_err = se; break; }  // (falls through to the drain of every stream below)
        }
        const poa_slot::out_set &S = W.slot[cur.slot].outs[cur.oset];
        if (launch_err != hipSuccess) break;
        bool pool_fail = false;
        for (uint64_t i = cur.i0; i < cur.i1; i++)
            if (S.h_outs.p[i - cur.i0].status == POA_ST_POOL) {
                if (cur.arena) too_big.push_back(order[i]);
                else pool_fail = true;
            }
        if (pool_fail) {
            slot_busy[cur.slot] = false;
            if (cur.i1 - cur.i0 == 1 && W.pool_scale >= 8.0) { rc_final = VGA_ERR_POOL; break; }
            W.pool_scale = std::min(16.0, W.pool_scale * 1.7);
            todo.push_back({cur.i0, cur.i1, cur.use32, false});  // enqueue it again, in smaller pieces
            fill();
            continue;
        }
        if (cur.raw_est > 0) {
            const double ratio = (double)W.h_next.p[cur.slot] / cur.raw_est;
            // conservative on purpose: a sub-batch that overflows its half takes its unfinished problems down with it
            W.pool_scale = std::max(ratio * 1.15, 0.6 * W.pool_scale + 0.4 * ratio * 1.25);
        }
        if (const char *dump = getenv("VGA_POA_DUMP_ROWS")) {  // diagnostics: the row records of the launch's first problem
            const poa_prob &pb0 = probs[order[cur.i0]];
            std::vector<poa_row> hr(pb0.N + 1);
            (void)hipMemcpy(hr.data(), W.slot[cur.slot].d_rows.p + pb0.row0, hr.size() * sizeof(poa_row), hipMemcpyDeviceToHost);
            FILE *f = fopen(dump, "w");
            if (f) {
                for (size_t r = 0; r < hr.size(); r++) fprintf(f, "%zu %d %d %d %d %u %u %llu\n", r, hr[r].beg, hr[r].end, hr[r].lmax, hr[r].rmax, hr[r].pred, hr[r].npred, (unsigned long long)hr[r].voff);
                fclose(f);
            }
        }
        if (tr.on) {
            double worst = 0;
            uint32_t mx = 0;
            for (uint64_t i = cur.i0; i < cur.i1; i++) {
                worst = std::max(worst, (double)S.h_outs.p[i - cur.i0].maxw / estw[order[i]]);
                mx = std::max(mx, S.h_outs.p[i - cur.i0].maxw);
            }
            {
                double lsum = 0, nsum = 0;
                uint32_t lmax = 0;
                for (uint64_t i = cur.i0; i < cur.i1; i++) { lsum += G[order[i]].life; nsum += (double)G[order[i]].n_ntab; lmax = std::max(lmax, G[order[i]].life); }
                double csum = 0, vsum = 0, rsum = 0, esum = 0;
                for (uint64_t i = cur.i0; i < cur.i1; i++) {
                    csum += (double)S.h_outs.p[i - cur.i0].cells; vsum += (double)S.h_outs.p[i - cur.i0].vcells; rsum += G[order[i]].N;
                    esum += est[order[i]];
                }
                const double nbd = (double)(cur.i1 - cur.i0);
                fprintf(stderr, "[vga-trace] poa:   edge span (nodes): mean %.1f, max %u; nodes %.0f; rows %.0f, cells %.1f M, value cells %.1f M, "
                                "estimate %.1f MB per problem, pool scale %.2f\n", lsum / nbd, lmax, nsum / nbd, rsum / nbd, csum / nbd / 1e6,
                        vsum / nbd / 1e6, esum / nbd / 1e6, W.pool_scale);
            }
            uint64_t tb = ~0ull, te = 0, tsum = 0;
            for (uint64_t i = cur.i0; i < cur.i1; i++) {
                const poa_out &ho = S.h_outs.p[i - cur.i0];
                if (ho.t_end > ho.t_begin) { tb = std::min(tb, ho.t_begin); te = std::max(te, ho.t_end); tsum += ho.t_end - ho.t_begin; }
            }
            {
                // the longest-running workgroup of the launch: what a single problem costs (its rows are sequential)
                uint64_t worst_i = cur.i0, worst_t = 0;
                for (uint64_t i = cur.i0; i < cur.i1; i++) {
                    const poa_out &ho = S.h_outs.p[i - cur.i0];
                    if (ho.t_end > ho.t_begin && ho.t_end - ho.t_begin > worst_t) { worst_t = ho.t_end - ho.t_begin; worst_i = i; }
                }
                const poa_out &ho = S.h_outs.p[worst_i - cur.i0];
                const poa_prep &g = G[order[worst_i]];
                fprintf(stderr, "[vga-trace] poa:   slowest problem: %.1f ms for %u rows (%.2f us per row), %u nodes, %.1f M cells (mean width %.0f, widest %u), "
                                "%.0f %% of them in kept rows, query %u\n", (double)worst_t / 1e5, g.N, (double)worst_t / 100.0 / (double)std::max(1u, g.N),
                        g.n_ntab - 1, (double)ho.cells / 1e6, (double)ho.cells / (double)std::max(1u, g.N), ho.maxw, 100.0 * (double)ho.vcells / (double)std::max<uint64_t>(1, ho.cells), g.qlen);
            }
            fprintf(stderr, "[vga-trace] poa: sub-batch [%llu, %llu) done, pool %.1f GB, widest row %u columns, worst width / estimate %.3f; "
                            "DP %.1f ms, mean %.1f workgroups resident, on GPU clock %.3f .. %.3f s\n",
                    (unsigned long long)cur.i0, (unsigned long long)cur.i1, (double)W.h_next.p[cur.slot] / 1e9, mx, worst,
                    te > tb ? (double)(te - tb) / 1e5 : 0.0, te > tb ? (double)tsum / (double)(te - tb) : 0.0, (double)(tb % 100000000000ull) / 1e8,
                    (double)(te % 100000000000ull) / 1e8);
        }
        // problems the 16-bit kernel stopped (a score near the int16 range) run again with 32-bit words
        for (uint64_t i = cur.i0; i < cur.i1; i++)
            if (S.h_outs.p[i - cur.i0].status == POA_ST_RANGE || S.h_outs.p[i - cur.i0].status == POA_ST_WIDE) retry32.push_back(order[i]);
        if (todo.empty() && inflight.empty() && !retry32.empty()) {
            const uint64_t a = order.size();
            for (uint32_t p : retry32) order.push_back(p);
            if (tr.on) fprintf(stderr, "[vga-trace] poa: %zu problems handed back (16-bit range / too wide for the register kernel): re-run\n", retry32.size());
            retry32.clear();
            todo.push_back({a, order.size(), true, n_arenas != 0});
        }
        if (todo.empty() && inflight.empty() && !too_big.empty()) {
            const uint64_t a = order.size();
            for (uint32_t p : too_big) order.push_back(p);
            if (tr.on) fprintf(stderr, "[vga-trace] poa: %zu problems did not fit an arena of %.1f MB: classic pass\n", too_big.size(), (double)arena_size / 1e6);
            too_big.clear();
            todo.push_back({a, order.size(), cur.use32, false});
        }
        // refill the GPU first (the new sub-batch's results go to the slot's other result set), then post-process
        slot_busy[cur.slot] = false;
        fill();
        {
            const uint64_t a0 = cur.i0, cnt = cur.i1 - cur.i0;
            parallel_for(cnt, [&](uint64_t t) { post_one(S, a0, a0 + t); });
            for (uint64_t i = cur.i0; i < cur.i1; i++) {
                const poa_out &ho = S.h_outs.p[i - cur.i0];
                if (ho.status == POA_ST_RANGE || ho.status == POA_ST_POOL || ho.status == POA_ST_WIDE) continue;
                all_cells += ho.cells; all_vcells += ho.vcells; all_ops += ho.nops;
                all_rows += G[order[i]].N; all_q += G[order[i]].qlen;
            }
        }
    }
    if (tr.on && feed.proxy) {
        for (uint64_t i = 0; i < order.size() && i < n; i += std::max<uint64_t>(1, n / 12))
            fprintf(stderr, "[vga-trace] poa:   launch position %llu: proxy %.3g, rows %u, longest path %d, query %u, estimate %.1f MB\n",
                    (unsigned long long)i, feed.proxy[order[i]], G[order[i]].N, G[order[i]].longest, G[order[i]].qlen, est[order[i]] / 1e6);
    }
    // drain both streams (also on the error paths: the slots belong to the context)
    for (int i = 1; i < n_slots; i++) (void)hipStreamSynchronize(sarr[i]);
    (void)hipStreamSynchronize(st);
    if (launch_err != hipSuccess) return vga_set_error(ctx, VGA_ERR_HIP, "POA launch failed: %s", hipGetErrorString(launch_err));
    if (malformed) return malformed_error();
    if (dev_failed) return dev_rc;  // (sg_prepare_rest has set the message)
    vga_timer_end(ctx, t_total);
    tr.mark("dp + traceback + cigar (pipelined sub-batches)");
    if (rc_final != VGA_OK)
        return vga_set_error(ctx, rc_final, "a single POA problem does not fit the %llu byte traceback pool",
                             (unsigned long long)W.pool_size);
    POA_CHECK(hipStreamSynchronize(st));
    vga_timers_collect(ctx);
    // byte model of the DP kernel: graph bases + query + 1 direction byte per cell
    // + the value rows kept in HBM (4 B per cell packed, 6 B otherwise), written once and read back at least once
    for (auto &a : ctx->last_times) {
        // (the traceback's 6 bytes per alignment column belong to whichever kernel walked: the DP kernel when fused)
        if (a.name == "poa_band_dp") a.bytes = all_rows + all_q + all_cells + (t4_any ? 12 : (h16_all ? 6 : (packed_all ? 8 : 12))) * all_vcells + (any_fused ? 6 * all_ops : 0);
        if (a.name == "poa_traceback") a.bytes = any_fused ? 0 : 6 * all_ops;
    }
    tm.ms_dp = vga_timer_sum(ctx, "poa_band_dp");
    tm.ms_tb = vga_timer_sum(ctx, "poa_traceback");
    tm.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
#undef POA_CHECK
    return VGA_OK;
}

extern "C" void vga_poa_result_free(vga_poa_result *r)
{
    if (!r) return;
    free(r->ok); free(r->best_score); free(r->path_off); free(r->abpoa_nodes); free(r->graph_nodes);
    free(r->aln_start_offset); free(r->aln_end_offset); free(r->n_aligned_bases); free(r->cigar_off);
    free(r->cigar); free(r->cs_off); free(r->cs); free(r->n_rows); free(r->n_cells); free(r->n_value_cells);
    free(r);
}

static int vga_poa_batch_impl(vga_ctx *ctx, uint64_t n, const uint64_t *node_ptr, const uint64_t *node_off,
                             const char *nodes_concat, const uint64_t *edge_ptr, const uint32_t *edge_src,
                             const uint32_t *edge_dst, const uint64_t *query_off, const char *queries_concat,
                             const vga_poa_params *params, vga_poa_result **out)
{
    if (!ctx || !out || !params || (n && (!node_ptr || !node_off || !nodes_concat || !edge_ptr || !query_off || !queries_concat)))
        return VGA_ERR_ARG;
    *out = nullptr;
    poa_feed feed;
    std::vector<poa_view> &views = feed.views;
    views.resize(n);
    for (uint64_t p = 0; p < n; p++) {
        const uint64_t ql = query_off[p + 1] - query_off[p];
        if (ql >= (1ull << 24)) return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "query %llu too long", (unsigned long long)p);
        views[p] = {node_off + node_ptr[p], nodes_concat, node_ptr[p + 1] - node_ptr[p], edge_src + edge_ptr[p], edge_dst + edge_ptr[p],
                    edge_ptr[p + 1] - edge_ptr[p], queries_concat + query_off[p], (uint32_t)ql};
    }
    std::vector<poa_item> items;
    poa_timing tm;
    int rc = poa_run(ctx, feed, params, items, tm);
    if (rc != VGA_OK) return rc;
    vga_poa_result *res = (vga_poa_result *)calloc(1, sizeof(vga_poa_result));
    if (!res) return vga_set_error(ctx, VGA_ERR_NOMEM, "out of host memory (POA result)");
    auto nomem = [&]() { vga_poa_result_free(res); return vga_set_error(ctx, VGA_ERR_NOMEM, "out of host memory (POA result of %llu problems)", (unsigned long long)n); };
    res->n = n;
    res->ok = pmalloc<uint8_t>(n);
    res->best_score = pmalloc<int32_t>(n);
    res->path_off = pmalloc<uint64_t>(n + 1);
    res->aln_start_offset = pmalloc<uint32_t>(n);
    res->aln_end_offset = pmalloc<uint32_t>(n);
    res->n_aligned_bases = pmalloc<uint32_t>(n);
    res->cigar_off = pmalloc<uint64_t>(n + 1);
    res->cs_off = pmalloc<uint64_t>(n + 1);
    res->n_rows = pmalloc<uint64_t>(n);
    res->n_cells = pmalloc<uint64_t>(n);
    res->n_value_cells = pmalloc<uint64_t>(n);
    if (!res->ok || !res->best_score || !res->path_off || !res->aln_start_offset || !res->aln_end_offset || !res->n_aligned_bases ||
        !res->cigar_off || !res->cs_off || !res->n_rows || !res->n_cells || !res->n_value_cells)
        return nomem();
    uint64_t tp = 0, tc = 0, ts = 0;
    for (uint64_t p = 0; p < n; p++) {
        res->path_off[p] = tp; res->cigar_off[p] = tc; res->cs_off[p] = ts;
        tp += items[p].rows.size(); tc += items[p].cigar.size() + 1; ts += items[p].cs.size() + 1;
    }
    res->path_off[n] = tp; res->cigar_off[n] = tc; res->cs_off[n] = ts;
    res->abpoa_nodes = pmalloc<uint32_t>(tp);
    res->graph_nodes = pmalloc<uint32_t>(tp);
    res->cigar = pmalloc<char>(tc);
    res->cs = pmalloc<char>(ts);
    if (!res->abpoa_nodes || !res->graph_nodes || !res->cigar || !res->cs) return nomem();
    for (uint64_t p = 0; p < n; p++) {
        const poa_item &it = items[p];
        res->ok[p] = it.ok; res->best_score[p] = it.score; res->aln_start_offset[p] = it.start_off;
        res->aln_end_offset[p] = it.end_off; res->n_aligned_bases[p] = it.aligned; res->n_rows[p] = it.n_rows;
        res->n_cells[p] = it.n_cells; res->n_value_cells[p] = it.n_vcells;
        if (!it.rows.empty()) {
            memcpy(res->abpoa_nodes + res->path_off[p], it.rows.data(), it.rows.size() * 4);
            memcpy(res->graph_nodes + res->path_off[p], it.gnodes.data(), it.gnodes.size() * 4);
        }
        memcpy(res->cigar + res->cigar_off[p], it.cigar.c_str(), it.cigar.size() + 1);
        memcpy(res->cs + res->cs_off[p], it.cs.c_str(), it.cs.size() + 1);
    }
    res->ms_dp = tm.ms_dp;
    res->ms_traceback = tm.ms_tb;
    res->ms_total = tm.ms_total;
    *out = res;
    return VGA_OK;
}

extern "C" int vga_poa_batch(vga_ctx *ctx, uint64_t n, const uint64_t *node_ptr, const uint64_t *node_off,
                             const char *nodes_concat, const uint64_t *edge_ptr, const uint32_t *edge_src,
                             const uint32_t *edge_dst, const uint64_t *query_off, const char *queries_concat,
                             const vga_poa_params *params, vga_poa_result **out)
{
    // nothing throws across the C ABI: an allocation failure inside becomes VGA_ERR_NOMEM
    try {
        return vga_poa_batch_impl(ctx, n, node_ptr, node_off, nodes_concat, edge_ptr, edge_src, edge_dst, query_off, queries_concat, params, out);
    } catch (const std::bad_alloc &) {
        return vga_set_error(ctx, VGA_ERR_NOMEM, "vga_poa_batch: out of host memory");
    } catch (const std::exception &e) {
        return vga_set_error(ctx, VGA_ERR_ARG, "vga_poa_batch: %s", e.what());
    }
}

