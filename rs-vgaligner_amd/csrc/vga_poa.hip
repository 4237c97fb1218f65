// vga_poa.hip -- banded partial-order alignment on gfx950: the replacement for the reference's only
// native component, abPOA behind  AbpoaAligner::create_align_safe(nodes, edges, query, Global)
// (src/align.rs:173-203; result fields consumed at src/align.rs:1107,1152-1165).
//
//   K4  k_poa_dp<NT>     one workgroup per (read, subgraph) problem.  Rows = graph bases in
//                        topological order, processed one after the other because the adaptive band
//                        of a row depends on where its predecessors' maxima fell (that dependency is
//                        also why an anti-diagonal wavefront cannot be used: a row's band is unknown
//                        until its predecessor rows are complete).  Lanes run across the band's
//                        columns, NT per step, all accesses coalesced.  The in-row insertion
//                        recurrence is a max-plus prefix scan (wave shuffles + one LDS exchange).
//   K4b k_poa_traceback  one lane per problem walking the 1-byte direction codes.
//
// HBM layout per problem (all carved from one pool by a bump allocator, 1 MiB chunks):
//   direction row  : 1 byte per cell (+3 predecessor-choice bytes per cell on the rare rows with
//                    more than one predecessor) -- the only per-cell data kept for the traceback;
//   value row      : int32 H + two 1-byte clamped gap deltas per cell, kept only for the LAST base of
//                    every graph node (the only rows a later, non-adjacent row can depend on);
//   ping-pong rows : the same 6 B/cell format for "previous row" hand-over inside a node.
// The deltas: a successor only ever needs max(H - (O+E), Ek - E); storing d = min(H - Ek, O) keeps
// exactly that quantity (H - E - d) and the open/extend decision (d == O) in one byte.
//
// Numerics are 32-bit integer and bit-exact against oracle/og_poa.c (see its header for the
// specification: recurrences, tie order, band rule).
#include "vga_common.hpp"

#include <algorithm>
#include <chrono>
#include <thread>

#define POA_NEG (-(1 << 29))
#define POA_IDENT (INT32_MIN / 2)
#define POA_CHUNK (1ull << 20)
#define POA_FLAG_LAST 1u
#define POA_FLAG_FIRST 2u

#define POA_ST_OK 0
#define POA_ST_POOL 1
#define POA_ST_NOALN 2
#define POA_ST_TRACE 3

struct poa_prob {
    uint64_t row0;   // first entry of the per-row arrays (rows 0..N)
    uint64_t pred0;  // first entry of the predecessor list
    uint64_t sink0;  // first entry of the sink predecessor list
    uint64_t q0;     // first query byte
    uint64_t ops0;   // first entry of the traceback output
    uint32_t n_sink;
    uint32_t qlen;
    uint32_t N;
    uint32_t w;      // adaptive band half-width: wb + floor(wf * qlen), computed on the host in double
};

struct poa_dev_params {
    int32_t match, mismatch, o1, e1, o2, e2, banded;
};

__device__ __forceinline__ int poa_sub(const poa_dev_params &P, uint8_t g, uint8_t q)
{
    const bool ga = (g == 'A') | (g == 'C') | (g == 'G') | (g == 'T');
    const bool qa = (q == 'A') | (q == 'C') | (q == 'G') | (q == 'T');
    if (!(ga && qa)) return 0;
    return g == q ? P.match : -P.mismatch;
}

template <int NT>
__global__ __launch_bounds__(NT) void k_poa_dp(
    const poa_prob *__restrict__ probs, const char *__restrict__ queries, const uint4 *__restrict__ row_meta,
    const uint32_t *__restrict__ preds, const uint32_t *__restrict__ sink_preds, poa_dev_params P,
    int32_t *row_beg, int32_t *row_end, uint64_t *row_doff, uint64_t *row_voff, int32_t *row_lmax, int32_t *row_rmax,
    uint8_t *pool, unsigned long long *pool_next, uint64_t pool_size, int32_t *__restrict__ out_score,
    uint32_t *__restrict__ out_row, int32_t *__restrict__ out_status, uint64_t *__restrict__ out_cells,
    uint64_t *__restrict__ out_vcells)
{
    constexpr int NW = NT / 64;
    __shared__ int32_t sA1[2][NT + 1];
    __shared__ int32_t sA2[2][NT + 1];
    __shared__ int32_t sW1[2][NW];
    __shared__ int32_t sW2[2][NW];
    __shared__ int32_t sRed[NW][3];
    __shared__ unsigned long long s_alloc;

    const poa_prob pb = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qlen = (int)pb.qlen;
    const char *query = queries + pb.q0;
    const uint4 *rmeta = row_meta + pb.row0;  // {base | flags << 8 | npred << 16, remain, pred_start, -}: one scalar load per row
    const uint32_t *plist = preds + pb.pred0;
    volatile int32_t *vbeg = row_beg + pb.row0;
    volatile int32_t *vend = row_end + pb.row0;
    volatile uint64_t *vvoff = row_voff + pb.row0;
    volatile int32_t *vlmax = row_lmax + pb.row0;
    volatile int32_t *vrmax = row_rmax + pb.row0;
    uint64_t *gdoff = row_doff + pb.row0;

    const int o1 = P.o1, e1 = P.e1, o2 = P.o2, e2 = P.e2;
    const int bw = (int)pb.w;

    // ---- bump allocation out of the pool (uniform control flow; thread 0 takes the chunk)
    uint64_t dcur = 0, dend = 0, vcur = 0, vendp = 0;
    bool failed = false;
    auto take_chunk = [&](uint64_t &cur, uint64_t &end) {
        __syncthreads();
        if (tid == 0) s_alloc = atomicAdd(pool_next, (unsigned long long)POA_CHUNK);
        __syncthreads();
        uint64_t b = s_alloc;
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

    const uint64_t pp_bytes = 6ull * (uint64_t)(qlen + 1);
    uint64_t pp[2];
    pp[0] = alloc(vcur, vendp, pp_bytes);
    pp[1] = alloc(vcur, vendp, pp_bytes);

    int prev_beg = 0, prev_end = 0, prev_lmax = 0, prev_rmax = 0;
    uint64_t prev_voff = 0;
    uint64_t cells = 0, vcells = 0;

    for (uint32_t r = 0; r <= pb.N && !failed; r++) {
        const uint4 mt = rmeta[r];
        const uint32_t flags = (mt.x >> 8) & 255u;
        const bool first = (flags & POA_FLAG_FIRST) != 0;
        const bool last = (flags & POA_FLAG_LAST) != 0;
        const int np = r == 0 ? 0 : (first ? (int)((mt.x >> 16) & 255u) : 1);
        const uint32_t ps = mt.z;
        // ---- band (abPOA adaptive band; pulls what the predecessors' maxima pushed)
        int mpl, mpr;
        if (r == 0) { mpl = 0; mpr = 0; }
        else if (!first) { mpl = prev_lmax + 1; mpr = prev_rmax + 1; }
        else {
            mpl = INT32_MAX; mpr = 0;
            for (int t = 0; t < np; t++) {
                const uint32_t p = plist[ps + t];
                const int lm = vlmax[p] + 1, rm = vrmax[p] + 1;
                mpl = lm < mpl ? lm : mpl;
                mpr = rm > mpr ? rm : mpr;
            }
        }
        int beg, end;
        if (!P.banded) { beg = 0; end = qlen; }
        else {
            const int diag = qlen - (int)mt.y;
            const int lo = mpl < diag ? mpl : diag;
            const int hi = mpr > diag ? mpr : diag;
            beg = lo - bw; if (beg < 0) beg = 0;
            end = hi + bw; if (end > qlen) end = qlen;
        }
        const int W = end - beg + 1;
        if (r > 0) cells += (uint64_t)W;
        if (last) vcells += (uint64_t)W;
        const uint64_t doff = alloc(dcur, dend, (uint64_t)W * (np > 1 ? 4u : 1u));
        if (failed) break;
        uint64_t voff;
        if (last) { voff = alloc(vcur, vendp, 6ull * (uint64_t)W); if (failed) break; }
        else voff = pp[r & 1u];
        if (tid == 0) {
            vbeg[r] = beg;
            vend[r] = end;
            gdoff[r] = doff;
            vvoff[r] = voff;
        }
        int32_t *Hrow = (int32_t *)(pool + voff);
        uint8_t *d1row = pool + voff + 4ull * (uint64_t)W;
        uint8_t *d2row = d1row + W;
        uint8_t *drow = pool + doff;
        const uint8_t gb = (uint8_t)(mt.x & 255u);

        int carry1 = POA_IDENT, carry2 = POA_IDENT, left1 = POA_IDENT, left2 = POA_IDENT;
        int best = INT32_MIN, lpos = beg, rpos = beg;
        int buf = 0;
        for (int c0 = 0; c0 < W; c0 += NT, buf ^= 1) {
            const int c = c0 + tid;
            const int j = beg + c;
            const bool act = j <= end;
            int m = POA_NEG, ev1 = POA_NEG, ev2 = POA_NEG;
            int pm = 0, p1 = 0, p2 = 0;
            int of1 = 0, of2 = 0;
            int ht, hts = 0;
            if (r > 0) {
                const uint8_t qc = (act && j >= 1) ? (uint8_t)query[j - 1] : (uint8_t)0;
                const int s = poa_sub(P, gb, qc);
                for (int t = 0; t < np; t++) {
                    int bp, ep;
                    uint64_t pv;
                    if (first) {
                        const uint32_t p = plist[ps + t];
                        bp = vbeg[p]; ep = vend[p]; pv = vvoff[p];
                    } else { bp = prev_beg; ep = prev_end; pv = prev_voff; }
                    const int Wp = ep - bp + 1;
                    const int32_t *Hp = (const int32_t *)(pool + pv);
                    const uint8_t *d1p = pool + pv + 4ull * (uint64_t)Wp;
                    const uint8_t *d2p = d1p + Wp;
                    if (act) {
                        const int jm = j - 1 - bp;
                        if (j >= 1 && jm >= 0 && j - 1 <= ep) {
                            const int cnd = Hp[jm] + s;
                            if (cnd > m) { m = cnd; pm = t; }
                        }
                        const int jj = j - bp;
                        if (jj >= 0 && j <= ep) {
                            const int hj = Hp[jj];
                            const int dd1 = d1p[jj], dd2 = d2p[jj];
                            const int c1 = hj - e1 - dd1;
                            if (c1 > ev1) { ev1 = c1; p1 = t; of1 = dd1 == o1; }
                            const int c2 = hj - e2 - dd2;
                            if (c2 > ev2) { ev2 = c2; p2 = t; of2 = dd2 == o2; }
                        }
                    }
                }
                ht = m;
                if (ev1 > ht) { ht = ev1; hts = 1; }
                if (ev2 > ht) { ht = ev2; hts = 2; }
            } else {
                ht = (j == 0) ? 0 : POA_NEG;
            }
            // ---- insertion recurrence as a max-plus prefix scan over the row
            const int a1 = act ? ht + e1 * j : POA_IDENT;
            const int a2 = act ? ht + e2 * j : POA_IDENT;
            int i1 = a1, i2 = a2;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int u1 = __shfl_up(i1, d, 64), u2 = __shfl_up(i2, d, 64);
                if (lane >= d) { i1 = u1 > i1 ? u1 : i1; i2 = u2 > i2 ? u2 : i2; }
            }
            sA1[buf][tid + 1] = a1;
            sA2[buf][tid + 1] = a2;
            if (lane == 63) { sW1[buf][wv] = i1; sW2[buf][wv] = i2; }
            __syncthreads();
            int x1 = __shfl_up(i1, 1, 64), x2 = __shfl_up(i2, 1, 64);
            if (lane == 0) { x1 = POA_IDENT; x2 = POA_IDENT; }
            int pre1 = carry1, pre2 = carry2, all1 = carry1, all2 = carry2;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const int t1 = sW1[buf][q], t2 = sW2[buf][q];
                if (q < wv) { pre1 = t1 > pre1 ? t1 : pre1; pre2 = t2 > pre2 ? t2 : pre2; }
                all1 = t1 > all1 ? t1 : all1;
                all2 = t2 > all2 ? t2 : all2;
            }
            const int P1 = pre1 > x1 ? pre1 : x1;
            const int P2 = pre2 > x2 ? pre2 : x2;
            const int la1 = tid == 0 ? left1 : sA1[buf][tid];
            const int la2 = tid == 0 ? left2 : sA2[buf][tid];
            carry1 = all1; carry2 = all2;
            left1 = sA1[buf][NT]; left2 = sA2[buf][NT];
            int f1 = POA_NEG, f2 = POA_NEG, fo1 = 0, fo2 = 0;
            if (j > beg) {
                f1 = P1 - o1 - e1 * j;
                f2 = P2 - o2 - e2 * j;
                fo1 = P1 == la1;
                fo2 = P2 == la2;
            }
            int h = ht, hs = hts;
            if (f1 > h) { h = f1; hs = 3; }
            if (f2 > h) { h = f2; hs = 4; }
            if (act) {
                const int lo4 = hs < 3 ? hs : 3 + (hs - 3) * 3 + hts;
                const int code = lo4 | (fo1 << 4) | (fo2 << 5) | (of1 << 6) | (of2 << 7);
                int dd1 = h - ev1; dd1 = dd1 < o1 ? dd1 : o1;
                int dd2 = h - ev2; dd2 = dd2 < o2 ? dd2 : o2;
                Hrow[c] = h;
                d1row[c] = (uint8_t)dd1;
                d2row[c] = (uint8_t)dd2;
                drow[c] = (uint8_t)code;
                if (np > 1) {
                    drow[(uint64_t)W + c] = (uint8_t)pm;
                    drow[2ull * W + c] = (uint8_t)p1;
                    drow[3ull * W + c] = (uint8_t)p2;
                }
                if (h > best) { best = h; lpos = j; rpos = j; }
                else if (h == best) rpos = j;
            }
        }
        // ---- row maximum: leftmost / rightmost column (feeds the successors' bands)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int ob = __shfl_xor(best, d, 64), ol = __shfl_xor(lpos, d, 64), orr = __shfl_xor(rpos, d, 64);
            if (ob > best) { best = ob; lpos = ol; rpos = orr; }
            else if (ob == best) { lpos = ol < lpos ? ol : lpos; rpos = orr > rpos ? orr : rpos; }
        }
        if (lane == 0) { sRed[wv][0] = best; sRed[wv][1] = lpos; sRed[wv][2] = rpos; }
        __syncthreads();
        best = sRed[0][0]; lpos = sRed[0][1]; rpos = sRed[0][2];
#pragma unroll
        for (int q = 1; q < NW; q++) {
            const int ob = sRed[q][0], ol = sRed[q][1], orr = sRed[q][2];
            if (ob > best) { best = ob; lpos = ol; rpos = orr; }
            else if (ob == best) { lpos = ol < lpos ? ol : lpos; rpos = orr > rpos ? orr : rpos; }
        }
        if (tid == 0) { vlmax[r] = lpos; vrmax[r] = rpos; }
        prev_beg = beg; prev_end = end; prev_lmax = lpos; prev_rmax = rpos; prev_voff = voff;
        __threadfence_block();
        __syncthreads();  // the row (values + row arrays) is complete and visible to the whole workgroup
    }

    if (tid == 0) {
        out_cells[blockIdx.x] = cells;
        out_vcells[blockIdx.x] = vcells;
        if (failed) {
            out_status[blockIdx.x] = POA_ST_POOL;
            out_score[blockIdx.x] = POA_NEG;
            out_row[blockIdx.x] = 0;
        } else {
            // sink: first predecessor (list order) with the best H at column qlen
            int bestv = INT32_MIN;
            uint32_t brow = 0;
            bool have = false;
            for (uint32_t t = 0; t < pb.n_sink; t++) {
                const uint32_t p = sink_preds[pb.sink0 + t];
                const int bp = vbeg[p], ep = vend[p];
                int val = POA_NEG;
                if (qlen >= bp && qlen <= ep) val = ((const volatile int32_t *)(pool + vvoff[p]))[qlen - bp];
                if (!have || val > bestv) { bestv = val; brow = p; have = true; }
            }
            out_score[blockIdx.x] = bestv;
            out_row[blockIdx.x] = brow;
            out_status[blockIdx.x] = (have && bestv > POA_NEG / 2) ? POA_ST_OK : POA_ST_NOALN;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// K4 (LDS form).  Same recurrences and outputs as k_poa_dp, but the row that was just filled stays in LDS,
// indexed by ABSOLUTE query column and overwritten in place by the next row:
//   Hs[j]  int32  H of the current "previous row" at column j
//   Ds[j]  uint16 d1 | d2 << 8
// so the common predecessor (the row directly above) costs two LDS reads instead of an L2 round trip, the
// workgroup barriers only wait for LDS (s_waitcnt lgkmcnt(0); s_barrier -- direction bytes and node-end value
// rows are fire-and-forget global stores), and a row needs ceil(W/NT) + 1 barriers.
// In-place hazard: inside a chunk every lane reads Hs[j-1], Hs[j] before the chunk's barrier and writes Hs[j]
// after it; the first lane of the NEXT chunk needs the old Hs of this chunk's last column, which the last lane
// parks in `edge` before the barrier.
// Rows whose predecessor is not the row directly above (bubble arms, multi-predecessor rows) read that
// predecessor's 6-byte value row from HBM; such a row starts with a full __syncthreads() (vmcnt(0)) so that the
// stores it depends on have landed.
// The query itself is staged in LDS too and the per-row metadata is one 16-byte scalar load, so the row loop
// issues no vector loads at all: gfx950 retires vector memory operations in order, and a load behind the
// direction-byte stores would wait for them to reach HBM.
// Requires 7 * (max_qlen + 1) + scratch bytes of dynamic LDS (2 workgroups per CU up to ~11 kbp reads).
#define POA_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int NT>
__global__ __launch_bounds__(NT) void k_poa_dp_lds(
    const poa_prob *__restrict__ probs, const char *__restrict__ queries, const uint4 *__restrict__ row_meta,
    const uint32_t *__restrict__ preds, const uint32_t *__restrict__ sink_preds, poa_dev_params P,
    int32_t *row_beg, int32_t *row_end, uint64_t *row_doff, uint64_t *row_voff, int32_t *row_lmax, int32_t *row_rmax,
    uint8_t *pool, unsigned long long *pool_next, uint64_t pool_size, int32_t *__restrict__ out_score,
    uint32_t *__restrict__ out_row, int32_t *__restrict__ out_status, uint64_t *__restrict__ out_cells,
    uint64_t *__restrict__ out_vcells, uint32_t lds_cols)
{
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int32_t *Hs = (int32_t *)smem;
    uint16_t *Ds = (uint16_t *)(smem + 4ull * lds_cols);
    uint8_t *Qs = smem + 6ull * lds_cols;  // the query, staged once: no vector loads in the row loop
    int32_t *scr = (int32_t *)(smem + ((7ull * lds_cols + 15ull) & ~15ull));
    int32_t *sW1 = scr;               // [2][NW] inclusive wave maxima of a1
    int32_t *sW2 = sW1 + 2 * NW;      // [2][NW]
    int32_t *sL1 = sW2 + 2 * NW;      // [2][NW] a1 of each wave's last lane
    int32_t *sL2 = sL1 + 2 * NW;      // [2][NW]
    int32_t *sRed = sL2 + 2 * NW;     // [NW][3]
    int32_t *edgeH = sRed + 3 * NW;   // [2]
    int32_t *edgeD = edgeH + 2;       // [2]
    unsigned long long *s_alloc = (unsigned long long *)(edgeD + 2);

    const poa_prob pb = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qlen = (int)pb.qlen;
    const char *query = queries + pb.q0;
    const uint4 *rmeta = row_meta + pb.row0;  // {base | flags << 8 | npred << 16, remain, pred_start, -}: one scalar load per row
    const uint32_t *plist = preds + pb.pred0;
    volatile int32_t *vbeg = row_beg + pb.row0;
    volatile int32_t *vend = row_end + pb.row0;
    volatile uint64_t *vvoff = row_voff + pb.row0;
    volatile int32_t *vlmax = row_lmax + pb.row0;
    volatile int32_t *vrmax = row_rmax + pb.row0;
    uint64_t *gdoff = row_doff + pb.row0;

    const int o1 = P.o1, e1 = P.e1, o2 = P.o2, e2 = P.e2;
    const int bw = (int)pb.w;

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

    for (int t = tid; t < qlen; t += NT) Qs[t] = (uint8_t)query[t];
    __syncthreads();

    int prev_beg = 0, prev_end = -1, prev_lmax = 0, prev_rmax = 0;
    uint64_t cells = 0, vcells = 0;

    for (uint32_t r = 0; r <= pb.N && !failed; r++) {
        const uint4 mt = rmeta[r];
        const uint32_t flags = (mt.x >> 8) & 255u;
        const bool first = (flags & POA_FLAG_FIRST) != 0;
        const bool last = (flags & POA_FLAG_LAST) != 0;
        const int np = r == 0 ? 0 : (first ? (int)((mt.x >> 16) & 255u) : 1);
        const uint32_t ps = mt.z;
        // does any predecessor live in HBM (i.e. is not the row directly above)?
        bool far = false;
        if (first)
            for (int t = 0; t < np; t++) far |= plist[ps + t] != r - 1;
        if (far) __syncthreads();  // vmcnt(0) + barrier: value rows / row arrays of far predecessors have landed
        int mpl, mpr;
        if (r == 0) { mpl = 0; mpr = 0; }
        else if (!first) { mpl = prev_lmax + 1; mpr = prev_rmax + 1; }
        else {
            mpl = INT32_MAX; mpr = 0;
            for (int t = 0; t < np; t++) {
                const uint32_t p = plist[ps + t];
                int lm, rm;
                if (p == r - 1) { lm = prev_lmax + 1; rm = prev_rmax + 1; }
                else { lm = vlmax[p] + 1; rm = vrmax[p] + 1; }
                mpl = lm < mpl ? lm : mpl;
                mpr = rm > mpr ? rm : mpr;
            }
        }
        int beg, end;
        if (!P.banded) { beg = 0; end = qlen; }
        else {
            const int diag = qlen - (int)mt.y;
            const int lo = mpl < diag ? mpl : diag;
            const int hi = mpr > diag ? mpr : diag;
            beg = lo - bw; if (beg < 0) beg = 0;
            end = hi + bw; if (end > qlen) end = qlen;
        }
        const int W = end - beg + 1;
        if (r > 0) cells += (uint64_t)W;
        if (last) vcells += (uint64_t)W;
        const uint64_t doff = alloc(dcur, dend, (uint64_t)W * (np > 1 ? 4u : 1u));
        if (failed) break;
        uint64_t voff = 0;
        if (last) { voff = alloc(vcur, vendp, 6ull * (uint64_t)W); if (failed) break; }
        if (tid == 0) {
            vbeg[r] = beg;
            vend[r] = end;
            gdoff[r] = doff;
            vvoff[r] = voff;
        }
        int32_t *Hrow = (int32_t *)(pool + voff);
        uint8_t *d1row = pool + voff + 4ull * (uint64_t)W;
        uint8_t *d2row = d1row + W;
        uint8_t *drow = pool + doff;
        const uint8_t gb = (uint8_t)(mt.x & 255u);

        int carry1 = POA_IDENT, carry2 = POA_IDENT, left1 = POA_IDENT, left2 = POA_IDENT;
        int best = INT32_MIN, lpos = beg, rpos = beg;
        int buf = 0;
        for (int c0 = 0; c0 < W; c0 += NT, buf ^= 1) {
            const int c = c0 + tid;
            const int j = beg + c;
            const bool act = j <= end;
            int m = POA_NEG, ev1 = POA_NEG, ev2 = POA_NEG;
            int pm = 0, p1 = 0, p2 = 0;
            int of1 = 0, of2 = 0;
            int ht, hts = 0;
            if (r > 0) {
                const uint8_t qc = (act && j >= 1) ? Qs[j - 1] : (uint8_t)0;
                const int s = poa_sub(P, gb, qc);
                for (int t = 0; t < np; t++) {
                    const uint32_t p = first ? plist[ps + t] : r - 1;
                    if (p == r - 1) {
                        // the row directly above: LDS, absolute columns
                        int hj = POA_NEG, dj = 0;
                        const bool inj = act && j >= prev_beg && j <= prev_end;
                        if (inj) { hj = Hs[j]; dj = Ds[j]; }
                        if (tid == NT - 1) { edgeH[buf] = hj; edgeD[buf] = inj ? 1 : 0; }
                        int hm = POA_NEG;
                        bool inm = act && j >= 1 && j - 1 >= prev_beg && j - 1 <= prev_end;
                        if (inm) {
                            if (tid == 0 && c0 > 0) { hm = edgeH[buf ^ 1]; inm = edgeD[buf ^ 1] != 0; }
                            else hm = Hs[j - 1];
                        }
                        if (inm) {
                            const int cnd = hm + s;
                            if (cnd > m) { m = cnd; pm = t; }
                        }
                        if (inj) {
                            const int dd1 = dj & 255, dd2 = dj >> 8;
                            const int c1 = hj - e1 - dd1;
                            if (c1 > ev1) { ev1 = c1; p1 = t; of1 = dd1 == o1; }
                            const int c2 = hj - e2 - dd2;
                            if (c2 > ev2) { ev2 = c2; p2 = t; of2 = dd2 == o2; }
                        }
                    } else {
                        const int bp = vbeg[p], ep = vend[p];
                        const uint64_t pv = vvoff[p];
                        const int Wp = ep - bp + 1;
                        const int32_t *Hp = (const int32_t *)(pool + pv);
                        const uint8_t *d1p = pool + pv + 4ull * (uint64_t)Wp;
                        const uint8_t *d2p = d1p + Wp;
                        if (act) {
                            const int jm = j - 1 - bp;
                            if (j >= 1 && jm >= 0 && j - 1 <= ep) {
                                const int cnd = Hp[jm] + s;
                                if (cnd > m) { m = cnd; pm = t; }
                            }
                            const int jj = j - bp;
                            if (jj >= 0 && j <= ep) {
                                const int hj = Hp[jj];
                                const int dd1 = d1p[jj], dd2 = d2p[jj];
                                const int c1 = hj - e1 - dd1;
                                if (c1 > ev1) { ev1 = c1; p1 = t; of1 = dd1 == o1; }
                                const int c2 = hj - e2 - dd2;
                                if (c2 > ev2) { ev2 = c2; p2 = t; of2 = dd2 == o2; }
                            }
                        }
                    }
                }
                ht = m;
                if (ev1 > ht) { ht = ev1; hts = 1; }
                if (ev2 > ht) { ht = ev2; hts = 2; }
            } else {
                ht = (j == 0) ? 0 : POA_NEG;
            }
            const int a1 = act ? ht + e1 * j : POA_IDENT;
            const int a2 = act ? ht + e2 * j : POA_IDENT;
            int i1 = a1, i2 = a2;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int u1 = __shfl_up(i1, d, 64), u2 = __shfl_up(i2, d, 64);
                if (lane >= d) { i1 = u1 > i1 ? u1 : i1; i2 = u2 > i2 ? u2 : i2; }
            }
            if (lane == 63) {
                sW1[buf * NW + wv] = i1; sW2[buf * NW + wv] = i2;
                sL1[buf * NW + wv] = a1; sL2[buf * NW + wv] = a2;
            }
            POA_LDS_BARRIER();
            int x1 = __shfl_up(i1, 1, 64), x2 = __shfl_up(i2, 1, 64);
            int la1 = __shfl_up(a1, 1, 64), la2 = __shfl_up(a2, 1, 64);
            if (lane == 0) {
                x1 = POA_IDENT; x2 = POA_IDENT;
                la1 = wv == 0 ? left1 : sL1[buf * NW + wv - 1];
                la2 = wv == 0 ? left2 : sL2[buf * NW + wv - 1];
            }
            int pre1 = carry1, pre2 = carry2, all1 = carry1, all2 = carry2;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const int t1 = sW1[buf * NW + q], t2 = sW2[buf * NW + q];
                if (q < wv) { pre1 = t1 > pre1 ? t1 : pre1; pre2 = t2 > pre2 ? t2 : pre2; }
                all1 = t1 > all1 ? t1 : all1;
                all2 = t2 > all2 ? t2 : all2;
            }
            const int P1 = pre1 > x1 ? pre1 : x1;
            const int P2 = pre2 > x2 ? pre2 : x2;
            carry1 = all1; carry2 = all2;
            left1 = sL1[buf * NW + NW - 1]; left2 = sL2[buf * NW + NW - 1];
            int f1 = POA_NEG, f2 = POA_NEG, fo1 = 0, fo2 = 0;
            if (j > beg) {
                f1 = P1 - o1 - e1 * j;
                f2 = P2 - o2 - e2 * j;
                fo1 = P1 == la1;
                fo2 = P2 == la2;
            }
            int h = ht, hs = hts;
            if (f1 > h) { h = f1; hs = 3; }
            if (f2 > h) { h = f2; hs = 4; }
            if (act) {
                const int lo4 = hs < 3 ? hs : 3 + (hs - 3) * 3 + hts;
                const int code = lo4 | (fo1 << 4) | (fo2 << 5) | (of1 << 6) | (of2 << 7);
                int dd1 = h - ev1; dd1 = dd1 < o1 ? dd1 : o1;
                int dd2 = h - ev2; dd2 = dd2 < o2 ? dd2 : o2;
                Hs[j] = h;
                Ds[j] = (uint16_t)(dd1 | (dd2 << 8));
                drow[c] = (uint8_t)code;
                if (last) {
                    Hrow[c] = h;
                    d1row[c] = (uint8_t)dd1;
                    d2row[c] = (uint8_t)dd2;
                }
                if (np > 1) {
                    drow[(uint64_t)W + c] = (uint8_t)pm;
                    drow[2ull * W + c] = (uint8_t)p1;
                    drow[3ull * W + c] = (uint8_t)p2;
                }
                if (h > best) { best = h; lpos = j; rpos = j; }
                else if (h == best) rpos = j;
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int ob = __shfl_xor(best, d, 64), ol = __shfl_xor(lpos, d, 64), orr = __shfl_xor(rpos, d, 64);
            if (ob > best) { best = ob; lpos = ol; rpos = orr; }
            else if (ob == best) { lpos = ol < lpos ? ol : lpos; rpos = orr > rpos ? orr : rpos; }
        }
        if (lane == 0) { sRed[wv * 3 + 0] = best; sRed[wv * 3 + 1] = lpos; sRed[wv * 3 + 2] = rpos; }
        POA_LDS_BARRIER();  // row complete in LDS; also fences the scratch buffers between rows
        best = sRed[0]; lpos = sRed[1]; rpos = sRed[2];
#pragma unroll
        for (int q = 1; q < NW; q++) {
            const int ob = sRed[q * 3], ol = sRed[q * 3 + 1], orr = sRed[q * 3 + 2];
            if (ob > best) { best = ob; lpos = ol; rpos = orr; }
            else if (ob == best) { lpos = ol < lpos ? ol : lpos; rpos = orr > rpos ? orr : rpos; }
        }
        if (tid == 0) { vlmax[r] = lpos; vrmax[r] = rpos; }
        prev_beg = beg; prev_end = end; prev_lmax = lpos; prev_rmax = rpos;
    }
    __syncthreads();
    if (tid == 0) {
        out_cells[blockIdx.x] = cells;
        out_vcells[blockIdx.x] = vcells;
        if (failed) {
            out_status[blockIdx.x] = POA_ST_POOL;
            out_score[blockIdx.x] = POA_NEG;
            out_row[blockIdx.x] = 0;
        } else {
            int bestv = INT32_MIN;
            uint32_t brow = 0;
            bool have = false;
            for (uint32_t t = 0; t < pb.n_sink; t++) {
                const uint32_t p = sink_preds[pb.sink0 + t];
                const int bp = vbeg[p], ep = vend[p];
                int val = POA_NEG;
                if (qlen >= bp && qlen <= ep) val = ((const volatile int32_t *)(pool + vvoff[p]))[qlen - bp];
                if (!have || val > bestv) { bestv = val; brow = p; have = true; }
            }
            out_score[blockIdx.x] = bestv;
            out_row[blockIdx.x] = brow;
            out_status[blockIdx.x] = (have && bestv > POA_NEG / 2) ? POA_ST_OK : POA_ST_NOALN;
        }
    }
}

static inline size_t poa_lds_bytes(uint32_t lds_cols, int nt)
{
    const int nw = nt / 64;
    return ((7ull * lds_cols + 15ull) & ~15ull) + (size_t)(8 * nw + 3 * nw + 4) * 4 + 16;
}

// K4b: one lane per problem.  ops are written in reverse (sink -> source) order.
__global__ __launch_bounds__(64) void k_poa_traceback(
    uint32_t n, const poa_prob *__restrict__ probs, const uint4 *__restrict__ row_meta,
    const uint32_t *__restrict__ preds, const int32_t *__restrict__ row_beg, const int32_t *__restrict__ row_end,
    const uint64_t *__restrict__ row_doff, const uint8_t *__restrict__ pool, const uint32_t *__restrict__ out_row,
    int32_t *__restrict__ out_status, uint8_t *__restrict__ ops, uint32_t *__restrict__ orow,
    uint32_t *__restrict__ out_nops)
{
    const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= n) return;
    out_nops[pi] = 0;
    if (out_status[pi] != POA_ST_OK) return;
    const poa_prob pb = probs[pi];
    const uint64_t cap = (uint64_t)pb.N + pb.qlen + 2;
    uint8_t *po = ops + pb.ops0;
    uint32_t *pr = orow + pb.ops0;
    uint32_t i = out_row[pi];
    int j = (int)pb.qlen;
    int st = 0;  // 0 H, 1 E1, 2 E2, 3 F1, 4 F2, 5 Ht
    uint64_t nops = 0;
    bool bad = false;
    while (i > 0 && !bad) {
        const uint64_t ri = pb.row0 + i;
        const uint4 mt = row_meta[ri];
        const uint32_t flags = (mt.x >> 8) & 255u;
        const bool first = (flags & POA_FLAG_FIRST) != 0;
        const int np = first ? (int)((mt.x >> 16) & 255u) : 1;
        const int beg = row_beg[ri], end = row_end[ri];
        const uint64_t W = (uint64_t)(end - beg + 1);
        const uint64_t doff = row_doff[ri];
        if (j < beg || j > end) { bad = true; break; }
        const uint64_t c = (uint64_t)(j - beg);
        const int code = pool[doff + c];
        const int lo4 = code & 15;
        const int hs = lo4 < 3 ? lo4 : 3 + (lo4 - 3) / 3;
        const int hts = lo4 < 3 ? lo4 : (lo4 - 3) % 3;
        const int src = st == 0 ? hs : (st == 5 ? hts : st);
        if (nops + 1 >= cap) { bad = true; break; }
        if (src == 0) {
            const int t = np > 1 ? pool[doff + W + c] : 0;
            const uint32_t p = first ? preds[pb.pred0 + mt.z + t] : i - 1;
            if (j < 1) { bad = true; break; }
            po[nops] = 0; pr[nops] = i; nops++;
            i = p; j -= 1; st = 0;
        } else if (src == 1 || src == 2) {
            const int t = np > 1 ? pool[doff + (src == 1 ? 2 : 3) * W + c] : 0;
            const uint32_t p = first ? preds[pb.pred0 + mt.z + t] : i - 1;
            const int open = (code >> (src == 1 ? 6 : 7)) & 1;
            po[nops] = 2; pr[nops] = i; nops++;
            st = open ? 0 : src;
            i = p;
        } else {
            const int open = (code >> (src == 3 ? 4 : 5)) & 1;
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
    if (bad) { out_status[pi] = POA_ST_TRACE; nops = 0; }
    out_nops[pi] = (uint32_t)nops;
}

// ============================================================================================ host
namespace {

struct poa_graph_host {
    uint32_t N = 0, qlen = 0;
    std::vector<uint8_t> base, flags, npred;
    std::vector<int32_t> remain;
    std::vector<uint32_t> pred_start, preds, sink, row_node, first_row;
};

// rows, predecessor lists (edge-list order), remain[]; mirrors the row construction of
// oracle/og_poa.c (which restates abPOA's graph of single-base nodes).  Returns false on bad input.
bool poa_prepare(const uint64_t *node_off, uint64_t n_nodes, const char *nodes_concat, const uint32_t *esrc,
                 const uint32_t *edst, uint64_t n_edges, uint32_t qlen, poa_graph_host &g)
{
    if (n_nodes == 0) return false;
    std::vector<uint32_t> last_row(n_nodes);
    g.first_row.resize(n_nodes);
    uint64_t N = 0;
    for (uint64_t v = 0; v < n_nodes; v++) {
        uint64_t len = node_off[v + 1] - node_off[v];
        if (len == 0) return false;
        g.first_row[v] = (uint32_t)(N + 1);
        N += len;
        last_row[v] = (uint32_t)N;
    }
    if (N >= (1ull << 31)) return false;
    g.N = (uint32_t)N;
    g.qlen = qlen;
    g.base.assign(N + 1, 0);
    g.flags.assign(N + 1, 0);
    g.npred.assign(N + 1, 1);
    g.remain.assign(N + 1, 0);
    g.pred_start.assign(N + 1, 0);
    g.row_node.assign(N + 1, 0);
    std::vector<uint32_t> in_off(n_nodes + 1, 0), out_off(n_nodes + 1, 0);
    for (uint64_t e = 0; e < n_edges; e++) {
        if (esrc[e] >= edst[e] || edst[e] >= n_nodes) return false;
        in_off[edst[e] + 1]++;
        out_off[esrc[e] + 1]++;
    }
    for (uint64_t v = 0; v < n_nodes; v++) { in_off[v + 1] += in_off[v]; out_off[v + 1] += out_off[v]; }
    std::vector<uint32_t> in_adj(n_edges ? n_edges : 1), out_adj(n_edges ? n_edges : 1), fi(n_nodes, 0), fo(n_nodes, 0);
    for (uint64_t e = 0; e < n_edges; e++) {
        in_adj[in_off[edst[e]] + fi[edst[e]]++] = esrc[e];
        out_adj[out_off[esrc[e]] + fo[esrc[e]]++] = edst[e];
    }
    g.flags[0] = POA_FLAG_LAST;
    g.npred[0] = 0;
    g.preds.clear();
    for (uint64_t v = 0; v < n_nodes; v++) {
        const char *s = nodes_concat + node_off[v];
        uint32_t fr = g.first_row[v], lr = last_row[v];
        for (uint32_t r = fr; r <= lr; r++) { g.base[r] = (uint8_t)s[r - fr]; g.row_node[r] = (uint32_t)v; }
        g.flags[fr] |= POA_FLAG_FIRST;
        g.flags[lr] |= POA_FLAG_LAST;
        g.pred_start[fr] = (uint32_t)g.preds.size();
        uint32_t deg = in_off[v + 1] - in_off[v];
        if (deg == 0) { g.preds.push_back(0); g.npred[fr] = 1; }
        else {
            if (deg > 255) return false;
            for (uint32_t t = in_off[v]; t < in_off[v + 1]; t++) g.preds.push_back(last_row[in_adj[t]]);
            g.npred[fr] = (uint8_t)deg;
        }
        if (out_off[v + 1] == out_off[v]) g.sink.push_back(lr);
    }
    for (uint64_t v = n_nodes; v-- > 0;) {
        int32_t rl = 0;
        for (uint32_t t = out_off[v]; t < out_off[v + 1]; t++) {
            int32_t c = 1 + g.remain[g.first_row[out_adj[t]]];
            if (c > rl) rl = c;
        }
        g.remain[last_row[v]] = rl;
        for (uint32_t r = last_row[v]; r-- > g.first_row[v];) g.remain[r] = g.remain[r + 1] + 1;
    }
    for (uint64_t v = 0; v < n_nodes; v++)
        if (in_off[v + 1] == in_off[v]) {
            int32_t c = 1 + g.remain[g.first_row[v]];
            if (c > g.remain[0]) g.remain[0] = c;
        }
    return true;
}

struct poa_ws {
    vga_dbuf<poa_prob> d_probs;
    vga_dbuf<uint4> d_meta;
    vga_dbuf<uint8_t> d_ops;
    vga_dbuf<int32_t> d_beg, d_end, d_lmax, d_rmax, d_score, d_status;
    vga_dbuf<uint32_t> d_preds, d_sink, d_row, d_orow, d_nops;
    vga_dbuf<uint64_t> d_doff, d_voff, d_cells, d_vcells;
    vga_dbuf<char> d_q;
    vga_dbuf<unsigned long long> d_next;
    uint8_t *pool = nullptr;
    uint64_t pool_size = 0;
    ~poa_ws() { if (pool) (void)hipFree(pool); }
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

}  // namespace

extern "C" void vga_poa_result_free(vga_poa_result *r)
{
    if (!r) return;
    free(r->ok); free(r->best_score); free(r->path_off); free(r->abpoa_nodes); free(r->graph_nodes);
    free(r->aln_start_offset); free(r->aln_end_offset); free(r->n_aligned_bases); free(r->cigar_off);
    free(r->cigar); free(r->cs_off); free(r->cs); free(r->n_rows); free(r->n_cells); free(r->n_value_cells);
    free(r);
}

extern "C" int vga_poa_batch(vga_ctx *ctx, uint64_t n, const uint64_t *node_ptr, const uint64_t *node_off,
                             const char *nodes_concat, const uint64_t *edge_ptr, const uint32_t *edge_src,
                             const uint32_t *edge_dst, const uint64_t *query_off, const char *queries_concat,
                             const vga_poa_params *params, vga_poa_result **out)
{
    if (!ctx || !out || !params || (n && (!node_ptr || !node_off || !nodes_concat || !edge_ptr || !query_off || !queries_concat)))
        return VGA_ERR_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    auto t_host0 = std::chrono::steady_clock::now();
    vga_trace tr("poa");
    if (params->gap_open1 < 0 || params->gap_open1 > 255 || params->gap_open2 < 0 || params->gap_open2 > 255 ||
        params->gap_ext1 < 0 || params->gap_ext2 < 0)
        return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "gap open penalties must be in 0..255");

    vga_poa_result *res = (vga_poa_result *)calloc(1, sizeof(vga_poa_result));
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
    res->path_off[0] = res->cigar_off[0] = res->cs_off[0] = 0;
    vga_timers_reset(ctx);
    if (n == 0) {
        res->abpoa_nodes = pmalloc<uint32_t>(0);
        res->graph_nodes = pmalloc<uint32_t>(0);
        res->cigar = pmalloc<char>(0);
        res->cs = pmalloc<char>(0);
        *out = res;
        return VGA_OK;
    }

    // ---- host: build the row graphs (threads over problems)
    std::vector<poa_graph_host> G(n);
    std::vector<uint8_t> okprep(n, 0);
    {
        unsigned nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 4;
        if (nt > 32) nt = 32;
        if ((uint64_t)nt > n) nt = (unsigned)n;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                for (uint64_t p = t; p < n; p += nt) {
                    const uint64_t nv = node_ptr[p + 1] - node_ptr[p];
                    const uint64_t ne = edge_ptr[p + 1] - edge_ptr[p];
                    const uint64_t ql = query_off[p + 1] - query_off[p];
                    if (ql >= (1ull << 18)) { okprep[p] = 0; continue; }  // 6*(qlen+1) must fit a pool chunk
                    // node_off entries are absolute offsets into nodes_concat
                    okprep[p] = poa_prepare(node_off + node_ptr[p], nv, nodes_concat, edge_src + edge_ptr[p],
                                            edge_dst + edge_ptr[p], ne, (uint32_t)ql, G[p])
                                    ? 1
                                    : 0;
                }
            });
        for (auto &x : th) x.join();
    }
    for (uint64_t p = 0; p < n; p++)
        if (!okprep[p]) {
            vga_poa_result_free(res);
            return vga_set_error(ctx, VGA_ERR_ARG,
                                 "vga_poa_batch: problem %llu is malformed (empty node, edge with src >= dst, in-degree > 255 "
                                 "or query longer than 262143)",
                                 (unsigned long long)p);
        }

    tr.mark("row graphs (host threads)");
    // ---- flatten
    std::vector<poa_prob> probs(n);
    uint64_t tot_rows = 0, tot_preds = 0, tot_sink = 0, tot_q = 0, tot_ops = 0;
    for (uint64_t p = 0; p < n; p++) {
        poa_prob &pb = probs[p];
        pb.row0 = tot_rows; pb.pred0 = tot_preds; pb.sink0 = tot_sink; pb.q0 = tot_q; pb.ops0 = tot_ops;
        pb.n_sink = (uint32_t)G[p].sink.size(); pb.qlen = G[p].qlen; pb.N = G[p].N;
        pb.w = params->wb < 0 ? G[p].qlen : (uint32_t)((int64_t)params->wb + (int64_t)(params->wf * (double)G[p].qlen));
        tot_rows += (uint64_t)G[p].N + 1;
        tot_preds += G[p].preds.size();
        tot_sink += G[p].sink.size();
        tot_q += G[p].qlen;
        tot_ops += (uint64_t)G[p].N + G[p].qlen + 2;
        res->n_rows[p] = G[p].N;
    }
    std::vector<uint4> h_meta(tot_rows);
    std::vector<uint32_t> h_preds(tot_preds ? tot_preds : 1), h_sink(tot_sink ? tot_sink : 1);
    std::vector<char> h_q(tot_q ? tot_q : 1);
    for (uint64_t p = 0; p < n; p++) {
        const poa_prob &pb = probs[p];
        const poa_graph_host &g = G[p];
        for (uint32_t r = 0; r <= g.N; r++)
            h_meta[pb.row0 + r] = make_uint4((uint32_t)g.base[r] | ((uint32_t)g.flags[r] << 8) | ((uint32_t)g.npred[r] << 16),
                                             (uint32_t)g.remain[r], g.pred_start[r], 0u);
        if (!g.preds.empty()) memcpy(&h_preds[pb.pred0], g.preds.data(), g.preds.size() * 4);
        if (!g.sink.empty()) memcpy(&h_sink[pb.sink0], g.sink.data(), g.sink.size() * 4);
        if (g.qlen) memcpy(&h_q[pb.q0], queries_concat + query_off[p], g.qlen);
    }

#define POA_CHECK(call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            vga_poa_result_free(res);                                                                \
            return vga_set_error(ctx, VGA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                                 __LINE__);                                                          \
        }                                                                                            \
    } while (0)

    tr.mark("flatten");
    if (!ctx->poa_ws) {
        ctx->poa_ws = new poa_ws();
        ctx->poa_ws_free = [](void *q) { delete (poa_ws *)q; };
    }
    poa_ws &W_ = *(poa_ws *)ctx->poa_ws;
    auto &d_probs = W_.d_probs; auto &d_meta = W_.d_meta; auto &d_ops = W_.d_ops;
    auto &d_beg = W_.d_beg; auto &d_end = W_.d_end; auto &d_lmax = W_.d_lmax; auto &d_rmax = W_.d_rmax;
    auto &d_score = W_.d_score; auto &d_status = W_.d_status; auto &d_preds = W_.d_preds;
    auto &d_sink = W_.d_sink; auto &d_row = W_.d_row; auto &d_orow = W_.d_orow; auto &d_nops = W_.d_nops; auto &d_doff = W_.d_doff;
    auto &d_voff = W_.d_voff; auto &d_cells = W_.d_cells; auto &d_vcells = W_.d_vcells; auto &d_q = W_.d_q; auto &d_next = W_.d_next;
    POA_CHECK(d_probs.reserve(n)); POA_CHECK(d_meta.reserve(tot_rows));
    POA_CHECK(d_preds.reserve(h_preds.size())); POA_CHECK(d_sink.reserve(h_sink.size())); POA_CHECK(d_q.reserve(h_q.size()));
    POA_CHECK(d_beg.reserve(tot_rows)); POA_CHECK(d_end.reserve(tot_rows)); POA_CHECK(d_doff.reserve(tot_rows));
    POA_CHECK(d_voff.reserve(tot_rows)); POA_CHECK(d_lmax.reserve(tot_rows)); POA_CHECK(d_rmax.reserve(tot_rows));
    POA_CHECK(d_score.reserve(n)); POA_CHECK(d_status.reserve(n)); POA_CHECK(d_row.reserve(n)); POA_CHECK(d_cells.reserve(n)); POA_CHECK(d_vcells.reserve(n));
    POA_CHECK(d_ops.reserve(tot_ops)); POA_CHECK(d_orow.reserve(tot_ops)); POA_CHECK(d_nops.reserve(n));
    POA_CHECK(d_next.reserve(1));
    POA_CHECK(hipMemcpyAsync(d_probs.p, probs.data(), n * sizeof(poa_prob), hipMemcpyHostToDevice, st));
    POA_CHECK(hipMemcpyAsync(d_meta.p, h_meta.data(), tot_rows * sizeof(uint4), hipMemcpyHostToDevice, st));
    POA_CHECK(hipMemcpyAsync(d_preds.p, h_preds.data(), h_preds.size() * 4, hipMemcpyHostToDevice, st));
    POA_CHECK(hipMemcpyAsync(d_sink.p, h_sink.data(), h_sink.size() * 4, hipMemcpyHostToDevice, st));
    POA_CHECK(hipMemcpyAsync(d_q.p, h_q.data(), h_q.size(), hipMemcpyHostToDevice, st));

    tr.mark("reserve + H2D enqueue");
    // ---- pool: persistent in the ctx, sized for this batch's estimated need (capped by free HBM) and only
    // ever grown.  Problems run in sub-batches that fit it.
    auto est_bytes = [&](uint64_t p) -> uint64_t {
        const poa_graph_host &g = G[p];
        uint64_t w = params->wb < 0 ? g.qlen : (uint64_t)params->wb + (uint64_t)(params->wf * (double)g.qlen);
        int64_t excess = (int64_t)g.remain[0] - (int64_t)g.qlen;
        if (excess < 0) excess = -excess;
        uint64_t width = std::min<uint64_t>((uint64_t)g.qlen + 1, 2 * w + 1 + (uint64_t)excess + 64);
        return (uint64_t)((double)g.N * (double)width * 2.6) + 4 * POA_CHUNK;
    };
    {
        uint64_t want = 0, biggest = 0;
        for (uint64_t p = 0; p < n; p++) { uint64_t e = est_bytes(p); want += e; biggest = std::max(biggest, e); }
        want = (uint64_t)((double)want * 1.15) + 64 * POA_CHUNK;
        const char *env_pool = getenv("VGA_POOL_BYTES");
        if (env_pool) want = std::min<uint64_t>(want, strtoull(env_pool, nullptr, 10));
        if (W_.pool_size < want) {
            size_t free_b = 0, total_b = 0;
            POA_CHECK(hipMemGetInfo(&free_b, &total_b));
            uint64_t avail = (uint64_t)((double)(free_b + W_.pool_size) * 0.85);
            uint64_t target = std::min(want, avail) & ~(POA_CHUNK - 1);
            if (target > W_.pool_size) {
                if (W_.pool) { (void)hipFree(W_.pool); W_.pool = nullptr; W_.pool_size = 0; }
                if (target < 64 * POA_CHUNK) {
                    vga_poa_result_free(res);
                    return vga_set_error(ctx, VGA_ERR_NOMEM, "vga_poa_batch: only %llu bytes of HBM free for the traceback pool",
                                         (unsigned long long)free_b);
                }
                POA_CHECK(hipMalloc((void **)&W_.pool, target));
                W_.pool_size = target;
            }
        }
    }
    uint8_t *d_pool = W_.pool;
    const uint64_t pool_size = W_.pool_size;
    tr.mark("pool hipMalloc");
    poa_dev_params P;
    P.match = params->match; P.mismatch = params->mismatch; P.o1 = params->gap_open1; P.e1 = params->gap_ext1;
    P.o2 = params->gap_open2; P.e2 = params->gap_ext2; P.banded = params->wb >= 0;

    std::vector<int32_t> h_status(n), h_score(n);
    std::vector<uint32_t> h_row(n), h_nops(n);
    std::vector<uint64_t> h_cells(n), h_vcells(n);
    std::vector<uint8_t> h_ops(tot_ops);
    std::vector<uint32_t> h_orow(tot_ops);

    int t_total = vga_timer_begin(ctx, "poa_total", 0);
    uint64_t p0 = 0;
    double shrink = 1.0;
    int rc_final = VGA_OK;
    while (p0 < n) {
        uint64_t budget = (uint64_t)((double)pool_size * 0.9 * shrink), used = 0, p1 = p0;
        while (p1 < n) {
            uint64_t e = est_bytes(p1);
            if (p1 > p0 && used + e > budget) break;
            used += e;
            p1++;
        }
        const uint32_t nb = (uint32_t)(p1 - p0);
        POA_CHECK(hipMemsetAsync(d_next.p, 0, sizeof(unsigned long long), st));
        int t_dp = vga_timer_begin(ctx, "poa_band_dp", 0);
        POA_CHECK(hipMemcpyAsync(d_probs.p + p0, probs.data() + p0, nb * sizeof(poa_prob), hipMemcpyHostToDevice, st));
        {
            uint32_t max_q = 0;
            for (uint64_t p = p0; p < p1; p++) max_q = std::max(max_q, G[p].qlen);
            const uint32_t lds_cols = max_q + 1;
            const char *force = getenv("VGA_POA_KERNEL");  // "gmem" | "lds256" | "lds512" | "lds1024" (testing)
            int nt = max_q >= 1536 ? 512 : 256;
            bool use_lds = poa_lds_bytes(lds_cols, nt) <= 160 * 1024 - 256;
            if (force) {
                if (!strcmp(force, "gmem")) use_lds = false;
                else if (!strcmp(force, "lds256")) nt = 256;
                else if (!strcmp(force, "lds512")) nt = 512;
                else if (!strcmp(force, "lds1024")) nt = 1024;
                if (strcmp(force, "gmem") && poa_lds_bytes(lds_cols, nt) > 160 * 1024 - 256) use_lds = false;
            }
#define POA_ARGS d_probs.p + p0, d_q.p, d_meta.p, d_preds.p, d_sink.p, P, d_beg.p, \
                 d_end.p, d_doff.p, d_voff.p, d_lmax.p, d_rmax.p, d_pool, d_next.p, pool_size, d_score.p + p0, d_row.p + p0,        \
                 d_status.p + p0, d_cells.p + p0, d_vcells.p + p0
            if (!use_lds) {
                hipLaunchKernelGGL(k_poa_dp<256>, dim3(nb), dim3(256), 0, st, POA_ARGS);
            } else {
                const size_t lds = poa_lds_bytes(lds_cols, nt);
                if (nt == 256) {
                    POA_CHECK(hipFuncSetAttribute((const void *)k_poa_dp_lds<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(k_poa_dp_lds<256>, dim3(nb), dim3(256), lds, st, POA_ARGS, lds_cols);
                } else if (nt == 512) {
                    POA_CHECK(hipFuncSetAttribute((const void *)k_poa_dp_lds<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(k_poa_dp_lds<512>, dim3(nb), dim3(512), lds, st, POA_ARGS, lds_cols);
                } else {
                    POA_CHECK(hipFuncSetAttribute((const void *)k_poa_dp_lds<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(k_poa_dp_lds<1024>, dim3(nb), dim3(1024), lds, st, POA_ARGS, lds_cols);
                }
            }
#undef POA_ARGS
            POA_CHECK(hipGetLastError());
        }
        vga_timer_end(ctx, t_dp);
        int t_tb = vga_timer_begin(ctx, "poa_traceback", 0);
        hipLaunchKernelGGL(k_poa_traceback, dim3((nb + 63) / 64), dim3(64), 0, st, nb, d_probs.p + p0, d_meta.p, d_preds.p, d_beg.p, d_end.p, d_doff.p, d_pool, d_row.p + p0, d_status.p + p0, d_ops.p,
                           d_orow.p, d_nops.p + p0);
        vga_timer_end(ctx, t_tb);
        POA_CHECK(hipMemcpyAsync(h_status.data() + p0, d_status.p + p0, nb * 4, hipMemcpyDeviceToHost, st));
        POA_CHECK(hipStreamSynchronize(st));
        bool pool_fail = false;
        for (uint64_t p = p0; p < p1; p++)
            if (h_status[p] == POA_ST_POOL) pool_fail = true;
        if (pool_fail) {
            if (nb == 1) { rc_final = VGA_ERR_POOL; break; }
            shrink *= 0.5;
            continue;  // rerun this sub-batch with fewer problems
        }
        p0 = p1;
    }
    tr.mark("dp + traceback (sub-batches)");
    vga_timer_end(ctx, t_total);
    if (rc_final != VGA_OK) {
        vga_poa_result_free(res);
        return vga_set_error(ctx, rc_final, "vga_poa_batch: a single problem does not fit the %llu byte traceback pool",
                             (unsigned long long)pool_size);
    }
    POA_CHECK(hipMemcpyAsync(h_score.data(), d_score.p, n * 4, hipMemcpyDeviceToHost, st));
    POA_CHECK(hipMemcpyAsync(h_row.data(), d_row.p, n * 4, hipMemcpyDeviceToHost, st));
    POA_CHECK(hipMemcpyAsync(h_nops.data(), d_nops.p, n * 4, hipMemcpyDeviceToHost, st));
    POA_CHECK(hipMemcpyAsync(h_cells.data(), d_cells.p, n * 8, hipMemcpyDeviceToHost, st));
    POA_CHECK(hipMemcpyAsync(h_vcells.data(), d_vcells.p, n * 8, hipMemcpyDeviceToHost, st));
    POA_CHECK(hipMemcpyAsync(h_ops.data(), d_ops.p, tot_ops, hipMemcpyDeviceToHost, st));
    POA_CHECK(hipMemcpyAsync(h_orow.data(), d_orow.p, tot_ops * 4, hipMemcpyDeviceToHost, st));
    POA_CHECK(hipStreamSynchronize(st));
    tr.mark("D2H ops");
    vga_timers_collect(ctx);

    // ---- host: CIGAR / cs / node path from the raw op stream (reverse order on the device)
    std::vector<std::string> cig(n), css(n);
    std::vector<std::vector<uint32_t>> prow(n);
    {
        unsigned nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 4;
        if (nt > 32) nt = 32;
        if ((uint64_t)nt > n) nt = (unsigned)n;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                for (uint64_t p = t; p < n; p += nt) {
                    res->ok[p] = h_status[p] == POA_ST_OK ? 1 : 0;
                    res->best_score[p] = h_score[p];
                    res->n_cells[p] = h_cells[p];
                    res->n_value_cells[p] = h_vcells[p];
                    res->aln_start_offset[p] = res->aln_end_offset[p] = res->n_aligned_bases[p] = 0;
                    if (!res->ok[p]) continue;
                    const poa_graph_host &g = G[p];
                    const uint8_t *po = &h_ops[probs[p].ops0];
                    const uint32_t *pr = &h_orow[probs[p].ops0];
                    const char *q = queries_concat + query_off[p];
                    const uint32_t nops = h_nops[p];
                    std::string &cg = cig[p], &cs = css[p];
                    cs = "cs:Z:";
                    uint64_t eq_run = 0, aligned = 0;
                    uint32_t qi = 0;  // next query base
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
                                const char gb = (char)g.base[pr[idx]], qb = q[qi++];
                                aligned++;
                                if (gb == qb) eq_run++;
                                else {
                                    if (eq_run) { cs.push_back(':'); append_u(cs, eq_run); eq_run = 0; }
                                    cs.push_back('*'); cs.push_back(lower(gb)); cs.push_back(lower(qb));
                                }
                                prow[p].push_back(pr[idx]);
                            } else if (op == 1) {
                                cs.push_back(lower(q[qi++]));
                            } else {
                                cs.push_back(lower((char)g.base[pr[idx]]));
                                prow[p].push_back(pr[idx]);
                            }
                        }
                        t2 = u;
                    }
                    if (eq_run) { cs.push_back(':'); append_u(cs, eq_run); }
                    res->n_aligned_bases[p] = (uint32_t)aligned;
                    if (!prow[p].empty()) {
                        uint32_t fr = prow[p].front(), lr = prow[p].back();
                        res->aln_start_offset[p] = fr - g.first_row[g.row_node[fr]];
                        res->aln_end_offset[p] = lr - g.first_row[g.row_node[lr]] + 1;
                    }
                }
            });
        for (auto &x : th) x.join();
    }
    tr.mark("cigar/cs (host threads)");
    uint64_t tp = 0, tc = 0, ts = 0;
    for (uint64_t p = 0; p < n; p++) {
        res->path_off[p] = tp; res->cigar_off[p] = tc; res->cs_off[p] = ts;
        tp += prow[p].size(); tc += cig[p].size() + 1; ts += css[p].size() + 1;
    }
    res->path_off[n] = tp; res->cigar_off[n] = tc; res->cs_off[n] = ts;
    res->abpoa_nodes = pmalloc<uint32_t>(tp);
    res->graph_nodes = pmalloc<uint32_t>(tp);
    res->cigar = pmalloc<char>(tc);
    res->cs = pmalloc<char>(ts);
    for (uint64_t p = 0; p < n; p++) {
        for (size_t t = 0; t < prow[p].size(); t++) {
            res->abpoa_nodes[res->path_off[p] + t] = prow[p][t];
            res->graph_nodes[res->path_off[p] + t] = G[p].row_node[prow[p][t]];
        }
        memcpy(res->cigar + res->cigar_off[p], cig[p].c_str(), cig[p].size() + 1);
        memcpy(res->cs + res->cs_off[p], css[p].c_str(), css[p].size() + 1);
    }
    // byte model for the DP kernel: N + L + C direction bytes + 12 B per node-end cell ...
    res->ms_dp = vga_timer_sum(ctx, "poa_band_dp");
    res->ms_traceback = vga_timer_sum(ctx, "poa_traceback");
    res->ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    // byte model of the DP kernel: graph bases + query + 1 direction byte per cell
    // + the 6-byte value rows of node-end bases, written once and read back at least once
    uint64_t all_cells = 0, all_vcells = 0, all_rows = 0, all_q = 0, all_ops = 0;
    for (uint64_t p = 0; p < n; p++) {
        all_cells += h_cells[p]; all_vcells += h_vcells[p]; all_rows += G[p].N; all_q += G[p].qlen; all_ops += h_nops[p];
    }
    for (auto &a : ctx->last_times) {
        if (a.name == "poa_band_dp") a.bytes = all_rows + all_q + all_cells + 12 * all_vcells;
        if (a.name == "poa_traceback") a.bytes = 6 * all_ops;
    }
#undef POA_CHECK
    tr.mark("pack results");
    *out = res;
    return VGA_OK;
}
