// vga_subgraph.hip -- chain -> subgraph -> POA node table on the GPU, one wave per chain.
//
// Stands in for (and is checked against the host walk of vga_align.hip, VGA_SUBGRAPH=host, and the oracle's og_align.c):
//   find_range_chain ............ src/align.rs:267-402   handles of the extreme anchor positions, every node id in between
//   extend_range_chain_2 ........ src/align.rs:523-665   walk up- / downstream of the range while the unaligned part of the
//                                                        read still has bases left
//   find_nodes_edges_for_abpoa .. src/align.rs:670-724   sorted handles, their sequences, edges i -> j with i < j
// and for the node-level part of the POA graph that poa_prepare (vga_poa.hip) builds for host graphs: first rows,
// predecessor rows, sink rows, `remain` (longest path to the sink) of every node.
//
// The work is pointer chasing over the index's CSR arrays (src/index.rs:388-606), a few hundred nodes per chain: latency
// bound and embarrassingly parallel over the chains, so a chain gets one wave and the waves of a launch cover the chains in
// a strided loop.  k_sg_mark builds the handle set as a bitmap (a bit per packed handle: sorted order and deduplication,
// align.rs:658-659, come for free) and counts what the set will need; the host turns the counts into offsets; k_sg_emit
// writes handles, first rows, bases, predecessor lists, sinks and the node table.
//
// The reference's extension keeps no visited set (its frontier revisits handles along every path); a handle ends up in the
// range iff it can be reached with a positive remaining budget, which is what the per-handle best-budget relaxation
// computes (level-synchronous over the wave, atomicMax per handle).
//
// Data written and read back inside a kernel by different lanes goes through sg_ld (agent-scope loads) after a barrier:
// the vector L1 is not coherent with stores of other lanes' earlier instructions that went to L2.
#include "vga_subgraph.hpp"

#include <algorithm>
#include <vector>

namespace {

#define SG_RING_SPAN 32u  // = POA_RING_SPAN (vga_poa.hip)

struct sg_index {
    const uint32_t *node_start, *edge_idx, *edges_to, *edges;
    const char *seq;
    uint32_t n_nodes;
};

__device__ __forceinline__ uint32_t sg_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// get_bv_rank (src/index.rs:427-439): node starts <= pos
__device__ inline uint32_t sg_rank(const sg_index &ix, uint32_t pos)
{
    uint32_t lo = 0, hi = ix.n_nodes + 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ix.node_start[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// get_bv_select (src/index.rs:461-480)
__device__ inline uint32_t sg_select(const sg_index &ix, uint32_t element_no)
{
    if (element_no == 0 || element_no > ix.n_nodes + 1) return 0;
    return ix.node_start[element_no - 1];
}
__device__ inline uint32_t sg_len(const sg_index &ix, uint32_t h)
{
    const uint32_t id = h >> 1;
    return ix.node_start[id] - ix.node_start[id - 1];
}
// incoming_edges_from_handle / outgoing_edges_from_handle (src/index.rs:559-606); a reverse handle takes the other list of
// its forward twin, flipped (the order inside a list never matters here)
template <typename F>
__device__ inline void sg_for_nb(const sg_index &ix, uint32_t h, bool incoming, F f)
{
    const uint32_t pos = (h >> 1) - 1;
    const bool inc = (h & 1u) ? !incoming : incoming;
    const uint32_t s0 = ix.edge_idx[pos], nin = ix.edges_to[pos];
    const uint32_t s = inc ? s0 : s0 + nin, e = inc ? s0 + nin : ix.edge_idx[pos + 1];
    const uint32_t fl = h & 1u;
    for (uint32_t i = s; i < e; i++) f(ix.edges[i] ^ fl);
}

__device__ inline bool sg_member(const uint32_t *bm, uint32_t nh, uint32_t x) { return x < nh && ((sg_ld(bm + (x >> 5)) >> (x & 31u)) & 1u); }

__device__ inline uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return v;
}
__device__ inline uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, o); v = t > v ? t : v; }
    return v;
}
__device__ inline uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, o); v = t < v ? t : v; }
    return v;
}
__device__ inline unsigned long long wave_sum64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += ((unsigned long long)(uint32_t)__shfl_xor((int)(v >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, o);
    return v;
}
// exclusive prefix sum over the wave; *total gets the sum
__device__ inline uint32_t wave_excl(uint32_t v, int lane, uint32_t *total)
{
    uint32_t s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)s, o);
        if (lane >= o) s += t;
    }
    *total = (uint32_t)__shfl((int)s, 63);
    return s - v;
}

// per handle of the set: what it contributes to the counts.  deg = in-neighbours inside the set that sort before it
// (RangeOrient::Forward keeps an edge i -> j only when i < j, align.rs:718), has_out = it has a successor in the set
struct sg_node_counts {
    uint32_t len, deg;
    bool has_out;
};
__device__ inline sg_node_counts sg_count_node(const sg_index &ix, const uint32_t *bm, uint32_t nh, uint32_t h)
{
    sg_node_counts c;
    c.len = sg_len(ix, h);
    c.deg = 0;
    c.has_out = false;
    sg_for_nb(ix, h, true, [&](uint32_t x) { if (x < h && sg_member(bm, nh, x)) c.deg++; });
    sg_for_nb(ix, h, false, [&](uint32_t y) { if (y > h && sg_member(bm, nh, y)) c.has_out = true; });
    return c;
}

// one direction of align.rs:551-591 / 616-656 for the whole wave
__device__ void sg_extend(const sg_index &ix, uint32_t from, uint32_t diff, bool incoming, uint32_t *best, uint32_t *stamp,
                          uint32_t *q0, uint32_t *q1, uint32_t *touched, uint32_t &level, uint32_t *s_cnt, uint32_t *s_touched, int lane)
{
    auto relax = [&](uint32_t x, uint32_t rem, uint32_t lvl, uint32_t *q, uint32_t *qn) {
        const uint32_t old = atomicMax(best + x, rem);
        if (old < rem) {
            if (old == 0) touched[atomicAdd(s_touched, 1u)] = x;
            if (atomicExch(stamp + x, lvl) != lvl) q[atomicAdd(qn, 1u)] = x;
        }
    };
    if (lane == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    level++;
    if (lane == 0) sg_for_nb(ix, from, incoming, [&](uint32_t x) { relax(x, diff, level, q0, s_cnt); });
    int cq = 0;
    for (;;) {
        __syncthreads();
        const uint32_t cnt = s_cnt[cq];
        if (cnt == 0) break;
        __syncthreads();
        if (lane == 0) s_cnt[cq ^ 1] = 0;
        __syncthreads();
        level++;
        uint32_t *q = cq ? q1 : q0, *qo = cq ? q0 : q1;
        for (uint32_t i = (uint32_t)lane; i < cnt; i += 64) {
            const uint32_t h = sg_ld(q + i);
            const uint32_t left = sg_ld(best + h);
            const uint32_t len = sg_len(ix, h);
            if (len < left) {
                const uint32_t rem = left - len;
                sg_for_nb(ix, h, incoming, [&](uint32_t x) { relax(x, rem, level, qo, s_cnt + (cq ^ 1)); });
            }
        }
        cq ^= 1;
    }
}

// ---- kernel 1: the handle set of every chain as a bitmap, and its counts
__global__ __launch_bounds__(64) void k_sg_mark(const sg_desc *__restrict__ descs, uint32_t n, sg_index ix, uint32_t k, uint32_t words,
                                                uint32_t nh, uint32_t *bitmaps, uint32_t *scratch, sg_sum *sums)
{
    __shared__ uint32_t s_cnt[2], s_touched;
    const int lane = threadIdx.x;
    uint32_t *best = scratch + (uint64_t)blockIdx.x * 5ull * nh, *stamp = best + nh, *q0 = stamp + nh, *q1 = q0 + nh, *touched = q1 + nh;
    for (uint32_t i = (uint32_t)lane; i < nh; i += 64) { best[i] = 0; stamp[i] = 0; }
    uint32_t level = 0;
    __syncthreads();
    for (uint32_t p = blockIdx.x; p < n; p += gridDim.x) {
        const sg_desc d = descs[p];
        uint32_t *bm = bitmaps + (uint64_t)p * words;
        // find_range_chain: the position -> handle map is monotonic, so the extreme positions give the extreme handles
        const uint32_t min_id = sg_rank(ix, d.pmin), max_id = sg_rank(ix, d.pmax);
        const uint32_t first_handle = min_id * 2, last_handle = max_id * 2;
        for (uint32_t w = (uint32_t)lane; w < words; w += 64) {
            // every forward handle first_handle .. last_handle (align.rs:358-364)
            const uint32_t lo = w * 32u, hi = lo + 31u;
            uint32_t m = 0;
            if (hi >= first_handle && lo <= last_handle) {
                m = 0x55555555u;
                if (first_handle > lo) m &= ~0u << (first_handle - lo);
                if (last_handle < hi) m &= ~0u >> (hi - last_handle);
            }
            bm[w] = m;
        }
        if (lane == 0) s_touched = 0;
        __syncthreads();
        // align.rs:536-547
        uint32_t prefix_diff = d.q_first;
        const uint32_t start_prefix_on_node = d.t_first - sg_select(ix, first_handle >> 1);
        if (start_prefix_on_node < prefix_diff) prefix_diff -= start_prefix_on_node; else prefix_diff = 0;
        uint32_t wlo = first_handle >> 5, whi = last_handle >> 5;
        // the handles a walk reached go into the set, and their budgets are cleared before the other walk starts (the two
        // walks are independent in the reference: a handle the upstream walk reached with a large budget must still be
        // expanded by the downstream walk -- they can only meet on cyclic or strand-mixing graphs)
        auto harvest = [&]() {
            __syncthreads();
            const uint32_t nt = s_touched;
            for (uint32_t i = (uint32_t)lane; i < nt; i += 64) {
                const uint32_t h = sg_ld(touched + i);
                atomicOr(bm + (h >> 5), 1u << (h & 31u));
                best[h] = 0;
                wlo = min(wlo, h >> 5);
                whi = max(whi, h >> 5);
            }
            __syncthreads();
            if (lane == 0) s_touched = 0;
            __syncthreads();
        };
        if (prefix_diff > 0) {
            sg_extend(ix, first_handle, prefix_diff, true, best, stamp, q0, q1, touched, level, s_cnt, &s_touched, lane);
            harvest();
        }
        // align.rs:593-612
        uint32_t suffix_diff = d.qlen - (d.q_last + k);
        const uint32_t end_suffix_on_node = sg_select(ix, (last_handle >> 1) + 1) - 1 - (d.te_last - 1);
        if (end_suffix_on_node > suffix_diff) suffix_diff = 0; else suffix_diff -= end_suffix_on_node;
        if (suffix_diff > 0) {
            sg_extend(ix, last_handle, suffix_diff, false, best, stamp, q0, q1, touched, level, s_cnt, &s_touched, lane);
            harvest();
        }
        wlo = wave_min(wlo);
        whi = wave_max(whi);
        __syncthreads();
        // counts: nodes, rows, predecessor entries (a node without one gets the virtual source), sink predecessors
        uint32_t c_nodes = 0, c_preds = 0, c_sinks = 0, bad = 0;
        unsigned long long c_rows = 0;
        for (uint32_t w = wlo + (uint32_t)lane; w <= whi; w += 64) {
            uint32_t bits = sg_ld(bm + w);
            while (bits) {
                const uint32_t h = w * 32u + (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1;
                const sg_node_counts c = sg_count_node(ix, bm, nh, h);
                c_nodes++;
                c_rows += c.len;
                c_preds += c.deg ? c.deg : 1u;
                c_sinks += c.has_out ? 0u : 1u;
                if (c.deg > 255 || c.len == 0 || c.len >= (1u << 24)) bad = 1;
            }
        }
        c_nodes = wave_sum(c_nodes);
        c_preds = wave_sum(c_preds);
        c_sinks = wave_sum(c_sinks);
        bad = wave_max(bad);
        const unsigned long long rows = wave_sum64(c_rows);
        if (rows >= (1ull << 31)) bad = 1;
        if (lane == 0) {
            sg_sum s = {};
            s.n_nodes = c_nodes; s.N = (uint32_t)rows; s.n_preds = c_preds; s.n_sinks = c_sinks; s.wlo = wlo; s.whi = whi; s.flags = bad;
            sums[p] = s;
        }
        __syncthreads();
    }
}

__device__ inline char sg_complement(char c)
{
    // dna.rs:19-33
    switch (c) {
    case 'a': return 't'; case 'c': return 'g'; case 't': return 'a'; case 'g': return 'c'; case 'u': return 'a';
    case 'A': return 'T'; case 'C': return 'G'; case 'T': return 'A'; case 'G': return 'C'; case 'U': return 'A';
    default: return 'N';
    }
}

// ---- kernel 2: handles, first rows, bases, predecessor rows, sinks, node table
__global__ __launch_bounds__(64) void k_sg_emit(uint32_t n, sg_index ix, uint32_t words, uint32_t nh, const uint32_t *bitmaps, uint32_t *scratch,
                                                sg_sum *sums, const sg_off *__restrict__ offs, uint32_t *handles, uint32_t *first_row, uint4 *ntab,
                                                uint32_t *preds, uint32_t *sinks, char *seq, int first_edge)
{
    const int lane = threadIdx.x;
    uint32_t *wb_rank = scratch + (uint64_t)blockIdx.x * 5ull * nh, *wb_row = wb_rank + words, *wb_pred = wb_row + words, *wb_sink = wb_pred + words;
    uint32_t *rf = scratch + (uint64_t)blockIdx.x * 5ull * nh + nh;  // remain of the first base of a node, by rank (4 * words <= nh)
    for (uint32_t p = blockIdx.x; p < n; p += gridDim.x) {
        const sg_sum sm = sums[p];
        if (sm.flags & 1u) continue;
        const sg_off of = offs[p];
        const uint32_t *bm = bitmaps + (uint64_t)p * words;
        const uint32_t wlo = sm.wlo, whi = sm.whi, n_sub = sm.n_nodes;
        uint32_t *hd = handles + of.node0, *fr = first_row + of.node0, *pl = preds + of.pred0, *sl = sinks + of.sink0;
        uint4 *nt = ntab + of.node0 + p;  // entry 0 = the virtual source
        char *sq = seq + of.seq0;
        // -- A: where every bitmap word starts (rank, row, predecessor entry, sink entry)
        {
            uint32_t run_rank = 0, run_row = 0, run_pred = 0, run_sink = 0;
            for (uint32_t w0 = wlo; w0 <= whi; w0 += 64) {
                const uint32_t w = w0 + (uint32_t)lane;
                uint32_t bits = w <= whi ? bm[w] : 0u;
                uint32_t cn = 0, cr = 0, cp = 0, cs = 0;
                while (bits) {
                    const uint32_t h = w * 32u + (uint32_t)__builtin_ctz(bits);
                    bits &= bits - 1;
                    const sg_node_counts c = sg_count_node(ix, bm, nh, h);
                    cn++; cr += c.len; cp += c.deg ? c.deg : 1u; cs += c.has_out ? 0u : 1u;
                }
                uint32_t tn, tr, tp, ts;
                const uint32_t en = wave_excl(cn, lane, &tn), er = wave_excl(cr, lane, &tr), ep = wave_excl(cp, lane, &tp), es = wave_excl(cs, lane, &ts);
                if (w <= whi) {
                    wb_rank[w - wlo] = run_rank + en; wb_row[w - wlo] = run_row + er; wb_pred[w - wlo] = run_pred + ep; wb_sink[w - wlo] = run_sink + es;
                }
                run_rank += tn; run_row += tr; run_pred += tp; run_sink += ts;
            }
        }
        __syncthreads();
        auto rank_of = [&](uint32_t x) -> uint32_t {  // position of a member handle in the sorted list
            const uint32_t w = x >> 5;
            return sg_ld(wb_rank + (w - wlo)) + (uint32_t)__builtin_popcount(bm[w] & ((1u << (x & 31u)) - 1u));
        };
        // -- B: handles, first rows, bases (seq_from_handle, src/index.rs:503-533)
        for (uint32_t w = wlo + (uint32_t)lane; w <= whi; w += 64) {
            uint32_t bits = bm[w];
            uint32_t i = sg_ld(wb_rank + (w - wlo)), row = sg_ld(wb_row + (w - wlo));
            while (bits) {
                const uint32_t h = w * 32u + (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1;
                const uint32_t id = h >> 1, s = ix.node_start[id - 1], e = ix.node_start[id];
                hd[i] = h;
                fr[i] = row + 1;
                if (!(h & 1u)) for (uint32_t t = s; t < e; t++) sq[row + (t - s)] = ix.seq[t];
                else for (uint32_t t = e; t-- > s;) sq[row + (e - 1 - t)] = sg_complement(ix.seq[t]);
                row += e - s;
                i++;
            }
        }
        __syncthreads();
        // -- C: predecessor rows in edge-list order (sources ascending), sinks, node table without `remain`
        uint32_t life = 1;
        for (uint32_t w = wlo + (uint32_t)lane; w <= whi; w += 64) {
            uint32_t bits = bm[w];
            uint32_t i = sg_ld(wb_rank + (w - wlo)), pcur = sg_ld(wb_pred + (w - wlo)), scur = sg_ld(wb_sink + (w - wlo));
            while (bits) {
                const uint32_t h = w * 32u + (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1;
                const uint32_t len = sg_len(ix, h), frow = sg_ld(fr + i);
                // in-neighbours by (handle, position in the list), smallest first
                uint32_t deg = 0, last_v = 0, last_k = 0, first_pred = 0;
                bool have_last = false;
                for (;;) {
                    uint32_t best_v = 0xFFFFFFFFu, best_k = 0xFFFFFFFFu, kk = 0;
                    sg_for_nb(ix, h, true, [&](uint32_t x) {
                        const uint32_t kx = kk++;
                        if (!(x < h && sg_member(bm, nh, x))) return;
                        if (have_last && (x < last_v || (x == last_v && kx <= last_k))) return;
                        if (x < best_v || (x == best_v && kx < best_k)) { best_v = x; best_k = kx; }
                    });
                    if (best_k == 0xFFFFFFFFu) break;
                    const uint32_t r = rank_of(best_v);
                    const uint32_t lrow = sg_ld(fr + r) + sg_len(ix, best_v) - 1;
                    pl[pcur + deg] = lrow;
                    if (deg == 0) first_pred = lrow;
                    deg++;
                    last_v = best_v; last_k = best_k; have_last = true;
                }
                if (deg == 0) pl[pcur] = 0;
                uint32_t reach = 0;
                bool has_out = false;
                sg_for_nb(ix, h, false, [&](uint32_t y) {
                    if (y > h && sg_member(bm, nh, y)) { has_out = true; const uint32_t sp = rank_of(y) - i; reach = sp > reach ? sp : reach; }
                });
                if (!has_out) sl[scur++] = frow + len - 1;
                if (reach <= SG_RING_SPAN) life = reach > life ? reach : life;
                nt[1 + i] = make_uint4(frow, len | ((deg ? deg : 1u) << 24), (has_out ? 0u : 0x80000000u) | (reach > SG_RING_SPAN ? 0x40000000u : 0u),
                                       deg <= 1 ? first_pred : pcur);
                pcur += deg ? deg : 1u;
                i++;
            }
        }
        life = wave_max(life);
        __syncthreads();
        // -- D: remain of the last base of every node, last node first: the longest path over its successors, or
        // (first_edge, VGA_REMAIN_FIRST_OUT_EDGE) the path through its first successor in edge-list order.  64 nodes at a
        // time: successors above the group are final; inside the group a lane only depends on lower lanes, so 64 broadcast
        // steps settle it
        uint32_t longest = 0, first_src = 0xFFFFFFFFu;
        for (uint32_t g0 = 0; g0 < n_sub; g0 += 64) {
            const uint32_t top = n_sub - 1 - g0;
            const bool valid = (uint32_t)lane <= top;
            const uint32_t i = valid ? top - (uint32_t)lane : 0u;
            uint32_t val = 0, len = 1;
            unsigned long long mask = 0;
            if (valid) {
                const uint32_t h = sg_ld(hd + i);
                len = sg_len(ix, h);
                bool taken = false;
                sg_for_nb(ix, h, false, [&](uint32_t y) {
                    if (!(y > h && sg_member(bm, nh, y))) return;
                    if (first_edge && taken) return;
                    taken = true;
                    const uint32_t j = rank_of(y);
                    if (j > top) { const uint32_t c = 1u + sg_ld(rf + j); val = c > val ? c : val; }
                    else mask |= 1ull << (top - j);
                });
            }
            for (int s = 0; s < 64; s++) {
                const uint32_t rfs = (uint32_t)__shfl((int)(val + len - 1), s);
                if ((mask >> s) & 1ull) { const uint32_t c = 1u + rfs; val = c > val ? c : val; }
            }
            if (valid) {
                rf[i] = val + len - 1;
                uint32_t *z = (uint32_t *)(nt + 1 + i);
                const uint32_t y = sg_ld(z + 1), wv = sg_ld(z + 3), zf = sg_ld(z + 2);
                z[2] = zf | val;
                if ((y >> 24) == 1u && wv == 0u) {  // a node without predecessor
                    const uint32_t c = 1u + val + len - 1;
                    if (first_edge) { if (i < first_src) { first_src = i; longest = c; } }  // (the source's first out-edge: the first such node)
                    else longest = c > longest ? c : longest;
                }
            }
            __syncthreads();
        }
        if (first_edge) {
            const uint32_t fs = wave_min(first_src);
            longest = wave_max(first_src == fs && fs != 0xFFFFFFFFu ? longest : 0u);
        } else
            longest = wave_max(longest);
        if (lane == 0) {
            nt[0] = make_uint4(0u, 1u, longest, 0u);  // the virtual source: row 0, remain = longest path
            sums[p].longest = longest;
            sums[p].life = life;
            if (longest >= (1u << 30)) sums[p].flags = 1u;
        }
        __syncthreads();
    }
}

struct sg_part_bufs {
    vga_dbuf<uint32_t> d_handles, d_first_row, d_preds, d_sinks;
    vga_dbuf<uint4> d_ntab;
    vga_dbuf<char> d_seq;
    vga_hbuf<uint32_t> h_handles, h_first_row;
};
struct sg_ws {
    vga_dbuf<sg_desc> d_desc;
    vga_dbuf<sg_sum> d_sum;
    vga_dbuf<sg_off> d_off;
    vga_dbuf<uint32_t> d_bitmaps, d_scratch;
    vga_hbuf<sg_desc> h_desc;
    vga_hbuf<sg_sum> h_sum;
    vga_hbuf<sg_off> h_off;
    sg_part_bufs part[2];
    std::vector<uint64_t> q_src;
    hipStream_t side = nullptr;
    uint64_t waves = 0;
    uint32_t nh = 0, words = 0;
    ~sg_ws()
    {
        if (side) (void)hipStreamDestroy(side);
    }
};

#define SG_CHECK(call)                                                                                                              \
    do {                                                                                                                            \
        hipError_t e_ = (call);                                                                                                     \
        if (e_ != hipSuccess)                                                                                                       \
            return vga_set_error(ctx, e_ == hipErrorOutOfMemory ? VGA_ERR_NOMEM : VGA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                 __FILE__, __LINE__);                                                                               \
    } while (0)

// mark + counts, offsets, emit and the copies back for problems [p0, p1) on stream st; waits for them
int sg_run_part(vga_ctx *ctx, sg_ws &W, sg_store &store, int k, uint64_t p0, uint64_t p1, hipStream_t st)
{
    sg_part &P = store.part[k];
    P.p0 = p0; P.p1 = p1;
    if (p1 <= p0) { P.ready = true; return VGA_OK; }
    const uint64_t n = p1 - p0;
    const vga_dev_index &dix = ctx->index;
    sg_index ix = {dix.d_node_start, dix.d_edge_idx, dix.d_edges_to, dix.d_edges, dix.d_seq_fwd, (uint32_t)dix.n_nodes};
    const uint32_t nh = W.nh, words = W.words;
    // the part that runs beside a DP launch keeps to two waves per CU: its waves are long-lived and every one of them
    // takes registers a DP workgroup (four waves at once) is waiting for -- 4 096 of them halve the DP's occupancy
    uint32_t waves = (uint32_t)std::min<uint64_t>(W.waves, n);
    if (k == 1) {
        uint32_t side_waves = 2u * (uint32_t)ctx->n_cu;  // (512: +3.7 % over one part on config 3; 256 starves the later launches, 1 024 and more slow the first)
        if (const char *e = getenv("VGA_SG_SIDE_WAVES")) side_waves = (uint32_t)std::max(1, atoi(e));
        waves = std::min(waves, side_waves);
    }
    sg_part_bufs &B = W.part[k];
    int t_mark = vga_timer_begin(ctx, "subgraph_mark", 0, st);
    hipLaunchKernelGGL(k_sg_mark, dim3(waves), dim3(64), 0, st, W.d_desc.p + p0, (uint32_t)n, ix, dix.k, words, nh, W.d_bitmaps.p + p0 * words,
                       W.d_scratch.p, W.d_sum.p + p0);
    vga_timer_end(ctx, t_mark);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipMemcpyAsync(W.h_sum.p + p0, W.d_sum.p + p0, n * sizeof(sg_sum), hipMemcpyDeviceToHost, st));
    SG_CHECK(hipStreamSynchronize(st));
    uint64_t tn = 0, tp = 0, ts = 0, tq = 0;
    for (uint64_t p = p0; p < p1; p++) {
        const sg_sum &s = W.h_sum.p[p];
        sg_off &o = W.h_off.p[p];
        o.node0 = tn; o.pred0 = tp; o.sink0 = ts; o.seq0 = tq; o.q_src = W.q_src[p];
        if (s.flags & 1u) continue;
        tn += s.n_nodes; tp += s.n_preds; ts += s.n_sinks; tq += ((uint64_t)s.N + 3) & ~3ull;
    }
    SG_CHECK(B.d_handles.reserve(tn + 1)); SG_CHECK(B.d_first_row.reserve(tn + 1)); SG_CHECK(B.d_ntab.reserve(tn + n));
    SG_CHECK(B.d_preds.reserve(tp + 1)); SG_CHECK(B.d_sinks.reserve(ts + 1)); SG_CHECK(B.d_seq.reserve(tq + 4));
    SG_CHECK(B.h_handles.reserve(tn + 1)); SG_CHECK(B.h_first_row.reserve(tn + 1));
    SG_CHECK(hipMemcpyAsync(W.d_off.p + p0, W.h_off.p + p0, n * sizeof(sg_off), hipMemcpyHostToDevice, st));
    int t_emit = vga_timer_begin(ctx, "subgraph_emit", 0, st);
    hipLaunchKernelGGL(k_sg_emit, dim3(waves), dim3(64), 0, st, (uint32_t)n, ix, words, nh, W.d_bitmaps.p + p0 * words, W.d_scratch.p, W.d_sum.p + p0,
                       W.d_off.p + p0, B.d_handles.p, B.d_first_row.p, B.d_ntab.p, B.d_preds.p, B.d_sinks.p, B.d_seq.p,
                       store.remain_rule == VGA_REMAIN_FIRST_OUT_EDGE ? 1 : 0);
    vga_timer_end(ctx, t_emit);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipMemcpyAsync(W.h_sum.p + p0, W.d_sum.p + p0, n * sizeof(sg_sum), hipMemcpyDeviceToHost, st));
    if (tn) {
        SG_CHECK(hipMemcpyAsync(B.h_handles.p, B.d_handles.p, tn * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        SG_CHECK(hipMemcpyAsync(B.h_first_row.p, B.d_first_row.p, tn * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    SG_CHECK(hipStreamSynchronize(st));
    P.d_ntab = B.d_ntab.p; P.d_preds = B.d_preds.p; P.d_sinks = B.d_sinks.p; P.d_seq = B.d_seq.p;
    P.h_handles = B.h_handles.p; P.h_first_row = B.h_first_row.p;
    P.ready = true;
    return VGA_OK;
}

}  // namespace

int sg_prepare(vga_ctx *ctx, const sg_desc *descs, const uint64_t *q_src, uint64_t n, uint64_t split, const char *d_reads, int remain_rule,
               sg_store &store)
{
    store = sg_store();
    store.remain_rule = remain_rule;
    if (n == 0) return VGA_OK;
    if (n >= (1ull << 31)) return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "too many chains in one call");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    if (!ctx->sg_ws) {
        ctx->sg_ws = new sg_ws();
        ctx->sg_ws_free = [](void *q) { delete (sg_ws *)q; };
    }
    sg_ws &W = *(sg_ws *)ctx->sg_ws;
    if (!W.side) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        (void)hi;
        SG_CHECK(hipStreamCreateWithPriority(&W.side, hipStreamNonBlocking, lo));  // (lowest priority: it must not displace DP workgroups)
    }
    const vga_dev_index &dix = ctx->index;
    W.nh = 2u * ((uint32_t)dix.n_nodes + 2u);
    W.words = (W.nh + 31u) / 32u;
    // waves in flight: enough to hide the latency of the pointer chasing, bounded by the scratch they need (5 words per handle)
    uint64_t waves = std::min<uint64_t>(n, 16ull * (uint64_t)ctx->n_cu);
    const uint64_t slab = 5ull * W.nh * sizeof(uint32_t);
    W.waves = std::max<uint64_t>(1, std::min<uint64_t>(waves, (4ull << 30) / slab));
    if (split > n) split = n;
    SG_CHECK(W.d_desc.reserve(n)); SG_CHECK(W.d_sum.reserve(n)); SG_CHECK(W.d_off.reserve(n));
    SG_CHECK(W.h_desc.reserve(n)); SG_CHECK(W.h_sum.reserve(n)); SG_CHECK(W.h_off.reserve(n));
    SG_CHECK(W.d_bitmaps.reserve(n * W.words));
    SG_CHECK(W.d_scratch.reserve(W.waves * 5ull * W.nh));
    memcpy(W.h_desc.p, descs, n * sizeof(sg_desc));
    W.q_src.assign(q_src, q_src + n);
    SG_CHECK(hipMemcpyAsync(W.d_desc.p, W.h_desc.p, n * sizeof(sg_desc), hipMemcpyHostToDevice, st));
    store.n = n;
    store.split = split;
    store.sum = W.h_sum.p; store.off = W.h_off.p; store.d_off = W.d_off.p;
    store.d_reads = d_reads;
    store.part[1].p0 = split; store.part[1].p1 = n;
    return sg_run_part(ctx, W, store, 0, 0, split, st);
}

int sg_prepare_rest(vga_ctx *ctx, sg_store &store)
{
    if (store.part[1].ready) return VGA_OK;
    (void)hipSetDevice(ctx->device);
    sg_ws &W = *(sg_ws *)ctx->sg_ws;
    return sg_run_part(ctx, W, store, 1, store.split, store.n, W.side);
}
