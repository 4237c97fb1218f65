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
#include <atomic>
#include <cstddef>
#include <type_traits>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "vga_poa_kernels.hpp"
#include "vga_poa_t4.hpp"
#include "vga_poa_t5.hpp"
#include "vga_poa_t6.hpp"
#include "vga_poa_t7.hpp"
#include "vga_poa_text.hpp"

// The dynamic LDS a kernel may ask for is an attribute of the function for the whole process (per device), and contexts on other
// host threads launch the same kernels with other sizes: the limit only ever goes up, under a lock, so that no launch meets a
// limit that another thread has just set lower than what it asks for
static hipError_t poa_allow_lds(int device, const void *fn, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> cur;
    std::lock_guard<std::mutex> g(mu);
    size_t &c = cur[std::make_pair(device, fn)];
    if (bytes <= c) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) c = bytes;
    return e;
}

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
// node (over the nodes, last first: the longest path, or with first_edge the path through the first out-edge in
// edge-list order -- vga_poa_params.remain_rule), sink predecessors.  Mirrors the row construction of oracle/og_poa.c.
void poa_prepare(const poa_view &v, poa_prep &g, bool first_edge)
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
        for (uint32_t t = out_off[i]; t < out_off[i + 1]; t++) {
            rl = std::max(rl, 1 + remain_first[out_adj[t]]);
            if (first_edge) break;
        }
        remain_last[i] = rl;
        remain_first[i] = rl + (int32_t)(last_row[i] - g.first_row[i]);
    }
    int32_t longest = 0;
    bool have_src = false;
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
            if (!(first_edge && have_src)) longest = std::max(longest, 1 + remain_first[i]);
            have_src = true;
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
    g.ntab[0] = make_uint4(0u, 1u, (uint32_t)longest, 0u);  // the virtual source: row 0, remain over the source nodes
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
    // k_poa_text: cs / CIGAR / node path of every problem as text in a compact arena (claimed through d_tcur), one record each
    vga_dbuf<char> d_text;
    vga_dbuf<poa_text_out> d_touts;
    vga_dbuf<unsigned long long> d_tcur;
    struct out_set {
        vga_hbuf<uint32_t> h_orow;
        vga_hbuf<uint8_t> h_ops;
        vga_hbuf<char> h_seq;     // device store: the bases of the sub-batch's problems (row r of a problem is byte seq0 + r - 1)
        vga_hbuf<poa_out> h_outs;
        vga_hbuf<char> h_text;
        char *text_p = nullptr;  // where this launch's text is: h_text.p, or a buffer of poa_ws::text_live (poa_feed::keep_text)
        vga_hbuf<poa_text_out> h_touts;
        vga_hbuf<unsigned long long> h_tcur;
        bool text = false;        // this sub-batch's strings were written on the device
        uint64_t tot_ops = 0, tot_seq = 0;
    } outs[2];
    uint32_t uses = 0;
};

// The traceback pool: SEGMENTS of HBM, allocated one after the other by a thread of its own (`grower`) so that the first launch
// does not wait for all of it -- on a GPU whose memory was used before, the driver clears what it hands out at ~40 GB/s, and
// rounds 1-2 spent 0.4-7 s in one 257 GB hipMalloc before the first kernel of a process (and 1.6 s in the hipFree at its end).
// k_poa_dp_t5 takes the segments' 1 MiB chunks through a device-side free list (vga_poa_kernels.hpp: poa_chunk_pool);
// classic launches (k_poa_dp_t4 / k_poa_dp_lds, problems the chunk mode hands back) bump-allocate inside whole segments --
// never at the same time as chunk-mode launches.
#define POA_SEG_LOG2 32  // 4 GiB segments (4 096 chunks)
#define POA_MAX_SEGS 80
struct poa_ws {
    poa_slot slot[POA_SLOTS];
    vga_dbuf<unsigned long long> d_next;
    vga_hbuf<unsigned long long> h_next;
    hipStream_t extra[POA_SLOTS] = {};  // streams of slots 1.. (slot 0 runs on the context's stream)
    double pool_scale = 1.35;  // measured pool bytes / estimated bytes, adapted after every sub-batch.  (Starts where config 3 ends up
                               // after a call: from 1.0 the second call of a process asked for a third more pool than the first --
                               // on memory the driver has to clear that is 0.4 s inside what bench.py times)
    // segments (guarded by mu)
    struct seg_t { uint8_t *p; uint64_t size; };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<seg_t> segs;
    uint64_t pool_size = 0;     // bytes in segs
    uint64_t grow_target = 0;   // the grower stops at this many bytes
    uint8_t *classic = nullptr; // the classic pool: one contiguous piece (launches that do not run in chunk-pool mode)
    uint64_t classic_size = 0;
    bool growing = false, grow_failed = false;
    std::thread grower;
    int device = 0;
    vga_ctx *owner = nullptr;  // (the grower holds back while owner->alloc_urgent: vga_common.hpp)
    uint64_t seg_bytes = 1ull << POA_SEG_LOG2;
    // chunk pool (device side)
    vga_dbuf<unsigned long long> d_head;  // the free-list heads (POA_LISTS of them, a cache line apart), then the statistics
    vga_dbuf<uint32_t> d_next_chunk, d_slot_flag, d_owner;
    vga_hbuf<uint32_t> h_short;           // poa_chunk_pool::short_flag
    vga_hbuf<uint64_t> h_seg_base;        // staging of ...
    vga_dbuf<uint64_t> d_seg_base;        // ... the segment table the kernels read (an entry is copied before its chunks are listed)
    hipStream_t add_stream = nullptr;
    uint32_t chunks_listed = 0;           // chunks of segments [0, segs_listed) are in the free list
    uint64_t polls = 0, empties_seen = 0; // how often the kernels found the free list empty (poa_chunk_pool::stats[1]), as last read
    size_t segs_listed = 0;
    uint8_t *state = nullptr;             // the state regions
    uint64_t state_bytes = 0;
    // poa_feed::keep_text: the text of every launch of a call in a pinned buffer of its own, alive until the next call
    std::vector<std::unique_ptr<vga_hbuf<char>>> text_live, text_free;
    hipError_t reset_lists()  // every free list empty, statistics zero
    {
        std::vector<unsigned long long> init(POA_LISTS * POA_LIST_STRIDE + 16, 0ull);
        for (int l = 0; l < POA_LISTS; l++) init[(size_t)l * POA_LIST_STRIDE] = (unsigned long long)POA_NIL;
        const hipError_t e = hipMemcpy(d_head.p, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice);
        return e != hipSuccess ? e : hipStreamSynchronize(nullptr);  // (the copy is on the device before anything is launched: see d_slot_flag)
    }
    // The small tables of the chunk pool (free-list heads, per-chunk links, state-region flags, the segment table and its pinned
    // staging, the kernels' shortage flag, the stream new segments are listed on).  Allocated BEFORE the grower is started: the
    // runtime serialises allocations, and a 16 KB hipMalloc or hipHostMalloc that queues behind the grower's 4 GiB segments
    // (0.1 s each while the driver clears them) held the first launch of a process back by 1.8-5.3 s
    bool tables_ready = false;
    hipError_t ensure_tables(uint32_t n_cu)
    {
        if (tables_ready) return hipSuccess;
        const uint32_t max_chunks = (uint32_t)(POA_MAX_SEGS * (1ull << (POA_SEG_LOG2 - 20)));
        hipError_t e;
        if ((e = d_head.reserve(POA_LISTS * POA_LIST_STRIDE + 16)) != hipSuccess) return e;
        if ((e = d_next_chunk.reserve(max_chunks)) != hipSuccess) return e;
        if ((e = d_slot_flag.reserve(16ull * (uint64_t)n_cu + 64)) != hipSuccess) return e;
        if ((e = h_seg_base.reserve(POA_MAX_SEGS)) != hipSuccess) return e;
        if ((e = h_short.reserve(16)) != hipSuccess) return e;
        h_short.p[0] = 0;
        if ((e = d_seg_base.reserve(POA_MAX_SEGS)) != hipSuccess) return e;
        if (!add_stream) {
            // (highest priority: when the pool does run short with the GPU full, the kernel that lists a new segment must be the
            // first to get the slot a workgroup frees)
            int pr_lo = 0, pr_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);
            if ((e = hipStreamCreateWithPriority(&add_stream, hipStreamNonBlocking, pr_hi)) != hipSuccess) return e;
            if ((e = reset_lists()) != hipSuccess) return e;
        }
        tables_ready = true;
        return hipSuccess;
    }
    // vga_align_prepare: the state regions and the first segments, allocated on a thread of its own before the first
    // vga_align_batch call needs them (on memory another process used the driver clears what it hands out: 13 GB = 0.2 s)
    std::thread preparer;
    void prepare_async(uint64_t state_want, uint64_t pool_want, uint32_t n_cu)
    {
        join_preparer();
        preparer = std::thread([this, state_want, pool_want, n_cu]() {
            (void)hipSetDevice(device);
            if (ensure_tables(n_cu) != hipSuccess) (void)hipGetLastError();  // (poa_run asks again, and reports)
            if (state_bytes < state_want) {
                uint8_t *q = nullptr;
                if (hipMalloc((void **)&q, state_want) == hipSuccess) {
                    if (state) (void)hipFree(state);
                    state = q;
                    state_bytes = state_want;
                } else
                    (void)hipGetLastError();  // (poa_run asks again, and reports)
            }
            if (pool_want) request(pool_want);
        });
    }
    void join_preparer() { if (preparer.joinable()) preparer.join(); }
    void stop_grower()
    {
        { std::lock_guard<std::mutex> lk(mu); grow_target = 0; }
        if (grower.joinable()) grower.join();
    }
    ~poa_ws()
    {
        join_preparer();
        stop_grower();
        for (auto &g : segs) (void)hipFree(g.p);
        if (classic) (void)hipFree(classic);
        if (state) (void)hipFree(state);
        if (add_stream) (void)hipStreamDestroy(add_stream);
        for (int i = 0; i < POA_SLOTS; i++)
            if (extra[i]) (void)hipStreamDestroy(extra[i]);
    }
    // asks for a pool of at least `target` bytes; returns at once (the grower thread allocates)
    void request(uint64_t target)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (target <= pool_size || (growing && target <= grow_target)) return;
        grow_target = target;
        grow_failed = false;
        if (growing) return;
        if (grower.joinable()) grower.join();
        growing = true;
        grower = std::thread([this]() {
            (void)hipSetDevice(device);
            for (;;) {
                uint64_t want;
                {
                    std::lock_guard<std::mutex> lk2(mu);
                    if (pool_size >= grow_target || segs.size() >= POA_MAX_SEGS) { growing = false; cv.notify_all(); return; }
                    want = std::min<uint64_t>(seg_bytes, (grow_target - pool_size + POA_CHUNK - 1) & ~(POA_CHUNK - 1));
                }
                // allocations of the context's calls go first: none in progress, and none for the last 3 ms (a call reserves its
                // buffers one after the other)
                if (owner) {
                    auto quiet_since = std::chrono::steady_clock::now();
                    for (;;) {
                        if (owner->alloc_urgent.load() > 0) quiet_since = std::chrono::steady_clock::now();
                        else if (std::chrono::steady_clock::now() - quiet_since >= std::chrono::milliseconds(3)) break;
                        { std::lock_guard<std::mutex> lk2(mu); if (grow_target == 0) break; }  // (stop_grower)
                        std::this_thread::sleep_for(std::chrono::microseconds(300));
                    }
                }
                uint8_t *q = nullptr;
                const hipError_t e = hipMalloc((void **)&q, want);
                std::lock_guard<std::mutex> lk2(mu);
                if (e != hipSuccess) { (void)hipGetLastError(); growing = false; grow_failed = true; cv.notify_all(); return; }
                segs.push_back({q, want});
                pool_size += want;
                cv.notify_all();
            }
        });
    }
    // waits until `bytes` of pool exist or the grower has stopped; returns what exists
    uint64_t wait_for(uint64_t bytes)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return pool_size >= bytes || !growing; });
        return pool_size;
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

// bytes of one state region of chunk-pool mode: the value-row ring, the wide-row scratch and a few kept value rows of a
// problem whose query has max_q bases
static uint64_t poa_state_size(uint32_t max_q)
{
    const uint64_t maxrow_all = (6ull * (uint64_t)((max_q + 8) & ~3u) + 15ull) & ~15ull;
    return (maxrow_all * (POA_RING_SPAN + 1) + 12ull * poa_lds_cols(max_q) + 4096ull + 65535ull) & ~65535ull;
}

// include/vga_hip.h.  Optional: what the first vga_align_batch call would allocate before its first kernel -- the state regions
// and about half of the chunk segments it is going to ask for -- starts to be allocated now, on a thread of its own.
extern "C" int vga_align_prepare(vga_ctx *ctx, uint64_t n_reads, uint32_t max_read_len)
{
    if (!ctx) return VGA_ERR_ARG;
    if (n_reads == 0 || max_read_len == 0 || max_read_len >= (1u << 24)) return VGA_OK;
    if (getenv("VGA_POA_ARENAS") && atoi(getenv("VGA_POA_ARENAS")) == 0) return VGA_OK;  // (classic mode sizes its pool itself)
    if (4ull * ((uint64_t)max_read_len + 8) > POA_CHUNK) return VGA_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return vga_set_error(ctx, VGA_ERR_HIP, "vga_align_prepare: hipSetDevice failed");
    vga_ctx_scope scope(ctx);
    if (!ctx->poa_ws) {
        ctx->poa_ws = new poa_ws();
        ctx->poa_ws_free = [](void *q) { delete (poa_ws *)q; };
    }
    poa_ws &W = *(poa_ws *)ctx->poa_ws;
    W.device = ctx->device;
    W.owner = ctx;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return VGA_OK;
    uint64_t avail = free_b > (16ull << 30) ? (uint64_t)((double)free_b * 0.85) : free_b / 4;
    double share = 1.0;  // of the GPU: its memory, and the workgroups that will be resident at a time
    {
        // contexts that share a GPU: vga_ctx_set_pool_fraction (the driver: 1 / their number); VGA_POOL_FRACTION is the diagnostic override
        double f = ctx->pool_fraction;
        if (const char *fr = getenv("VGA_POOL_FRACTION")) f = atof(fr);
        if (f > 0.0 && f < 1.0) { avail = (uint64_t)((double)avail * f); share = f; }
    }
    if (const char *env_pool = getenv("VGA_POOL_BYTES")) avail = std::min<uint64_t>(avail, strtoull(env_pool, nullptr, 10));
    const uint64_t state_size = poa_state_size(max_read_len);
    // (a context that shares the GPU with others has its share of the resident workgroups, which is what the chunk pool is sized
    // for below: eight contexts that each provided for a whole GPU spent 5.3 s of a 12 s run in allocations that the driver
    // serialises and clears at 40 GB/s)
    uint64_t ns = std::min<uint64_t>(16ull * (uint64_t)ctx->n_cu, std::max<uint64_t>(n_reads, 64));  // (state regions: one per workgroup of a launch -- fewer, and the rest of a launch spins for one on CUs the holders need)
    while (ns > 1 && ns * state_size > avail / 4) ns /= 2;
    if (ns * state_size > avail / 2) return VGA_OK;
    // the direction rows of a read of L bases against its subgraph: about 1.7 L rows of a band about 0.25 L wide plus the kept
    // value rows -- half of what the resident problems of such a call will hold (poa_run asks for the rest, from its probe)
    const double per_problem = 0.5 * (double)max_read_len * (double)max_read_len + 2.0 * (double)POA_CHUNK;
    const uint64_t resident = std::min<uint64_t>(n_reads, std::max<uint64_t>(32, (uint64_t)(6.0 * (double)ctx->n_cu * share)));
    uint64_t pool_want = (uint64_t)std::min<double>((double)resident * per_problem * 0.35, (double)avail / 4.0) & ~(POA_CHUNK - 1);
    if (pool_want < 16 * POA_CHUNK) pool_want = 0;
    W.prepare_async(ns * state_size, pool_want, (uint32_t)ctx->n_cu);
    return VGA_OK;
}

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
    if (params->remain_rule != VGA_REMAIN_LONGEST_PATH && params->remain_rule != VGA_REMAIN_FIRST_OUT_EDGE)
        return vga_set_error(ctx, VGA_ERR_ARG, "vga_poa_params.remain_rule %d: not one of VGA_REMAIN_*", params->remain_rule);
    if (feed.dev && feed.dev->remain_rule != params->remain_rule)
        return vga_set_error(ctx, VGA_ERR_ARG, "the device subgraph store was built for another remain_rule");
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
        // k_poa_dp_t4 can shrink its window of the row state down to 512 columns; k_poa_dp_lds keeps every column
        const size_t need = unpacked ? poa_lds_bytes(lds_cols_all, 128) : poa_t4_lds_bytes(std::min<uint32_t>(lds_cols_all, 512), lds_cols_all, 128);
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
                poa_prepare(views[p], G[p], params->remain_rule == VGA_REMAIN_FIRST_OUT_EDGE);
            // footprint in the pool: a direction byte per cell plus the value-row ring
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
    // (poa_feed::keep_text: what the previous call's items pointed into has been read by now)
    for (auto &hb : W.text_live) W.text_free.push_back(std::move(hb));
    W.text_live.clear();
    W.join_preparer();
#define POA_CHECK(call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return vga_set_error(ctx, VGA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                                 __LINE__);                                                          \
    } while (0)
    POA_CHECK(W.h_next.reserve(POA_SLOTS));
    POA_CHECK(W.d_next.reserve(POA_SLOTS));
    // ---- footprint probe: the first prepared problems (the largest, in launch order)
    const uint64_t n_probe = std::min<uint64_t>(n, 512);
    ensure(0, n_probe);
    if (malformed) return malformed_error();
    double probe_sum = 0, probe_big = 0;
    for (uint64_t i = 0; i < n_probe; i++) { probe_sum += est[order[i]]; probe_big = std::max(probe_big, est[order[i]]); }
    const double probe_mean = probe_sum / (double)n_probe;
    W.device = ctx->device;
    W.owner = ctx;
    // what this context may take of the GPU: everything else it allocates (staging of three sub-batches, the subgraph store,
    // the map workspace) keeps 15 % of what is free, at least 16 GB -- two processes sharing a GPU otherwise starve each other
    uint64_t avail_pool = 0;
    {
        size_t free_b = 0, total_b = 0;
        POA_CHECK(hipMemGetInfo(&free_b, &total_b));
        uint64_t have;
        { std::lock_guard<std::mutex> lk(W.mu); have = free_b + W.pool_size; }
        have += W.classic_size;
        const uint64_t reserve = std::max<uint64_t>((uint64_t)((double)have * 0.15), 16ull << 30);
        avail_pool = have > reserve ? have - reserve : have / 4;
        // several contexts on one GPU (vgaligner map --devices 0,0: the driver sets each one's share to 1 / their number) share it
        {
            double f = ctx->pool_fraction;
            if (const char *fr = getenv("VGA_POOL_FRACTION")) f = atof(fr);
            if (f > 0.0 && f < 1.0) avail_pool = (uint64_t)((double)avail_pool * f);
        }
        if (const char *env_pool = getenv("VGA_POOL_BYTES")) avail_pool = std::min<uint64_t>(avail_pool, strtoull(env_pool, nullptr, 10));
    }
    // Two sub-batches are in flight at any time, one per stream, each carving from its own half of the pool: while one
    // drains (its last workgroups, then the latency-bound traceback and the copies back) the other one's
    // workgroups fill the CUs.
    const char *force_k = getenv("VGA_POA_KERNEL");
    // k_poa_dp_t4 (vga_poa_t4.hpp), the default: scores scaled by 4 with argmax tags, G bytes 4 g - 1 / 4 g
    const bool t4_k = !(force_k && strstr(force_k, "unpacked")) && 4 * (params->gap_open1 + params->gap_ext1) - 1 <= 255 && 4 * (params->gap_open2 + params->gap_ext2) <= 255 &&
                      params->gap_ext1 >= 1 && params->match + params->mismatch >= 0 && params->match + params->mismatch < (1 << 20);
    // k_poa_dp_lds hands pool space out in 1 MiB chunks and assumes that a request fits one (k_poa_dp_t4 takes
    // whole chunks for a larger one): their two wide-row scratch rows (8 B per column) and an unbanded direction row with its
    // three predecessor planes (4 B per column) must stay below that
    if (!t4_k && 8ull * (uint64_t)poa_lds_cols(max_q) > POA_CHUNK)
        return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "query of %u bases: only k_poa_dp_t4 (default penalties range) handles queries beyond ~131 kbp", max_q);
    // traceback: fused into the DP kernel (default), or VGA_POA_TB=wave: a kernel of its own after the DP
    const bool tb_fused = !getenv("VGA_POA_TB") || strstr(getenv("VGA_POA_TB"), "fused");
    // k_poa_dp_t5 (vga_poa_t5.hpp), the default: the same rows under a leaderless row loop; its packed gap-byte arithmetic
    // needs 4 o_k + 1 <= 128.  VGA_POA_KERNEL=t4 selects k_poa_dp_t4
    const bool t5_k = t4_k && !(force_k && strstr(force_k, "t4")) && params->gap_open1 <= 31 && params->gap_open2 <= 31 &&
                      4 * (params->gap_open2 + params->gap_ext2) + 1 <= 255;
    // chunk-pool mode (k_poa_dp_t5 with its fused traceback; VGA_POA_ARENAS=0 switches it off): every row of a problem must
    // fit a chunk (a multi-predecessor row has four planes), and a state region holds the ring, the scratch rows and a few
    // kept value rows
    const uint64_t state_size = poa_state_size(max_q);
    const bool arena_wanted = t5_k && tb_fused && !(getenv("VGA_POA_ARENAS") && atoi(getenv("VGA_POA_ARENAS")) == 0) &&
                              4ull * ((uint64_t)max_q + 8) <= POA_CHUNK;
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
    // ... and when the call's problems are narrow-band (k_poa_dp_t6's launches: 2 048 single-wave workgroups each, two of them are
    // exactly the GPU's 4 096 wave slots): while the first of two launches drains, its freed slots stay empty until it has ended
    // and the next one is staged -- with a third launch in flight they are taken at once (config 5: 58 200 -> 66 400 reads/s,
    // same box; four: 62 700; config 3, wide bands: 9 410 with two, 9 330 with three)
    {
        double wsum = 0;
        for (uint64_t i = 0; i < n_probe; i++) wsum += estw[order[i]];
        if (arena_wanted && wsum / (double)n_probe <= 800.0) n_slots = 3;
    }
    if (const char *e = getenv("VGA_POA_SLOTS")) n_slots = std::max(1, std::min(POA_SLOTS, atoi(e)));
    hipStream_t sarr[POA_SLOTS];
    sarr[0] = st;
    for (int i = 1; i < n_slots; i++) {
        if (!W.extra[i]) POA_CHECK(hipStreamCreateWithFlags(&W.extra[i], hipStreamNonBlocking));
        sarr[i] = W.extra[i];
    }
    // ---- chunk-pool mode: the state regions now, the chunk segments on the grower thread (the first launch starts as soon
    // as one segment is there; its workgroups all begin with empty hands).  The pool should hold what the resident
    // workgroups have written so far: about six per CU, each on average two thirds through a problem of the probe's mean
    // size -- a workgroup that finds the free list empty waits for chunks to come back (and gives its problem up after a
    // bounded wait: the classic pass takes it).
    uint32_t n_arenas = 0;  // state regions (0: classic mode for the whole call)
    poa_chunk_pool CP = {};
    if (arena_wanted) {
        uint64_t ns = std::min<uint64_t>(16ull * (uint64_t)ctx->n_cu, std::max<uint64_t>(n, 64));  // (16 two-wave workgroups per CU at most)
        if (const char *e = getenv("VGA_POA_ARENAS")) ns = std::min<uint64_t>(ns, std::max<uint64_t>(1, strtoull(e, nullptr, 10)));
        while (ns > 1 && ns * state_size > avail_pool / 4) ns /= 2;
        // (not scaled by the context's share of the GPU: a context whose slice holds a call's longest problems needs their whole
        // footprint whatever its share -- scaled, eight contexts on one GPU waited seconds for chunks and gave problems up)
        const uint64_t resident = std::min<uint64_t>(n, 6ull * (uint64_t)ctx->n_cu);
        double fill = 0.7;
        if (const char *e = getenv("VGA_POOL_FILL")) fill = atof(e);
        // what is resident at a time: every long problem of the call (a launch of their own, one CU each: poa_feed::klass) and
        // `resident` workgroups of the others.  The probe is the head of the launch order, where the long problems stand: their
        // footprints are summed, not taken for the mean of the rest (config 4, 12 000 reads: 217 GB asked for where 75 GB do)
        double long_sum = 0, bulk_sum = 0;
        uint64_t long_cnt = 0, bulk_cnt = 0, long_all = 0;
        if (feed.klass) {
            for (uint64_t p = 0; p < n; p++) long_all += feed.klass[p] != 0;
            for (uint64_t i = 0; i < n_probe; i++) {
                if (feed.klass[order[i]]) { long_sum += est[order[i]]; long_cnt++; }
                else { bulk_sum += est[order[i]]; bulk_cnt++; }
            }
            if (long_cnt && long_all > long_cnt) long_sum *= (double)long_all / (double)long_cnt;
        } else { bulk_sum = probe_sum; bulk_cnt = n_probe; }
        const double bulk_mean = bulk_cnt ? bulk_sum / (double)bulk_cnt : probe_mean;
        const uint64_t bulk_resident = std::min<uint64_t>(n - std::min<uint64_t>(n, long_all), resident);
        uint64_t want = (uint64_t)((long_sum + (double)bulk_resident * bulk_mean) * W.pool_scale * fill) + 64 * POA_CHUNK;
        want = std::max<uint64_t>(want, (uint64_t)(probe_big * W.pool_scale * 1.5));
        want = std::min<uint64_t>(want, avail_pool > ns * state_size ? avail_pool - ns * state_size : avail_pool / 2);
        want = (want + POA_CHUNK - 1) & ~(POA_CHUNK - 1);
        if (ns * state_size <= avail_pool / 2 && want >= 16 * POA_CHUNK) {
            if (W.state_bytes < ns * state_size) {
                if (W.state) { (void)hipFree(W.state); W.state = nullptr; W.state_bytes = 0; }
                POA_CHECK(hipMalloc((void **)&W.state, ns * state_size));
                W.state_bytes = ns * state_size;
            }
            if (W.classic && want > 0) {  // (memory the classic pool holds is memory the segments cannot have)
                size_t free_b = 0, total_b = 0;
                POA_CHECK(hipMemGetInfo(&free_b, &total_b));
                uint64_t have;
                { std::lock_guard<std::mutex> lk(W.mu); have = W.pool_size; }
                if (have < want && free_b < want - have + (8ull << 30)) { (void)hipFree(W.classic); W.classic = nullptr; W.classic_size = 0; }
            }
            {
                std::lock_guard<std::mutex> lk(W.mu);  // (the grower may be at work already: vga_align_prepare)
                W.seg_bytes = std::min<uint64_t>(1ull << POA_SEG_LOG2, std::max<uint64_t>(want, 16 * POA_CHUNK));
                if (const char *e = getenv("VGA_POOL_SEG")) W.seg_bytes = std::max<uint64_t>(16 * POA_CHUNK, strtoull(e, nullptr, 10) & ~(POA_CHUNK - 1));
                W.seg_bytes = std::min<uint64_t>(W.seg_bytes, 1ull << POA_SEG_LOG2);  // (chunks are numbered segment << 12 | chunk in segment)
            }
            tr.mark("pool: state regions");
            POA_CHECK(W.ensure_tables((uint32_t)ctx->n_cu));
            tr.mark("pool: tables");
            W.request(want);
            // The launches start when the pool holds what their resident workgroups need: on memory that was used before, the driver
            // clears a segment as it hands it out (0.1 s per 4 GiB), and launches that fill the GPU with workgroups waiting for chunks
            // leave the kernel that lists new segments no slot to run in (a 12 000-read call of config 4 that started with a sixth
            // of its pool took 19 s).  On fresh memory this waits a few milliseconds.
            const uint64_t got = W.wait_for(want);
            tr.mark("pool: segments");
            if (got >= 16 * POA_CHUNK) {
                const uint32_t max_chunks = (uint32_t)(POA_MAX_SEGS * (1ull << (POA_SEG_LOG2 - 20)));
                W.h_short.p[0] = 0;
                // (on the context's stream, and waited for: hipMemset runs on the null stream and may return before the device has
                // done it -- no stream of this library waits for the null stream, and on a GPU that other contexts keep full the
                // flags were cleared AFTER the first workgroups of slots 1 and 2 had taken their state regions: a second workgroup
                // took the same region, and both problems came back with wrong alignments)
                POA_CHECK(hipMemsetAsync(W.d_slot_flag.p, 0, ns * sizeof(uint32_t), st));
                POA_CHECK(hipStreamSynchronize(st));
                tr.mark("pool: flags cleared");
                n_arenas = (uint32_t)ns;
                CP.head = W.d_head.p; CP.next = W.d_next_chunk.p; CP.seg_base = W.d_seg_base.p;
                CP.cps_log2 = POA_SEG_LOG2 - 20; CP.n_slots = n_arenas; CP.state_base = W.state; CP.state_size = state_size;
                CP.slot_flag = W.d_slot_flag.p; CP.stats = W.d_head.p + POA_LISTS * POA_LIST_STRIDE;
                CP.short_flag = W.h_short.p;
                if (getenv("VGA_POOL_CHECK") && atoi(getenv("VGA_POOL_CHECK")) != 0) {  // (diagnostics: vga_poa_kernels.hpp, poa_chunk_pool::owner)
                    if (!W.d_owner.p) {
                        POA_CHECK(W.d_owner.reserve(max_chunks));
                        POA_CHECK(hipMemsetAsync(W.d_owner.p, 0, W.d_owner.cap * sizeof(uint32_t), st));
                        POA_CHECK(hipStreamSynchronize(st));
                    }
                    CP.owner = W.d_owner.p;
                }
            }
        }
    }
    // new segments' chunks join the free list (a tiny kernel on a stream of its own), and requests that found every list empty
    // make the pool grow.  Called before every launch and, every millisecond, by the keeper thread below
    std::mutex list_mu;
    auto list_new_segments = [&]() -> hipError_t {
        std::lock_guard<std::mutex> list_lk(list_mu);
        std::vector<poa_ws::seg_t> fresh;
        {
            std::lock_guard<std::mutex> lk(W.mu);
            for (size_t k = W.segs_listed; k < W.segs.size(); k++) fresh.push_back(W.segs[k]);
        }
        for (const poa_ws::seg_t &g : fresh) {
            const uint32_t first = (uint32_t)W.segs_listed << (POA_SEG_LOG2 - 20), cnt = (uint32_t)(g.size >> 20);
            W.h_seg_base.p[W.segs_listed] = (uint64_t)g.p;
            (void)hipMemcpyAsync(W.d_seg_base.p + W.segs_listed, W.h_seg_base.p + W.segs_listed, sizeof(uint64_t), hipMemcpyHostToDevice, W.add_stream);
            hipLaunchKernelGGL(k_poa_chunks_add, dim3(1), dim3(64), 0, W.add_stream, CP, first, cnt);
            W.segs_listed++;
            W.chunks_listed += cnt;
            if (tr.on && W.segs_listed > 1) fprintf(stderr, "[vga-trace] poa: segment %zu listed (%u chunks)\n", W.segs_listed, cnt);
        }
        hipError_t e = fresh.empty() ? hipSuccess : hipStreamSynchronize(W.add_stream);
        // requests that found the list empty: the pool is short of what the resident workgroups need -- more segments.
        // (The kernels raise a flag in pinned host memory: reading it costs no GPU work.)
        volatile uint32_t *flag = W.h_short.p;
        // one step at a time: what is raised while a step is still being allocated and listed is the shortage that step
        // answers -- without this the target runs away, +50 % every few milliseconds
        bool settled;
        { std::lock_guard<std::mutex> lk(W.mu); settled = !W.growing && W.segs_listed == W.segs.size(); }
        if (e == hipSuccess && *flag) {
            *flag = 0;
            if (settled) {
                uint64_t ps; { std::lock_guard<std::mutex> lk(W.mu); ps = std::max(W.pool_size, W.grow_target); }
                const uint64_t more = std::min<uint64_t>(ps + ps / 2 + (4ull << 30), avail_pool);
                if (more > ps) W.request(more);
                if (tr.on) fprintf(stderr, "[vga-trace] poa: requests have found every free list empty: pool target %.1f -> %.1f GB\n", (double)ps / 1e9, (double)more / 1e9);
            }
        }
        return e;
    };
    // the keeper: the thread that runs this call may be held up for as long as a launch takes (a staging buffer that grows, the
    // look-ahead preparation waiting for its kernel), and workgroups that wait for chunks meanwhile keep the launch from ending --
    // so the pool is looked after by a thread that does nothing else
    struct keeper_t {
        std::atomic<bool> stop{false};
        std::thread t;
        ~keeper_t() { stop = true; if (t.joinable()) t.join(); }
    } keeper;
    if (n_arenas && !getenv("VGA_POOL_NOPOLL"))
        keeper.t = std::thread([&]() {
            (void)hipSetDevice(ctx->device);
            vga_ctx_scope scope(ctx);
            while (!keeper.stop) {
                (void)list_new_segments();
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
        });
    // ---- the classic pool: one contiguous piece, cut into a part per slot; allocated when a classic launch is first needed
    uint64_t half_pool = 0;
    double classic_need = probe_mean * (double)n;  // estimated bytes of the problems that will run in classic mode (the whole call, or what chunk mode handed back)
    auto ensure_classic = [&]() -> int {
        if (half_pool) return VGA_OK;
        const double want_d = classic_need * W.pool_scale * 1.3 + std::min<double>((double)n, classic_need / std::max(1.0, probe_mean) + 64.0) * 3.0 * (double)POA_CHUNK;
        const uint64_t want = (uint64_t)want_d + 64 * POA_CHUNK;
        uint64_t target = std::min(std::max<uint64_t>(2 * want, n_arenas ? 1ull << 30 : 8ull << 30), avail_pool) & ~(POA_CHUNK - 1);
        if (W.classic_size < std::min<uint64_t>(want, target)) {
            if (W.classic) { (void)hipFree(W.classic); W.classic = nullptr; W.classic_size = 0; }
            if (target < 64 * POA_CHUNK) return vga_set_error(ctx, VGA_ERR_NOMEM, "only %llu bytes of HBM for the traceback pool", (unsigned long long)target);
            // the chunk segments give way (no chunk-mode launch is in flight when a classic one starts)
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            if (free_b < target + (4ull << 30)) {
                std::lock_guard<std::mutex> list_lk(list_mu);  // (the keeper is not listing segments meanwhile)
                W.stop_grower();
                std::lock_guard<std::mutex> lk(W.mu);
                for (auto &g : W.segs) (void)hipFree(g.p);
                W.segs.clear(); W.pool_size = 0; W.segs_listed = 0; W.chunks_listed = 0; W.empties_seen = 0;
                if (W.d_head.p) (void)W.reset_lists();
                (void)hipMemGetInfo(&free_b, &total_b);
                target = std::min<uint64_t>(target, free_b > (4ull << 30) ? (free_b - (4ull << 30)) & ~(POA_CHUNK - 1) : target);
            }
            const hipError_t e = hipMalloc((void **)&W.classic, target);
            if (e != hipSuccess) return vga_set_error(ctx, VGA_ERR_NOMEM, "hipMalloc of the %llu byte traceback pool failed: %s", (unsigned long long)target, hipGetErrorString(e));
            W.classic_size = target;
        }
        half_pool = (W.classic_size / (uint64_t)n_slots) & ~(POA_CHUNK - 1);
        return VGA_OK;
    };
    if (!n_arenas) { const int rc = ensure_classic(); if (rc != VGA_OK) return rc; }
    tr.mark("pool");
    if (tr.on) {
        uint64_t ps; { std::lock_guard<std::mutex> lk(W.mu); ps = W.pool_size; }
        fprintf(stderr, "[vga-trace] poa: %s; chunk segments so far %.1f GB, %u state regions of %.2f MB, classic pool %.1f GB\n", n_arenas ? "chunk-pool mode" : "classic mode",
                (double)ps / 1e9, n_arenas, (double)state_size / 1e6, (double)W.classic_size / 1e9);
    }

    poa_dev_params P;
    P.match = params->match; P.mismatch = params->mismatch; P.o1 = params->gap_open1; P.e1 = params->gap_ext1;
    P.o2 = params->gap_open2; P.e2 = params->gap_ext2; P.banded = params->wb >= 0;

    bool t4_any = false;     // ... and 6 B with k_poa_dp_t4
    bool any_fused = false;  // the DP kernel walked the alignments back itself
    int t_total = vga_timer_begin(ctx, "poa_total", 0);
    struct sub_t { uint64_t i0, i1; double raw_est; int slot; int oset; bool general = false; bool arena = false; };
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
    auto launch = [&](uint64_t i0, uint64_t cap, int slot, bool general, bool arena) -> sub_t {
        hipStream_t st = sarr[slot];  // shadows the context's stream inside this lambda
        poa_slot &S = W.slot[slot];
        if (!arena) {
            const int rcc = ensure_classic();
            if (rcc != VGA_OK) { dev_failed = true; dev_rc = rcc; return {i0, i0, 0.0, slot, 0}; }
        }
        const auto t_launch0 = std::chrono::steady_clock::now();
        auto lt = [&](const char *what) {  // (VGA_TRACE: where the host's time goes between a launch ending and the next one starting)
            if (tr.on) fprintf(stderr, "[vga-trace] poa:     launch set-up: %-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_launch0).count());
        };
        if (arena && launch_err == hipSuccess) launch_err = list_new_segments();
        lt("free lists looked after");
        uint8_t *pool_base = arena ? nullptr : W.classic + (uint64_t)slot * half_pool;
        const double budget = (double)half_pool * 0.92;
        double used_est = 0, raw_est = 0, cells_est = 0;
        uint64_t i1 = i0;
        // (device store: a launch gathers from one part of it)
        if (feed.dev && feed.dev->split > i0 && feed.dev->split < cap) cap = feed.dev->split;
        while (i1 < cap) {
            // (device store: a problem's preparation is a copy of its sizes -- a whole launch's worth in one fan-out; 256 at a time
            // cost eight thread fan-outs per launch, 6-7 ms before each of a call's first two launches with the GPU idle)
            if (!ready[order[i1]]) ensure(i1, std::min<uint64_t>(cap, i1 + (feed.dev ? 4096 : 256)));
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
            raw_est += est[order[i1]];
            cells_est += (double)G[order[i1]].N * estw[order[i1]];
            i1++;
        }
        lt("problems chosen (and prepared)");
        auto chk = [&](hipError_t e) { if (e != hipSuccess && launch_err == hipSuccess) launch_err = e; };
        bool sub_t4 = false;   // ... k_poa_dp_t4 (its own direction-byte encoding)
        bool sub_t5 = false;   // ... k_poa_dp_t5 (direction dwords)
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
            pb.flags = 0u;  // (bit 0: not for the chunk pool -- every problem of a chunk-mode call fits it by construction)
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
        // cs / CIGAR / node path on the device (K4c) when the caller does not need the per-base rows; VGA_POA_TEXT=host keeps the host's
        const bool text_on_device = dev && !feed.want_rows && !(getenv("VGA_POA_TEXT") && strstr(getenv("VGA_POA_TEXT"), "host"));
        chk(S.h_probs.reserve(nb));
        if (dev) chk(S.h_ids.reserve(2 * (size_t)nb));
        else {
            chk(S.h_ntab.reserve(tot_nodes)); chk(S.h_seq32.reserve(tot_seq / 4 + 1));
            chk(S.h_preds.reserve(tot_preds + 1)); chk(S.h_sink.reserve(tot_sink + 1)); chk(S.h_q.reserve(tot_q + 1));
        }
        chk(O.h_outs.reserve(nb));
        if (!text_on_device) {  // (K4c: the operations stay on the device; the fallback reserves these when it needs them)
            chk(O.h_ops.reserve(tot_ops)); chk(O.h_orow.reserve(tot_ops));
            if (feed.dev) chk(O.h_seq.reserve(tot_seq + 4));
        }
        chk(S.d_probs.reserve(nb)); chk(S.d_ntab.reserve(tot_nodes)); chk(S.d_seq32.reserve(tot_seq / 4 + 1));
        chk(S.d_preds.reserve(tot_preds + 1)); chk(S.d_sink.reserve(tot_sink + 1)); chk(S.d_q.reserve(tot_q + 1));
        chk(S.d_rows.reserve(tot_rows)); chk(S.d_outs.reserve(nb)); chk(S.d_ops.reserve(tot_ops)); chk(S.d_orow.reserve(tot_ops));
        if (dev) chk(S.d_ids.reserve(2 * (size_t)nb));
        if (launch_err != hipSuccess) return {i0, i0, 0.0, slot, 0};
        lt("buffers reserved");
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
        lt("graphs staged");
        int t_dp = vga_timer_begin(ctx, "poa_band_dp", 0, st);
        {
            uint32_t mq = 0;
            double mw = 0;
            double sum_w = 0;
            for (uint64_t i = i0; i < i1; i++) { mq = std::max(mq, G[order[i]].qlen); mw = std::max(mw, estw[order[i]]); sum_w += estw[order[i]]; }
            const double mean_w = sum_w / (double)nb;
            const uint32_t lds_cols = poa_lds_cols(mq);
            // VGA_POA_KERNEL (testing): "unpacked" selects k_poa_dp_lds, "128" / "256" / "512" pin the workgroup size,
            // "full" keeps every column in LDS, "generic" the run-time penalties; VGA_POA_WINDOW=<power of two> pins the LDS
            // column window
            const char *force = getenv("VGA_POA_KERNEL");
            const bool t4 = t4_k;
            t4_any = t4_any || t4;
            const bool def_pen = P.o1 == 4 && P.e1 == 2 && P.o2 == 24 && P.e2 == 1 && !(force && strstr(force, "generic"));
            sub_fused = t4 && tb_fused;
            any_fused = any_fused || sub_fused;
            const bool giant = feed.klass && feed.klass[order[i0]] && !getenv("VGA_POA_NO_GIANTS");
            // LDS column window (k_poa_dp_t4): 4096 columns keep almost every row of a 10 kbp read resident (its widest
            // rows, a few per cent, take the HBM detour described in the kernel) and let five workgroups share a CU.
            // Queries that fit a smaller array anyway keep every column.
            uint32_t hg_cols = lds_cols, win_mask = 0xFFFFFFFFu;
            auto set_window = [&](uint32_t want) {
                hg_cols = lds_cols; win_mask = 0xFFFFFFFFu;
                if (want >= 16 && (want & (want - 1)) == 0 && want < lds_cols) { hg_cols = want; win_mask = want - 1; }
            };
            uint32_t want = 4096;
            if (t4 && !(force && strstr(force, "full"))) {
                // narrow bands: a window that just covers the launch's widest estimated row (rows that turn out wider take
                // the HBM detour) leaves room for more two-wave workgroups per CU -- such launches are bound by the latency
                // of the per-row chain, not by instruction issue (config 5: +20 %)
                if (mean_w <= 800.0) {
                    uint32_t w2 = 512;
                    while (w2 < 4096 && (double)w2 < mw * 1.25 + 16.0) w2 <<= 1;
                    want = w2;
                }
                // the longest problems of a call have a CU almost to themselves (16 waves, 55-66 KB of LDS) and are often as wide as the
                // query: every column in LDS (62 % of the cells of config 4's 107 000-row problem lie in rows wider than 8 192 columns,
                // 100 % of those of its 34 000-row problems: 24-27 us per row through the HBM detour against 7.6).  A query too long for
                // that falls back to the largest window that fits (below)
                if (giant) want = 0;
                if (giant && getenv("VGA_POA_GIANT_WINDOW")) want = (uint32_t)strtoul(getenv("VGA_POA_GIANT_WINDOW"), nullptr, 10);  // (0: every column)
                const char *ew = getenv("VGA_POA_WINDOW");
                if (ew) want = (uint32_t)strtoul(ew, nullptr, 10);
                set_window(want);
            }
            // workgroup size
            int nt = mq >= 3072 ? 512 : (mq >= 768 ? 256 : 128);
            // k_poa_dp_t5 (vga_poa_t5.hpp), the default: the same rows under a leaderless row loop; its packed gap-byte arithmetic
            // needs 4 o_k + 1 <= 128 and 4 (o2 + e2) + 1 <= 255.  VGA_POA_KERNEL=t4 selects k_poa_dp_t4
            const bool t5 = t4 && !(force && strstr(force, "t4")) && P.o1 <= 31 && P.o2 <= 31 && 4 * (P.o2 + P.e2) + 1 <= 255 && 4 * (P.o1 + P.e1) <= 255;
            auto lds_of = [&](int t) { return t5 ? poa_t5_lds_bytes(hg_cols, lds_cols, t) : (t4 ? poa_t4_lds_bytes(hg_cols, lds_cols, t) : poa_lds_bytes(lds_cols, t)); };
            const size_t lds_limit = 160 * 1024 - 256;
            if (t4) {
                // the one that keeps the most waves resident (LDS and 16 wave slots per CU at this kernel's register count
                // bound the workgroups per CU; the problems still to be run -- this sub-batch and the ones that will overlap
                // it -- bound how many there are); ties go to the smaller workgroup, whose barriers are cheaper
                size_t best_waves = 0;
                for (int t = 128; t <= 512; t += 64) {
                    const size_t by_lds = std::max<size_t>(1, (160 * 1024) / (lds_of(t) + 256));
                    const size_t per_cu = std::min<size_t>(by_lds, (size_t)(16 / (t / 64)));
                    const size_t waves = std::min<size_t>(order.size() - i0 + in_flight_other, per_cu * (size_t)ctx->n_cu) * (size_t)(t / 64);
                    if (waves > best_waves) { best_waves = waves; nt = t; }
                }
                // narrow bands (one step of a 128-thread workgroup covers a typical row): the per-row set-up and the
                // barriers dominate, and they are per wave -- config 5 (mean width 340): +6 % with 128 threads
                {
                    static const double nt128_w = getenv("VGA_POA_NT128_W") ? atof(getenv("VGA_POA_NT128_W")) : 800.0;  // (experiments)
                    if (mean_w <= nt128_w) nt = 128;  // (the estimate is of a problem's widest rows: about twice its mean band)
                }
                if (giant) nt = getenv("VGA_POA_GIANT_NT") ? atoi(getenv("VGA_POA_GIANT_NT")) : 1024;  // (config 4: +5 % over 512, same-box)
                const char *ent = getenv("VGA_POA_NT");
                if (ent) nt = atoi(ent);
                if (nt < 128 || (nt > 512 && nt != 768 && nt != 1024) || nt % 64) nt = 512;
            }
            if (force) {
                if (strstr(force, "128")) nt = 128;
                else if (strstr(force, "256")) nt = 256;
                else if (strstr(force, "512")) nt = 512;
            }
            if (t4) {
                // a query whose column codes leave no room for the chosen window: first halve the window (down to 512
                // columns), then step the workgroup down through the instantiated sizes
                while (lds_of(nt) > lds_limit && win_mask != 0xFFFFFFFFu && hg_cols > 512) set_window(hg_cols / 2);
                while (lds_of(nt) > lds_limit && lds_cols > 512 && win_mask == 0xFFFFFFFFu && hg_cols > 512) {
                    uint32_t w2 = 1u << 30;
                    while (w2 >= lds_cols) w2 >>= 1;
                    set_window(w2);
                }
                while (nt > 128 && lds_of(nt) > lds_limit) nt = nt > 768 ? 768 : (nt > 512 ? 512 : nt - 64);
            } else
                while (nt > 128 && lds_of(nt) > lds_limit) nt /= 2;
            const size_t lds = lds_of(nt);
            if (tr.on)
                fprintf(stderr, "[vga-trace] poa: (at %.1f ms) launch %u problems, NT %d, %s, window %u of %u columns, width estimate mean %.0f max %.0f, LDS %zu B\n",
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count(), nb, nt,
                        t5 ? "k_poa_dp_t5" : (t4 ? "k_poa_dp_t4" : "k_poa_dp_lds"), hg_cols, lds_cols, mean_w, mw, lds);
            (void)hipGetLastError();  // a launch failure below must be this launch's, not an older ignored status
#define POA_ARGS S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, S.d_sink.p, P, S.d_rows.p, pool_base,          \
                 W.d_next.p + slot, half_pool, S.d_outs.p, lds_cols
            sub_t4 = t4;
            sub_t5 = t5;
            if (t4) {
                poa_chunk_pool cp_arg = CP;
                if (!arena) cp_arg.n_slots = 0;
                const poa_t5_args t5a = {S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, S.d_rows.p, pool_base, W.d_next.p + slot, half_pool,
                                         S.d_outs.p, (sub_fused ? S.d_ops.p : nullptr), (sub_fused ? S.d_orow.p : nullptr), cp_arg, lds_cols, hg_cols, win_mask, P,
                                         (giant && !(getenv("VGA_POA_GIANT_PRIO") && atoi(getenv("VGA_POA_GIANT_PRIO")) == 0)) ? 1u : 0u};
                (void)t5a;
#define POA_T4_ARGS S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, P, S.d_rows.p, pool_base, W.d_next.p + slot, half_pool,   \
                    S.d_outs.p, lds_cols, hg_cols, win_mask, (sub_fused ? S.d_ops.p : nullptr), (sub_fused ? S.d_orow.p : nullptr),       \
                    0u, (uint64_t)0, (unsigned long long *)nullptr, (uint32_t *)nullptr
#define POA_T4_LAUNCH(T)                                                                                                     \
    case T:                                                                                                                  \
        if (t5 && def_pen) {                                                                                                 \
            chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_t5<T, true>, lds)); \
            hipLaunchKernelGGL((k_poa_dp_t5<T, true>), dim3(nb), dim3(T), lds, st, S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, t5a);                                    \
        } else if (t5) {                                                                                                     \
            chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_t5<T, false>, lds)); \
            hipLaunchKernelGGL((k_poa_dp_t5<T, false>), dim3(nb), dim3(T), lds, st, S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, t5a);                                   \
        } else if (def_pen) {                                                                                                     \
            chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_t4<T, true>, lds)); \
            hipLaunchKernelGGL((k_poa_dp_t4<T, true>), dim3(nb), dim3(T), lds, st, POA_T4_ARGS);                            \
        } else {                                                                                                             \
            chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_t4<T, false>, lds)); \
            hipLaunchKernelGGL((k_poa_dp_t4<T, false>), dim3(nb), dim3(T), lds, st, POA_T4_ARGS);                           \
        }                                                                                                                    \
        break;
                // k_poa_dp_t6 (vga_poa_t6.hpp): one wave per problem, the row in registers -- launches of narrow bands in chunk-pool
                // mode with the fused traceback; what does not fit its window comes back with POA_ST_RETRY and runs below
                const bool t6_forced = force && strstr(force, "t6");
                const bool t6 = t5 && arena && sub_fused && !general && !giant && !(force && strstr(force, "t5")) && !(getenv("VGA_POA_T6") && atoi(getenv("VGA_POA_T6")) == 0) &&
                                (t6_forced || (mean_w <= 800.0 && mw <= 1000.0));
                // k_poa_dp_t7 (vga_poa_t7.hpp): t6's eight-columns-per-lane row for bands that need several waves.  The launch of a call's
                // longest problems runs it (1 024 threads: 8 192 columns per step, every column in LDS): their rows are a serial chain
                // on a CU of their own, and a t7 row takes 4.9-5.9 us where a t5 row of the same 6 000-column band takes 7.6 (config 4:
                // 10 900 -> 13 600-13 850 reads/s).  On ordinary launches it issues as many instructions per cell as t5 and loses to the
                // problems that leave its window (config 3: 7 700 against 9 450 reads/s), so there it is opt-in (VGA_POA_KERNEL=t7)
                const bool t7 = t5 && arena && sub_fused && !general && !t6 &&
                                ((force && strstr(force, "t7")) ||
                                 (giant && !(force && strstr(force, "t5")) && !(getenv("VGA_POA_T7_GIANTS") && atoi(getenv("VGA_POA_T7_GIANTS")) == 0)));
                if (t7) {
                    int nt7 = giant ? 1024 : 256;
                    if (const char *e = getenv("VGA_POA_T7_NT")) nt7 = atoi(e);
                    uint32_t w7 = 4096;
                    while (w7 < lds_cols && (giant || (double)w7 < mw * 1.6 + 64.0)) w7 <<= 1;  // (a power of two that holds the launch's widest expected row)
                    if (const char *e = getenv("VGA_POA_T7_WINDOW")) w7 = (uint32_t)strtoul(e, nullptr, 10);
                    while (poa_t5_lds_bytes(w7, lds_cols, nt7) > lds_limit && w7 > 1024) w7 >>= 1;
                    const size_t lds7 = poa_t5_lds_bytes(w7, lds_cols, nt7);
                    poa_t5_args t7a = t5a;
                    t7a.hg_cols = w7; t7a.win_mask = w7 - 1;
                    if (tr.on) fprintf(stderr, "[vga-trace] poa:     k_poa_dp_t7<%d>: window %u columns, LDS %zu B\n", nt7, w7, lds7);
#define POA_T7_LAUNCH(T)                                                                                                         \
    case T:                                                                                                                      \
        if (def_pen) {                                                                                                           \
            chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_t7<T, true>, lds7)); \
            hipLaunchKernelGGL((k_poa_dp_t7<T, true>), dim3(nb), dim3(T), lds7, st, S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, t7a); \
        } else {                                                                                                                 \
            chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_t7<T, false>, lds7)); \
            hipLaunchKernelGGL((k_poa_dp_t7<T, false>), dim3(nb), dim3(T), lds7, st, S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, t7a); \
        }                                                                                                                        \
        break;
                    switch (nt7) {
                        POA_T7_LAUNCH(128) POA_T7_LAUNCH(256) POA_T7_LAUNCH(512) POA_T7_LAUNCH(1024)
                    default: chk(hipErrorInvalidValue);
                    }
#undef POA_T7_LAUNCH
                } else if (t6) {
                    const size_t lds6 = poa_t6_lds_bytes<8>(lds_cols);
                    if (tr.on) fprintf(stderr, "[vga-trace] poa:     k_poa_dp_t6<8>: one wave per problem, LDS %zu B\n", lds6);
                    if (def_pen) {
                        chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_t6<8, true>, lds6));
                        hipLaunchKernelGGL((k_poa_dp_t6<8, true>), dim3(nb), dim3(64), lds6, st, S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, t5a);
                    } else {
                        chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_t6<8, false>, lds6));
                        hipLaunchKernelGGL((k_poa_dp_t6<8, false>), dim3(nb), dim3(64), lds6, st, S.d_probs.p, S.d_q.p, S.d_ntab.p, S.d_seq32.p, S.d_preds.p, t5a);
                    }
                } else
                switch (nt) {
                    POA_T4_LAUNCH(128) POA_T4_LAUNCH(192) POA_T4_LAUNCH(256) POA_T4_LAUNCH(320)
                    POA_T4_LAUNCH(384) POA_T4_LAUNCH(448) POA_T4_LAUNCH(512) POA_T4_LAUNCH(768) POA_T4_LAUNCH(1024)
                default: chk(hipErrorInvalidValue);
                }
#undef POA_T4_LAUNCH
#undef POA_T4_ARGS
            } else if (nt == 128) {
                chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_lds<128, 4>, lds));
                hipLaunchKernelGGL((k_poa_dp_lds<128, 4>), dim3(nb), dim3(128), lds, st, POA_ARGS);
            } else if (nt == 256) {
                chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_lds<256, 4>, lds));
                hipLaunchKernelGGL((k_poa_dp_lds<256, 4>), dim3(nb), dim3(256), lds, st, POA_ARGS);
            } else {
                chk(poa_allow_lds(ctx->device, (const void *)k_poa_dp_lds<512, 4>, lds));
                hipLaunchKernelGGL((k_poa_dp_lds<512, 4>), dim3(nb), dim3(512), lds, st, POA_ARGS);
            }
#undef POA_ARGS
            chk(hipGetLastError());
        }
        vga_timer_end(ctx, t_dp);
        int t_tb = vga_timer_begin(ctx, "poa_traceback", 0, st);
        if (sub_fused) {
            // the DP kernel's first wave already walked each problem back
        }
        else if (sub_t5)
            hipLaunchKernelGGL(k_poa_traceback_wave<2>, dim3(nb), dim3(64), 0, st, nb, S.d_probs.p, S.d_rows.p, S.d_preds.p,
                               pool_base, S.d_outs.p, S.d_ops.p, S.d_orow.p, 0);
        else if (sub_t4)
            hipLaunchKernelGGL(k_poa_traceback_wave<1>, dim3(nb), dim3(64), 0, st, nb, S.d_probs.p, S.d_rows.p, S.d_preds.p,
                               pool_base, S.d_outs.p, S.d_ops.p, S.d_orow.p, 0);
        else
            hipLaunchKernelGGL(k_poa_traceback_wave<0>, dim3(nb), dim3(64), 0, st, nb, S.d_probs.p, S.d_rows.p, S.d_preds.p,
                               pool_base, S.d_outs.p, S.d_ops.p, S.d_orow.p, 0);
        vga_timer_end(ctx, t_tb);
        chk(hipMemcpyAsync(O.h_outs.p, S.d_outs.p, nb * sizeof(poa_out), hipMemcpyDeviceToHost, st));
        chk(hipMemcpyAsync(W.h_next.p + slot, W.d_next.p + slot, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        O.text = text_on_device;
        O.tot_ops = tot_ops; O.tot_seq = tot_seq;
        if (text_on_device) {
            // K4c: the strings and the node path are written where the operations are (vga_poa_text.hpp); what crosses PCIe now is a
            // record per problem and the counter of the arena -- the text itself follows when the host knows how much there is
            uint64_t arena = std::min<uint64_t>(2ull * tot_ops + 64ull * nb + 4096ull, 0xF0000000ull);
            if (const char *e = getenv("VGA_POA_TEXT_ARENA")) arena = std::min<uint64_t>(arena, strtoull(e, nullptr, 10));  // (testing: the overflow path)
            chk(S.d_text.reserve(arena + 16)); chk(S.d_touts.reserve(nb)); chk(S.d_tcur.reserve(1));
            chk(O.h_touts.reserve(nb)); chk(O.h_tcur.reserve(1));
            if (launch_err == hipSuccess) {
                int t_tx = vga_timer_begin(ctx, "poa_text", 0, st);
                chk(hipMemsetAsync(S.d_tcur.p, 0, sizeof(unsigned long long), st));
                hipLaunchKernelGGL(k_poa_text, dim3(nb), dim3(64), 0, st, nb, S.d_probs.p, S.d_outs.p, S.d_ops.p, S.d_orow.p, S.d_rows.p, S.d_ntab.p,
                                   (const char *)S.d_seq32.p, S.d_q.p, S.d_text.p, (uint32_t)arena, S.d_tcur.p, S.d_touts.p);
                chk(hipGetLastError());
                vga_timer_end(ctx, t_tx);
                chk(hipMemcpyAsync(O.h_touts.p, S.d_touts.p, nb * sizeof(poa_text_out), hipMemcpyDeviceToHost, st));
                chk(hipMemcpyAsync(O.h_tcur.p, S.d_tcur.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            }
        } else {
            chk(hipMemcpyAsync(O.h_ops.p, S.d_ops.p, tot_ops, hipMemcpyDeviceToHost, st));
            chk(hipMemcpyAsync(O.h_orow.p, S.d_orow.p, tot_ops * 4, hipMemcpyDeviceToHost, st));
            // (device store: the cs strings need the graph bases of the aligned rows -- the gathered node sequences come back too,
            // 17 KB per problem, instead of a handle lookup per aligned base)
            if (feed.dev) chk(hipMemcpyAsync(O.h_seq.p, S.d_seq32.p, tot_seq, hipMemcpyDeviceToHost, st));
        }
        return {i0, i1, raw_est, slot, oset, false, arena};
    };
    // host: CIGAR / cs / node path of one problem from the raw op stream (reverse order on the device)
    auto post_one = [&](const poa_slot::out_set &S, uint64_t i0, uint64_t i) {
        const uint32_t p = order[i];
        poa_item &it = out[p];
        const poa_out &ho = S.h_outs.p[i - i0];
        if (ho.status == POA_ST_POOL || ho.status == POA_ST_RETRY) return;  // re-run later
        it.ok = ho.status == POA_ST_OK ? 1 : 0;
        it.score = ho.score;
        it.n_cells = ho.cells;
        it.n_vcells = ho.vcells;
        if (!it.ok) return;
        const poa_prep &g = G[p];
        const poa_prob &pb = probs[p];
        if (S.text && S.h_touts.p[i - i0].flags == 1u) {
            // K4c wrote the fields (vga_poa_text.hpp): they are copied, not derived
            const poa_text_out &t = S.h_touts.p[i - i0];
            const uint32_t *runs = (const uint32_t *)(S.text_p + t.runs_off);
            if (feed.keep_text) {
                it.cs_p = S.text_p + t.cs_off; it.cs_n = t.cs_len;
                it.cigar_p = S.text_p + t.cg_off; it.cigar_n = t.cg_len;
                it.gnodes_p = runs; it.gnodes_n = t.n_runs;
                it.cs.clear(); it.cigar.clear(); it.gnodes.clear();
            } else {
                it.cs.assign(S.text_p + t.cs_off, t.cs_len);
                it.cigar.assign(S.text_p + t.cg_off, t.cg_len);
                it.gnodes.assign(runs, runs + t.n_runs);
            }
            it.rows.clear();
            it.deduped = true;
            it.n_path = t.n_path; it.start_off = t.start_off; it.end_off = t.end_off; it.aligned = t.aligned;
            return;
        }
        const uint8_t *po = S.h_ops.p + pb.ops0;
        const uint32_t *pr = S.h_orow.p + pb.ops0;
        const char *q = views[p].query;
        // base of graph row r: bases[r - 1] -- the node strings of a host graph, or (device store) the sub-batch's gathered
        // node sequences, which came back with the operations
        const char *bases = feed.dev ? S.h_seq.p + pb.seq0 : views[p].nodes + views[p].node_off[0];
        const uint32_t *frow = g.first_row_p;
        const size_t nv = (size_t)g.n_ntab - 1;
        const uint32_t nops = ho.nops;
        // one pass over the operations (stored sink -> source), writing through raw pointers into buffers of the largest
        // possible size: 3 characters per operation for cs ("*ac"), a run of one per operation for the CIGAR ("1M").  Those
        // are scratch of the worker thread; the problem keeps copies of the exact size (the largest possible size is four
        // times what a 10 kbp read uses: 2.4 GB of touched, unused capacity per 10 000 reads, which the process then carries
        // to its exit)
        struct scratch_t { std::vector<char> cs, cg; std::vector<uint32_t> rows; };
        static thread_local scratch_t sc;
        if (sc.cs.size() < 5 + 3 * (size_t)nops + 24) sc.cs.resize(5 + 3 * (size_t)nops + 24 + 4096);
        if (sc.cg.size() < 2 * (size_t)nops + 24) sc.cg.resize(2 * (size_t)nops + 24 + 4096);
        if (sc.rows.size() < nops) sc.rows.resize((size_t)nops + 1024);
        char *const cs0 = sc.cs.data(), *const cg0 = sc.cg.data();
        char *c = cs0, *d = cg0;
        uint32_t *rowp = sc.rows.data();
        memcpy(c, "cs:Z:", 5);
        c += 5;
        auto put_u = [](char *&w, uint64_t v) {
            char t[24];
            int k = 0;
            do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v);
            while (k) *w++ = t[--k];
        };
        uint64_t eq_run = 0, aligned = 0;
        uint32_t qi = 0, n_rows = 0;
        uint32_t t2 = nops;
        while (t2 > 0) {
            const uint8_t op = po[t2 - 1];
            uint32_t u = t2 - 1;
            while (u > 0 && po[u - 1] == op) u--;
            put_u(d, t2 - u);
            *d++ = op == 0 ? 'M' : (op == 1 ? 'I' : 'D');
            if (op != 0 && eq_run) { *c++ = ':'; put_u(c, eq_run); eq_run = 0; }
            if (op == 0) {
                for (uint32_t x = t2; x > u; x--) {
                    const uint32_t r = pr[x - 1];
                    const char gb = bases[r - 1], qb = q[qi++];
                    rowp[n_rows++] = r;
                    if (gb == qb) eq_run++;
                    else {
                        if (eq_run) { *c++ = ':'; put_u(c, eq_run); eq_run = 0; }
                        *c++ = '*'; *c++ = lower(gb); *c++ = lower(qb);
                    }
                }
                aligned += t2 - u;
            } else if (op == 1) {
                *c++ = '+';
                for (uint32_t x = t2; x > u; x--) *c++ = lower(q[qi++]);
            } else {
                *c++ = '-';
                for (uint32_t x = t2; x > u; x--) {
                    const uint32_t r = pr[x - 1];
                    rowp[n_rows++] = r;
                    *c++ = lower(bases[r - 1]);
                }
            }
            t2 = u;
        }
        if (eq_run) { *c++ = ':'; put_u(c, eq_run); }
        it.cs.assign(cs0, (size_t)(c - cs0));
        it.cigar.assign(cg0, (size_t)(d - cg0));
        it.rows.assign(rowp, rowp + n_rows);
        it.n_path = n_rows;
        it.deduped = false;
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
    struct seg_t { uint64_t first, second; bool general; bool arena; };
    std::vector<seg_t> todo;  // used as a stack of [begin, end) ranges of launch positions, front = back()
    todo.push_back({0, n, false, n_arenas != 0});
    std::vector<uint32_t> retry;  // problems a specialised DP kernel handed back (POA_ST_RETRY): re-run with the general one
    std::vector<uint32_t> too_big;  // problems chunk mode gave up on twice: classic mode once the chunk-mode launches are done
    std::vector<uint32_t> again;    // ... once: they run again in chunk mode when the others are through (the pool has grown, fewer compete)
    std::vector<uint8_t> gave_up(n, 0);
    std::vector<sub_t> inflight;
    bool slot_busy[POA_SLOTS] = {};
    uint64_t all_cells = 0, all_vcells = 0, all_rows = 0, all_q = 0, all_ops = 0;
    uint64_t text_bytes = 0;  // what came back over PCIe for the strings and paths: K4c's text, or the raw operations
    auto fill = [&]() {
        while ((int)inflight.size() < n_slots && !todo.empty() && !malformed && !dev_failed && launch_err == hipSuccess) {
            int slot = 0;
            while (slot_busy[slot]) slot++;
            auto &seg = todo.back();
            // arena launches use the whole pool, classic ones its per-slot segments: never both at a time
            if (!inflight.empty() && inflight.front().arena != seg.arena) break;
            in_flight_other = 0;
            for (const sub_t &o : inflight) in_flight_other += o.i1 - o.i0;
            sub_t sb = launch(seg.first, seg.second, slot, seg.general, seg.arena);
            sb.general = seg.general;
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
            if (tr.on) fprintf(stderr, "[vga-trace] poa: (at %.1f ms) the launch of [%llu, %llu) has finished\n",
                               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count(), (unsigned long long)cur.i0, (unsigned long long)cur.i1);
        }
        poa_slot::out_set &S = W.slot[cur.slot].outs[cur.oset];
        if (launch_err != hipSuccess) break;
        if (S.text) {
            // K4c, second half: the text the kernel wrote (its length is known now); problems that found the arena full fall back
            // to the operations, which are still on the device
            poa_slot &SL = W.slot[cur.slot];
            const uint64_t used = std::min<uint64_t>(S.h_tcur.p[0], SL.d_text.cap);
            bool overflow = false;
            for (uint64_t i = cur.i0; i < cur.i1; i++) overflow = overflow || S.h_touts.p[i - cur.i0].flags == 2u;
            hipError_t ce = hipSuccess;
            if (feed.keep_text) {
                // a buffer of the context's that holds the text: the smallest free one that fits, else the largest free one grows
                size_t pickb = W.text_free.size(), big = W.text_free.size();
                for (size_t k = 0; k < W.text_free.size(); k++) {
                    if (W.text_free[k]->cap >= used + 16 && (pickb == W.text_free.size() || W.text_free[k]->cap < W.text_free[pickb]->cap)) pickb = k;
                    if (big == W.text_free.size() || W.text_free[k]->cap > W.text_free[big]->cap) big = k;
                }
                if (pickb == W.text_free.size()) pickb = big;
                std::unique_ptr<vga_hbuf<char>> hb;
                if (pickb < W.text_free.size()) { hb = std::move(W.text_free[pickb]); W.text_free.erase(W.text_free.begin() + (long)pickb); }
                else hb.reset(new vga_hbuf<char>());
                ce = hb->reserve(used + 16);
                S.text_p = hb->p;
                W.text_live.push_back(std::move(hb));
            } else {
                ce = S.h_text.reserve(used + 16);
                S.text_p = S.h_text.p;
            }
            if (ce == hipSuccess && used) {
                void *hd = nullptr;
                if (!getenv("VGA_POA_TEXT_MEMCPY") && hipHostGetDevicePointer(&hd, S.text_p, 0) == hipSuccess && hd) {
                    const uint64_t n16 = (used + 15) / 16;  // (both buffers are 16-byte aligned and hold 16 bytes of slack)
                    hipLaunchKernelGGL(k_poa_text_to_host, dim3((unsigned)std::min<uint64_t>(256, (n16 + 255) / 256)), dim3(256), 0, sarr[cur.slot],
                                       (const uint4 *)SL.d_text.p, (uint4 *)hd, n16);
                    ce = hipGetLastError();
                } else {
                    (void)hipGetLastError();
                    ce = hipMemcpyAsync(S.text_p, SL.d_text.p, used, hipMemcpyDeviceToHost, sarr[cur.slot]);
                }
            }
            if (ce == hipSuccess && overflow) {
                if (tr.on) fprintf(stderr, "[vga-trace] poa:   the text arena was too small for some problems: their operations come back\n");
                ce = S.h_ops.reserve(S.tot_ops);
                if (ce == hipSuccess) ce = S.h_orow.reserve(S.tot_ops);
                if (ce == hipSuccess) ce = S.h_seq.reserve(S.tot_seq + 4);
                if (ce == hipSuccess) ce = hipMemcpyAsync(S.h_ops.p, SL.d_ops.p, S.tot_ops, hipMemcpyDeviceToHost, sarr[cur.slot]);
                if (ce == hipSuccess) ce = hipMemcpyAsync(S.h_orow.p, SL.d_orow.p, S.tot_ops * 4, hipMemcpyDeviceToHost, sarr[cur.slot]);
                if (ce == hipSuccess) ce = hipMemcpyAsync(S.h_seq.p, SL.d_seq32.p, S.tot_seq, hipMemcpyDeviceToHost, sarr[cur.slot]);
            }
            const auto t_tx0 = std::chrono::steady_clock::now();
            if (ce == hipSuccess) ce = hipStreamSynchronize(sarr[cur.slot]);
            if (ce != hipSuccess) { launch_err = ce; break; }
            if (tr.on) fprintf(stderr, "[vga-trace] poa:   text of the sub-batch: %.1f MB copied back in %.2f ms\n", (double)used / 1e6,
                               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tx0).count());
            text_bytes += used + (cur.i1 - cur.i0) * sizeof(poa_text_out);
            if (overflow) text_bytes += 5 * S.tot_ops + S.tot_seq;
        } else
            text_bytes += 5 * S.tot_ops + (feed.dev ? S.tot_seq : 0);
        bool pool_fail = false;
        for (uint64_t i = cur.i0; i < cur.i1; i++)
            if (S.h_outs.p[i - cur.i0].status == POA_ST_POOL) {
                if (cur.arena) {
                    if (gave_up[order[i]]++ == 0) again.push_back(order[i]);
                    else too_big.push_back(order[i]);
                } else pool_fail = true;
            }
        if (pool_fail) {
            slot_busy[cur.slot] = false;
            if (cur.i1 - cur.i0 == 1 && W.pool_scale >= 8.0) { rc_final = VGA_ERR_POOL; break; }
            W.pool_scale = std::min(16.0, W.pool_scale * 1.7);
            todo.push_back({cur.i0, cur.i1, cur.general, false});  // enqueue it again, in smaller pieces
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
            fprintf(stderr, "[vga-trace] poa: (at %.1f ms) sub-batch [%llu, %llu) done, pool %.1f GB, widest row %u columns, worst width / estimate %.3f; "
                            "DP %.1f ms, mean %.1f workgroups resident, on GPU clock %.3f .. %.3f s\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count(),
                    (unsigned long long)cur.i0, (unsigned long long)cur.i1, (double)W.h_next.p[cur.slot] / 1e9, mx, worst,
                    te > tb ? (double)(te - tb) / 1e5 : 0.0, te > tb ? (double)tsum / (double)(te - tb) : 0.0, (double)(tb % 100000000000ull) / 1e8,
                    (double)(te % 100000000000ull) / 1e8);
        }
        // problems the 16-bit kernel stopped (a score near the int16 range) run again with 32-bit words
        for (uint64_t i = cur.i0; i < cur.i1; i++)
            if (S.h_outs.p[i - cur.i0].status == POA_ST_RETRY) retry.push_back(order[i]);
        if (!retry.empty()) {
            // at once, beside the launches that are still to come: a pass of its own at the end would run for as long as its
            // longest problem has rows with most of the GPU idle (config 5, first build of k_poa_dp_t6: 45 ms of a 170 ms call)
            const uint64_t a = order.size();
            for (uint32_t p : retry) order.push_back(p);
            if (tr.on) fprintf(stderr, "[vga-trace] poa: %zu problems handed back by the specialised DP kernel: re-run with the general one\n", retry.size());
            retry.clear();
            todo.push_back({a, order.size(), true, n_arenas != 0});
        }
        if (todo.empty() && inflight.empty() && !again.empty()) {
            const uint64_t a = order.size();
            for (uint32_t p : again) order.push_back(p);
            if (tr.on) fprintf(stderr, "[vga-trace] poa: %zu problems gave up waiting for chunks: they run again\n", again.size());
            again.clear();
            todo.push_back({a, order.size(), cur.general, true});
        }
        if (todo.empty() && inflight.empty() && !too_big.empty()) {
            classic_need = 0;
            for (uint32_t p : too_big) classic_need += est[p];
            const uint64_t a = order.size();
            for (uint32_t p : too_big) order.push_back(p);
            if (tr.on) fprintf(stderr, "[vga-trace] poa: %zu problems gave up waiting for chunks (or need more contiguous state than a region holds): classic pass\n", too_big.size());
            too_big.clear();
            todo.push_back({a, order.size(), cur.general, false});
        }
        // refill the GPU first (the new sub-batch's results go to the slot's other result set), then post-process
        slot_busy[cur.slot] = false;
        fill();
        {
            const uint64_t a0 = cur.i0, cnt = cur.i1 - cur.i0;
            const auto t_post0 = std::chrono::steady_clock::now();
            parallel_for(cnt, [&](uint64_t t) { post_one(S, a0, a0 + t); });
            if (tr.on)
                fprintf(stderr, "[vga-trace] poa:   CIGAR / cs strings of sub-batch [%llu, %llu): %.1f ms on the host, %zu launches in flight meanwhile, at %.1f ms\n",
                        (unsigned long long)cur.i0, (unsigned long long)cur.i1, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_post0).count(),
                        inflight.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count());
            for (uint64_t i = cur.i0; i < cur.i1; i++) {
                const poa_out &ho = S.h_outs.p[i - cur.i0];
                if (ho.status == POA_ST_POOL || ho.status == POA_ST_RETRY) continue;
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
    if (n_arenas && CP.owner) {
        unsigned long long bad[4] = {0, 0, 0, 0};
        (void)hipMemcpy(bad, CP.stats + 4, sizeof bad, hipMemcpyDeviceToHost);
        if (bad[0] || bad[1])
            return vga_set_error(ctx, VGA_ERR_HIP, "chunk pool check: %llu chunks were handed out while somebody held them, %llu (+ %llu broken chains) came back from somebody else "
                                 "(the first: chunk %llu held by %llu, pushed by %llu, position %llu of its chain, %llu threads)",
                                 bad[0], bad[1] & 0xFFFFFFFFull, bad[1] >> 32, bad[2] & 0xFFFFFFFFull, bad[2] >> 32, bad[3] & 0xFFFFFFFFull, (bad[3] >> 32) & 0xFFFFull, bad[3] >> 48);
    }
    if (tr.on && n_arenas) {
        unsigned long long empties = 0;
        (void)hipMemcpy(&empties, W.d_head.p + POA_LISTS * POA_LIST_STRIDE, sizeof empties, hipMemcpyDeviceToHost);
        uint64_t ps; { std::lock_guard<std::mutex> lk(W.mu); ps = W.pool_size; }
        fprintf(stderr, "[vga-trace] poa: chunk pool %.1f GB in %zu segments (%u chunks listed); since the context began %llu requests found every free list empty\n",
                (double)ps / 1e9, W.segs_listed, W.chunks_listed, empties);
    }
    if (rc_final != VGA_OK)
        return vga_set_error(ctx, rc_final, "a single POA problem does not fit the %llu byte traceback pool",
                             (unsigned long long)W.pool_size);
    POA_CHECK(hipStreamSynchronize(st));
    vga_timers_collect(ctx);
    // byte model of the DP kernel: graph bases + query + 1 direction byte per cell
    // + the value rows kept in HBM (6 B per cell), written once and read back at least once
    for (auto &a : ctx->last_times) {
        // (the traceback's 6 bytes per alignment column belong to whichever kernel walked: the DP kernel when fused)
        if (a.name == "poa_band_dp") a.bytes = all_rows + all_q + all_cells + 12 * all_vcells + (any_fused ? 6 * all_ops : 0);
        if (a.name == "poa_traceback") a.bytes = any_fused ? 0 : 6 * all_ops;
    }
    tm.ms_dp = vga_timer_sum(ctx, "poa_band_dp");
    tm.ms_tb = vga_timer_sum(ctx, "poa_traceback");
    tm.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    tm.result_bytes = text_bytes;
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
    (void)hipSetDevice(ctx->device);
    vga_ctx_scope scope(ctx);
    vga_release_deferred(ctx);  // (buffers of this context that grew during an earlier call: freed now, while it has nothing in flight)
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

