// vga_align.hip -- chain -> subgraph -> POA problem -> one alignment record per read.
//
// Stands in for best_alignment_for_query / obtain_base_level_alignment (src/align.rs:34-145):
//   find_range_chain ............ src/align.rs:267-402   (u64::from(Handle) read as the node id)
//   extend_range_chain_2 ........ src/align.rs:523-665
//   find_nodes_edges_for_abpoa .. src/align.rs:670-724
//   create_align_safe ........... src/align.rs:202        -> vga_poa_batch (vga_poa.hip)
//   generate_alignment .......... src/align.rs:1096-1168  (fields only; the GAF text is host code)
// The subgraph extraction and the POA node tables are built on the GPU (vga_subgraph.hip: one wave per chain over the
// index's CSR arrays in HBM); the host only reduces each chain's anchors to their extremes -- the same pass that fixes
// the launch order.  VGA_SUBGRAPH=host selects the host-thread walk below instead (build_subgraph + poa_prepare), which
// the GPU tests use as a second opinion.  The ./subgraphs/*.gfa export side effect of align.rs:104-111 is a debugging
// aid and is not reproduced.
#include "vga_common.hpp"
#include "vga_poa_internal.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

namespace {

typedef uint32_t handle_t;  // (id << 1) | is_reverse

struct index_view {
    const vga_dev_index &ix;
    explicit index_view(const vga_dev_index &i) : ix(i) {}

    // get_bv_rank over the node-start bit vector (src/index.rs:427-439): starts <= pos
    uint32_t rank(uint32_t pos) const
    {
        return (uint32_t)(std::upper_bound(ix.node_start.begin(), ix.node_start.end(), pos) - ix.node_start.begin());
    }
    // handle_from_seqpos for a Forward position (src/index.rs:415-423)
    handle_t handle_from_fwd_pos(uint32_t pos) const { return rank(pos) * 2; }
    // get_bv_select (src/index.rs:461-480)
    uint32_t select(uint32_t element_no) const
    {
        if (element_no == 0 || element_no > ix.n_nodes + 1) return 0;
        return ix.node_start[element_no - 1];
    }
    uint32_t node_len(handle_t h) const
    {
        uint32_t id = h >> 1;
        return ix.node_start[id] - ix.node_start[id - 1];
    }
    // incoming_edges_from_handle / outgoing_edges_from_handle (src/index.rs:559-606)
    void incoming(handle_t h, std::vector<handle_t> &out) const
    {
        out.clear();
        uint32_t pos = (h >> 1) - 1;
        if (!(h & 1)) {
            uint32_t s = ix.edge_idx[pos], n = ix.edges_to[pos];
            for (uint32_t i = 0; i < n; i++) out.push_back(ix.edges[s + i]);
        } else {
            outgoing(h ^ 1, out);
            for (auto &x : out) x ^= 1;
            std::reverse(out.begin(), out.end());
        }
    }
    void outgoing(handle_t h, std::vector<handle_t> &out) const
    {
        out.clear();
        uint32_t pos = (h >> 1) - 1;
        if (!(h & 1)) {
            uint32_t s = ix.edge_idx[pos] + ix.edges_to[pos], e = ix.edge_idx[pos + 1];
            for (uint32_t i = s; i < e; i++) out.push_back(ix.edges[i]);
        } else {
            incoming(h ^ 1, out);
            for (auto &x : out) x ^= 1;
            std::reverse(out.begin(), out.end());
        }
    }
    // dna.rs:19-33
    static char complement(char c)
    {
        switch (c) {
        case 'a': return 't'; case 'c': return 'g'; case 't': return 'a'; case 'g': return 'c'; case 'u': return 'a';
        case 'A': return 'T'; case 'C': return 'G'; case 'T': return 'A'; case 'G': return 'C'; case 'U': return 'A';
        default: return 'N';
        }
    }
    // seq_from_handle (src/index.rs:503-533); the reverse strand is derived as dna.rs:19-33 does
    void append_seq(handle_t h, std::string &out) const
    {
        uint32_t id = h >> 1;
        uint32_t s = ix.node_start[id - 1], e = ix.node_start[id];
        if (!(h & 1)) out.append(ix.seq_fwd.data() + s, e - s);
        else
            for (uint32_t i = e; i-- > s;) out.push_back(complement(ix.seq_fwd[i]));
    }
};

struct subgraph_t {
    std::vector<handle_t> handles;  // sorted, deduplicated (src/align.rs:658-659)
    std::vector<uint64_t> node_off; // per node: offset into seqs (n+1)
    std::string seqs;
    std::vector<uint32_t> esrc, edst;
};

struct scratch_t {
    std::vector<uint32_t> best;      // per packed handle: largest remaining budget seen
    std::vector<handle_t> touched;
    std::vector<std::pair<uint32_t, handle_t>> cur, next;
    std::vector<handle_t> nb, lo, hi;
};

// one direction of src/align.rs:551-591 / 616-656.  The reference's walk keeps no visited set (its
// frontier grows exponentially on bubble chains); a handle ends up in the range iff it is reachable
// with a positive remaining budget, which is what the per-handle best-budget relaxation computes (a handle is
// re-expanded only when it is reached with a larger budget than before).
void extend_dir(const index_view &iv, handle_t from, uint32_t diff, bool incoming, std::vector<handle_t> &hs, scratch_t &sc)
{
    const vga_dev_index &ix = iv.ix;
    // neighbours of a handle in walk direction, straight from the edge lists when the handle is forward (it always is
    // with only_forward; the general accessor covers the rest)
    auto for_each_nb = [&](handle_t h, auto &&f) {
        if (!(h & 1)) {
            const uint32_t pos = (h >> 1) - 1;
            const uint32_t s = incoming ? ix.edge_idx[pos] : ix.edge_idx[pos] + ix.edges_to[pos];
            const uint32_t e = incoming ? ix.edge_idx[pos] + ix.edges_to[pos] : ix.edge_idx[pos + 1];
            for (uint32_t i = s; i < e; i++) f(ix.edges[i]);
        } else {
            if (incoming) iv.incoming(h, sc.nb); else iv.outgoing(h, sc.nb);
            for (handle_t x : sc.nb) f(x);
        }
    };
    sc.cur.clear();
    for_each_nb(from, [&](handle_t x) { sc.cur.emplace_back(diff, x); });
    while (!sc.cur.empty()) {
        sc.next.clear();
        for (auto &it : sc.cur) {
            const uint32_t left = it.first;
            const handle_t h = it.second;
            if (sc.best[h] >= left) continue;
            if (sc.best[h] == 0) { hs.push_back(h); sc.touched.push_back(h); }
            sc.best[h] = left;
            const uint32_t len = iv.node_len(h);
            if (len < left) {
                const uint32_t rem = left - len;
                for_each_nb(h, [&](handle_t x) { if (sc.best[x] < rem) sc.next.emplace_back(rem, x); });
            }
        }
        sc.cur.swap(sc.next);
    }
    for (handle_t h : sc.touched) sc.best[h] = 0;
    sc.touched.clear();
}

// find_range_chain + extend_range_chain_2 + find_nodes_edges_for_abpoa for one chain
static std::atomic<long long> g_ns_range{0}, g_ns_extend{0}, g_ns_seq{0}, g_ns_edges{0};  // VGA_TRACE: where the time goes
void build_subgraph(const index_view &iv, const vga_map_result *m, uint64_t read, uint64_t chain, uint32_t k, uint32_t qlen,
                    subgraph_t &sg, scratch_t &sc)
{
    auto tnow = []() { return std::chrono::steady_clock::now(); };
    auto t_a = tnow();
    const uint64_t a0 = m->anchor_off[read];
    const uint64_t c0 = m->chain_anchor_off[chain], c1 = m->chain_anchor_off[chain + 1];
    // smallest / largest handle over the anchors' begin and inclusive end positions (align.rs:286-308).  The position ->
    // handle map is monotonic, so it is enough to look the extreme positions up.
    uint32_t pmin = 0xFFFFFFFFu, pmax = 0;
    for (uint64_t t = c0; t < c1; t++) {
        const uint64_t ai = a0 + m->chain_anchor_idx[t];
        const uint32_t s = m->target_begin[ai], e = m->target_end[ai] - 1;  // get_end_seqpos_inclusive, chain.rs:65-70
        pmin = std::min(pmin, std::min(s, e));
        pmax = std::max(pmax, std::max(s, e));
    }
    const handle_t min_h = iv.handle_from_fwd_pos(pmin), max_h = iv.handle_from_fwd_pos(pmax);
    sg.handles.clear();
    for (uint32_t x = min_h >> 1; x <= (max_h >> 1); x++) sg.handles.push_back(x * 2);  // align.rs:358-364
    const handle_t first_handle = sg.handles.front(), last_handle = sg.handles.back();
    auto t_b = tnow();
    const uint64_t fa = a0 + m->chain_anchor_idx[c0], la = a0 + m->chain_anchor_idx[c1 - 1];
    // align.rs:536-547
    uint32_t prefix_diff = m->query_begin[fa];
    uint32_t start_prefix_on_node = m->target_begin[fa] - iv.select(first_handle >> 1);
    if (start_prefix_on_node < prefix_diff) prefix_diff -= start_prefix_on_node; else prefix_diff = 0;
    if (prefix_diff > 0) extend_dir(iv, first_handle, prefix_diff, true, sg.handles, sc);
    // align.rs:593-612
    uint32_t suffix_diff = qlen - (m->query_begin[la] + k);
    uint32_t end_suffix_on_node = iv.select((last_handle >> 1) + 1) - 1 - (m->target_end[la] - 1);
    if (end_suffix_on_node > suffix_diff) suffix_diff = 0; else suffix_diff -= end_suffix_on_node;
    if (suffix_diff > 0) extend_dir(iv, last_handle, suffix_diff, false, sg.handles, sc);
    // sort + dedup (align.rs:658-659).  The id range is sorted already and the walks only add forward handles, so the
    // result is  sorted(added below the range) + range + sorted(added above it);  anything else takes the general route.
    {
        const size_t n_range = (size_t)((last_handle - first_handle) / 2 + 1);
        bool simple = true;
        sc.lo.clear();
        sc.hi.clear();
        for (size_t i = n_range; i < sg.handles.size(); i++) {
            const handle_t h = sg.handles[i];
            if (h < first_handle) sc.lo.push_back(h);
            else if (h > last_handle) sc.hi.push_back(h);
            else if (h & 1) simple = false;  // a reverse handle inside the range (not reachable with only_forward)
        }
        if (simple) {
            std::sort(sc.lo.begin(), sc.lo.end());
            sc.lo.erase(std::unique(sc.lo.begin(), sc.lo.end()), sc.lo.end());
            std::sort(sc.hi.begin(), sc.hi.end());
            sc.hi.erase(std::unique(sc.hi.begin(), sc.hi.end()), sc.hi.end());
            sg.handles.resize(n_range);
            sg.handles.insert(sg.handles.begin(), sc.lo.begin(), sc.lo.end());
            sg.handles.insert(sg.handles.end(), sc.hi.begin(), sc.hi.end());
        } else {
            std::sort(sg.handles.begin(), sg.handles.end());
            sg.handles.erase(std::unique(sg.handles.begin(), sg.handles.end()), sg.handles.end());
        }
    }
    auto t_c = tnow();
    // align.rs:670-724
    sg.seqs.clear();
    sg.node_off.assign(1, 0);
    for (handle_t h : sg.handles) { iv.append_seq(h, sg.seqs); sg.node_off.push_back(sg.seqs.size()); }
    auto t_d = tnow();
    sg.esrc.clear();
    sg.edst.clear();
    // position of a handle in the sorted list: a dense map over the handles (sc.best is free between extensions),
    // stored as position + 1
    for (uint32_t i = 0; i < sg.handles.size(); i++) sc.best[sg.handles[i]] = i + 1;
    for (uint32_t i = 0; i < sg.handles.size(); i++) {
        iv.outgoing(sg.handles[i], sc.nb);
        for (handle_t t : sc.nb) {
            const uint32_t e1 = sc.best[t];
            if (e1 == 0) continue;  // the neighbour is not in the range
            if (i < e1 - 1) { sg.esrc.push_back(i); sg.edst.push_back(e1 - 1); }  // RangeOrient::Forward, align.rs:718
        }
    }
    for (handle_t h : sg.handles) sc.best[h] = 0;
    auto t_e = tnow();
    g_ns_range += std::chrono::duration_cast<std::chrono::nanoseconds>(t_b - t_a).count();
    g_ns_extend += std::chrono::duration_cast<std::chrono::nanoseconds>(t_c - t_b).count();
    g_ns_seq += std::chrono::duration_cast<std::chrono::nanoseconds>(t_d - t_c).count();
    g_ns_edges += std::chrono::duration_cast<std::chrono::nanoseconds>(t_e - t_d).count();
}

template <typename T>
T *amalloc(size_t n)
{
    return (T *)malloc((n ? n : 1) * sizeof(T));
}

}  // namespace

extern "C" void vga_align_result_free(vga_align_result *r)
{
    if (!r) return;
    free(r->aligned); free(r->path_off); free(r->path_handles); free(r->path_length); free(r->path_start);
    free(r->path_end); free(r->block_length); free(r->best_score); free(r->cigar_off); free(r->cigar);
    free(r->cs_off); free(r->cs);
    free(r);
}

static int vga_align_batch_impl(vga_batch *b, const vga_map_result *m, uint32_t align_best_n, const vga_poa_params *params,
                               vga_align_result **out)
{
    if (!b || !m || !params || !out || !b->ctx) return VGA_ERR_ARG;  // b->ctx == nullptr: the context was destroyed
    vga_ctx *ctx = b->ctx;
    *out = nullptr;
    if (!ctx->index.loaded) return vga_set_error(ctx, VGA_ERR_NO_INDEX, "vga_align_batch: no index uploaded");
    (void)hipSetDevice(ctx->device);
    vga_ctx_scope scope(ctx);
    vga_release_deferred(ctx);  // (buffers of this context that grew during an earlier call: freed now, while it has nothing in flight)
    if (m->n_reads != b->n_reads) return vga_set_error(ctx, VGA_ERR_ARG, "vga_align_batch: chains belong to a different batch");
    const uint64_t R = b->n_reads;
    const uint32_t k = ctx->index.k;
    auto t0 = std::chrono::steady_clock::now();
    vga_trace tr("align");

    // ---- which (read, chain) pairs become POA problems: first min(best_n, len) chains (align.rs:43-50)
    std::vector<uint64_t> prob_read, prob_chain;
    std::vector<uint64_t> read_prob0(R + 1, 0);
    for (uint64_t r = 0; r < R; r++) {
        read_prob0[r] = prob_read.size();
        uint64_t c0 = m->chain_off[r], c1 = m->chain_off[r + 1];
        uint64_t take = std::min<uint64_t>(align_best_n, c1 - c0);
        for (uint64_t c = c0; c < c0 + take; c++)
            if (!m->chain_placeholder[c]) { prob_read.push_back(r); prob_chain.push_back(c); }
    }
    read_prob0[R] = prob_read.size();
    const uint64_t n = prob_read.size();
    // slot_of[q]: where the q-th (read, chain) pair of the list above sits in the launch order (filled below)
    std::vector<uint32_t> slot_of(n);

    // ---- subgraphs: built by the host threads on request, one sub-batch ahead of the GPU (see poa_feed).  The launch
    // order is fixed up front from the span of each chain on the linearised graph.
    const char *sg_env = getenv("VGA_SUBGRAPH");
    const bool on_device = !(sg_env && strstr(sg_env, "host"));
    std::vector<subgraph_t> SG(on_device ? 0 : n);
    poa_feed feed;
    feed.views.resize(n);
    std::vector<double> proxy(n, 0.0);
    std::vector<float> west(n, 650.0f);  // the band-width term of the proxy (its floor: a graph about as long as the read)
    std::vector<uint8_t> klass(n, 0);  // 1: a very long problem (by its actual rows, known for the first part of the store), launched apart
    std::vector<uint32_t> launch_order(n);
    std::vector<sg_desc> descs(on_device ? n : 0);
    std::vector<uint64_t> q_src(on_device ? n : 0);
    std::atomic<int> has_reverse{0};
    // (rows: 24 000 and more -- config 3's longest problems have 21 000-22 000 rows and stay in the ordinary launches; config 4's
    // bubble-rich problems of 24 000-38 000 rows, whose band is as wide as the query on every row, do not)
    const uint32_t giant_rows = getenv("VGA_GIANT_ROWS") ? (uint32_t)atol(getenv("VGA_GIANT_ROWS")) : 24000u;
    vga_parallel_for(n, [&](uint64_t p) {
        const uint64_t r = prob_read[p], c = prob_chain[p];
        const uint64_t a0 = m->anchor_off[r];
        uint32_t lo = 0xFFFFFFFFu, hi = 0, pmin = 0xFFFFFFFFu, pmax = 0;
        for (uint64_t t = m->chain_anchor_off[c]; t < m->chain_anchor_off[c + 1]; t++) {
            const uint64_t ai = a0 + m->chain_anchor_idx[t];
            lo = std::min(lo, m->target_begin[ai]);
            hi = std::max(hi, m->target_end[ai]);
            if ((m->target_begin[ai] | m->target_end[ai]) >> 31) has_reverse = 1;  // (vga_map_params.only_forward = 0)
            // smallest / largest position over the anchors' begins and inclusive ends (align.rs:286-308; chain.rs:65-70)
            const uint32_t s = m->target_begin[ai], e = m->target_end[ai] - 1;
            pmin = std::min(pmin, std::min(s, e));
            pmax = std::max(pmax, std::max(s, e));
        }
        const uint32_t ql = (uint32_t)(b->read_off[r + 1] - b->read_off[r]);
        // footprint ~ rows x mean band width.  Rows: the chain's span on the linearised graph plus what the extension
        // adds for the part of the read the chain does not cover (it walks every allele, ~1.6 graph bases per read
        // base on DRB1-3123); the longest path is ~0.85 of the rows (width estimate).
        const uint64_t c0 = m->chain_anchor_off[c], c1 = m->chain_anchor_off[c + 1];
        const double q_first = c1 > c0 ? (double)m->query_begin[a0 + m->chain_anchor_idx[c0]] : 0.0;
        const double q_last = c1 > c0 ? (double)m->query_begin[a0 + m->chain_anchor_idx[c1 - 1]] + (double)k : (double)ql;
        const double uncovered = q_first + std::max(0.0, (double)ql - q_last);
        const double rows = (double)(hi > lo ? hi - lo : 0) + 1.6 * uncovered;
        west[p] = (float)(650.0 + 0.3 * std::max(0.0, 0.85 * rows - (double)ql));
        proxy[p] = rows * (double)west[p];
        // a chain whose span alone makes it a long problem goes to the front of the order whatever its width term: it must be in the
        // first part of the store, where its actual rows are known before the launch of the long problems starts -- config 4: 21
        // problems of 24 000-37 000 rows sat behind position 2 048, were launched as a second group of long problems 80 ms into
        // the call and, 1 024-thread workgroups that need a CU's 16 wave slots at once, only got their CUs when the bulk launch
        // beside them had nothing left to dispatch: they ended last, 512 ms after their launch
        if (rows >= 0.85 * (double)giant_rows) proxy[p] += 1e13;
        feed.views[p] = {nullptr, nullptr, 0, nullptr, nullptr, 0, b->reads.data() + b->read_off[r], ql};
        if (on_device) {
            const uint64_t fa = a0 + m->chain_anchor_idx[c0], la = a0 + m->chain_anchor_idx[c1 - 1];
            descs[p] = {pmin, pmax, m->query_begin[fa], m->target_begin[fa], m->query_begin[la], m->target_end[la], ql, 0u};
            q_src[p] = b->read_off[r];
        }
    });
    // Launch order = largest footprint first (poa_run's stable sort by proxy).  The device store is filled in that order,
    // in two parts: what the first launch takes before it, the rest beside it -- so the problems are permuted into launch
    // order here and poa_run's sort keeps them.
    for (uint64_t q = 0; q < n; q++) slot_of[q] = (uint32_t)q;
    if (on_device && n > 1) {
        std::vector<uint32_t> ord(n);
        for (uint64_t q = 0; q < n; q++) ord[q] = (uint32_t)q;
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return proxy[x] > proxy[y]; });
        auto permute = [&](auto &v) {
            auto w = v;
            for (uint64_t i = 0; i < n; i++) w[i] = v[ord[i]];
            v.swap(w);
        };
        permute(prob_read); permute(prob_chain); permute(proxy); permute(west); permute(feed.views); permute(descs); permute(q_src);
        for (uint64_t i = 0; i < n; i++) slot_of[ord[i]] = (uint32_t)i;
    }
    if (has_reverse)
        return vga_set_error(ctx, VGA_ERR_UNSUPPORTED,
                             "a chain to be aligned holds reverse-strand anchors: RangeOrient::Reverse / Both (src/align.rs:365-387) is not supported");
    feed.proxy = proxy.data();
    double sub_ms = 0;
    const unsigned n_thr = std::max(1u, vga_host_threads(n));
    std::vector<scratch_t> scratch(n_thr);
    if (!on_device)
        for (auto &sc : scratch) sc.best.assign((size_t)(ctx->index.n_nodes + 2) * 2, 0);
    sg_store store;
    // ... and so do problems with fewer rows whose band will be as wide as the query (a subgraph several times longer than the
    // read: the band spans what separates the path from the diagonal `qlen - remain`): in an ordinary launch every row of
    // theirs is wider than the LDS window and takes the HBM detour -- 24-27 us per row, 0.83-0.99 s for 34 000-37 000 rows on
    // config 4, as long as the 107 000-row problem takes in the launch of the long ones.  The product rows x expected width
    // (poa_run's estimate: from the longest source-sink path) decides.
    const double giant_cells = getenv("VGA_GIANT_CELLS") ? atof(getenv("VGA_GIANT_CELLS")) : 1.5e8;
    std::function<uint8_t(uint64_t)> is_giant = [&](uint64_t i) -> uint8_t {  // (function scope: feed.dev_rest calls it from inside poa_run)
        const sg_sum &sm = store.sum[i];
        if (sm.N >= giant_rows) return 1;
        const double ql = (double)feed.views[i].qlen;
        const double w = params->wb < 0 ? ql : (double)params->wb + (double)(uint64_t)(params->wf * ql);
        const double ew = std::min(ql + 1.0, 2.0 * w + 431.0 + 0.3 * std::abs((double)sm.longest - ql));
        return (double)sm.N * ew >= giant_cells ? 1 : 0;
    };
    index_view iv_all(ctx->index);
    if (on_device) {
        vga_timers_reset(ctx);
        feed.keep_timers = true;
        auto ta = std::chrono::steady_clock::now();
        // the first launch takes 2 048 problems (poa_run, arena mode); small calls are prepared in one go -- and so are calls of
        // narrow-band problems (rows ~ read length: the width estimate below stays at its floor; config 5), whose DP launches are
        // short and whose subgraphs are cheap: with the second part beside the first launches on a throttled stream, those
        // launches waited 2 x 160 ms for it in the command line tool (25 000 reads; the chains' path text is written on the GPU at
        // the same time), against 13 ms for all of it up front
        uint64_t split = n > 3072 ? 2048 : n;
        {
            double wsum = 0;
            for (uint64_t i = 0; i < n; i++) wsum += west[i];
            // (config 5: 650-740 from chunk to chunk; config 3: 1 500-2 500; config 4: in between and above)
            if (n && wsum / (double)n <= 900.0) split = n;
            if (getenv("VGA_TRACE") && atoi(getenv("VGA_TRACE")) != 0)
                fprintf(stderr, "[vga-trace] align: mean width term of the launch-order proxy %.0f: the subgraph store is built in %s\n", n ? wsum / (double)n : 0.0,
                        split == n ? "one part" : "two parts");
        }
        if (const char *e = getenv("VGA_SG_SPLIT")) { const long v = atol(e); split = v <= 0 ? n : std::min<uint64_t>(n, (uint64_t)v); }  // (0: one part)
        const int rc = sg_prepare(ctx, descs.data(), q_src.data(), n, split, b->d_reads, params->remain_rule, store);
        if (rc != VGA_OK) return rc;
        sub_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count();
        feed.dev = &store;
        feed.want_rows = false;  // (this caller reads the deduplicated path, its length and the strings: k_poa_text may write them)
        feed.keep_text = true;   // ... and copies each string once, from where it came back (poa_item::cs_p) into the result below
        // The rows of the first part's problems are known now.  Very long ones (config 3's longest have 21 000 rows; a chain
        // that spans 100 kbp of the linearisation has 110 000, all sequential) decide how long the call takes: they go
        // first, in a launch of their own with 512 threads and an 8 192-column window (poa_run).  The order inside the
        // first part is free -- the store is addressed by problem index.
        for (uint64_t i = 0; i < n; i++) launch_order[i] = (uint32_t)i;
        for (uint64_t i = 0; i < store.split; i++) klass[i] = is_giant(i);
        std::stable_sort(launch_order.begin(), launch_order.begin() + (long)store.split, [&](uint32_t x, uint32_t y) { return klass[x] > klass[y]; });
        feed.order = launch_order.data();
        feed.klass = klass.data();
        feed.dev_rest = [&]() -> int {
            auto tb = std::chrono::steady_clock::now();
            const int rc2 = sg_prepare_rest(ctx, store);
            sub_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
            if (rc2 == VGA_OK) {  // none of the second part's problems has been staged yet: its very long ones move to its front
                for (uint64_t i = store.split; i < n; i++) klass[i] = is_giant(i);
                std::stable_sort(launch_order.begin() + (long)store.split, launch_order.end(), [&](uint32_t x, uint32_t y) { return klass[x] > klass[y]; });
            }
            return rc2;
        };
        tr.mark("subgraphs on the GPU");
    }
    if (!on_device) feed.prepare = [&](const uint32_t *ids, uint64_t cnt) {
        auto ta = std::chrono::steady_clock::now();
        const unsigned nt = std::min<uint64_t>(n_thr, cnt);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                index_view iv(ctx->index);
                scratch_t &sc = scratch[t];
                for (uint64_t q = t; q < cnt; q += nt) {
                    const uint32_t p = ids[q];
                    const uint64_t r = prob_read[p];
                    build_subgraph(iv, m, r, prob_chain[p], k, (uint32_t)(b->read_off[r + 1] - b->read_off[r]), SG[p], sc);
                    poa_view &v = feed.views[p];
                    v.node_off = SG[p].node_off.data(); v.nodes = SG[p].seqs.data(); v.n_nodes = SG[p].handles.size();
                    v.esrc = SG[p].esrc.data(); v.edst = SG[p].edst.data(); v.n_edges = SG[p].esrc.size();
                }
            });
        for (auto &x : th) x.join();
        sub_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count();
    };
    tr.mark("launch order");

    std::vector<poa_item> items;
    poa_timing tm;
    if (n > 0) {
        int rc = poa_run(ctx, feed, params, items, tm);
        if (rc != VGA_OK) return rc;
    }
    tr.mark("poa_run");

    // ---- per read: keep the candidate with the longest path (stable, align.rs:52-54)
    vga_align_result *res = (vga_align_result *)calloc(1, sizeof(vga_align_result));
    if (!res) return vga_set_error(ctx, VGA_ERR_NOMEM, "out of host memory (align result)");
    auto nomem = [&]() { vga_align_result_free(res); return vga_set_error(ctx, VGA_ERR_NOMEM, "out of host memory (align result of %llu reads)", (unsigned long long)R); };
    res->n_reads = R;
    res->aligned = amalloc<uint8_t>(R);
    res->path_off = amalloc<uint64_t>(R + 1);
    res->path_length = amalloc<uint32_t>(R);
    res->path_start = amalloc<uint32_t>(R);
    res->path_end = amalloc<uint32_t>(R);
    res->block_length = amalloc<uint32_t>(R);
    res->best_score = amalloc<int32_t>(R);
    res->cigar_off = amalloc<uint64_t>(R + 1);
    res->cs_off = amalloc<uint64_t>(R + 1);
    if (!res->aligned || !res->path_off || !res->path_length || !res->path_start || !res->path_end || !res->block_length || !res->best_score ||
        !res->cigar_off || !res->cs_off)
        return nomem();
    std::vector<int64_t> pick(R, -1);
    std::vector<uint32_t> path_n(R, 0);
    vga_parallel_for(R, [&](uint64_t r) {
        int64_t best = -1;
        for (uint64_t q = read_prob0[r]; q < read_prob0[r + 1]; q++) {
            const uint64_t p = slot_of[q];
            if (!items[p].ok) continue;
            if (best < 0 || items[p].n_path > items[best].n_path) best = (int64_t)p;
        }
        pick[r] = best;
        if (best >= 0) {
            // graph_nodes.dedup() (align.rs:1114): count the runs
            const poa_item &ib = items[best];
            const uint32_t *gn = ib.gnodes_p ? ib.gnodes_p : ib.gnodes.data();
            const size_t gnn = ib.gnodes_p ? ib.gnodes_n : ib.gnodes.size();
            uint32_t c = 0;
            for (size_t t = 0; t < gnn; t++) c += (t == 0 || gn[t] != gn[t - 1]);
            path_n[r] = c;
        }
    });
    uint64_t tp = 0, tc = 0, ts = 0;
    for (uint64_t r = 0; r < R; r++) {
        res->path_off[r] = tp; res->cigar_off[r] = tc; res->cs_off[r] = ts;
        tp += path_n[r];
        tc += pick[r] >= 0 ? (items[pick[r]].cs_p ? items[pick[r]].cigar_n : items[pick[r]].cigar.size()) + 1 : 1;
        ts += pick[r] >= 0 ? (items[pick[r]].cs_p ? items[pick[r]].cs_n : items[pick[r]].cs.size()) + 1 : 1;
    }
    res->path_off[R] = tp; res->cigar_off[R] = tc; res->cs_off[R] = ts;
    res->path_handles = amalloc<uint64_t>(tp);
    res->cigar = amalloc<char>(tc);
    res->cs = amalloc<char>(ts);
    if (!res->path_handles || !res->cigar || !res->cs) return nomem();
    vga_parallel_for(R, [&](uint64_t r) {
        res->aligned[r] = pick[r] >= 0;
        res->path_length[r] = res->path_start[r] = res->path_end[r] = res->block_length[r] = 0;
        res->best_score[r] = 0;
        if (pick[r] < 0) {
            res->cigar[res->cigar_off[r]] = 0;
            res->cs[res->cs_off[r]] = 0;
            return;
        }
        const uint64_t p = (uint64_t)pick[r];
        const poa_item &it = items[p];
        uint64_t o = res->path_off[r];
        const uint32_t *gn = it.gnodes_p ? it.gnodes_p : it.gnodes.data();
        const size_t gnn = it.gnodes_p ? it.gnodes_n : it.gnodes.size();
        for (size_t t = 0; t < gnn; t++)
            if (t == 0 || gn[t] != gn[t - 1])  // align.rs:1120-1123
                res->path_handles[o++] = on_device ? store.of(p).h_handles[store.off[p].node0 + gn[t]] : SG[p].handles[gn[t]];
        res->path_length[r] = it.n_path;
        res->path_start[r] = it.start_off;
        res->path_end[r] = it.end_off;
        res->block_length[r] = it.aligned;
        res->best_score[r] = it.score;
        if (it.cs_p) {  // (device text, where it came back: the one copy of the strings on the host)
            memcpy(res->cigar + res->cigar_off[r], it.cigar_p, it.cigar_n); res->cigar[res->cigar_off[r] + it.cigar_n] = 0;
            memcpy(res->cs + res->cs_off[r], it.cs_p, it.cs_n); res->cs[res->cs_off[r] + it.cs_n] = 0;
        } else {
            memcpy(res->cigar + res->cigar_off[r], it.cigar.c_str(), it.cigar.size() + 1);
            memcpy(res->cs + res->cs_off[r], it.cs.c_str(), it.cs.size() + 1);
        }
    });
    res->poa_problems = n;
    for (uint64_t p = 0; p < n; p++) {
        res->poa_rows += items[p].n_rows; res->poa_cells += items[p].n_cells; res->poa_value_cells += items[p].n_vcells;
    }
    res->ms_subgraph = (float)sub_ms;  // GPU kernels before the first DP launch (VGA_SUBGRAPH=host: host threads, overlapped with the GPU)
    res->ms_dp = tm.ms_dp;
    res->ms_traceback = tm.ms_tb;
    res->result_bytes = tm.result_bytes;
    res->ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    tr.mark("assemble records");
    if (tr.on && !on_device) {
        fprintf(stderr, "[vga-trace] align: subgraph thread time: range %.1f ms, extension %.1f ms, node strings %.1f ms, edges %.1f ms\n",
                g_ns_range.exchange(0) / 1e6, g_ns_extend.exchange(0) / 1e6, g_ns_seq.exchange(0) / 1e6, g_ns_edges.exchange(0) / 1e6);
    }
    *out = res;
    return VGA_OK;
}

extern "C" int vga_align_batch(vga_batch *b, const vga_map_result *m, uint32_t align_best_n, const vga_poa_params *params,
                               vga_align_result **out)
{
    // nothing throws across the C ABI: an allocation failure inside becomes VGA_ERR_NOMEM
    try {
        return vga_align_batch_impl(b, m, align_best_n, params, out);
    } catch (const std::bad_alloc &) {
        return vga_set_error((b ? b->ctx : nullptr), VGA_ERR_NOMEM, "vga_align_batch: out of host memory");
    } catch (const std::exception &e) {
        return vga_set_error((b ? b->ctx : nullptr), VGA_ERR_ARG, "vga_align_batch: %s", e.what());
    }
}

