// vga_subgraph.hpp -- device-side chain -> subgraph -> POA node table (vga_subgraph.hip), as vga_align.hip and poa_run see it.
//
// Replaces, on the GPU, the host walk of find_range_chain / extend_range_chain_2 / find_nodes_edges_for_abpoa
// (src/align.rs:267-402, 523-665, 670-724) and the node-table construction poa_prepare does for host graphs.
#pragma once

#include <cstdint>

#include "vga_common.hpp"

// one chain, as the host hands it over (the extremes of its anchors; the reduction over the anchors is the one the launch
// order needs anyway)
struct sg_desc {
    uint32_t pmin, pmax;        // smallest / largest forward position among the anchors' begins and inclusive ends
    uint32_t q_first, t_first;  // first anchor: query_begin, target_begin
    uint32_t q_last, te_last;   // last anchor: query_begin, target_end (exclusive)
    uint32_t qlen, pad;
};

// what the kernels report per problem
struct sg_sum {
    uint32_t n_nodes;  // handles of the subgraph
    uint32_t N;        // rows = graph bases
    uint32_t n_preds, n_sinks;
    uint32_t wlo, whi;  // words of the handle bitmap that hold set bits
    uint32_t longest;   // `remain` of the virtual source: graph bases on the source-sink path the remain rule follows
    uint32_t life;      // largest edge span (in nodes) among the nodes that use the value-row ring
    uint32_t flags;     // bit 0: malformed for the POA kernels (in-degree > 255, too many rows)
    uint32_t pad[3];
};

// where a problem's pieces live in the store
struct sg_off {
    uint64_t node0;  // handles / first_row: node0 .. node0 + n_nodes;  node table: node0 + problem index (one source entry each)
    uint64_t pred0, sink0, seq0;
    uint64_t q_src;  // first base of the query in the batch's device copy of the reads
};

// The prepared problems of one vga_align_batch call (device store + what the host needs of it).  The problems come in
// launch order and are prepared in two parts: [0, split) before the first DP launch, [split, n) beside it on a stream
// of its own (sg_prepare_rest), so that only the first part's kernels sit in front of the DP.
struct sg_part {
    uint64_t p0 = 0, p1 = 0;
    const uint4 *d_ntab = nullptr;
    const uint32_t *d_preds = nullptr, *d_sinks = nullptr;
    const char *d_seq = nullptr;
    const uint32_t *h_handles = nullptr, *h_first_row = nullptr;  // host copies (pinned), indexed from off[p].node0
    bool ready = false;
};
struct sg_store {
    uint64_t n = 0, split = 0;
    int remain_rule = 0;            // vga_poa_params.remain_rule the node tables are built for
    const sg_sum *sum = nullptr;    // host, n entries (valid for a part once it is ready)
    const sg_off *off = nullptr;    // host, n entries, offsets inside the problem's part
    const sg_off *d_off = nullptr;  // device copy
    const char *d_reads = nullptr;
    sg_part part[2];
    const sg_part &of(uint64_t p) const { return part[p >= split ? 1 : 0]; }
};

// Runs the extraction for chains [0, split) of n on ctx->stream and waits for it.  `store` points into the context's
// workspace and stays valid until the next call.  Returns VGA_OK or a negative VGA_ERR_*.
int sg_prepare(vga_ctx *ctx, const sg_desc *descs, const uint64_t *q_src, uint64_t n, uint64_t split, const char *d_reads, int remain_rule,
               sg_store &store);
// ... and for chains [split, n), on the workspace's side stream (it may run beside DP launches); waits for it.
int sg_prepare_rest(vga_ctx *ctx, sg_store &store);
