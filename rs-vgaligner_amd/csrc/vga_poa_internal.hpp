// vga_poa_internal.hpp -- the POA engine as the rest of the library sees it (vga_poa_batch and vga_align_batch
// are both thin packers around poa_run).
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "vga_common.hpp"
#include "vga_subgraph.hpp"

// one create_align_safe(nodes, edges, query, Global) problem, by reference
struct poa_view {
    const uint64_t *node_off;  // n_nodes + 1 offsets into `nodes`
    const char *nodes;
    uint64_t n_nodes;
    const uint32_t *esrc, *edst;  // 0-based node indices, src < dst
    uint64_t n_edges;
    const char *query;
    uint32_t qlen;
};

struct poa_item {
    uint8_t ok = 0;
    int32_t score = 0;
    uint32_t start_off = 0, end_off = 0, aligned = 0;
    uint64_t n_rows = 0, n_cells = 0, n_vcells = 0;
    uint32_t n_path = 0;           // abpoa_nodes.len(): graph-consuming alignment columns
    bool deduped = false;          // the fields came from the device (k_poa_text): `rows` is empty and `gnodes` holds every node the
                                   // alignment enters once, in order (graph_nodes.dedup(), align.rs:1114) -- vga_align_batch's input
    std::vector<uint32_t> rows;    // AbpoaAlignmentResult.abpoa_nodes: 1-based base-row id per graph-consuming column
    std::vector<uint32_t> gnodes;  // AbpoaAlignmentResult.graph_nodes
    std::string cigar, cs;
    // (device text, poa_feed::keep_text) views into the launch's text as it came back -- valid until the context's next POA call;
    // when cs_p is set, cs / cigar / gnodes above are empty
    const char *cs_p = nullptr, *cigar_p = nullptr;
    const uint32_t *gnodes_p = nullptr;
    uint32_t cs_n = 0, cigar_n = 0, gnodes_n = 0;
};

struct poa_timing {
    float ms_dp = 0, ms_tb = 0, ms_total = 0;
    uint64_t result_bytes = 0;  // bytes copied back for strings and node paths (K4c's text, or the raw operations)
};

// How poa_run obtains its problems.  `views` has one entry per problem with query / qlen valid.  When `prepare` is
// set the graph part of a view (nodes, edges) is filled on request, for the listed problems, shortly before they are
// staged -- poa_run calls it while earlier sub-batches are on the GPU -- and `proxy` (one value per problem, larger
// = bigger) fixes the launch order up front.  Without `prepare` every view is complete and the order is by footprint.
// With `dev` set the graphs never exist on the host: the node tables, predecessor lists, sinks and bases of all problems
// already sit in the device store sg_prepare filled (vga_subgraph.hip) and are gathered into a sub-batch's buffers device to
// device; the queries come from the batch's device copy of the reads (the gathered node sequences come back with a sub-batch's
// results for the cs strings).
struct poa_feed {
    std::vector<poa_view> views;
    std::function<void(const uint32_t *ids, uint64_t cnt)> prepare;
    const double *proxy = nullptr;
    const sg_store *dev = nullptr;
    bool keep_text = false;  // device text (K4c) stays where it came back, one pinned buffer per launch, and the items point into it
                             // (poa_item::cs_p ...): the caller copies each string once, into its result.  The buffers belong to the
                             // context and are recycled by its next POA call
    std::function<int()> dev_rest;  // prepares the store's second part (problems >= dev->split); called once, when they are first needed
    // optional: the launch order itself (n problem indices; poa_run then does not sort by proxy) and a class per problem
    // (1 = very long: such problems are launched apart, with the largest workgroup and window)
    const uint32_t *order = nullptr;
    const uint8_t *klass = nullptr;
    bool want_rows = true;    // false (vga_align_batch): cs, CIGAR and the deduplicated node path may come from the device (k_poa_text)
    bool keep_timers = false;  // the caller has reset the context's kernel timers and recorded some of its own
};

// Runs every problem on the GPU (sub-batched to fit the pool).  Returns VGA_OK or a negative VGA_ERR_*.
int poa_run(vga_ctx *ctx, poa_feed &feed, const vga_poa_params *params, std::vector<poa_item> &out, poa_timing &tm);

