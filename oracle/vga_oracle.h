/*
 * vga_oracle.h -- CPU ORACLE for the map -> chain -> align path of AlgoLab/rs-vgaligner.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (rs-vgaligner_amd/)
 * never links, imports or calls anything in oracle/.
 *
 * It is a single-threaded plain-C restatement of the reference's algorithm; every
 * function cites the reference file:line it follows (paths relative to the reference
 * checkout).  The Rust reference cannot be compiled in this environment (no cargo /
 * rustc, three git dependencies), so the restatement is pinned by the reference's own
 * known-answer unit tests (tests/test_oracle_reference_vectors.py lists each one).
 *
 * PARITY STATUS
 *   - index / k-mer table / anchors / rank-select / edges: pinned by reference tests.
 *   - chaining f(i), predecessors, chain membership: restated from src/chain.rs; the
 *     reference has no test asserting these values (only score_anchor's rejection rule).
 *   - POA (og_poa_align): PARITY UNPINNED.  The reference calls abPOA through the
 *     un-pinned git crate ab_poa (Cargo.toml:38, Cargo.lock:5-15; call site
 *     src/align.rs:202) whose source is not in the tree and no reference test ever
 *     calls it.  og_poa_align restates abPOA's *published* algorithm (global mode,
 *     convex gap 4/2 + 24/1, match 2, mismatch 4, adaptive band b=10 f=0.01); its
 *     tie-breaking and result encoding are this repo's specification.
 */
#ifndef VGA_ORACLE_H
#define VGA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ---- */
#define OG_OK 0
#define OG_ERR_IO (-1)
#define OG_ERR_PARSE (-2)
#define OG_ERR_NODE_IDS (-3)   /* ids are not exactly 1..n (index.rs:489-491 assumes it) */
#define OG_ERR_NOT_DNA (-4)    /* dna.rs:9-12 panics on it */
#define OG_ERR_NO_KMERS (-5)   /* kmer.rs:828 unwrap() on empty list panics */
#define OG_ERR_ARG (-6)
#define OG_ERR_NOMEM (-7)

/* handle = (node_id << 1) | is_reverse, as handlegraph::handle::Handle */
typedef uint64_t og_handle;
#define OG_H_ID(h) ((h) >> 1)
#define OG_H_REV(h) ((int)((h)&1))
#define OG_H_PACK(id, rev) ((((uint64_t)(id)) << 1) | (uint64_t)((rev) ? 1 : 0))
#define OG_H_FLIP(h) ((h) ^ 1)

#define OG_FORWARD 0
#define OG_REVERSE 1

/* ---------------- graph (stand-in for handlegraph::hashgraph::HashGraph) ---------------- */
typedef struct og_graph og_graph;

og_graph *og_graph_new(void);
void og_graph_free(og_graph *g);
/* HashGraph::create_handle */
int og_graph_add_node(og_graph *g, uint64_t id, const char *seq, size_t len);
/* HashGraph::create_edge(Edge(left,right)) -- insertion order defines neighbour order */
int og_graph_add_edge(og_graph *g, og_handle left, og_handle right);
int og_graph_add_path(og_graph *g, const char *name, const og_handle *steps, size_t n);
/* GFAParser::parse_file + HashGraph::from_gfa: all S, then all L, then all P, each in file order */
int og_graph_load_gfa(const char *path, og_graph **out);
size_t og_graph_n_nodes(const og_graph *g);
size_t og_graph_n_paths(const og_graph *g);
const char *og_graph_path_name(const og_graph *g, size_t i);
size_t og_graph_path_len(const og_graph *g, size_t i);
const og_handle *og_graph_path_steps(const og_graph *g, size_t i);
/* graph.sequence(handle): reverse handles give the reverse complement; out must hold node length */
size_t og_graph_node_len(const og_graph *g, uint64_t id);
size_t og_graph_sequence(const og_graph *g, og_handle h, char *out);
/* graph.handle_edges_iter(h, Left|Right); returns count, writes up to cap handles */
size_t og_graph_neighbors(const og_graph *g, og_handle h, int go_left, og_handle *out, size_t cap);

/* ---------------- index (src/index.rs, src/kmer.rs, src/utils.rs) ---------------- */
typedef struct {
    uint64_t seq_idx, edge_idx, edges_to_node;
} og_noderef; /* utils.rs:15-22 */

typedef struct {
    uint8_t orient; /* OG_FORWARD / OG_REVERSE */
    uint64_t position;
} og_seqpos; /* kmer.rs:27-31 */

typedef struct {
    og_seqpos start, end;
} og_kmerpos; /* kmer.rs:733-738; delimiter = (Reverse,u64::MAX)x2, kmer.rs:740-749 */

typedef struct {
    /* kmer.rs:48-65 */
    const char *seq; /* k bytes, not NUL terminated */
    og_seqpos begin_offset, end_offset;
    og_handle first_handle, last_handle;
    int handle_orient;
    uint64_t forks;
} og_graphkmer_view;

typedef struct og_index og_index;

/* Index::build (index.rs:109-281) without the .idx side effect; sampling_rate is None */
int og_index_build(const og_graph *g, uint64_t k, uint64_t max_furcations, uint64_t max_degree,
                   og_index **out);
void og_index_free(og_index *ix);

uint64_t og_index_k(const og_index *ix);
uint64_t og_index_seq_length(const og_index *ix);
uint64_t og_index_n_nodes(const og_index *ix);
uint64_t og_index_n_edges(const og_index *ix);
const char *og_index_seq_fwd(const og_index *ix);
const char *og_index_seq_rev(const og_index *ix);
const uint8_t *og_index_seq_bv(const og_index *ix); /* seq_length+1 bytes of 0/1 */
const og_handle *og_index_edges(const og_index *ix);
const og_noderef *og_index_node_ref(const og_index *ix); /* n_nodes+1 entries */
uint64_t og_index_n_kmers(const og_index *ix);
uint64_t og_index_n_kmer_pos(const og_index *ix);       /* table length incl. delimiters */
const char *og_index_kmer_keys(const og_index *ix);     /* n_kmers * k bytes, sorted */
const uint64_t *og_index_kmer_starts(const og_index *ix); /* offsets into the table */
const og_kmerpos *og_index_kmer_pos_table(const og_index *ix);

/* generate_kmers_parallel output (kmer.rs:277-304), for the k-mer count vectors */
uint64_t og_index_n_graph_kmers(const og_index *ix);
int og_index_graph_kmer(const og_index *ix, uint64_t i, og_graphkmer_view *out);
/* kmer.rs:93-273 generate_kmers (tests only): returns the number of k-mers after sort+dedup */
int64_t og_generate_kmers_count(const og_graph *g, uint64_t k, uint64_t edge_max, uint64_t degree_max);

/* index.rs:353-382; returns the number of positions, *out points into the table */
/* cost mode of bench.py's cpu_baseline_faithful leg (see og_index.c); results are identical either way */
void og_set_reference_faithful_costs(int on);
size_t og_index_find_positions(const og_index *ix, const char *kmer, size_t kmer_len,
                               const og_kmerpos **out);
/* index.rs:427-480, 388-423 */
uint64_t og_index_bv_rank(const og_index *ix, uint64_t pos);
uint64_t og_index_bv_inverse_rank(const og_index *ix, uint64_t pos);
uint64_t og_index_bv_select(const og_index *ix, uint64_t element_no);
uint64_t og_index_node_id_from_seqpos(const og_index *ix, og_seqpos p);
og_handle og_index_handle_from_seqpos(const og_index *ix, og_seqpos p);
/* index.rs:503-533; returns length, writes into out (cap >= node length) */
size_t og_index_seq_from_handle(const og_index *ix, og_handle h, char *out, size_t cap);
/* index.rs:536-606; return counts */
size_t og_index_edges_from_handle(const og_index *ix, og_handle h, og_handle *out, size_t cap);
size_t og_index_incoming_edges(const og_index *ix, og_handle h, og_handle *out, size_t cap);
size_t og_index_outgoing_edges(const og_index *ix, og_handle h, og_handle *out, size_t cap);

/* ---------------- anchors + chaining (src/chain.rs) ---------------- */
typedef struct {
    uint64_t id;
    uint64_t query_begin, query_end;
    og_seqpos target_begin, target_end;
    double max_chain_score;
    int64_t best_predecessor_id; /* -1 == None */
} og_anchor; /* chain.rs:29-44 */

/* io.rs:41-56 + chain.rs:134-173; *out is malloc'd (free with og_free) */
size_t og_anchors_for_query(const og_index *ix, const char *query, size_t qlen, int only_forward,
                            og_anchor **out);
/* chain.rs:274-368 */
double og_score_anchor(const og_anchor *a, const og_anchor *b, uint64_t seed_length, uint64_t max_gap);

typedef struct {
    og_anchor *anchors; /* ascending order (chain.rs:546) */
    size_t n;
    int is_placeholder;
} og_chain;

typedef struct {
    og_chain *chains;
    size_t n;
    double curr_max;
} og_chain_set;

/* chain.rs:370-655.  anchors is sorted in place and left in its post-backtracking state.
 * If dp_snapshot != NULL it receives a copy of the n anchors after STEP 1 (sorted, f(i) and
 * predecessor ids set, before backtracking consumes predecessors). */
int og_chain_anchors(og_anchor *anchors, size_t n, uint64_t seed_length, uint64_t bandwidth,
                     uint64_t max_gap, uint64_t chain_min_n_anchors, og_chain_set *out,
                     og_anchor *dp_snapshot);
void og_chain_set_free(og_chain_set *cs);

/* ---------------- subgraph extraction (src/align.rs:267-402, 523-665, 670-724) ---------------- */
#define OG_RANGE_FORWARD 0
#define OG_RANGE_REVERSE 1
#define OG_RANGE_BOTH 2
typedef struct {
    int orient;
    og_handle *handles;
    size_t n;
} og_range;
void og_range_free(og_range *r);

int og_find_range_chain(const og_index *ix, const og_chain *chain, og_range *out);
int og_extend_range_chain_2(const og_index *ix, const og_chain *chain, uint64_t query_len,
                            const og_range *old_range, og_range *out);

typedef struct {
    size_t n_nodes;
    char **seqs;
    size_t *seq_lens;
    size_t n_edges;
    size_t *edge_src, *edge_dst;
} og_subgraph;
void og_subgraph_free(og_subgraph *sg);
int og_find_nodes_edges_for_abpoa(const og_index *ix, const og_range *range, og_subgraph *out);

/* ---------------- POA (stands in for ab_poa::AbpoaAligner::create_align_safe, align.rs:202) -------- */
typedef struct {
    int32_t match;      /* +2 */
    int32_t mismatch;   /* 4 (penalty) */
    int32_t gap_open1;  /* 4 */
    int32_t gap_ext1;   /* 2 */
    int32_t gap_open2;  /* 24 */
    int32_t gap_ext2;   /* 1 */
    int32_t wb;         /* 10; <0 disables banding */
    double wf;          /* 0.01 */
    int32_t remain_rule; /* which path `remain` (the diagonal term of the adaptive band) follows: OG_REMAIN_* */
} og_poa_params;
#define OG_REMAIN_LONGEST_PATH 0   /* graph bases after the row on the LONGEST path to the sink */
#define OG_REMAIN_FIRST_OUT_EDGE 1 /* ... on the path that always takes the heaviest out-edge, first one on a tie -- with the
                                      unit edge weights of a graph that was only ever built from node strings and an edge
                                      list: the FIRST out-edge in edge-list order */
void og_poa_default_params(og_poa_params *p);

typedef struct {
    int ok;                 /* 0: no alignment inside the band */
    int32_t best_score;
    size_t n_abpoa_nodes;   /* graph bases on the alignment path (M and D columns) */
    uint32_t *abpoa_nodes;  /* 1-based base-row ids, path order */
    uint32_t *graph_nodes;  /* per path base: index of the input node string */
    uint64_t aln_start_offset; /* offset of the first path base inside its node */
    uint64_t aln_end_offset;   /* offset+1 of the last path base inside its node */
    uint64_t n_aligned_bases;  /* number of M columns */
    char *cigar;            /* run-length M/I/D */
    char *cs_string;        /* "cs:Z:" + short cs */
    uint64_t n_rows;        /* N: graph bases in the subgraph */
    uint64_t n_cells;       /* C: sum of band widths over the base rows */
} og_poa_result;
void og_poa_result_free(og_poa_result *r);

int og_poa_align(const char *const *nodes, const size_t *node_lens, size_t n_nodes,
                 const size_t *edge_src, const size_t *edge_dst, size_t n_edges, const char *query,
                 size_t qlen, const og_poa_params *params, og_poa_result *out);

/* ---------------- GAF records (src/align.rs:746-1028, 1096-1168) ---------------- */
/* each returns a malloc'd line including the trailing '\n' */
char *og_gaf_from_chain(const og_index *ix, const og_chain *chain, const char *qname, uint64_t qlen);
char *og_gaf_from_placeholder(const char *qname, uint64_t qlen);
char *og_gaf_from_poa(const og_range *range, const og_poa_result *res, const char *qname, uint64_t qlen);

/* ---------------- the whole path (src/map.rs:27-216) ---------------- */
typedef struct {
    uint64_t bandwidth;           /* 50  (map_main.rs:103) */
    uint64_t max_gap;             /* 1000 */
    uint64_t chain_min_n_anchors; /* 3 */
    uint64_t align_best_n;        /* 1 */
    int also_align;
    og_poa_params poa;
} og_map_params;
void og_map_default_params(og_map_params *p);

typedef struct {
    uint64_t n_reads, n_anchors, n_hits, n_chains, n_placeholder_reads, n_aligned_reads;
    uint64_t poa_rows, poa_cells, cigar_ops, path_bases;
    double t_anchor_s, t_chain_s, t_subgraph_s, t_poa_s;
} og_map_stats;

/* chains_gaf / alignments_gaf: malloc'd concatenations (map.rs:219-226 joins with "") */
int og_map_reads(const og_index *ix, const char *const *names, const char *const *seqs, size_t n_reads,
                 const og_map_params *params, char **chains_gaf, char **alignments_gaf,
                 og_map_stats *stats);

void og_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
