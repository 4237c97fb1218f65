/*
 * vga_hip.h -- C ABI of libvga_hip.so: the MI355X (gfx950) implementation of rs-vgaligner's
 * per-read hot path  map.rs -> chain.rs -> align.rs.
 *
 * Plain C, plain pointers and sizes, int error codes.  No C++/torch types cross this boundary.
 * Every entry point names the reference interface it stands in for (paths relative to the
 * AlgoLab/rs-vgaligner checkout).  A Rust `extern "C"` block binding exactly these symbols is shown
 * in INTEGRATION.md.
 *
 * Ownership: inputs are borrowed for the duration of a call; result objects are allocated by the
 * library and released with the matching *_free.  Device memory belongs to the ctx.  One ctx per
 * GPU; calls on one ctx must be serialised by the caller, different ctxs may be driven from
 * different threads/processes concurrently.  Nothing throws or aborts across the ABI: a negative
 * return code plus vga_last_error(ctx) replaces the reference's panic!/unwrap().
 */
#ifndef VGA_HIP_H
#define VGA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGA_OK 0
#define VGA_ERR_ARG (-1)
#define VGA_ERR_HIP (-2)          /* a HIP runtime call failed (message has the hipError string) */
#define VGA_ERR_NOMEM (-3)
#define VGA_ERR_UNSUPPORTED (-4)  /* k > 15, non-ACGT k-mer in the table, bandwidth > 64, ... */
#define VGA_ERR_NO_INDEX (-5)
#define VGA_ERR_NO_DEVICE (-6)    /* no gfx950 device visible: the library never falls back to a CPU path */
#define VGA_ERR_POOL (-7)         /* traceback pool exhausted; retry with a smaller sub-batch */

#define VGA_NO_PRED (-1)

typedef struct vga_ctx vga_ctx;

/* ---- context ---------------------------------------------------------------------------- */
/* Creates a context on HIP device `device` with a private stream.  Fails with VGA_ERR_NO_DEVICE
 * when no GPU is present. */
int vga_ctx_create(int device, vga_ctx **out);
void vga_ctx_destroy(vga_ctx *ctx);
/* Message of the last error of `ctx`.  The pointer is a copy that belongs to the calling thread (valid until that thread calls
 * vga_last_error again); two entry points of one context may run on two threads (vga_chain_paths_text beside vga_align_batch)
 * and each may read it. */
const char *vga_last_error(const vga_ctx *ctx);
/* Per-context settings (replace the VGA_POOL_FRACTION / VGA_HOST_THREADS environment hand-off of ABI <= 5; src/map.rs has no
 * counterpart: the reference is single-threaded and holds no device memory).
 *   vga_ctx_set_pool_fraction: the share (0, 1] of the device memory that is free when vga_align_batch sizes its traceback pool
 *     that THIS context may take -- 1 / n for n contexts on one GPU (`vgaligner map --devices 0,0`).  Default 1.
 *   vga_ctx_set_host_threads: host threads the calls of this context fan out to (subgraph fallback, CIGAR / cs strings, result
 *     copies); 0 = the default (VGA_HOST_THREADS if set, else the hardware's concurrency, at most 32).
 * Both take effect with the next call on the context; VGA_ERR_ARG outside those ranges. */
int vga_ctx_set_pool_fraction(vga_ctx *ctx, double fraction);
int vga_ctx_set_host_threads(vga_ctx *ctx, uint32_t n_threads);
/* Blocks until all work queued on the ctx's stream has finished. */
int vga_ctx_synchronize(vga_ctx *ctx);
/* ABI version of the library (bumped on any signature change). */
int vga_abi_version(void);

/* ---- index upload ------------------------------------------------------------------------ */
/* KmerPos, src/kmer.rs:733-738 (SeqPos = {orient, position}, src/kmer.rs:27-31).
 * The delimiter record is {1, UINT64_MAX, 1, UINT64_MAX} (src/kmer.rs:740-749). */
typedef struct {
    uint64_t start;
    uint64_t end;
    uint8_t start_orient; /* 0 Forward, 1 Reverse */
    uint8_t end_orient;
} vga_kmerpos;

/* Host view of the fields of `Index` the query side reads (src/index.rs:37-90).  The boomphf MPHF
 * and the ahash keys are replaced by the k-mer strings themselves: the reference checks exact
 * membership before it consults the MPHF (src/index.rs:319-320), so an exact map is equivalent. */
typedef struct {
    uint32_t kmer_length;          /* Index.kmer_length */
    uint64_t seq_length;           /* Index.seq_length */
    const char *seq_fwd;           /* Index.seq_fwd, seq_length bytes */
    uint64_t n_nodes;              /* Index.n_nodes */
    const uint64_t *node_seq_idx;  /* NodeRef.seq_idx,       n_nodes+1 (src/utils.rs:15-22) */
    const uint64_t *node_edge_idx; /* NodeRef.edge_idx,      n_nodes+1 */
    const uint64_t *node_edges_to; /* NodeRef.edges_to_node, n_nodes+1 */
    uint64_t n_edges;              /* Index.n_edges */
    const uint64_t *edges;         /* Index.edges as packed handles (id<<1 | is_reverse) */
    uint64_t n_kmers;              /* Index.n_kmers */
    const char *kmer_keys;         /* n_kmers * kmer_length bytes, the distinct k-mers */
    const uint64_t *kmer_starts;   /* per k-mer: first record in kmer_pos_table (the MPHF's values) */
    uint64_t n_kmer_pos;           /* Index.n_kmer_pos (records incl. delimiters) */
    const vga_kmerpos *kmer_pos_table; /* Index.kmer_pos_table */
} vga_index_desc;

/* Stands in for Index::load_from_file + every per-read Index accessor
 * (src/index.rs:296-305, 309-382, 388-606): copies the index to HBM once. */
int vga_index_upload(vga_ctx *ctx, const vga_index_desc *desc);

/* ---- read batches -------------------------------------------------------------------------- */
typedef struct vga_batch vga_batch;
/* reads_concat: all read sequences back to back; read_off[i]..read_off[i+1] delimits read i
 * (n_reads+1 offsets).  Copies the reads to HBM (the Vec<QuerySequence> of src/io.rs:74-162). */
int vga_batch_create(vga_ctx *ctx, const char *reads_concat, const uint64_t *read_off, uint64_t n_reads,
                     vga_batch **out);
/* May be called before or after vga_ctx_destroy of the batch's context: destroying the context releases the
 * batch's device memory and detaches it (later map / align calls on it return VGA_ERR_ARG). */
void vga_batch_destroy(vga_batch *b);

/* ---- map: anchors + chains ------------------------------------------------------------------ */
typedef struct {
    uint32_t bandwidth;           /* 50   (src/subcommands/map_main.rs:103) */
    uint64_t max_gap;             /* 1000 (map_main.rs:30-34) */
    uint32_t chain_min_n_anchors; /* 3    (map_main.rs:42-46) */
    int only_forward;             /* 1    (src/map.rs:62).  0 = anchors_for_query(.., false) (src/chain.rs:154-155): every
                                   * k-mer record becomes an anchor and bit 31 of target_begin / target_end is the
                                   * orientation of that end (1 = reverse strand); k <= 13; such chains cannot be passed
                                   * to vga_align_batch unless they are all-forward */
    int emit_dp;                  /* 1: the result also carries Anchor.id, f(i) and the best predecessor of every anchor
                                   *    (what the reference keeps inside chain_anchors; parity checks read them).
                                   * 0: anchor_id / max_chain_score / best_pred_id are NULL -- the GAF writers and
                                   *    vga_align_batch only read the anchor coordinates and the chain membership, and
                                   *    16 of the 40 bytes per anchor stay on the GPU. */
} vga_map_params;
void vga_map_default_params(vga_map_params *p);

/* Result of anchors_for_query + chain_anchors for every read of a batch.
 * Anchors of read r occupy [anchor_off[r], anchor_off[r+1]) and are in the order chain_anchors
 * leaves them in (stable sort by target_end.position, src/chain.rs:386-389).
 * Chains of read r are chain_off[r]..chain_off[r+1]; a read with no chain has exactly one
 * placeholder entry (src/chain.rs:644-649) with chain_len == 0. */
typedef struct {
    uint64_t n_reads;
    uint64_t n_anchors;
    uint64_t *anchor_off;    /* n_reads+1 */
    uint32_t *anchor_id;     /* Anchor.id (src/chain.rs:146,162); NULL when emit_dp = 0 */
    uint32_t *query_begin;   /* Anchor.query_begin; query_end = query_begin + k */
    uint32_t *target_begin;  /* Anchor.target_begin.position (Forward) */
    uint32_t *target_end;    /* Anchor.target_end.position   (Forward) */
    double *max_chain_score; /* f(i) after src/chain.rs:403-450; NULL when emit_dp = 0 */
    int32_t *best_pred_id;   /* id of the best predecessor after the DP, VGA_NO_PRED for None; NULL when emit_dp = 0 */
    double *curr_max;        /* per read */
    uint64_t n_chains;
    uint64_t *chain_off;        /* n_reads+1 */
    uint8_t *chain_placeholder; /* n_chains */
    uint64_t *chain_anchor_off; /* n_chains+1, into chain_anchor_idx */
    uint32_t *chain_anchor_idx; /* index (within the read's sorted anchors) of each chain member, ascending */
    /* timing of the last call, milliseconds, measured with hipEvents on the ctx stream */
    float ms_probe, ms_sort, ms_chain, ms_total;
    uint64_t n_hits;            /* position records touched (H of the byte model) */
} vga_map_result;

/* Stands in for the pass-1 loop of map_reads: anchors_for_query (src/map.rs:62 -> src/chain.rs:134)
 * followed by chain_anchors (src/map.rs:79-89 -> src/chain.rs:370). */
int vga_map_batch(vga_batch *b, const vga_map_params *params, vga_map_result **out);
void vga_map_result_free(vga_map_result *r);

/* The path column of every chain's GAF record -- GAFAlignment::from_chain (src/align.rs:762-911) with AnchorPosOnGraph::new
 * (src/chain.rs:90-127): for each anchor of a chain, in order, "(>N:off,>N:off)," = node id and offset inside the node of
 * target_begin and of the inclusive target_end.  `text` holds the fields of all chains back to back, without terminators;
 * chain c's field is text[text_off[c] .. text_off[c + 1]) (empty for a placeholder chain).  Forward-strand anchors only
 * (vga_map_params.only_forward = 1, the reference's only live setting).
 * Threads: the one call on a context that may run beside another -- one thread may be in vga_chain_paths_text while
 * another is in vga_align_batch on the same context (it works on a stream of its own and shares only the index and, on
 * failure, the error string). */
typedef struct {
    uint64_t n_chains;
    uint64_t *text_off; /* n_chains + 1 */
    char *text;
    float ms_total;
} vga_chain_text;
int vga_chain_paths_text(vga_ctx *ctx, const vga_map_result *chains, vga_chain_text **out);
void vga_chain_text_free(vga_chain_text *t);

/* ---- align: chain -> subgraph -> POA ---------------------------------------------------------- */
typedef struct {
    int32_t match;     /* 2  */
    int32_t mismatch;  /* 4  */
    int32_t gap_open1; /* 4  */
    int32_t gap_ext1;  /* 2  */
    int32_t gap_open2; /* 24 */
    int32_t gap_ext2;  /* 1  */
    int32_t wb;        /* 10, adaptive band: w = wb + floor(wf * qlen); wb < 0 disables banding */
    int32_t remain_rule; /* VGA_REMAIN_*: which path the diagonal term  qlen - remain[row]  of the adaptive band follows
                          * (ABI 4; it occupies what was padding before `wf`, so sizes and offsets are unchanged) */
    double wf;         /* 0.01 */
} vga_poa_params;
/* remain[row] = graph bases after the row on one path to the sink.  abPOA's source is not in the reference tree, so which
 * path it takes is unverified; both readings are implemented and parity-tested (oracle/og_poa.c):
 *   LONGEST_PATH   the longest path to the sink (the default of rounds 1-3 of this library);
 *   FIRST_OUT_EDGE the path that follows the heaviest out-edge, the first on a tie -- abPOA's abpoa_BFS_set_node_remain
 *                  as remembered; with the unit weights of a graph built from node strings + an edge list that is the
 *                  first out-edge in edge-list order.  vga_poa_default_params sets this one since round 4: it is the
 *                  reading most likely to reproduce the reference's scores and CIGARs. */
#define VGA_REMAIN_LONGEST_PATH 0
#define VGA_REMAIN_FIRST_OUT_EDGE 1
void vga_poa_default_params(vga_poa_params *p);

/* The fields of ab_poa's AbpoaAlignmentResult that the reference consumes
 * (src/align.rs:205-206, 1107, 1152-1165), for n problems. */
typedef struct {
    uint64_t n;
    uint8_t *ok;                /* 0: placeholder / no alignment inside the band */
    int32_t *best_score;
    uint64_t *path_off;         /* n+1: per problem, range of graph-consuming alignment columns */
    uint32_t *abpoa_nodes;      /* AbpoaAlignmentResult.abpoa_nodes: 1-based base-row id per column */
    uint32_t *graph_nodes;      /* AbpoaAlignmentResult.graph_nodes: index of the node string per column */
    uint32_t *aln_start_offset;
    uint32_t *aln_end_offset;
    uint32_t *n_aligned_bases;
    uint64_t *cigar_off;        /* n+1 */
    char *cigar;                /* concatenated, each NUL terminated inside its range */
    uint64_t *cs_off;           /* n+1 */
    char *cs;                   /* "cs:Z:..." */
    uint64_t *n_rows;           /* N of the byte model: graph bases in the subgraph */
    uint64_t *n_cells;          /* C of the byte model: sum of band widths */
    uint64_t *n_value_cells;    /* cells of the rows whose values are kept in HBM (last base of each node) */
    float ms_dp, ms_traceback, ms_total;
} vga_poa_result;
void vga_poa_result_free(vga_poa_result *r);

/* The reference's one real FFI seam:
 *   AbpoaAligner::create_align_safe(&Vec<&str> nodes, &Vec<(usize,usize)> edges, &str query, Global)
 * (src/align.rs:173-203), batched over n independent problems.
 * Problem p owns nodes node_ptr[p] .. node_ptr[p+1]-1; node v is nodes_concat[node_off[v] .. node_off[v+1])
 * (node_off holds node_ptr[n]+1 absolute offsets: node strings are laid out back to back);
 * edges edge_ptr[p]..edge_ptr[p+1] are 0-based (src,dst) pairs with src < dst; the query is
 * queries_concat[query_off[p] .. query_off[p+1]). */
int vga_poa_batch(vga_ctx *ctx, uint64_t n, const uint64_t *node_ptr, const uint64_t *node_off,
                  const char *nodes_concat, const uint64_t *edge_ptr, const uint32_t *edge_src,
                  const uint32_t *edge_dst, const uint64_t *query_off, const char *queries_concat,
                  const vga_poa_params *params, vga_poa_result **out);

/* One alignment record per read: best_alignment_for_query (src/align.rs:34-55) over the chains of
 * vga_map_batch, i.e. find_range_chain + extend_range_chain_2 + find_nodes_edges_for_abpoa +
 * create_align_safe + the fields generate_alignment needs (src/align.rs:267-402, 523-665, 670-724,
 * 202, 1096-1168). */
typedef struct {
    uint64_t n_reads;
    uint8_t *aligned;            /* 0 => placeholder record (src/align.rs:913-930) */
    uint64_t *path_off;          /* n_reads+1 */
    uint64_t *path_handles;      /* packed handles of the node path after dedup (src/align.rs:1114-1123) */
    uint32_t *path_length;       /* abpoa_nodes.len()  (src/align.rs:1152) */
    uint32_t *path_start;        /* aln_start_offset   (src/align.rs:1155) */
    uint32_t *path_end;          /* aln_end_offset     (src/align.rs:1156) */
    uint32_t *block_length;      /* n_aligned_bases    (src/align.rs:1158) */
    int32_t *best_score;
    uint64_t *cigar_off;
    char *cigar;
    uint64_t *cs_off;
    char *cs;
    uint64_t poa_rows, poa_cells, poa_value_cells, poa_problems; /* totals for the byte model */
    float ms_subgraph, ms_dp, ms_traceback, ms_total;
    uint64_t result_bytes;       /* what crossed PCIe for cs / CIGAR / node paths: their text (K4c) or the raw traceback operations (ABI 6) */
} vga_align_result;

int vga_align_batch(vga_batch *b, const vga_map_result *chains, uint32_t align_best_n,
                    const vga_poa_params *params, vga_align_result **out);
void vga_align_result_free(vga_align_result *r);
/* Optional, returns at once.  The first vga_align_batch of a context allocates its traceback memory before its first kernel
 * (tens of GB of HBM; on memory another process has used the driver clears what it hands out, 0.2 s and more).  A caller that
 * knows it is going to align n_reads reads of up to max_read_len bases says so early -- before vga_map_batch, say -- and the
 * allocation runs on a thread of its own meanwhile.  No counterpart in the reference (its abPOA allocates per call on the
 * host, src/align.rs:1032-1040). */
int vga_align_prepare(vga_ctx *ctx, uint64_t n_reads, uint32_t max_read_len);

/* Per-kernel timing of the most recent vga_map_batch / vga_poa_batch / vga_align_batch on this ctx:
 * name[i] / total milliseconds / launches, measured with hipEvents on the stream each launch ran on.
 * The POA sub-batches run two at a time on two streams: `ms` sums every launch's own duration (what
 * rocprofv3 --kernel-trace --stats reports), `busy_ms` is the wall time during which at least one launch
 * of that kernel was executing (the union of the launch intervals; equal to `ms` when nothing overlaps).
 * Returns the number of kernels (at most cap entries are written). */
typedef struct {
    const char *name;
    float ms;
    uint32_t launches;
    uint64_t algorithmic_bytes; /* byte model of DESIGN.md for the units those launches processed */
    float busy_ms;
    uint32_t reserved;
} vga_kernel_time;
int vga_last_kernel_times(const vga_ctx *ctx, vga_kernel_time *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
