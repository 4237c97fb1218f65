/* og_internal.h -- private definitions shared by the oracle's C files (test infrastructure). */
#ifndef OG_INTERNAL_H
#define OG_INTERNAL_H

#include "vga_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    char *seq;
    size_t len;
    og_handle *left;  /* handles x with an edge x -> node+ (insertion order) */
    size_t nleft, cleft;
    og_handle *right; /* handles y with an edge node+ -> y (insertion order) */
    size_t nright, cright;
    int present;
} og_node;

struct og_graph {
    og_node *nodes; /* indexed by node id */
    size_t cap;
    uint64_t min_id, max_id;
    size_t n_nodes;
    char **path_names;
    og_handle **path_steps;
    size_t *path_lens;
    size_t n_paths, c_paths;
};

typedef struct {
    char *seq; /* owned, k bytes when complete (shorter while incomplete) */
    uint32_t seq_len;
    og_seqpos begin_offset, end_offset;
    og_handle first_handle, last_handle;
    int handle_orient;
    uint64_t forks;
} og_graphkmer;

struct og_index {
    uint64_t k, seq_length, n_nodes, n_edges;
    char *seq_fwd, *seq_rev;
    uint8_t *seq_bv;       /* seq_length + 1 */
    uint32_t *rank_prefix; /* rank_prefix[i] = number of set bits in seq_bv[0..=i] */
    og_handle *edges;
    og_noderef *node_ref;  /* n_nodes + 1 */
    uint64_t n_kmers, n_kmer_pos;
    char *kmer_keys;
    uint64_t *kmer_starts;
    og_kmerpos *table;
    og_graphkmer *gkmers;
    uint64_t n_gkmers;
};

#define OG_GROW(ptr, n, cap, T)                                              \
    do {                                                                     \
        if ((n) >= (cap)) {                                                  \
            size_t ncap_ = (cap) ? (cap)*2 : 16;                             \
            while (ncap_ <= (n)) ncap_ *= 2;                                 \
            (ptr) = (T *)realloc((ptr), ncap_ * sizeof(T));                  \
            (cap) = ncap_;                                                   \
        }                                                                    \
    } while (0)

static inline int og_seqpos_cmp(og_seqpos a, og_seqpos b)
{ /* derive(Ord) on (orient, position): kmer.rs:27-31 */
    if (a.orient != b.orient) return a.orient < b.orient ? -1 : 1;
    if (a.position != b.position) return a.position < b.position ? -1 : 1;
    return 0;
}

static inline int og_kmerpos_cmp(const og_kmerpos *a, const og_kmerpos *b)
{ /* derive(Ord) on (start, end): kmer.rs:732-738 */
    int c = og_seqpos_cmp(a->start, b->start);
    if (c) return c;
    return og_seqpos_cmp(a->end, b->end);
}

char og_complement(char c);
double og_now(void);

#endif
