/* og_map.c -- ORACLE (test infrastructure): the per-read driver.
 *   map_reads ..................... src/map.rs:27-216 (pass 1 chains 56-111, chains GAF 123-145,
 *                                   pass 2 alignments 154-184)
 *   best_alignment_for_query ...... src/align.rs:34-55
 *   obtain_base_level_alignment ... src/align.rs:58-145 (without the per-read subgraph GFA export side effect)
 * No debug printing (the reference prints inside its hot loops; that is not part of the contract).
 */
#include "og_internal.h"

void og_map_default_params(og_map_params *p)
{
    p->bandwidth = 50;            /* map_main.rs:103 */
    p->max_gap = 1000;            /* map_main.rs:30-34 */
    p->chain_min_n_anchors = 3;   /* map_main.rs:42-46 */
    p->align_best_n = 1;          /* map_main.rs:48-52 */
    p->also_align = 1;
    og_poa_default_params(&p->poa);
}

typedef struct {
    char *s;
    size_t n, cap;
} og_buf;
static void og_buf_add(og_buf *b, const char *s)
{
    size_t len = strlen(s);
    if (b->n + len + 1 > b->cap) {
        size_t nc = b->cap ? b->cap : 4096;
        while (nc < b->n + len + 1) nc *= 2;
        b->s = (char *)realloc(b->s, nc);
        b->cap = nc;
    }
    memcpy(b->s + b->n, s, len + 1);
    b->n += len;
}

int og_map_reads(const og_index *ix, const char *const *names, const char *const *seqs, size_t n_reads,
                 const og_map_params *params, char **chains_gaf, char **alignments_gaf,
                 og_map_stats *stats)
{
    og_buf cb = {0}, ab = {0};
    og_map_stats st;
    memset(&st, 0, sizeof st);
    st.n_reads = n_reads;
    og_buf_add(&cb, "");
    og_buf_add(&ab, "");
    for (size_t r = 0; r < n_reads; r++) {
        const char *q = seqs[r];
        size_t qlen = strlen(q);
        double t0 = og_now();
        og_anchor *anchors = NULL;
        size_t na = og_anchors_for_query(ix, q, qlen, 1, &anchors); /* map.rs:62 only_forward=true */
        double t1 = og_now();
        st.t_anchor_s += t1 - t0;
        st.n_anchors += na;
        og_chain_set cs;
        og_chain_anchors(anchors, na, ix->k, params->bandwidth, params->max_gap,
                         params->chain_min_n_anchors, &cs, NULL);
        double t2 = og_now();
        st.t_chain_s += t2 - t1;
        st.n_chains += cs.n;
        if (cs.n == 1 && cs.chains[0].is_placeholder) st.n_placeholder_reads++;
        for (size_t c = 0; c < cs.n; c++) {
            char *line = og_gaf_from_chain(ix, &cs.chains[c], names[r], qlen);
            og_buf_add(&cb, line);
            free(line);
        }
        if (params->also_align) {
            /* align.rs:43-54: align the first min(best_n, len) chains, keep the longest path_length */
            size_t take = cs.n < params->align_best_n ? cs.n : (size_t)params->align_best_n;
            char *best_line = NULL;
            int best_has = 0;      /* Option<u64>: None < Some */
            uint64_t best_plen = 0;
            int aligned = 0;
            for (size_t c = 0; c < take; c++) {
                char *line = NULL;
                int has = 0;
                uint64_t plen = 0;
                if (cs.chains[c].is_placeholder) {
                    line = og_gaf_from_placeholder(names[r], qlen);
                } else {
                    double ta = og_now();
                    og_range r0, r1;
                    og_subgraph sg;
                    og_find_range_chain(ix, &cs.chains[c], &r0);
                    og_extend_range_chain_2(ix, &cs.chains[c], qlen, &r0, &r1);
                    og_find_nodes_edges_for_abpoa(ix, &r1, &sg);
                    double tb = og_now();
                    st.t_subgraph_s += tb - ta;
                    og_poa_result res;
                    int rc = og_poa_align((const char *const *)sg.seqs, sg.seq_lens, sg.n_nodes, sg.edge_src,
                                          sg.edge_dst, sg.n_edges, q, qlen, &params->poa, &res);
                    st.t_poa_s += og_now() - tb;
                    if (rc == OG_OK && res.ok) {
                        line = og_gaf_from_poa(&r1, &res, names[r], qlen);
                        has = 1;
                        plen = res.n_abpoa_nodes;
                        st.poa_rows += res.n_rows;
                        st.poa_cells += res.n_cells;
                        st.path_bases += res.n_abpoa_nodes;
                        st.cigar_ops += strlen(res.cigar);
                        aligned = 1;
                    } else {
                        line = og_gaf_from_placeholder(names[r], qlen);
                    }
                    if (rc == OG_OK) og_poa_result_free(&res);
                    og_subgraph_free(&sg);
                    og_range_free(&r0);
                    og_range_free(&r1);
                }
                /* stable sort by path_length descending, take the first */
                int better = 0;
                if (!best_line) better = 1;
                else if (has && !best_has) better = 1;
                else if (has && best_has && plen > best_plen) better = 1;
                if (better) {
                    free(best_line);
                    best_line = line;
                    best_has = has;
                    best_plen = plen;
                } else {
                    free(line);
                }
            }
            if (aligned) st.n_aligned_reads++;
            if (best_line) {
                og_buf_add(&ab, best_line);
                free(best_line);
            }
        }
        og_chain_set_free(&cs);
        free(anchors);
    }
    if (chains_gaf) *chains_gaf = cb.s; else free(cb.s);
    if (alignments_gaf) *alignments_gaf = ab.s; else free(ab.s);
    if (stats) *stats = st;
    return OG_OK;
}
