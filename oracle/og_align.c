/* og_align.c -- ORACLE (test infrastructure): chain -> subgraph, and GAF records.
 *   find_range_chain ............ src/align.rs:267-402
 *   extend_range_chain_2 ........ src/align.rs:523-665
 *   find_nodes_edges_for_abpoa .. src/align.rs:670-724
 *   AnchorPosOnGraph::new ....... src/chain.rs:90-127
 *   GAFAlignment::from_chain .... src/align.rs:762-911
 *   from_placeholder_chain ...... src/align.rs:913-930
 *   generate_alignment .......... src/align.rs:1096-1168
 *   GAFAlignment::to_string ..... src/align.rs:971-1027
 *
 * u64::from(Handle) at align.rs:359-378 is read as the NODE ID (the code packs it back with x*2).
 *
 * extend_range_chain_2's breadth-first walk keeps no visited set, so its frontier (not its result)
 * grows exponentially along chains of bubbles.  A handle is pushed iff it can be reached from the
 * range's first (last) handle with strictly less than prefix_diff (suffix_diff) bases of
 * intermediate nodes in between; the restatement computes exactly that set by keeping, per handle,
 * the largest remaining budget seen so far.  The final sort+dedup (align.rs:658-659) makes the two
 * indistinguishable.
 */
#include "og_internal.h"

void og_range_free(og_range *r)
{
    if (!r) return;
    free(r->handles);
    r->handles = NULL;
    r->n = 0;
}

static og_seqpos og_end_inclusive(const og_anchor *a)
{ /* chain.rs:65-70 */
    og_seqpos p = a->target_end;
    p.position -= 1;
    return p;
}

int og_find_range_chain(const og_index *ix, const og_chain *chain, og_range *out)
{
    memset(out, 0, sizeof(*out));
    if (!chain || chain->n == 0) return OG_ERR_ARG;
    og_handle min_h = UINT64_MAX, max_h = 0;
    for (size_t i = 0; i < chain->n; i++) {
        og_handle s = og_index_handle_from_seqpos(ix, chain->anchors[i].target_begin);
        og_handle e = og_index_handle_from_seqpos(ix, og_end_inclusive(&chain->anchors[i]));
        if (s < min_h) min_h = s;
        if (s > max_h) max_h = s;
        if (e < min_h) min_h = e;
        if (e > max_h) max_h = e;
    }
    uint64_t lo = OG_H_ID(min_h), hi = OG_H_ID(max_h);
    size_t cap = 0;
    if (!OG_H_REV(min_h) && !OG_H_REV(max_h)) {
        for (uint64_t x = lo; x <= hi; x++) {
            OG_GROW(out->handles, out->n, cap, og_handle);
            out->handles[out->n++] = x * 2;
        }
        out->orient = OG_RANGE_FORWARD;
    } else if (OG_H_REV(min_h) && OG_H_REV(max_h)) {
        for (uint64_t x = lo; x <= hi; x++) {
            OG_GROW(out->handles, out->n, cap, og_handle);
            out->handles[out->n++] = x * 2 + 1;
        }
        out->orient = OG_RANGE_REVERSE;
    } else {
        /* fwd list ++ rev list, then sorted: interleaves x*2, x*2+1 */
        for (uint64_t x = lo; x <= hi; x++) {
            OG_GROW(out->handles, out->n, cap, og_handle);
            out->handles[out->n++] = x * 2;
            OG_GROW(out->handles, out->n, cap, og_handle);
            out->handles[out->n++] = x * 2 + 1;
        }
        out->orient = OG_RANGE_BOTH;
    }
    if (out->n == 0 && min_h == max_h) { /* align.rs:394-396 */
        OG_GROW(out->handles, out->n, cap, og_handle);
        out->handles[out->n++] = min_h;
    }
    return OG_OK;
}

static int og_handle_qcmp(const void *a, const void *b)
{
    og_handle x = *(const og_handle *)a, y = *(const og_handle *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

typedef struct {
    uint64_t budget;
    og_handle h;
} og_front;

/* one direction of align.rs:551-591 / 616-656 */
static void og_extend_dir(const og_index *ix, og_handle from, uint64_t diff, int incoming, og_handle **hs,
                          size_t *n, size_t *cap)
{
    /* best remaining budget per packed handle */
    size_t nh = (size_t)(ix->n_nodes + 1) * 2 + 2;
    uint64_t *best = (uint64_t *)calloc(nh, sizeof(uint64_t));
    og_front *cur = NULL, *next = NULL;
    size_t ncur = 0, ccur = 0, nnext = 0, cnext = 0;
    og_handle nb[4096];
    size_t k = incoming ? og_index_incoming_edges(ix, from, nb, 4096) : og_index_outgoing_edges(ix, from, nb, 4096);
    for (size_t i = 0; i < k; i++) {
        OG_GROW(cur, ncur, ccur, og_front);
        cur[ncur].budget = diff;
        cur[ncur].h = nb[i];
        ncur++;
    }
    while (ncur > 0) {
        nnext = 0;
        for (size_t i = 0; i < ncur; i++) {
            og_handle h = cur[i].h;
            uint64_t left = cur[i].budget;
            if (h < nh && best[h] >= left) continue; /* already visited with at least this budget */
            if (h < nh) {
                if (best[h] == 0) {
                    OG_GROW(*hs, *n, *cap, og_handle);
                    (*hs)[(*n)++] = h;
                }
                best[h] = left;
            }
            uint64_t len = og_index_seq_from_handle(ix, h, NULL, 0);
            if (len < left) {
                uint64_t rem = left - len;
                size_t m = incoming ? og_index_incoming_edges(ix, h, nb, 4096) : og_index_outgoing_edges(ix, h, nb, 4096);
                for (size_t t = 0; t < m; t++) {
                    OG_GROW(next, nnext, cnext, og_front);
                    next[nnext].budget = rem;
                    next[nnext].h = nb[t];
                    nnext++;
                }
            }
        }
        og_front *tf = cur; cur = next; next = tf;
        size_t tc = ccur; ccur = cnext; cnext = tc;
        ncur = nnext;
    }
    free(cur);
    free(next);
    free(best);
}

int og_extend_range_chain_2(const og_index *ix, const og_chain *chain, uint64_t query_len,
                            const og_range *old_range, og_range *out)
{
    memset(out, 0, sizeof(*out));
    if (!chain || chain->n == 0 || old_range->n == 0) return OG_ERR_ARG;
    size_t cap = old_range->n + 16;
    out->handles = (og_handle *)malloc(cap * sizeof(og_handle));
    memcpy(out->handles, old_range->handles, old_range->n * sizeof(og_handle));
    out->n = old_range->n;
    out->orient = old_range->orient;

    const og_anchor *first_anchor = &chain->anchors[0];
    const og_anchor *last_anchor = &chain->anchors[chain->n - 1];

    /* align.rs:536-547 */
    uint64_t prefix_diff = first_anchor->query_begin;
    og_handle first_handle = old_range->handles[0];
    uint64_t start_prefix_on_node =
        first_anchor->target_begin.position - og_index_bv_select(ix, OG_H_ID(first_handle));
    if (start_prefix_on_node < prefix_diff) prefix_diff -= start_prefix_on_node;
    else prefix_diff = 0;
    if (prefix_diff > 0) og_extend_dir(ix, first_handle, prefix_diff, 1, &out->handles, &out->n, &cap);

    /* align.rs:593-612 */
    uint64_t suffix_diff = query_len - last_anchor->query_end;
    og_handle last_handle = old_range->handles[old_range->n - 1];
    uint64_t end_suffix_on_node =
        og_index_bv_select(ix, OG_H_ID(last_handle) + 1) - 1 - og_end_inclusive(last_anchor).position;
    if (end_suffix_on_node > suffix_diff) suffix_diff = 0;
    else suffix_diff -= end_suffix_on_node;
    if (suffix_diff > 0) og_extend_dir(ix, last_handle, suffix_diff, 0, &out->handles, &out->n, &cap);

    /* align.rs:658-659 */
    qsort(out->handles, out->n, sizeof(og_handle), og_handle_qcmp);
    size_t o = 0;
    for (size_t i = 0; i < out->n; i++)
        if (o == 0 || out->handles[i] != out->handles[o - 1]) out->handles[o++] = out->handles[i];
    out->n = o;
    return OG_OK;
}

void og_subgraph_free(og_subgraph *sg)
{
    if (!sg) return;
    for (size_t i = 0; i < sg->n_nodes; i++) free(sg->seqs[i]);
    free(sg->seqs);
    free(sg->seq_lens);
    free(sg->edge_src);
    free(sg->edge_dst);
    memset(sg, 0, sizeof(*sg));
}

/* first index of h in the (sorted, deduplicated) range, or -1 */
static int64_t og_range_position(const og_range *r, og_handle h)
{
    size_t lo = 0, hi = r->n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (r->handles[mid] < h) lo = mid + 1;
        else hi = mid;
    }
    if (lo < r->n && r->handles[lo] == h) return (int64_t)lo;
    /* the un-extended range of find_range_chain is sorted too; fall back to a scan otherwise */
    for (size_t i = 0; i < r->n; i++)
        if (r->handles[i] == h) return (int64_t)i;
    return -1;
}

int og_find_nodes_edges_for_abpoa(const og_index *ix, const og_range *range, og_subgraph *out)
{
    memset(out, 0, sizeof(*out));
    out->n_nodes = range->n;
    out->seqs = (char **)calloc(range->n ? range->n : 1, sizeof(char *));
    out->seq_lens = (size_t *)calloc(range->n ? range->n : 1, sizeof(size_t));
    for (size_t i = 0; i < range->n; i++) {
        size_t len = og_index_seq_from_handle(ix, range->handles[i], NULL, 0);
        out->seqs[i] = (char *)malloc(len + 1);
        og_index_seq_from_handle(ix, range->handles[i], out->seqs[i], len);
        out->seqs[i][len] = 0;
        out->seq_lens[i] = len;
    }
    size_t cap = 0, n = 0;
    og_handle nb[4096];
    for (size_t i = 0; i < range->n; i++) {
        size_t m = og_index_outgoing_edges(ix, range->handles[i], nb, 4096);
        for (size_t t = 0; t < m; t++) {
            int64_t e = og_range_position(range, nb[t]);
            if (e < 0) continue;
            size_t s = i; /* position(|x| x == handle): handles are unique after dedup */
            int keep = 1;
            if (range->orient == OG_RANGE_FORWARD) keep = s < (size_t)e;
            else if (range->orient == OG_RANGE_REVERSE) keep = (size_t)e < s;
            if (!keep) continue;
            OG_GROW(out->edge_src, n, cap, size_t);
            out->edge_dst = (size_t *)realloc(out->edge_dst, cap * sizeof(size_t));
            out->edge_src[n] = s;
            out->edge_dst[n] = (size_t)e;
            n++;
        }
    }
    out->n_edges = n;
    return OG_OK;
}

/* ---------------- GAF ---------------- */
typedef struct {
    char *s;
    size_t n, cap;
} og_str;

static void og_str_add(og_str *b, const char *s, size_t len)
{
    if (b->n + len + 1 > b->cap) {
        size_t nc = b->cap ? b->cap : 256;
        while (nc < b->n + len + 1) nc *= 2;
        b->s = (char *)realloc(b->s, nc);
        b->cap = nc;
    }
    memcpy(b->s + b->n, s, len);
    b->n += len;
    b->s[b->n] = 0;
}
static void og_str_adds(og_str *b, const char *s) { og_str_add(b, s, strlen(s)); }
static void og_str_addu(og_str *b, uint64_t v)
{
    char t[32];
    int n = snprintf(t, sizeof t, "%llu", (unsigned long long)v);
    og_str_add(b, t, (size_t)n);
}

char *og_gaf_from_placeholder(const char *qname, uint64_t qlen)
{ /* align.rs:913-930 + 971-1027 */
    og_str b = {0};
    og_str_adds(&b, qname);
    og_str_adds(&b, "\t");
    og_str_addu(&b, qlen);
    og_str_adds(&b, "\t*\t*\t*\t*\t*\t*\t*\t*\t*\t0\t*\n");
    return b.s;
}

char *og_gaf_from_chain(const og_index *ix, const og_chain *chain, const char *qname, uint64_t qlen)
{
    if (chain->is_placeholder) return og_gaf_from_placeholder(qname, qlen);
    og_str b = {0};
    og_str_adds(&b, qname);
    og_str_adds(&b, "\t");
    og_str_addu(&b, qlen);
    og_str_adds(&b, "\t");
    og_str_addu(&b, chain->anchors[0].query_begin);
    og_str_adds(&b, "\t");
    og_str_addu(&b, chain->anchors[chain->n - 1].query_end);
    og_str_adds(&b, "\t+\t");
    for (size_t i = 0; i < chain->n; i++) {
        const og_anchor *a = &chain->anchors[i];
        /* chain.rs:90-127 */
        og_handle fh = og_index_handle_from_seqpos(ix, a->target_begin);
        uint64_t fstart = og_index_bv_select(ix, OG_H_ID(fh));
        uint64_t foff = a->target_begin.position - fstart;
        og_seqpos ei = og_end_inclusive(a);
        og_handle lh = og_index_handle_from_seqpos(ix, ei);
        uint64_t lstart = og_index_bv_select(ix, OG_H_ID(lh));
        uint64_t loff = ei.position - lstart;
        og_str_adds(&b, "(");
        og_str_adds(&b, OG_H_REV(fh) ? "<" : ">");
        og_str_addu(&b, OG_H_ID(fh));
        og_str_adds(&b, ":");
        og_str_addu(&b, foff);
        og_str_adds(&b, ",");
        og_str_adds(&b, OG_H_REV(lh) ? "<" : ">");
        og_str_addu(&b, OG_H_ID(lh));
        og_str_adds(&b, ":");
        og_str_addu(&b, loff);
        og_str_adds(&b, "),");
    }
    /* plen pstart pend residue block = 0; mapq = min(f64::MIN as u64, 254) = 0 */
    og_str_adds(&b, "\t0\t0\t0\t0\t0\t0\tta:Z:chain,n_anchors: ");
    og_str_addu(&b, chain->n);
    og_str_adds(&b, "\n");
    return b.s;
}

char *og_gaf_from_poa(const og_range *range, const og_poa_result *res, const char *qname, uint64_t qlen)
{ /* align.rs:1096-1168 */
    og_str b = {0};
    og_str_adds(&b, qname);
    og_str_adds(&b, "\t");
    og_str_addu(&b, qlen);
    og_str_adds(&b, "\t0\t");
    og_str_addu(&b, qlen);
    og_str_adds(&b, "\t+\t");
    /* graph_nodes.dedup() -> range.handles[idx] */
    uint32_t prev = UINT32_MAX;
    for (size_t i = 0; i < res->n_abpoa_nodes; i++) {
        uint32_t gnode = res->graph_nodes[i];
        if (i > 0 && gnode == prev) continue;
        prev = gnode;
        og_handle h = range->handles[gnode];
        og_str_adds(&b, OG_H_REV(h) ? "<" : ">");
        og_str_addu(&b, OG_H_ID(h));
    }
    og_str_adds(&b, "\t");
    og_str_addu(&b, res->n_abpoa_nodes);
    og_str_adds(&b, "\t");
    og_str_addu(&b, res->aln_start_offset);
    og_str_adds(&b, "\t");
    og_str_addu(&b, res->aln_end_offset);
    og_str_adds(&b, "\t0\t");
    og_str_addu(&b, res->n_aligned_bases);
    og_str_adds(&b, "\t255\tas:i:-30 ");
    og_str_adds(&b, res->cs_string ? res->cs_string : "");
    og_str_adds(&b, ",cg:Z:");
    og_str_adds(&b, res->cigar ? res->cigar : "");
    og_str_adds(&b, "\n");
    return b.s;
}
