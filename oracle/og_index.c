/* og_index.c -- ORACLE (test infrastructure): CPU restatement of the reference's index.
 *   Index::build ............ src/index.rs:109-281
 *   find_forward_sequence ... src/utils.rs:81-146
 *   reverse_complement ...... src/dna.rs:5-33
 *   k-mer enumeration ....... src/kmer.rs:277-505 (generate_kmers_parallel, the one build() calls)
 *                             src/kmer.rs:93-273  (generate_kmers, used by reference tests only)
 *   positions table ......... src/kmer.rs:752-770, 816-928
 *   lookup .................. src/index.rs:309-382
 *   rank / select / handles . src/index.rs:388-480
 *   sequence + edges ........ src/index.rs:489-606
 *
 * boomphf + ahash are not restated: the reference tests exact membership (index.rs:319) before it
 * asks the MPHF, so any exact k-mer -> offset map gives identical answers.  Here the distinct
 * k-mers are kept sorted (they come out of kmer.rs:298 sorted) and searched by bisection.
 * sampling_rate is None (index_main.rs:39-42 default); the sampled mode depends on ahash values.
 */
#include "og_internal.h"

/* ---------- dna.rs:5-33 ---------- */
static int og_is_dna(char b)
{
    switch (b) {
    case 'A': case 'a': case 'C': case 'c': case 'G': case 'g':
    case 'T': case 't': case 'U': case 'u': case 'N':
        return 1;
    default:
        return 0;
    }
}
static char og_switch_base(char c)
{
    switch (c) {
    case 'a': return 't';
    case 'c': return 'g';
    case 't': return 'a';
    case 'g': return 'c';
    case 'u': return 'a';
    case 'A': return 'T';
    case 'C': return 'G';
    case 'T': return 'A';
    case 'G': return 'C';
    case 'U': return 'A';
    default: return 'N';
    }
}

/* ---------- k-mer enumeration ---------- */
typedef struct {
    og_graphkmer *a;
    size_t n, cap;
} og_kvec;

static void og_kmer_free(og_graphkmer *k) { free(k->seq); }

static og_graphkmer og_kmer_clone(const og_graphkmer *k, uint64_t kk)
{
    og_graphkmer c = *k;
    c.seq = (char *)malloc(kk + 1);
    memcpy(c.seq, k->seq, k->seq_len);
    return c;
}

static void og_kvec_push(og_kvec *v, og_graphkmer k)
{
    OG_GROW(v->a, v->n, v->cap, og_graphkmer);
    v->a[v->n++] = k;
}

static int og_has_N(const char *s, size_t n)
{
    return memchr(s, 'N', n) != NULL;
}

/* kmer.rs:347-505 (abort_on_N=1) and the per-handle body of kmer.rs:103-261 (abort_on_N=0).
 * Appends the complete k-mers of one (handle, orientation) to out.  With abort_on_N the whole
 * call yields nothing as soon as any examined k-mer prefix contains 'N' (kmer.rs:401-403,459-461). */
static void og_kmers_from_handle(const og_graph *g, og_handle handle_in, int orient_flag, uint64_t k,
                                 uint64_t edge_max, uint64_t degree_max, int abort_on_N, og_kvec *out)
{
    og_handle nb[4096];
    og_handle handle = handle_in;
    size_t out_start = out->n;
    og_kvec inc = {0};

    /* kmer.rs:361-372 */
    size_t cnt = og_graph_neighbors(g, handle, 0, nb, 4096);
    if ((uint64_t)cnt > degree_max) return;

    size_t hlen = og_graph_node_len(g, OG_H_ID(handle));
    char *hseq = (char *)malloc(hlen + 1);
    og_graph_sequence(g, handle, hseq);
    int aborted = 0;

    /* kmer.rs:383-446 */
    for (uint64_t i = 0; i < hlen && !aborted; i++) {
        uint64_t begin = i;
        uint64_t end = (i + k < hlen) ? i + k : hlen;
        og_graphkmer km;
        km.seq = (char *)malloc(k + 1);
        km.seq_len = (uint32_t)(end - begin);
        memcpy(km.seq, hseq + begin, km.seq_len);
        km.begin_offset.orient = (uint8_t)OG_H_REV(handle);
        km.begin_offset.position = begin;
        km.end_offset.orient = (uint8_t)OG_H_REV(handle);
        km.end_offset.position = end;
        km.first_handle = handle;
        km.last_handle = handle;
        km.handle_orient = orient_flag;
        km.forks = 0;

        if (og_has_N(km.seq, km.seq_len)) {
            og_kmer_free(&km);
            if (abort_on_N) { aborted = 1; break; }
            continue;
        }
        if (km.seq_len == k) {
            og_kvec_push(out, km);
        } else {
            uint64_t next_count = og_graph_neighbors(g, handle, 0, nb, 4096);
            if (next_count < degree_max || km.forks < edge_max) {
                for (uint64_t t = 0; t < next_count; t++) {
                    og_graphkmer c = og_kmer_clone(&km, k);
                    c.last_handle = nb[t];
                    if (next_count > 1) c.forks += 1;
                    og_kvec_push(&inc, c);
                }
            }
            og_kmer_free(&km);
        }
    }
    free(hseq);

    /* kmer.rs:449-497: Vec::pop() == LIFO */
    while (!aborted && inc.n > 0) {
        og_graphkmer km = inc.a[--inc.n];
        handle = km.last_handle;
        hlen = og_graph_node_len(g, OG_H_ID(handle));
        hseq = (char *)malloc(hlen + 1);
        og_graph_sequence(g, handle, hseq);
        uint64_t end = (k - km.seq_len < hlen) ? k - km.seq_len : hlen;
        /* extend_kmer, kmer.rs:80-84 */
        memcpy(km.seq + km.seq_len, hseq, end);
        km.seq_len += (uint32_t)end;
        km.end_offset.orient = (uint8_t)OG_H_REV(handle);
        km.end_offset.position = end;
        km.last_handle = handle;
        free(hseq);

        if (og_has_N(km.seq, km.seq_len)) {
            og_kmer_free(&km);
            if (abort_on_N) { aborted = 1; break; }
            continue;
        }
        if (km.seq_len == k) {
            og_kvec_push(out, km);
        } else {
            uint64_t next_count = og_graph_neighbors(g, handle, 0, nb, 4096);
            for (uint64_t t = 0; t < next_count; t++) {
                if (next_count < degree_max || km.forks < edge_max) {
                    og_graphkmer c = og_kmer_clone(&km, k);
                    c.last_handle = nb[t];
                    if (next_count > 1) c.forks += 1;
                    og_kvec_push(&inc, c);
                }
            }
            og_kmer_free(&km);
        }
    }
    for (size_t i = 0; i < inc.n; i++) og_kmer_free(&inc.a[i]);
    free(inc.a);
    if (aborted) { /* return Vec::new() */
        for (size_t i = out_start; i < out->n; i++) og_kmer_free(&out->a[i]);
        out->n = out_start;
    }
}

static int og_gk_equal(const og_graphkmer *a, const og_graphkmer *b)
{ /* derive(PartialEq) on GraphKmer, kmer.rs:47-65 */
    return a->seq_len == b->seq_len && memcmp(a->seq, b->seq, a->seq_len) == 0 &&
           a->begin_offset.orient == b->begin_offset.orient &&
           a->begin_offset.position == b->begin_offset.position &&
           a->end_offset.orient == b->end_offset.orient &&
           a->end_offset.position == b->end_offset.position && a->first_handle == b->first_handle &&
           a->last_handle == b->last_handle && a->handle_orient == b->handle_orient &&
           a->forks == b->forks;
}

/* stable merge sort of k-mers by sequence bytes (Vec::sort_by is stable; String::cmp is bytewise) */
static void og_gk_msort(og_graphkmer *a, og_graphkmer *tmp, size_t n, uint64_t k)
{
    if (n < 2) return;
    size_t m = n / 2;
    og_gk_msort(a, tmp, m, k);
    og_gk_msort(a + m, tmp, n - m, k);
    size_t i = 0, j = m, o = 0;
    while (i < m && j < n) {
        if (memcmp(a[j].seq, a[i].seq, k) < 0) tmp[o++] = a[j++];
        else tmp[o++] = a[i++];
    }
    while (i < m) tmp[o++] = a[i++];
    while (j < n) tmp[o++] = a[j++];
    memcpy(a, tmp, n * sizeof(og_graphkmer));
}

static void og_sort_dedup(og_kvec *v, uint64_t k)
{
    if (v->n == 0) return;
    og_graphkmer *tmp = (og_graphkmer *)malloc(v->n * sizeof(og_graphkmer));
    og_gk_msort(v->a, tmp, v->n, k);
    free(tmp);
    size_t o = 1;
    for (size_t i = 1; i < v->n; i++) { /* Vec::dedup: drop an element equal to the last KEPT one */
        if (og_gk_equal(&v->a[i], &v->a[o - 1])) og_kmer_free(&v->a[i]);
        else v->a[o++] = v->a[i];
    }
    v->n = o;
}

/* kmer.rs:277-304 */
static void og_generate_kmers_parallel(const og_graph *g, uint64_t k, uint64_t edge_max,
                                       uint64_t degree_max, og_kvec *out)
{
    for (uint64_t id = g->min_id; id <= g->max_id && g->n_nodes; id++) {
        if (id >= g->cap || !g->nodes[id].present) continue;
        og_handle fwd = OG_H_PACK(id, 0);
        /* kmer.rs:316: [true, false] -> forward handle first, then its flip */
        og_kmers_from_handle(g, fwd, 1, k, edge_max, degree_max, 1, out);
        og_kmers_from_handle(g, OG_H_FLIP(fwd), 0, k, edge_max, degree_max, 1, out);
    }
    og_sort_dedup(out, k);
}

int64_t og_generate_kmers_count(const og_graph *g, uint64_t k, uint64_t edge_max, uint64_t degree_max)
{ /* kmer.rs:93-273 */
    og_kvec v = {0};
    for (uint64_t id = g->min_id; id <= g->max_id && g->n_nodes; id++) {
        if (id >= g->cap || !g->nodes[id].present) continue;
        og_handle fwd = OG_H_PACK(id, 0);
        og_kmers_from_handle(g, fwd, 1, k, edge_max, degree_max, 0, &v);
        og_kmers_from_handle(g, OG_H_FLIP(fwd), 0, k, edge_max, degree_max, 0, &v);
    }
    og_sort_dedup(&v, k);
    int64_t n = (int64_t)v.n;
    for (size_t i = 0; i < v.n; i++) og_kmer_free(&v.a[i]);
    free(v.a);
    return n;
}

/* kmer.rs:752-770 */
static uint64_t og_get_seq_pos(og_handle h, const og_noderef *node_ref, uint64_t ref_length,
                               uint64_t handle_length)
{
    uint64_t rank = OG_H_ID(h) - 1;
    uint64_t node_start = node_ref[rank].seq_idx;
    return OG_H_REV(h) ? ref_length - node_start - handle_length : node_start;
}

static int og_kmerpos_qcmp(const void *a, const void *b)
{
    return og_kmerpos_cmp((const og_kmerpos *)a, (const og_kmerpos *)b);
}

int og_index_build(const og_graph *g, uint64_t k, uint64_t max_furcations, uint64_t max_degree,
                   og_index **out)
{
    if (!g || !out || k == 0) return OG_ERR_ARG;
    /* index.rs:489-491: node_ref is addressed by id-1 => ids must be exactly 1..n */
    if (g->n_nodes == 0 || g->min_id != 1 || g->max_id != g->n_nodes) return OG_ERR_NODE_IDS;

    og_index *ix = (og_index *)calloc(1, sizeof(og_index));
    ix->k = k;
    ix->n_nodes = g->n_nodes;
    /* utils.rs:25-31 */
    uint64_t seq_length = 0;
    for (uint64_t id = 1; id <= g->max_id; id++) seq_length += g->nodes[id].len;
    ix->seq_length = seq_length;
    ix->seq_bv = (uint8_t *)calloc(seq_length + 1, 1);
    ix->seq_fwd = (char *)malloc(seq_length + 1);
    ix->node_ref = (og_noderef *)calloc(g->n_nodes + 1, sizeof(og_noderef));

    /* utils.rs:81-146 */
    size_t ecap = 0, en = 0;
    og_handle nb[4096];
    uint64_t bv_pos = 0;
    for (uint64_t id = 1; id <= g->max_id; id++) {
        og_handle h = OG_H_PACK(id, 0);
        const og_node *nd = &g->nodes[id];
        memcpy(ix->seq_fwd + bv_pos, nd->seq, nd->len);
        size_t nl = og_graph_neighbors(g, h, 1, nb, 4096);
        ix->node_ref[id - 1].seq_idx = bv_pos;
        ix->node_ref[id - 1].edge_idx = en;
        ix->node_ref[id - 1].edges_to_node = nl;
        for (size_t i = 0; i < nl; i++) {
            OG_GROW(ix->edges, en, ecap, og_handle);
            ix->edges[en++] = nb[i];
        }
        size_t nr = og_graph_neighbors(g, h, 0, nb, 4096);
        for (size_t i = 0; i < nr; i++) {
            OG_GROW(ix->edges, en, ecap, og_handle);
            ix->edges[en++] = nb[i];
        }
        ix->seq_bv[bv_pos] = 1;
        bv_pos += nd->len;
    }
    ix->seq_bv[bv_pos] = 1;
    ix->node_ref[g->n_nodes].seq_idx = bv_pos;
    ix->node_ref[g->n_nodes].edge_idx = en;
    ix->node_ref[g->n_nodes].edges_to_node = 0;
    ix->n_edges = en;
    ix->seq_fwd[seq_length] = 0;

    ix->rank_prefix = (uint32_t *)malloc((seq_length + 1) * sizeof(uint32_t));
    uint32_t r = 0;
    for (uint64_t i = 0; i <= seq_length; i++) {
        r += ix->seq_bv[i];
        ix->rank_prefix[i] = r;
    }

    /* index.rs:143 + dna.rs:5-17 */
    ix->seq_rev = (char *)malloc(seq_length + 1);
    for (uint64_t i = 0; i < seq_length; i++) {
        char b = ix->seq_fwd[seq_length - 1 - i];
        if (!og_is_dna(b)) { og_index_free(ix); return OG_ERR_NOT_DNA; }
        ix->seq_rev[i] = og_switch_base(b);
    }
    ix->seq_rev[seq_length] = 0;

    /* index.rs:162-168 */
    og_kvec kv = {0};
    og_generate_kmers_parallel(g, k, max_furcations, max_degree, &kv);
    ix->gkmers = kv.a;
    ix->n_gkmers = kv.n;
    if (kv.n == 0) { og_index_free(ix); return OG_ERR_NO_KMERS; }

    /* kmer.rs:816-928: group by sequence, sort every group, flatten with a delimiter per group */
    ix->table = (og_kmerpos *)malloc((2 * kv.n + 1) * sizeof(og_kmerpos));
    ix->kmer_keys = (char *)malloc(kv.n * k + 1);
    ix->kmer_starts = (uint64_t *)malloc(kv.n * sizeof(uint64_t));
    const og_kmerpos delim = {{OG_REVERSE, UINT64_MAX}, {OG_REVERSE, UINT64_MAX}};
    uint64_t tn = 0, nk = 0;
    size_t i = 0;
    while (i < kv.n) {
        size_t j = i;
        uint64_t group_start = tn;
        while (j < kv.n && memcmp(kv.a[j].seq, kv.a[i].seq, k) == 0) {
            const og_graphkmer *km = &kv.a[j];
            uint64_t first_len = og_graph_node_len(g, OG_H_ID(km->first_handle));
            uint64_t last_len = og_graph_node_len(g, OG_H_ID(km->last_handle));
            og_kmerpos p;
            p.start.orient = km->begin_offset.orient;
            p.start.position =
                og_get_seq_pos(km->first_handle, ix->node_ref, seq_length, first_len) + km->begin_offset.position;
            p.end.orient = km->end_offset.orient;
            p.end.position =
                og_get_seq_pos(km->last_handle, ix->node_ref, seq_length, last_len) + km->end_offset.position;
            ix->table[tn++] = p;
            j++;
        }
        qsort(ix->table + group_start, tn - group_start, sizeof(og_kmerpos), og_kmerpos_qcmp);
        memcpy(ix->kmer_keys + nk * k, kv.a[i].seq, k);
        ix->kmer_starts[nk] = group_start;
        nk++;
        ix->table[tn++] = delim;
        i = j;
    }
    ix->n_kmers = nk;
    ix->n_kmer_pos = tn;
    *out = ix;
    return OG_OK;
}

void og_index_free(og_index *ix)
{
    if (!ix) return;
    free(ix->seq_fwd);
    free(ix->seq_rev);
    free(ix->seq_bv);
    free(ix->rank_prefix);
    free(ix->edges);
    free(ix->node_ref);
    free(ix->kmer_keys);
    free(ix->kmer_starts);
    free(ix->table);
    for (uint64_t i = 0; i < ix->n_gkmers; i++) free(ix->gkmers[i].seq);
    free(ix->gkmers);
    free(ix);
}

uint64_t og_index_k(const og_index *ix) { return ix->k; }
uint64_t og_index_seq_length(const og_index *ix) { return ix->seq_length; }
uint64_t og_index_n_nodes(const og_index *ix) { return ix->n_nodes; }
uint64_t og_index_n_edges(const og_index *ix) { return ix->n_edges; }
const char *og_index_seq_fwd(const og_index *ix) { return ix->seq_fwd; }
const char *og_index_seq_rev(const og_index *ix) { return ix->seq_rev; }
const uint8_t *og_index_seq_bv(const og_index *ix) { return ix->seq_bv; }
const og_handle *og_index_edges(const og_index *ix) { return ix->edges; }
const og_noderef *og_index_node_ref(const og_index *ix) { return ix->node_ref; }
uint64_t og_index_n_kmers(const og_index *ix) { return ix->n_kmers; }
uint64_t og_index_n_kmer_pos(const og_index *ix) { return ix->n_kmer_pos; }
const char *og_index_kmer_keys(const og_index *ix) { return ix->kmer_keys; }
const uint64_t *og_index_kmer_starts(const og_index *ix) { return ix->kmer_starts; }
const og_kmerpos *og_index_kmer_pos_table(const og_index *ix) { return ix->table; }
uint64_t og_index_n_graph_kmers(const og_index *ix) { return ix->n_gkmers; }

int og_index_graph_kmer(const og_index *ix, uint64_t i, og_graphkmer_view *out)
{
    if (i >= ix->n_gkmers) return OG_ERR_ARG;
    const og_graphkmer *k = &ix->gkmers[i];
    out->seq = k->seq;
    out->begin_offset = k->begin_offset;
    out->end_offset = k->end_offset;
    out->first_handle = k->first_handle;
    out->last_handle = k->last_handle;
    out->handle_orient = k->handle_orient;
    out->forks = k->forks;
    return OG_OK;
}

/* "Reference-faithful" cost mode (baseline B1): the answers never change, but the lookups also do the
 * work the reference does for them -- the linear Vec::contains over the k-mer keys before the MPHF is asked
 * (src/index.rs:319), the bit-by-bit rank / inverse rank / select loops (src/index.rs:427-480) and the clone of the whole
 * linearised sequence per node lookup (src/index.rs:516-519).  Off by default; bench.py's cpu_baseline_faithful leg turns it on. */
static int og_faithful = 0;
static volatile uint64_t og_faithful_sink = 0;
void og_set_reference_faithful_costs(int on) { og_faithful = on; }

static uint64_t og_pack_key(const char *s, uint64_t k)
{
    uint64_t v = 0;
    for (uint64_t i = 0; i < k && i < 21; i++) v = v * 5 + (uint64_t)(s[i] == 'A' ? 0 : s[i] == 'C' ? 1 : s[i] == 'G' ? 2 : s[i] == 'T' ? 3 : 4);
    return v;
}

/* index.rs:309-382 */
size_t og_index_find_positions(const og_index *ix, const char *kmer, size_t kmer_len,
                               const og_kmerpos **out)
{
    if (out) *out = NULL;
    if (kmer_len != ix->k) return 0; /* index.rs:310-312 */
    if (og_faithful) { /* kmer_pos_ref.contains(&hash): one u64 per distinct k-mer, scanned from the front (index.rs:319) */
        static const og_index *cached_ix = NULL;
        static uint64_t *cached = NULL;
        if (cached_ix != ix) {
            free(cached);
            cached = (uint64_t *)malloc((ix->n_kmers ? ix->n_kmers : 1) * sizeof(uint64_t));
            for (uint64_t i = 0; i < ix->n_kmers; i++) cached[i] = og_pack_key(ix->kmer_keys + i * ix->k, ix->k);
            cached_ix = ix;
        }
        const uint64_t want = og_pack_key(kmer, ix->k);
        uint64_t hit = 0;
        for (uint64_t i = 0; i < ix->n_kmers; i++)
            if (cached[i] == want) { hit = i + 1; break; }
        og_faithful_sink += hit;
    }
    uint64_t lo = 0, hi = ix->n_kmers;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        int c = memcmp(ix->kmer_keys + mid * ix->k, kmer, ix->k);
        if (c == 0) {
            uint64_t s = ix->kmer_starts[mid];
            uint64_t e = s;
            /* index.rs:328-348: walk to the delimiter */
            while (!(ix->table[e].start.orient == OG_REVERSE && ix->table[e].start.position == UINT64_MAX &&
                     ix->table[e].end.orient == OG_REVERSE && ix->table[e].end.position == UINT64_MAX))
                e++;
            if (out) *out = ix->table + s;
            return (size_t)(e - s);
        }
        if (c < 0) lo = mid + 1;
        else hi = mid;
    }
    return 0;
}

/* index.rs:427-439: set bits in seq_bv[0..=pos] */
uint64_t og_index_bv_rank(const og_index *ix, uint64_t pos)
{
    if (pos > ix->seq_length) return 0;
    if (og_faithful) { /* index.rs:427-439: the loop over seq_bv[0..=pos] */
        uint64_t r = 0;
        for (uint64_t i = 0; i <= pos; i++) r += ix->seq_bv[i] != 0;
        og_faithful_sink += r;
    }
    return ix->rank_prefix[pos];
}

/* index.rs:443-458: set bits in seq_bv[len-1-pos ..= len-1], len = seq_length+1 */
uint64_t og_index_bv_inverse_rank(const og_index *ix, uint64_t pos)
{
    if (pos > ix->seq_length) return 0;
    uint64_t start_point = ix->seq_length;
    uint64_t lo = start_point - pos;
    if (og_faithful) { /* index.rs:443-458: the loop over seq_bv[len-1-pos ..= len-1] */
        uint64_t r = 0;
        for (uint64_t i = lo; i <= start_point; i++) r += ix->seq_bv[i] != 0;
        og_faithful_sink += r;
    }
    uint64_t total = ix->rank_prefix[start_point];
    return total - (lo > 0 ? ix->rank_prefix[lo - 1] : 0);
}

/* index.rs:461-480 */
uint64_t og_index_bv_select(const og_index *ix, uint64_t element_no)
{
    if (element_no == 0) return 0; /* reference panics */
    if (og_faithful) { /* index.rs:461-480: walks the bit vector until the element_no-th set bit */
        uint64_t seen = 0, i = 0;
        for (; i <= ix->seq_length; i++) {
            seen += ix->seq_bv[i] != 0;
            if (seen == element_no) break;
        }
        og_faithful_sink += i;
    }
    if (element_no <= ix->n_nodes + 1) return ix->node_ref[element_no - 1].seq_idx;
    return 0; /* loop falls through with start_pos = 0 */
}

/* index.rs:388-411 */
uint64_t og_index_node_id_from_seqpos(const og_index *ix, og_seqpos p)
{
    if (p.orient == OG_FORWARD) return og_index_bv_rank(ix, p.position);
    return ix->n_nodes - og_index_bv_inverse_rank(ix, p.position) + 1;
}

/* index.rs:415-423 */
og_handle og_index_handle_from_seqpos(const og_index *ix, og_seqpos p)
{
    uint64_t id = og_index_node_id_from_seqpos(ix, p);
    return p.orient == OG_FORWARD ? id * 2 : id * 2 + 1;
}

/* index.rs:503-533 */
size_t og_index_seq_from_handle(const og_index *ix, og_handle h, char *out, size_t cap)
{
    uint64_t pos = OG_H_ID(h) - 1;
    if (OG_H_ID(h) == 0 || pos >= ix->n_nodes) return 0; /* index.rs:508-513 assert */
    uint64_t cur = ix->node_ref[pos].seq_idx, nxt = ix->node_ref[pos + 1].seq_idx;
    uint64_t start, end;
    const char *ref;
    if (!OG_H_REV(h)) {
        ref = ix->seq_fwd; start = cur; end = nxt;
    } else {
        ref = ix->seq_rev; start = ix->seq_length - nxt; end = ix->seq_length - cur;
    }
    size_t n = (size_t)(end - start);
    if (og_faithful) { /* index.rs:516-519: `self.seq_fwd.clone()` (the whole linearisation) before the substring is taken */
        char *clone = (char *)malloc(ix->seq_length + 1);
        memcpy(clone, ref, ix->seq_length);
        og_faithful_sink += (uint64_t)(unsigned char)clone[start < ix->seq_length ? start : 0];
        free(clone);
    }
    if (out && n <= cap) memcpy(out, ref + start, n);
    return n;
}

size_t og_index_edges_from_handle(const og_index *ix, og_handle h, og_handle *out, size_t cap)
{ /* index.rs:536-554 */
    uint64_t pos = OG_H_ID(h) - 1;
    uint64_t s = ix->node_ref[pos].edge_idx, e = ix->node_ref[pos + 1].edge_idx;
    for (uint64_t i = s; i < e && i - s < cap; i++) out[i - s] = ix->edges[i];
    return (size_t)(e - s);
}

size_t og_index_outgoing_edges(const og_index *ix, og_handle h, og_handle *out, size_t cap);

/* index.rs:559-579 */
size_t og_index_incoming_edges(const og_index *ix, og_handle h, og_handle *out, size_t cap)
{
    uint64_t pos = OG_H_ID(h) - 1;
    if (!OG_H_REV(h)) {
        uint64_t s = ix->node_ref[pos].edge_idx;
        uint64_t n = ix->node_ref[pos].edges_to_node;
        for (uint64_t i = 0; i < n && i < cap; i++) out[i] = ix->edges[s + i];
        return (size_t)n;
    }
    /* outgoing(flip).map(flip).rev() */
    size_t n = og_index_outgoing_edges(ix, OG_H_FLIP(h), out, cap);
    size_t m = n < cap ? n : cap;
    for (size_t i = 0; i < m; i++) out[i] = OG_H_FLIP(out[i]);
    for (size_t i = 0; i < m / 2; i++) {
        og_handle t = out[i]; out[i] = out[m - 1 - i]; out[m - 1 - i] = t;
    }
    return n;
}

/* index.rs:584-606 */
size_t og_index_outgoing_edges(const og_index *ix, og_handle h, og_handle *out, size_t cap)
{
    uint64_t pos = OG_H_ID(h) - 1;
    if (!OG_H_REV(h)) {
        uint64_t s = ix->node_ref[pos].edge_idx + ix->node_ref[pos].edges_to_node;
        uint64_t e = ix->node_ref[pos + 1].edge_idx;
        for (uint64_t i = s; i < e && i - s < cap; i++) out[i - s] = ix->edges[i];
        return (size_t)(e - s);
    }
    size_t n = og_index_incoming_edges(ix, OG_H_FLIP(h), out, cap);
    size_t m = n < cap ? n : cap;
    for (size_t i = 0; i < m; i++) out[i] = OG_H_FLIP(out[i]);
    for (size_t i = 0; i < m / 2; i++) {
        og_handle t = out[i]; out[i] = out[m - 1 - i]; out[m - 1 - i] = t;
    }
    return n;
}
