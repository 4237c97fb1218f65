/* og_chain.c -- ORACLE (test infrastructure): anchors and chaining.
 *   split_into_kmers ....... src/io.rs:41-56
 *   anchors_for_query ...... src/chain.rs:134-173
 *   score_anchor ........... src/chain.rs:274-368
 *   chain_anchors .......... src/chain.rs:370-655 (sort 386-389, DP 398-450, backtracking 455-558,
 *                            chain sort 563, placeholder 644-649)
 * Floating point: f64 throughout, same operation order as the reference, no FMA contraction
 * (compile with -ffp-contract=off); log2 and round are the host libm's, as Rust's f64::log2 /
 * f64::round are on the same machine.
 */
#include "og_internal.h"

#include <float.h>
#include <math.h>

size_t og_anchors_for_query(const og_index *ix, const char *query, size_t qlen, int only_forward,
                            og_anchor **out)
{
    size_t k = (size_t)ix->k;
    og_anchor *a = NULL;
    size_t n = 0, cap = 0;
    uint64_t id = 0;
    if (k <= qlen) { /* io.rs:47 */
        for (size_t i = 0; i + k <= qlen; i++) {
            const og_kmerpos *pos;
            size_t np = og_index_find_positions(ix, query + i, k, &pos);
            for (size_t t = 0; t < np; t++) {
                if ((only_forward && pos[t].start.orient == OG_FORWARD && pos[t].end.orient == OG_FORWARD) ||
                    !only_forward) {
                    OG_GROW(a, n, cap, og_anchor);
                    og_anchor *an = &a[n++];
                    an->id = id++;
                    an->query_begin = i;
                    an->query_end = i + k;
                    an->target_begin = pos[t].start;
                    an->target_end = pos[t].end;
                    an->max_chain_score = (double)ix->k;
                    an->best_predecessor_id = -1;
                }
            }
        }
    }
    *out = a;
    return n;
}

double og_score_anchor(const og_anchor *a, const og_anchor *b, uint64_t seed_length, uint64_t max_gap)
{
    double score = a->max_chain_score;
    if (a->query_end >= b->query_end ||
        (a->target_end.orient == b->target_end.orient && a->target_end.position >= b->target_end.position) ||
        !(a->target_end.orient == b->target_end.orient && a->target_begin.orient == b->target_begin.orient &&
          a->target_end.orient == b->target_begin.orient && a->target_begin.orient == b->target_end.orient)) {
        score = -DBL_MAX;
    } else {
        uint64_t d1 = b->query_begin - a->query_begin, d2 = b->query_end - a->query_end;
        uint64_t query_length = d1 < d2 ? d1 : d2; /* chain.rs:313 */
        uint64_t query_overlap = a->query_end > b->query_end ? a->query_end - b->query_end : 0;
        uint64_t tbd = og_seqpos_cmp(b->target_begin, a->target_begin) > 0
                           ? b->target_begin.position - a->target_begin.position
                           : a->target_begin.position - b->target_begin.position;
        uint64_t ted = og_seqpos_cmp(b->target_end, a->target_end) > 0
                           ? b->target_end.position - a->target_end.position
                           : a->target_end.position - b->target_end.position;
        uint64_t target_length = tbd < ted ? tbd : ted;
        uint64_t gap_length = query_length > target_length ? query_length - target_length
                                                           : target_length - query_length;
        if (gap_length > max_gap) {
            score = -DBL_MAX;
        } else {
            double gap_cost = 0.0;
            if (gap_length != 0)
                gap_cost = 0.01 * (double)seed_length * (double)gap_length + 0.5 * log2((double)gap_length);
            uint64_t ml = query_length < target_length ? query_length : target_length;
            if (seed_length < ml) ml = seed_length;
            score = round((a->max_chain_score + (double)ml - gap_cost) * 1000.0) / 1000.0 +
                    (double)query_overlap;
        }
    }
    return score;
}

/* stable sort by (target_end.orient DESC, target_end.position ASC): chain.rs:386-389 */
static int og_anchor_sort_less(const og_anchor *x, const og_anchor *y)
{ /* returns 1 iff y must precede x strictly */
    if (y->target_end.orient != x->target_end.orient) return y->target_end.orient > x->target_end.orient;
    return y->target_end.position < x->target_end.position;
}

static void og_anchor_msort(og_anchor *a, og_anchor *tmp, size_t n)
{
    if (n < 2) return;
    size_t m = n / 2;
    og_anchor_msort(a, tmp, m);
    og_anchor_msort(a + m, tmp, n - m);
    size_t i = 0, j = m, o = 0;
    while (i < m && j < n) {
        if (og_anchor_sort_less(&a[i], &a[j])) tmp[o++] = a[j++];
        else tmp[o++] = a[i++];
    }
    while (i < m) tmp[o++] = a[i++];
    while (j < n) tmp[o++] = a[j++];
    memcpy(a, tmp, n * sizeof(og_anchor));
}

int og_chain_anchors(og_anchor *anchors, size_t n, uint64_t seed_length, uint64_t bandwidth,
                     uint64_t max_gap, uint64_t chain_min_n_anchors, og_chain_set *out,
                     og_anchor *dp_snapshot)
{
    memset(out, 0, sizeof(*out));
    if (n > 1) {
        og_anchor *tmp = (og_anchor *)malloc(n * sizeof(og_anchor));
        og_anchor_msort(anchors, tmp, n);
        free(tmp);
    }
    double curr_max = 0.0;
    /* STEP 1, chain.rs:403-450 */
    for (size_t i = 1; i < n; i++) {
        size_t min_j = (bandwidth > (uint64_t)i) ? 0 : i - (size_t)bandwidth;
        for (size_t j = i; j-- > min_j;) {
            double p = og_score_anchor(&anchors[j], &anchors[i], seed_length, max_gap);
            if (p > anchors[i].max_chain_score) {
                anchors[i].max_chain_score = p;
                anchors[i].best_predecessor_id = (int64_t)anchors[j].id;
            }
            if (p > curr_max) curr_max = p;
        }
    }
    out->curr_max = curr_max;
    if (dp_snapshot) memcpy(dp_snapshot, anchors, n * sizeof(og_anchor));

    /* STEP 2, chain.rs:455-558 */
    size_t ccap = 0;
    if (n > 0) {
        /* id -> sorted position (the reference searches linearly, chain.rs:493) */
        size_t *pos_of_id = (size_t *)malloc(n * sizeof(size_t));
        for (size_t i = 0; i < n; i++) pos_of_id[anchors[i].id] = i;
        og_anchor *buf = (og_anchor *)malloc((n + 1) * sizeof(og_anchor));
        for (size_t i = n; i-- > 0;) {
            og_anchor *cur = &anchors[i];
            if (cur->best_predecessor_id >= 0 && cur->max_chain_score == curr_max) {
                size_t len = 0;
                while (cur->best_predecessor_id >= 0) {
                    int64_t pred = cur->best_predecessor_id;
                    cur->best_predecessor_id = -1;
                    buf[len++] = *cur;
                    cur = &anchors[pos_of_id[pred]];
                }
                buf[len++] = *cur;
                if (len >= (size_t)chain_min_n_anchors) {
                    OG_GROW(out->chains, out->n, ccap, og_chain);
                    og_chain *c = &out->chains[out->n++];
                    c->anchors = (og_anchor *)malloc(len * sizeof(og_anchor));
                    for (size_t t = 0; t < len; t++) c->anchors[t] = buf[len - 1 - t];
                    c->n = len;
                    c->is_placeholder = 0;
                }
            }
        }
        free(buf);
        free(pos_of_id);
    }
    /* STEP 3: the stable sort by score (chain.rs:563) is a no-op, every Chain.score is 0.0 */
    if (out->n == 0) { /* chain.rs:644-649 */
        OG_GROW(out->chains, out->n, ccap, og_chain);
        og_chain *c = &out->chains[out->n++];
        c->anchors = NULL;
        c->n = 0;
        c->is_placeholder = 1;
    }
    return OG_OK;
}

void og_chain_set_free(og_chain_set *cs)
{
    if (!cs) return;
    for (size_t i = 0; i < cs->n; i++) free(cs->chains[i].anchors);
    free(cs->chains);
    cs->chains = NULL;
    cs->n = 0;
}
