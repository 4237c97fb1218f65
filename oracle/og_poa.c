/* og_poa.c -- ORACLE (test infrastructure): sequence-to-graph partial order alignment.
 *
 * Stands in for ab_poa::abpoa_wrapper::AbpoaAligner::create_align_safe(nodes, edges, query, Global)
 * (reference call site src/align.rs:202; result fields consumed at src/align.rs:1107,1152-1165).
 *
 * PARITY UNPINNED: abPOA reaches the reference through the git crate ab_poa 1.5.0
 * (HopedWall/rs-abpoa, no rev in Cargo.toml:38, no source in Cargo.lock:5-15); neither the crate
 * nor abPOA's C source is in the reference tree and no reference test calls it.  This file
 * restates abPOA's PUBLISHED algorithm (Gao et al. 2021, "abPOA: an SIMD-based C library for fast
 * partial order alignment using adaptive band"):
 *   - graph of single-base nodes in topological order, virtual source and sink;
 *   - global alignment, convex (two-piece affine) gap cost, defaults M=2 X=4 O1=4 E1=2 O2=24 E2=1;
 *   - adaptive band per row r:  w = b + floor(f*qlen), b=10, f=0.01,
 *       beg = max(0,    min(max_pos_left[r],  qlen - remain[r]) - w)
 *       end = min(qlen, max(max_pos_right[r], qlen - remain[r]) + w)
 *     remain[r] = graph bases after r on ONE path to the sink -- which one is an OPEN CHOICE with first-order effects
 *     (it moves the band, hence results, and it sets the band's width, hence cost), exposed as
 *     og_poa_params.remain_rule:
 *       OG_REMAIN_LONGEST_PATH   (0) the longest path (rounds 1-2 of this repository: the reading of the paper's
 *                                    "number of remaining bases");
 *       OG_REMAIN_FIRST_OUT_EDGE (1) the path that follows, from every base, its heaviest out-edge, the first one on a
 *                                    tie -- what abPOA's public source is remembered to do (abpoa_graph.c,
 *                                    abpoa_BFS_set_node_remain: "max weight out_id", strict >); a graph that was only
 *                                    built from node strings and an edge list has unit weights everywhere, so this is
 *                                    the FIRST out-edge in edge-list order (for the virtual source: the first node
 *                                    without a predecessor).  Neither abPOA nor the rs-abpoa wrapper is in the
 *                                    reference tree, so which of the two the reference computes is UNVERIFIED.
 *     The DEFAULT is OG_REMAIN_FIRST_OUT_EDGE since round 4 (rounds 1-3: the longest path): it is what abPOA's source is
 *     remembered to compute -- by the author of this restatement and, independently, by the reviewer of rounds 2 and 3 --
 *     and the default should be the reading most likely to reproduce the reference's scores and CIGARs.
 *     After a row is filled the leftmost/rightmost column of its maximum (+1) is pushed to every successor row.
 * What abPOA leaves to its SIMD implementation (band rounding to vector width, tie order in the
 * traceback) is fixed here as this repository's specification:
 *   - H = max(M, E1, E2, F1, F2); ties resolve in that order; among predecessors the first in
 *     list order wins (list order = order of the input edge list);
 *   - a gap state prefers "open" over "extend" on a tie;
 *   - bases other than upper-case A/C/G/T score 0 against anything;
 *   - cells outside a predecessor's band read as OG_NEG; there is no clamping;
 *   - recurrences, with Ht = max(M, E1, E2) ("H before insertions"):
 *       M [r][j] = max_p H[p][j-1] + s(base_r, q_j)
 *       Ek[r][j] = max_p max(H[p][j] - (Ok+Ek), Ek[p][j] - Ek)          k = 1, 2
 *       Fk[r][j] = max_{beg <= j' < j} Ht[r][j'] - Ok - Ek*(j - j')      (OG_IDENT - Ok - Ek*j at j = beg)
 *       H [r][j] = max(Ht, F1, F2)
 *     i.e. an insertion run opens from Ht only.  Opening it from a cell whose own value came from
 *     an insertion is dominated under a convex gap cost (O1,O2 >= 0, E2 <= E1), so H is unchanged,
 *     and Fk becomes a max-plus prefix scan of the row -- the form the GPU kernel evaluates;
 *   - traceback is a state machine over {H, Ht, E1, E2, F1, F2}: leaving an insertion run by its
 *     "open" edge lands in state Ht of the cell to the left.
 */
#include "og_internal.h"

#define OG_IDENT (INT32_MIN / 2)
#define OG_NEG (-(1 << 21)) /* "minus infinity": far below any real score of a read up to ~100 kbp, small enough that the GPU kernel keeps H in 23 signed bits */

void og_poa_default_params(og_poa_params *p)
{
    p->match = 2;
    p->mismatch = 4;
    p->gap_open1 = 4;
    p->gap_ext1 = 2;
    p->gap_open2 = 24;
    p->gap_ext2 = 1;
    p->wb = 10;
    p->wf = 0.01;
    p->remain_rule = OG_REMAIN_FIRST_OUT_EDGE;  /* (the default since round 4: see the header) */
}

void og_poa_result_free(og_poa_result *r)
{
    if (!r) return;
    free(r->abpoa_nodes);
    free(r->graph_nodes);
    free(r->cigar);
    free(r->cs_string);
    memset(r, 0, sizeof(*r));
}

static inline int og_is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

static inline int32_t og_sub(const og_poa_params *p, char g, char q)
{
    if (!og_is_acgt(g) || !og_is_acgt(q)) return 0;
    return g == q ? p->match : -p->mismatch;
}

typedef struct {
    char *s;
    size_t n, cap;
} og_sb;
static void og_sb_add(og_sb *b, const char *s, size_t len)
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
static void og_sb_addu(og_sb *b, uint64_t v)
{
    char t[32];
    int n = snprintf(t, sizeof t, "%llu", (unsigned long long)v);
    og_sb_add(b, t, (size_t)n);
}
static char og_lower(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c + 32) : c; }

int og_poa_align(const char *const *nodes, const size_t *node_lens, size_t n_nodes,
                 const size_t *edge_src, const size_t *edge_dst, size_t n_edges, const char *query,
                 size_t qlen_, const og_poa_params *P, og_poa_result *out)
{
    memset(out, 0, sizeof(*out));
    if (n_nodes == 0) return OG_ERR_ARG;
    const int64_t qlen = (int64_t)qlen_;
    /* ---- rows ---- */
    size_t N = 0;
    size_t *first_row = (size_t *)malloc(n_nodes * sizeof(size_t));
    size_t *last_row = (size_t *)malloc(n_nodes * sizeof(size_t));
    for (size_t v = 0; v < n_nodes; v++) {
        if (node_lens[v] == 0) { free(first_row); free(last_row); return OG_ERR_ARG; }
        first_row[v] = N + 1;
        N += node_lens[v];
        last_row[v] = N;
    }
    for (size_t e = 0; e < n_edges; e++)
        if (edge_src[e] >= edge_dst[e] || edge_dst[e] >= n_nodes) { free(first_row); free(last_row); return OG_ERR_ARG; }

    char *row_base = (char *)malloc(N + 2);
    uint32_t *row_node = (uint32_t *)malloc((N + 2) * sizeof(uint32_t));
    for (size_t v = 0; v < n_nodes; v++)
        for (size_t t = 0; t < node_lens[v]; t++) {
            row_base[first_row[v] + t] = nodes[v][t];
            row_node[first_row[v] + t] = (uint32_t)v;
        }
    /* in / out adjacency of nodes in edge-list order */
    size_t *in_off = (size_t *)calloc(n_nodes + 1, sizeof(size_t));
    size_t *out_off = (size_t *)calloc(n_nodes + 1, sizeof(size_t));
    for (size_t e = 0; e < n_edges; e++) { in_off[edge_dst[e] + 1]++; out_off[edge_src[e] + 1]++; }
    for (size_t v = 0; v < n_nodes; v++) { in_off[v + 1] += in_off[v]; out_off[v + 1] += out_off[v]; }
    size_t *in_adj = (size_t *)malloc((n_edges ? n_edges : 1) * sizeof(size_t));
    size_t *out_adj = (size_t *)malloc((n_edges ? n_edges : 1) * sizeof(size_t));
    size_t *in_fill = (size_t *)calloc(n_nodes, sizeof(size_t));
    size_t *out_fill = (size_t *)calloc(n_nodes, sizeof(size_t));
    for (size_t e = 0; e < n_edges; e++) {
        in_adj[in_off[edge_dst[e]] + in_fill[edge_dst[e]]++] = edge_src[e];
        out_adj[out_off[edge_src[e]] + out_fill[edge_src[e]]++] = edge_dst[e];
    }
    free(in_fill);
    free(out_fill);

    /* remain[r]: graph bases after r on the path to the sink that remain_rule names (see the header) */
    if (P->remain_rule != OG_REMAIN_LONGEST_PATH && P->remain_rule != OG_REMAIN_FIRST_OUT_EDGE) {
        free(first_row); free(last_row); free(row_base); free(row_node);
        free(in_off); free(out_off); free(in_adj); free(out_adj);
        return OG_ERR_ARG;
    }
    const int first_edge = P->remain_rule == OG_REMAIN_FIRST_OUT_EDGE;
    int64_t *remain = (int64_t *)calloc(N + 2, sizeof(int64_t));
    for (size_t v = n_nodes; v-- > 0;) {
        int64_t rl = 0;
        for (size_t t = out_off[v]; t < out_off[v + 1]; t++) {
            int64_t c = 1 + remain[first_row[out_adj[t]]];
            if (first_edge) { rl = c; break; }
            if (c > rl) rl = c;
        }
        remain[last_row[v]] = rl;
        for (size_t r = last_row[v]; r-- > first_row[v];) remain[r] = remain[r + 1] + 1;
    }
    for (size_t v = 0; v < n_nodes; v++)
        if (in_off[v + 1] == in_off[v]) {
            int64_t c = 1 + remain[first_row[v]];
            if (first_edge) { remain[0] = c; break; }
            if (c > remain[0]) remain[0] = c;
        }

    const int32_t oe1 = P->gap_open1 + P->gap_ext1, e1 = P->gap_ext1;
    const int32_t oe2 = P->gap_open2 + P->gap_ext2, e2 = P->gap_ext2;
    const int64_t w = P->wb < 0 ? qlen : (int64_t)P->wb + (int64_t)(P->wf * (double)qlen);

    int64_t *mpl = (int64_t *)malloc((N + 2) * sizeof(int64_t));
    int64_t *mpr = (int64_t *)malloc((N + 2) * sizeof(int64_t));
    for (size_t r = 0; r <= N + 1; r++) { mpl[r] = INT64_MAX; mpr[r] = 0; }
    mpl[0] = 0;
    mpr[0] = 0;

    int64_t *beg = (int64_t *)malloc((N + 2) * sizeof(int64_t));
    int64_t *end = (int64_t *)malloc((N + 2) * sizeof(int64_t));
    size_t *off = (size_t *)malloc((N + 2) * sizeof(size_t));
    size_t cells = 0, ccap = 0;
    int32_t *H = NULL, *HT = NULL, *E1 = NULL, *E2 = NULL, *F1 = NULL, *F2 = NULL;

    size_t pbuf_cap = 64;
    size_t *preds = (size_t *)malloc(pbuf_cap * sizeof(size_t));
    uint64_t n_cells_rows = 0;

    /* diagnostics (OG_POA_WIDTHS=<file>): band-width statistics of every problem, appended as one line */
    const char *width_stat = getenv("OG_POA_WIDTHS");
    size_t ws_max = 0, ws_sum = 0, ws_over512 = 0, ws_over1024 = 0;
    for (size_t r = 0; r <= N; r++) {
        int64_t diag = qlen - remain[r];
        int64_t b, e;
        if (P->wb < 0) { b = 0; e = qlen; }
        else {
            int64_t lo = mpl[r] < diag ? mpl[r] : diag;
            int64_t hi = mpr[r] > diag ? mpr[r] : diag;
            b = lo - w; if (b < 0) b = 0;
            e = hi + w; if (e > qlen) e = qlen;
        }
        beg[r] = b; end[r] = e; off[r] = cells;
        size_t width = (size_t)(e - b + 1);
        if (width_stat) { if (width > ws_max) ws_max = width; ws_sum += width; if ((size_t)(e - (b & ~7)) + 1 > 512) ws_over512++; if ((size_t)(e - (b & ~7)) + 1 > 1024) ws_over1024++; }
        if (cells + width > ccap) {
            size_t nc = ccap ? ccap : (1u << 20);
            while (nc < cells + width) nc *= 2;
            H = (int32_t *)realloc(H, nc * sizeof(int32_t));
            HT = (int32_t *)realloc(HT, nc * sizeof(int32_t));
            E1 = (int32_t *)realloc(E1, nc * sizeof(int32_t));
            E2 = (int32_t *)realloc(E2, nc * sizeof(int32_t));
            F1 = (int32_t *)realloc(F1, nc * sizeof(int32_t));
            F2 = (int32_t *)realloc(F2, nc * sizeof(int32_t));
            ccap = nc;
        }
        cells += width;
        if (r > 0) n_cells_rows += width;
        int32_t *h = H + off[r], *ht = HT + off[r], *pe1 = E1 + off[r], *pe2 = E2 + off[r], *f1 = F1 + off[r], *f2 = F2 + off[r];

        /* predecessors of this row */
        size_t np = 0;
        if (r > 0) {
            uint32_t v = row_node[r];
            if (r != first_row[v]) preds[np++] = r - 1;
            else if (in_off[v + 1] == in_off[v]) preds[np++] = 0;
            else {
                size_t deg = in_off[v + 1] - in_off[v];
                if (deg > pbuf_cap) { pbuf_cap = deg * 2; preds = (size_t *)realloc(preds, pbuf_cap * sizeof(size_t)); }
                for (size_t t = in_off[v]; t < in_off[v + 1]; t++) preds[np++] = last_row[in_adj[t]];
            }
        }

        int32_t rmax = INT32_MIN;
        int64_t lmax = b, rmaxpos = b;
        /* running maxima of Ht[j'] + Ek*j' over the columns already visited (the prefix scan) */
        int have_run = 0;
        int64_t run1 = 0, run2 = 0;
        for (int64_t j = b; j <= e; j++) {
            size_t c = (size_t)(j - b);
            int32_t m = OG_NEG, ve1 = OG_NEG, ve2 = OG_NEG, vf1 = OG_NEG, vf2 = OG_NEG, vht, vh;
            if (r == 0) {
                vht = j == 0 ? 0 : OG_NEG;
            } else {
                char gb = row_base[r];
                for (size_t t = 0; t < np; t++) {
                    size_t p = preds[t];
                    if (j >= 1 && j - 1 >= beg[p] && j - 1 <= end[p]) {
                        int32_t cnd = H[off[p] + (size_t)(j - 1 - beg[p])] + og_sub(P, gb, query[j - 1]);
                        if (cnd > m) m = cnd;
                    }
                    if (j >= beg[p] && j <= end[p]) {
                        size_t q = off[p] + (size_t)(j - beg[p]);
                        int32_t a1 = H[q] - oe1, b1 = E1[q] - e1;
                        int32_t c1 = a1 > b1 ? a1 : b1;
                        if (c1 > ve1) ve1 = c1;
                        int32_t a2 = H[q] - oe2, b2 = E2[q] - e2;
                        int32_t c2 = a2 > b2 ? a2 : b2;
                        if (c2 > ve2) ve2 = c2;
                    }
                }
                vht = m;
                if (ve1 > vht) vht = ve1;
                if (ve2 > vht) vht = ve2;
            }
            /* the first column of a band has no left neighbour: its run is OG_IDENT ("minus infinity" of the scan),
             * which keeps F far below any real or OG_NEG-derived value without a special case */
            vf1 = (int32_t)((have_run ? run1 : (int64_t)OG_IDENT) - P->gap_open1 - (int64_t)e1 * j);
            vf2 = (int32_t)((have_run ? run2 : (int64_t)OG_IDENT) - P->gap_open2 - (int64_t)e2 * j);
            vh = vht;
            if (vf1 > vh) vh = vf1;
            if (vf2 > vh) vh = vf2;
            {
                int64_t a1 = (int64_t)vht + (int64_t)e1 * j, a2 = (int64_t)vht + (int64_t)e2 * j;
                if (!have_run || a1 > run1) run1 = a1;
                if (!have_run || a2 > run2) run2 = a2;
                have_run = 1;
            }
            h[c] = vh; ht[c] = vht; pe1[c] = ve1; pe2[c] = ve2; f1[c] = vf1; f2[c] = vf2;
            if (vh > rmax) { rmax = vh; lmax = j; rmaxpos = j; }
            else if (vh == rmax) rmaxpos = j;
        }
        /* push max positions to successors */
        if (r == 0) {
            for (size_t v = 0; v < n_nodes; v++)
                if (in_off[v + 1] == in_off[v]) {
                    size_t s = first_row[v];
                    if (lmax + 1 < mpl[s]) mpl[s] = lmax + 1;
                    if (rmaxpos + 1 > mpr[s]) mpr[s] = rmaxpos + 1;
                }
        } else {
            uint32_t v = row_node[r];
            if (r != last_row[v]) {
                size_t s = r + 1;
                if (lmax + 1 < mpl[s]) mpl[s] = lmax + 1;
                if (rmaxpos + 1 > mpr[s]) mpr[s] = rmaxpos + 1;
            } else {
                for (size_t t = out_off[v]; t < out_off[v + 1]; t++) {
                    size_t s = first_row[out_adj[t]];
                    if (lmax + 1 < mpl[s]) mpl[s] = lmax + 1;
                    if (rmaxpos + 1 > mpr[s]) mpr[s] = rmaxpos + 1;
                }
            }
        }
    }
    out->n_rows = N;
    out->n_cells = n_cells_rows;

    /* ---- sink: best predecessor at column qlen ---- */
    int32_t best = INT32_MIN;
    size_t best_row = 0;
    int have = 0;
    for (size_t v = 0; v < n_nodes; v++) {
        if (out_off[v + 1] != out_off[v]) continue;
        size_t p = last_row[v];
        int32_t val = (qlen >= beg[p] && qlen <= end[p]) ? H[off[p] + (size_t)(qlen - beg[p])] : OG_NEG;
        if (!have || val > best) { best = val; best_row = p; have = 1; }
    }
    out->best_score = best;
    if (!have || best <= OG_NEG / 2) {
        out->ok = 0;
        goto cleanup;
    }

    /* ---- traceback ---- */
    {
        size_t ocap = N + (size_t)qlen + 8, on = 0;
        char *ops = (char *)malloc(ocap);          /* 'M','I','D' in reverse */
        uint32_t *orow = (uint32_t *)malloc(ocap * sizeof(uint32_t));
        uint32_t *oq = (uint32_t *)malloc(ocap * sizeof(uint32_t));
        size_t i = best_row;
        int64_t j = qlen;
        int st = 0; /* 0=H 1=E1 2=E2 3=F1 4=F2 5=Ht */
        int bad = 0;
        while (i > 0 && !bad) {
            /* predecessors of row i */
            size_t np = 0;
            uint32_t v = row_node[i];
            if (i != first_row[v]) preds[np++] = i - 1;
            else if (in_off[v + 1] == in_off[v]) preds[np++] = 0;
            else for (size_t t = in_off[v]; t < in_off[v + 1]; t++) preds[np++] = last_row[in_adj[t]];
            size_t c = off[i] + (size_t)(j - beg[i]);
            if (st == 0 || st == 5) {
                int32_t hv = st == 0 ? H[c] : HT[c];
                int found = 0;
                if (j >= 1) {
                    int32_t s = og_sub(P, row_base[i], query[j - 1]);
                    for (size_t t = 0; t < np && !found; t++) {
                        size_t p = preds[t];
                        if (j - 1 >= beg[p] && j - 1 <= end[p] && H[off[p] + (size_t)(j - 1 - beg[p])] + s == hv) {
                            ops[on] = 'M'; orow[on] = (uint32_t)i; oq[on] = (uint32_t)(j - 1); on++;
                            i = p; j = j - 1; found = 1; st = 0;
                        }
                    }
                }
                if (!found) {
                    if (E1[c] == hv) st = 1;
                    else if (E2[c] == hv) st = 2;
                    else if (st == 0 && F1[c] == hv) st = 3;
                    else if (st == 0 && F2[c] == hv) st = 4;
                    else bad = 1;
                }
            } else if (st == 1 || st == 2) {
                const int32_t *E = st == 1 ? E1 : E2;
                int32_t oe = st == 1 ? oe1 : oe2, ee = st == 1 ? e1 : e2;
                int32_t ev = E[c];
                int found = 0;
                for (size_t t = 0; t < np && !found; t++) {
                    size_t p = preds[t];
                    if (j >= beg[p] && j <= end[p]) {
                        size_t q = off[p] + (size_t)(j - beg[p]);
                        int32_t a = H[q] - oe, b2 = E[q] - ee;
                        int32_t mx = a > b2 ? a : b2;
                        if (mx == ev) {
                            ops[on] = 'D'; orow[on] = (uint32_t)i; oq[on] = 0; on++;
                            if (a == ev) st = 0;
                            i = p; found = 1;
                        }
                    }
                }
                if (!found) bad = 1;
            } else {
                const int32_t *F = st == 3 ? F1 : F2;
                int32_t oe = st == 3 ? oe1 : oe2;
                int32_t fv = F[c];
                if (j - 1 < beg[i]) { bad = 1; break; }
                ops[on] = 'I'; orow[on] = 0; oq[on] = (uint32_t)(j - 1); on++;
                if (HT[c - 1] - oe == fv) st = 5;
                j = j - 1;
            }
            if (on + 2 >= ocap) { bad = 1; }
        }
        while (!bad && j > 0) { ops[on] = 'I'; orow[on] = 0; oq[on] = (uint32_t)(j - 1); on++; j--; }
        if (bad) {
            out->ok = 0;
            free(ops); free(orow); free(oq);
            goto cleanup;
        }
        /* ---- encode (forward order) ---- */
        out->ok = 1;
        size_t npath = 0;
        for (size_t t = 0; t < on; t++) if (ops[t] != 'I') npath++;
        out->n_abpoa_nodes = npath;
        out->abpoa_nodes = (uint32_t *)malloc((npath ? npath : 1) * sizeof(uint32_t));
        out->graph_nodes = (uint32_t *)malloc((npath ? npath : 1) * sizeof(uint32_t));
        og_sb cg = {0}, cs = {0};
        og_sb_add(&cs, "cs:Z:", 5);
        size_t pi = 0;
        size_t t = on;
        uint64_t eq_run = 0;
        while (t > 0) {
            char op = ops[t - 1];
            size_t run = 0;
            size_t u = t;
            while (u > 0 && ops[u - 1] == op) { u--; run++; }
            og_sb_addu(&cg, run);
            og_sb_add(&cg, &op, 1);
            if (op != 'M' && eq_run) { og_sb_add(&cs, ":", 1); og_sb_addu(&cs, eq_run); eq_run = 0; }
            if (op == 'I') og_sb_add(&cs, "+", 1);
            if (op == 'D') og_sb_add(&cs, "-", 1);
            for (size_t x = t; x > u; x--) {
                size_t idx = x - 1;
                if (op == 'M') {
                    char gb = row_base[orow[idx]], qb = query[oq[idx]];
                    out->n_aligned_bases++;
                    if (gb == qb) eq_run++;
                    else {
                        if (eq_run) { og_sb_add(&cs, ":", 1); og_sb_addu(&cs, eq_run); eq_run = 0; }
                        char tmp[3] = {'*', og_lower(gb), og_lower(qb)};
                        og_sb_add(&cs, tmp, 3);
                    }
                } else if (op == 'I') {
                    char ch = og_lower(query[oq[idx]]);
                    og_sb_add(&cs, &ch, 1);
                } else {
                    char ch = og_lower(row_base[orow[idx]]);
                    og_sb_add(&cs, &ch, 1);
                }
                if (op != 'I') {
                    out->abpoa_nodes[pi] = orow[idx];
                    out->graph_nodes[pi] = row_node[orow[idx]];
                    pi++;
                }
            }
            t = u;
        }
        if (eq_run) { og_sb_add(&cs, ":", 1); og_sb_addu(&cs, eq_run); }
        if (!cg.s) og_sb_add(&cg, "", 0);
        out->cigar = cg.s;
        out->cs_string = cs.s;
        if (npath) {
            uint32_t fr = out->abpoa_nodes[0], lr = out->abpoa_nodes[npath - 1];
            out->aln_start_offset = fr - first_row[row_node[fr]];
            out->aln_end_offset = lr - first_row[row_node[lr]] + 1;
        }
        free(ops); free(orow); free(oq);
    }

cleanup:
    free(first_row); free(last_row); free(row_base); free(row_node);
    free(in_off); free(out_off); free(in_adj); free(out_adj);
    if (width_stat) {
        FILE *wf = fopen(width_stat, "a");
        if (wf) { fprintf(wf, "rows %zu qlen %lld max %zu mean %.1f over512 %zu over1024 %zu\n", N, (long long)qlen, ws_max, (double)ws_sum / (double)(N + 1), ws_over512, ws_over1024); fclose(wf); }
    }
    free(remain); free(mpl); free(mpr); free(beg); free(end); free(off);
    free(H); free(HT); free(E1); free(E2); free(F1); free(F2); free(preds);
    return OG_OK;
}
