/* og_graph.c -- ORACLE (test infrastructure): stand-in for handlegraph 0.5.0's HashGraph and the
 * gfa 0.8.0 parser as the reference uses them (src/subcommands/index_main.rs:72-74).
 *
 * Neither crate's source is in the reference tree.  What this file restates is the behaviour
 * the reference relies on and that its own tests pin:
 *   - nodes are stored by id, each with a left-edge and a right-edge list in INSERTION order;
 *   - from_gfa inserts every S line, then every L line, then every P line, in file order;
 *   - handle_edges_iter(h, Right) on a forward handle yields the right list as stored, on a
 *     reverse handle the left list with every handle flipped (and mirrored for Left);
 *     order pinned by src/index.rs:1261-1367.
 */
#include "og_internal.h"

#include <time.h>

double og_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void og_free(void *p) { free(p); }

char og_complement(char c)
{
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'a': return 't';
    case 'c': return 'g';
    case 'g': return 'c';
    case 't': return 'a';
    default: return c; /* N and anything else map to themselves */
    }
}

og_graph *og_graph_new(void)
{
    og_graph *g = (og_graph *)calloc(1, sizeof(og_graph));
    if (g) g->min_id = UINT64_MAX;
    return g;
}

void og_graph_free(og_graph *g)
{
    if (!g) return;
    for (size_t i = 0; i < g->cap; i++) {
        free(g->nodes[i].seq);
        free(g->nodes[i].left);
        free(g->nodes[i].right);
    }
    free(g->nodes);
    for (size_t i = 0; i < g->n_paths; i++) {
        free(g->path_names[i]);
        free(g->path_steps[i]);
    }
    free(g->path_names);
    free(g->path_steps);
    free(g->path_lens);
    free(g);
}

int og_graph_add_node(og_graph *g, uint64_t id, const char *seq, size_t len)
{
    if (!g || id == 0 || id > (1ull << 40)) return OG_ERR_ARG;
    if (id >= g->cap) {
        size_t ncap = g->cap ? g->cap : 64;
        while (ncap <= id) ncap *= 2;
        g->nodes = (og_node *)realloc(g->nodes, ncap * sizeof(og_node));
        memset(g->nodes + g->cap, 0, (ncap - g->cap) * sizeof(og_node));
        g->cap = ncap;
    }
    og_node *nd = &g->nodes[id];
    if (nd->present) return OG_ERR_ARG;
    nd->seq = (char *)malloc(len + 1);
    memcpy(nd->seq, seq, len);
    nd->seq[len] = 0;
    nd->len = len;
    nd->present = 1;
    g->n_nodes++;
    if (id < g->min_id) g->min_id = id;
    if (id > g->max_id) g->max_id = id;
    return OG_OK;
}

static og_node *og_node_get(const og_graph *g, uint64_t id)
{
    if (id >= g->cap || !g->nodes[id].present) return NULL;
    return &g->nodes[id];
}

/* HashGraph::create_edge.  The duplicate test only looks at the left node's right list
 * (whatever the orientation), as the crate does. */
int og_graph_add_edge(og_graph *g, og_handle left, og_handle right)
{
    og_node *ln = og_node_get(g, OG_H_ID(left));
    og_node *rn = og_node_get(g, OG_H_ID(right));
    if (!ln || !rn) return OG_ERR_ARG;
    for (size_t i = 0; i < ln->nright; i++)
        if (ln->right[i] == right) return OG_OK;
    if (OG_H_REV(left)) {
        OG_GROW(ln->left, ln->nleft, ln->cleft, og_handle);
        ln->left[ln->nleft++] = OG_H_FLIP(right);
    } else {
        OG_GROW(ln->right, ln->nright, ln->cright, og_handle);
        ln->right[ln->nright++] = right;
    }
    if (left != OG_H_FLIP(right)) {
        rn = og_node_get(g, OG_H_ID(right));
        if (OG_H_REV(right)) {
            OG_GROW(rn->right, rn->nright, rn->cright, og_handle);
            rn->right[rn->nright++] = OG_H_FLIP(left);
        } else {
            OG_GROW(rn->left, rn->nleft, rn->cleft, og_handle);
            rn->left[rn->nleft++] = left;
        }
    }
    return OG_OK;
}

int og_graph_add_path(og_graph *g, const char *name, const og_handle *steps, size_t n)
{
    OG_GROW(g->path_names, g->n_paths, g->c_paths, char *);
    g->path_steps = (og_handle **)realloc(g->path_steps, g->c_paths * sizeof(og_handle *));
    g->path_lens = (size_t *)realloc(g->path_lens, g->c_paths * sizeof(size_t));
    g->path_names[g->n_paths] = strdup(name);
    g->path_steps[g->n_paths] = (og_handle *)malloc((n ? n : 1) * sizeof(og_handle));
    memcpy(g->path_steps[g->n_paths], steps, n * sizeof(og_handle));
    g->path_lens[g->n_paths] = n;
    g->n_paths++;
    return OG_OK;
}

size_t og_graph_n_nodes(const og_graph *g) { return g->n_nodes; }
size_t og_graph_n_paths(const og_graph *g) { return g->n_paths; }
const char *og_graph_path_name(const og_graph *g, size_t i) { return g->path_names[i]; }
size_t og_graph_path_len(const og_graph *g, size_t i) { return g->path_lens[i]; }
const og_handle *og_graph_path_steps(const og_graph *g, size_t i) { return g->path_steps[i]; }

size_t og_graph_node_len(const og_graph *g, uint64_t id)
{
    og_node *nd = og_node_get(g, id);
    return nd ? nd->len : 0;
}

size_t og_graph_sequence(const og_graph *g, og_handle h, char *out)
{
    og_node *nd = og_node_get(g, OG_H_ID(h));
    if (!nd) return 0;
    if (!OG_H_REV(h)) {
        memcpy(out, nd->seq, nd->len);
    } else {
        for (size_t i = 0; i < nd->len; i++) out[i] = og_complement(nd->seq[nd->len - 1 - i]);
    }
    return nd->len;
}

size_t og_graph_neighbors(const og_graph *g, og_handle h, int go_left, og_handle *out, size_t cap)
{
    og_node *nd = og_node_get(g, OG_H_ID(h));
    if (!nd) return 0;
    int rev = OG_H_REV(h);
    /* (Left,rev)->right list, (Left,fwd)->left list, (Right,rev)->left list, (Right,fwd)->right list */
    int use_left = (go_left != 0) != (rev != 0);
    const og_handle *lst = use_left ? nd->left : nd->right;
    size_t n = use_left ? nd->nleft : nd->nright;
    for (size_t i = 0; i < n && i < cap; i++) out[i] = rev ? OG_H_FLIP(lst[i]) : lst[i];
    return n;
}

/* ---------------- GFA1 reader ---------------- */
typedef struct {
    char *buf;
    size_t len;
} og_filebuf;

static int og_read_file(const char *path, og_filebuf *fb)
{
    FILE *f = fopen(path, "rb");
    if (!f) return OG_ERR_IO;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    fb->buf = (char *)malloc((size_t)sz + 2);
    if (fread(fb->buf, 1, (size_t)sz, f) != (size_t)sz) {
        fclose(f);
        free(fb->buf);
        return OG_ERR_IO;
    }
    fclose(f);
    fb->buf[sz] = '\n';
    fb->buf[sz + 1] = 0;
    fb->len = (size_t)sz + 1;
    return OG_OK;
}

/* split a line (in place) on tabs; returns field count */
static size_t og_split_tabs(char *line, char **fields, size_t maxf)
{
    size_t n = 0;
    char *p = line;
    while (n < maxf) {
        fields[n++] = p;
        char *t = strchr(p, '\t');
        if (!t) break;
        *t = 0;
        p = t + 1;
    }
    return n;
}

static int og_parse_u64(const char *s, uint64_t *out)
{
    if (!*s) return 0;
    uint64_t v = 0;
    for (const char *p = s; *p; p++) {
        if (*p < '0' || *p > '9') return 0;
        v = v * 10 + (uint64_t)(*p - '0');
    }
    *out = v;
    return 1;
}

int og_graph_load_gfa(const char *path, og_graph **out)
{
    og_filebuf fb;
    int rc = og_read_file(path, &fb);
    if (rc) return rc;
    og_graph *g = og_graph_new();
    /* three passes over the lines: S, L, P (HashGraph::from_gfa order) */
    for (int pass = 0; pass < 3; pass++) {
        char want = pass == 0 ? 'S' : (pass == 1 ? 'L' : 'P');
        char *copy = (char *)malloc(fb.len + 1);
        memcpy(copy, fb.buf, fb.len + 1);
        char *line = copy;
        while (line < copy + fb.len) {
            char *nl = strchr(line, '\n');
            if (!nl) break;
            *nl = 0;
            size_t ll = (size_t)(nl - line);
            if (ll && line[ll - 1] == '\r') line[ll - 1] = 0;
            if (line[0] == want && line[1] == '\t') {
                char *f[8];
                size_t nf = og_split_tabs(line, f, 8);
                if (want == 'S') {
                    uint64_t id;
                    if (nf < 3 || !og_parse_u64(f[1], &id)) { rc = OG_ERR_PARSE; }
                    else rc = og_graph_add_node(g, id, f[2], strlen(f[2]));
                } else if (want == 'L') {
                    uint64_t a, b;
                    if (nf < 5 || !og_parse_u64(f[1], &a) || !og_parse_u64(f[3], &b) ||
                        (f[2][0] != '+' && f[2][0] != '-') || (f[4][0] != '+' && f[4][0] != '-'))
                        rc = OG_ERR_PARSE;
                    else
                        rc = og_graph_add_edge(g, OG_H_PACK(a, f[2][0] == '-'), OG_H_PACK(b, f[4][0] == '-'));
                } else {
                    if (nf < 3) rc = OG_ERR_PARSE;
                    else {
                        size_t cap = 16, n = 0;
                        og_handle *steps = (og_handle *)malloc(cap * sizeof(og_handle));
                        char *p = f[2];
                        while (*p && rc == OG_OK) {
                            char *c = strchr(p, ',');
                            if (c) *c = 0;
                            size_t sl = strlen(p);
                            if (sl < 2 || (p[sl - 1] != '+' && p[sl - 1] != '-')) { rc = OG_ERR_PARSE; break; }
                            int rev = p[sl - 1] == '-';
                            p[sl - 1] = 0;
                            uint64_t id;
                            if (!og_parse_u64(p, &id)) { rc = OG_ERR_PARSE; break; }
                            OG_GROW(steps, n, cap, og_handle);
                            steps[n++] = OG_H_PACK(id, rev);
                            if (!c) break;
                            p = c + 1;
                        }
                        if (rc == OG_OK) rc = og_graph_add_path(g, f[1], steps, n);
                        free(steps);
                    }
                }
                if (rc) { free(copy); free(fb.buf); og_graph_free(g); return rc; }
            }
            line = nl + 1;
        }
        free(copy);
    }
    free(fb.buf);
    *out = g;
    return OG_OK;
}
