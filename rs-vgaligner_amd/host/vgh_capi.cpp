// vgh_capi.cpp -- C wrappers over the C++ host (for ctypes-driven tests and bench.py).
#include "vgh.hpp"

#include <cstring>

using namespace vgh;

namespace {
thread_local std::string g_err;
char *dup_str(const std::string &s)
{
    char *p = (char *)malloc(s.size() + 1);
    memcpy(p, s.c_str(), s.size() + 1);
    return p;
}
struct IndexBox {
    Index ix;
    Index::DescScratch scratch;
    vga_index_desc desc;
};
}  // namespace

extern "C" {

const char *vgh_last_error(void) { return g_err.c_str(); }

// GFAParser::parse_file + HashGraph::from_gfa + Index::build (src/subcommands/index_main.rs:72-86)
void *vgh_index_build_from_gfa(const char *gfa, uint64_t k, uint64_t max_furcations, uint64_t max_degree)
{
    try {
        IndexBox *b = new IndexBox();
        b->ix = Index::build(HashGraph::from_gfa(gfa), k, max_furcations, max_degree);
        b->ix.describe(b->desc, b->scratch);
        return b;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

void *vgh_index_load(const char *path)
{
    try {
        IndexBox *b = new IndexBox();
        b->ix = Index::load(path);
        b->ix.describe(b->desc, b->scratch);
        return b;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

int vgh_index_store(void *h, const char *path)
{
    try { ((IndexBox *)h)->ix.store(path); return 0; } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

void vgh_index_free(void *h) { delete (IndexBox *)h; }
const vga_index_desc *vgh_index_desc(void *h) { return &((IndexBox *)h)->desc; }
int vgh_index_upload(void *h, vga_ctx *ctx) { return vga_index_upload(ctx, &((IndexBox *)h)->desc); }

// GAFAlignment of one aligned read (src/align.rs:1145-1167) from the fields of a vga_align_result, appended to `prefix`:
// what the driver's text threads do per read (for CPU tests of the record writer).  Returns malloc'd text (vgh_free).
char *vgh_gaf_alignment_record(const char *prefix, const char *name, uint64_t seq_len, int aligned, const uint64_t *handles, uint64_t n_handles,
                               uint32_t path_length, uint32_t path_start, uint32_t path_end, uint32_t block_length, const char *cs, const char *cigar)
{
    try {
        QuerySequence q{name, std::string((size_t)seq_len, 'A')};
        vga_align_result a;
        memset(&a, 0, sizeof a);
        uint8_t al = aligned ? 1 : 0;
        uint64_t path_off[2] = {0, n_handles}, cs_off[2] = {0, strlen(cs) + 1}, cg_off[2] = {0, strlen(cigar) + 1};
        a.n_reads = 1;
        a.aligned = &al;
        a.path_off = path_off;
        a.path_handles = const_cast<uint64_t *>(handles);
        a.path_length = &path_length; a.path_start = &path_start; a.path_end = &path_end; a.block_length = &block_length;
        a.cs_off = cs_off; a.cs = const_cast<char *>(cs);
        a.cigar_off = cg_off; a.cigar = const_cast<char *>(cigar);
        std::string out = prefix;
        gaf_from_alignment(out, q, &a, 0);
        if (out != std::string(prefix) + gaf_from_alignment(q, &a, 0)) throw Error("the appending and the returning record writers differ");
        return dup_str(out);
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

// map_reads over in-memory reads; returns 0 and malloc'd GAF texts (free with vgh_free)
int vgh_map_reads(vga_ctx *ctx, void *h, uint64_t n, const char *const *names, const char *const *seqs, uint64_t max_gap,
                  uint64_t chain_min_n_anchors, int also_align, uint64_t align_best_n, const char *out_prefix,
                  char **chains_gaf, char **alignments_gaf, uint64_t *n_aligned)
{
    try {
        std::vector<QuerySequence> in(n);
        for (uint64_t i = 0; i < n; i++) in[i] = {names[i], seqs[i]};
        MapOptions opt;
        opt.max_gap = max_gap;
        opt.chain_min_n_anchors = chain_min_n_anchors;
        opt.also_align = also_align != 0;
        opt.align_best_n = align_best_n;
        if (const char *e = getenv("VGH_CHUNK_READS")) opt.chunk_reads = strtoull(e, nullptr, 10);  // (tests)
        if (const char *e = getenv("VGH_POA_REMAIN")) opt.poa_remain_rule = atoi(e);                  // (tests)
        MapOutput o = map_reads(ctx, ((IndexBox *)h)->ix, in, opt, out_prefix ? out_prefix : "");
        if (chains_gaf) *chains_gaf = dup_str(o.chains_gaf);
        if (alignments_gaf) *alignments_gaf = dup_str(o.alignments_gaf);
        if (n_aligned) *n_aligned = o.n_aligned;
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// validation records (src/validate.rs:36-102, 127-145) of a whole alignments GAF; *out is malloc'd
int vgh_validation_records(void *h, const char *alignments_gaf, uint64_t n, const char *const *names, const char *const *seqs, char **out)
{
    try {
        std::vector<QuerySequence> in(n);
        for (uint64_t i = 0; i < n; i++) in[i] = {names[i], seqs[i]};
        const std::string g = alignments_gaf;
        std::string res;
        size_t p0 = 0;
        while (p0 < g.size()) {
            size_t p1 = g.find('\n', p0);
            if (p1 == std::string::npos) p1 = g.size();
            res += validation_record(((IndexBox *)h)->ix, g.substr(p0, p1 - p0), in);
            p0 = p1 + 1;
        }
        *out = dup_str(res);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// read_seqs_from_file: returns the number of reads or -1; names/seqs are malloc'd arrays of malloc'd strings
int64_t vgh_read_seqs_from_file(const char *path, char ***names, char ***seqs)
{
    try {
        std::vector<QuerySequence> v = read_seqs_from_file(path);
        *names = (char **)malloc((v.size() + 1) * sizeof(char *));
        *seqs = (char **)malloc((v.size() + 1) * sizeof(char *));
        for (size_t i = 0; i < v.size(); i++) { (*names)[i] = dup_str(v[i].name); (*seqs)[i] = dup_str(v[i].seq); }
        return (int64_t)v.size();
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// plan_shards: fills begin/end/slot (capacity cap entries); returns the number of shards (or the number needed when > cap)
uint64_t vgh_plan_shards(const uint64_t *lengths, uint64_t n, uint32_t n_slots, uint64_t chunk_reads, uint64_t *begin, uint64_t *end,
                         uint32_t *slot, uint64_t cap)
{
    std::vector<uint64_t> len(lengths, lengths + n);
    const std::vector<Shard> plan = plan_shards(len, n_slots, chunk_reads);
    for (uint64_t i = 0; i < plan.size() && i < cap; i++) { begin[i] = plan[i].begin; end[i] = plan[i].end; slot[i] = plan[i].slot; }
    return plan.size();
}

// map_reads_multi over in-memory reads: one context per entry of devices[] (n_devices == 0: device 0)
int vgh_map_reads_multi(void *h, uint64_t n, const char *const *names, const char *const *seqs, uint64_t max_gap,
                        uint64_t chain_min_n_anchors, int also_align, uint64_t align_best_n, const int *devices, uint32_t n_devices,
                        uint64_t chunk_reads, const char *out_prefix, char **chains_gaf, char **alignments_gaf, uint64_t *n_aligned,
                        uint64_t *n_chunks)
{
    try {
        std::vector<QuerySequence> in(n);
        for (uint64_t i = 0; i < n; i++) in[i] = {names[i], seqs[i]};
        MapOptions opt;
        opt.max_gap = max_gap;
        opt.chain_min_n_anchors = chain_min_n_anchors;
        opt.also_align = also_align != 0;
        opt.align_best_n = align_best_n;
        opt.devices.assign(devices, devices + n_devices);
        opt.chunk_reads = chunk_reads;
        if (const char *e = getenv("VGH_POA_REMAIN")) opt.poa_remain_rule = atoi(e);  // (tests)
        MapOutput o = map_reads_multi(((IndexBox *)h)->ix, in, opt, out_prefix ? out_prefix : "");
        if (chains_gaf) *chains_gaf = dup_str(o.chains_gaf);
        if (alignments_gaf) *alignments_gaf = dup_str(o.alignments_gaf);
        if (n_aligned) *n_aligned = o.n_aligned;
        if (n_chunks) *n_chunks = o.n_chunks;
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// diagnostics (tests/prof_textpath.py): GAF text + file appends of one chunk, repeated; out[0..5] = bytes, chains bytes, seconds,
// seconds in text, seconds in appends, threads
int vgh_textpath_replay(vga_ctx *ctx, void *h, uint64_t n, const char *const *names, const char *const *seqs, const char *out_prefix, uint32_t repeat,
                        uint32_t n_threads, double *out)
{
    try {
        std::vector<QuerySequence> in(n);
        for (uint64_t i = 0; i < n; i++) in[i] = {names[i], seqs[i]};
        MapOptions opt;
        opt.also_align = true;
        const TextReplay r = textpath_replay(ctx, ((IndexBox *)h)->ix, in, opt, out_prefix, repeat, n_threads);
        out[0] = (double)r.bytes; out[1] = (double)r.chains_bytes; out[2] = r.seconds; out[3] = r.text_seconds; out[4] = r.write_seconds; out[5] = r.threads;
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

void vgh_free(void *p) { free(p); }
}
