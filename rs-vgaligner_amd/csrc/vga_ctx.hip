// vga_ctx.hip -- context, error reporting, hipEvent kernel timing, index upload, read batches.
//
// Index upload stands in for Index::load_from_file and the query-side accessors of src/index.rs
// (309-382 lookup, 388-480 rank/select, 489-606 sequence/edges).  The boomphf MPHF + ahash keys
// (src/index.rs:71,236, src/kmer.rs:931-934) are replaced by a direct-address table on the 2-bit
// packed k-mer: the reference tests exact membership (index.rs:319) before asking the MPHF, so any
// exact map gives the same answers, and 4^11 * 4 B = 16 MiB sits in the 256 MiB Infinity Cache.
#include "vga_common.hpp"

#include <mutex>

#include <algorithm>

#include <malloc.h>

int vga_set_error(vga_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) { std::lock_guard<std::mutex> lk(ctx->err_mu); ctx->err = buf; }
    return code;
}

unsigned vga_host_threads(uint64_t n)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 32) nt = 32;
    const char *e = getenv("VGA_HOST_THREADS");
    if (e && atoi(e) > 0) nt = (unsigned)atoi(e);
    if (const vga_ctx *c = vga_current_ctx())
        if (c->host_threads) nt = c->host_threads;
    if ((uint64_t)nt > n) nt = (unsigned)(n ? n : 1);
    return nt;
}

extern "C" int vga_abi_version(void) { return VGA_ABI_VERSION; }

namespace {
thread_local vga_ctx *t_current_ctx = nullptr;
void release_one(const vga_deferred &d)
{
    if (d.dev) (void)hipFree(d.dev);
    if (d.pinned) (void)hipHostFree(d.pinned);
    if (d.reg) { (void)hipHostUnregister(d.reg); (void)munmap(d.reg, d.reg_bytes); }
}
}  // namespace

vga_ctx *vga_current_ctx() { return t_current_ctx; }
vga_alloc_urgent::vga_alloc_urgent() : ctx_(t_current_ctx) { if (ctx_) ((vga_ctx *)ctx_)->alloc_urgent.fetch_add(1); }
vga_alloc_urgent::~vga_alloc_urgent() { if (ctx_) ((vga_ctx *)ctx_)->alloc_urgent.fetch_sub(1); }
vga_ctx_scope::vga_ctx_scope(vga_ctx *ctx) : prev(t_current_ctx) { t_current_ctx = ctx; }
vga_ctx_scope::~vga_ctx_scope() { t_current_ctx = prev; }

void vga_defer_release(void *device_ptr, void *pinned_ptr, void *registered_ptr, size_t registered_bytes)
{
    const vga_deferred d = {device_ptr, pinned_ptr, registered_ptr, registered_bytes};
    vga_ctx *ctx = t_current_ctx;
    if (!ctx) { release_one(d); return; }  // (no context to keep it for: the old behaviour, a release that waits for the device)
    std::lock_guard<std::mutex> lk(ctx->deferred_mu);
    ctx->deferred.push_back(d);
}

// the calling context is idle (an entry point that has not launched anything yet, or its destruction): what ITS buffers gave up
// goes back now.  Other contexts' lists are theirs: a hipFree here would wait for their launches, and stall them.
void vga_release_deferred(vga_ctx *ctx)
{
    if (!ctx) return;
    std::vector<vga_deferred> todo;
    {
        std::lock_guard<std::mutex> lk(ctx->deferred_mu);
        todo.swap(ctx->deferred);
    }
    if (todo.empty()) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != ctx->device) (void)hipSetDevice(ctx->device);
    for (const vga_deferred &d : todo) release_one(d);
    if (cur != ctx->device) (void)hipSetDevice(cur);
}

extern "C" int vga_ctx_set_pool_fraction(vga_ctx *ctx, double fraction)
{
    if (!ctx) return VGA_ERR_ARG;
    if (!(fraction > 0.0) || fraction > 1.0) return vga_set_error(ctx, VGA_ERR_ARG, "vga_ctx_set_pool_fraction: %g is not in (0, 1]", fraction);
    ctx->pool_fraction = fraction;
    return VGA_OK;
}

extern "C" int vga_ctx_set_host_threads(vga_ctx *ctx, uint32_t n_threads)
{
    if (!ctx) return VGA_ERR_ARG;
    if (n_threads > 1024) return vga_set_error(ctx, VGA_ERR_ARG, "vga_ctx_set_host_threads: %u threads", n_threads);
    ctx->host_threads = n_threads;
    return VGA_OK;
}

extern "C" const char *vga_last_error(const vga_ctx *ctx)
{
    if (!ctx) return "null ctx";
    // a copy taken under the lock, kept per calling thread: the pointer stays valid until this thread asks again
    static thread_local std::string copy;
    {
        std::lock_guard<std::mutex> lk(const_cast<vga_ctx *>(ctx)->err_mu);
        copy = ctx->err;
    }
    return copy.c_str();
}

extern "C" int vga_ctx_create(int device, vga_ctx **out)
{
    if (!out) return VGA_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device < 0 || device >= n) return VGA_ERR_NO_DEVICE;
    {
        // Every call moves gigabytes of results through malloc'd arrays.  With glibc's defaults those come from mmap and go
        // back to the kernel on free, so each call pays the page faults again (hundreds of ms per 10k-read batch).  A host
        // process that wants them kept in the heap opts in with VGA_TUNE_MALLOC=1 (the `vgaligner` CLI and bench.py do):
        // the library does not change the allocator behaviour of a process that embeds it unasked.
        static std::once_flag tuned;
        const char *tm = getenv("VGA_TUNE_MALLOC");
        if (tm && atoi(tm) != 0)
            std::call_once(tuned, []() {
                mallopt(M_MMAP_THRESHOLD, INT32_MAX);
                mallopt(M_TRIM_THRESHOLD, -1);  // (never: with a finite threshold a step that frees more than it -- 25 000 config-5 reads:
                                                // 3 x 0.6 GB of anchor arrays -- gives the pages back and faults them in again next step)
                mallopt(M_TOP_PAD, 256 << 20);
            });
    }
    vga_ctx *ctx = new vga_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return VGA_ERR_NO_DEVICE; }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return VGA_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    *out = ctx;
    return VGA_OK;
}

static void vga_index_release(vga_dev_index &ix)
{
    if (ix.d_table) (void)hipFree(ix.d_table);
    if (ix.d_pos) (void)hipFree(ix.d_pos);
    if (ix.d_table_all) (void)hipFree(ix.d_table_all);
    if (ix.d_pos_all) (void)hipFree(ix.d_pos_all);
    if (ix.d_seq_fwd) (void)hipFree(ix.d_seq_fwd);
    if (ix.d_node_start) (void)hipFree(ix.d_node_start);
    if (ix.d_edge_idx) (void)hipFree(ix.d_edge_idx);
    if (ix.d_edges_to) (void)hipFree(ix.d_edges_to);
    if (ix.d_edges) (void)hipFree(ix.d_edges);
    ix = vga_dev_index();
}

extern "C" void vga_ctx_destroy(vga_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    vga_ctx_scope scope(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (vga_batch *b : ctx->batches) {
        if (b->d_reads) (void)hipFree(b->d_reads);
        if (b->d_read_off) (void)hipFree(b->d_read_off);
        b->d_reads = nullptr;
        b->d_read_off = nullptr;
        b->ctx = nullptr;
    }
    if (ctx->map_ws && ctx->map_ws_free) ctx->map_ws_free(ctx->map_ws);
    if (ctx->poa_ws && ctx->poa_ws_free) ctx->poa_ws_free(ctx->poa_ws);
    if (ctx->sg_ws && ctx->sg_ws_free) ctx->sg_ws_free(ctx->sg_ws);
    if (ctx->gaf_ws && ctx->gaf_ws_free) ctx->gaf_ws_free(ctx->gaf_ws);
    vga_index_release(ctx->index);
    for (hipEvent_t ev : ctx->event_pool) (void)hipEventDestroy(ev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    vga_release_deferred(ctx);
    delete ctx;
}

extern "C" int vga_ctx_synchronize(vga_ctx *ctx)
{
    if (!ctx) return VGA_ERR_ARG;
    VGA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VGA_OK;
}

// ---------------------------------------------------------------- timing
static hipEvent_t vga_event_get(vga_ctx *ctx)
{
    if (ctx->events_used == ctx->event_pool.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        ctx->event_pool.push_back(ev);
    }
    return ctx->event_pool[ctx->events_used++];
}

void vga_timers_reset(vga_ctx *ctx)
{
    ctx->timers.clear();
    ctx->events_used = 0;
}

int vga_timer_begin(vga_ctx *ctx, const char *name, uint64_t bytes, hipStream_t stream)
{
    vga_timer_entry t;
    t.name = name;
    t.bytes = bytes;
    t.stream = stream ? stream : ctx->stream;
    t.e0 = vga_event_get(ctx);
    t.e1 = vga_event_get(ctx);
    if (t.e0) (void)hipEventRecord(t.e0, t.stream);
    ctx->timers.push_back(t);
    return (int)ctx->timers.size() - 1;
}

void vga_timer_end(vga_ctx *ctx, int idx)
{
    if (idx < 0 || (size_t)idx >= ctx->timers.size()) return;
    if (ctx->timers[idx].e1) (void)hipEventRecord(ctx->timers[idx].e1, ctx->timers[idx].stream);
}

void vga_timers_collect(vga_ctx *ctx)
{
    ctx->last_times.clear();
    if (ctx->timers.empty()) return;
    // launch intervals relative to the first recorded event (event pairs may sit on different streams)
    const hipEvent_t ref = ctx->timers[0].e0;
    std::vector<std::vector<std::pair<float, float>>> spans;
    for (const vga_timer_entry &t : ctx->timers) {
        float ms = 0.f, a = 0.f;
        if (t.e0 && t.e1) (void)hipEventElapsedTime(&ms, t.e0, t.e1);
        if (ref && t.e0) (void)hipEventElapsedTime(&a, ref, t.e0);
        size_t k = 0;
        for (; k < ctx->last_times.size(); k++)
            if (ctx->last_times[k].name == t.name) break;
        if (k == ctx->last_times.size()) {
            ctx->last_times.push_back({t.name, 0.f, 0u, 0ull, 0.f});
            spans.emplace_back();
        }
        ctx->last_times[k].ms += ms;
        ctx->last_times[k].launches += 1;
        ctx->last_times[k].bytes += t.bytes;
        spans[k].push_back({a, a + ms});
    }
    for (size_t k = 0; k < spans.size(); k++) {
        std::sort(spans[k].begin(), spans[k].end());
        float busy = 0.f, lo = 0.f, hi = -1.f;
        for (const auto &iv : spans[k]) {
            if (hi < lo || iv.first > hi) {
                if (hi >= lo) busy += hi - lo;
                lo = iv.first;
                hi = iv.second;
            } else if (iv.second > hi) hi = iv.second;
        }
        if (hi >= lo) busy += hi - lo;
        ctx->last_times[k].busy_ms = busy;
    }
}

float vga_timer_sum(const vga_ctx *ctx, const char *prefix)
{
    float s = 0.f;
    size_t n = strlen(prefix);
    for (const auto &a : ctx->last_times)
        if (a.name.compare(0, n, prefix) == 0) s += a.ms;
    return s;
}

extern "C" int vga_last_kernel_times(const vga_ctx *ctx, vga_kernel_time *out, int cap)
{
    if (!ctx) return 0;
    int n = (int)ctx->last_times.size();
    for (int i = 0; i < n && i < cap; i++) {
        out[i].name = ctx->last_times[i].name.c_str();
        out[i].ms = ctx->last_times[i].ms;
        out[i].launches = ctx->last_times[i].launches;
        out[i].algorithmic_bytes = ctx->last_times[i].bytes;
        out[i].busy_ms = ctx->last_times[i].busy_ms;
        out[i].reserved = 0;
    }
    return n;
}

// ---------------------------------------------------------------- index upload
static inline int vga_base_code(char c)
{
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
    }
}

static int vga_index_upload_impl(vga_ctx *ctx, const vga_index_desc *d)
{
    if (!ctx || !d) return VGA_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    if (d->kmer_length == 0 || d->kmer_length > 15)
        return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "kmer_length %u: the direct-address probe table supports 1..15",
                             d->kmer_length);
    if (d->seq_length >= (1ull << 31) || d->n_nodes >= (1ull << 30) || d->n_edges >= (1ull << 31))
        return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "graph too large for 32-bit device coordinates");
    if (!d->seq_fwd || !d->node_seq_idx || !d->node_edge_idx || !d->node_edges_to || (!d->edges && d->n_edges) ||
        !d->kmer_keys || !d->kmer_starts || !d->kmer_pos_table)
        return vga_set_error(ctx, VGA_ERR_ARG, "null array in vga_index_desc");

    vga_index_release(ctx->index);
    vga_dev_index &ix = ctx->index;
    ix.k = d->kmer_length;
    ix.seq_length = d->seq_length;
    ix.n_nodes = d->n_nodes;
    ix.n_edges = d->n_edges;
    ix.seq_fwd.assign(d->seq_fwd, d->seq_fwd + d->seq_length);
    ix.node_start.resize(d->n_nodes + 1);
    ix.edge_idx.resize(d->n_nodes + 1);
    ix.edges_to.resize(d->n_nodes + 1);
    for (uint64_t i = 0; i <= d->n_nodes; i++) {
        ix.node_start[i] = (uint32_t)d->node_seq_idx[i];
        ix.edge_idx[i] = (uint32_t)d->node_edge_idx[i];
        ix.edges_to[i] = (uint32_t)d->node_edges_to[i];
    }
    ix.edges.resize(d->n_edges);
    for (uint64_t i = 0; i < d->n_edges; i++) ix.edges[i] = (uint32_t)d->edges[i];

    // Build the probe tables on the host, then copy once: the forward/forward records only (what map_reads asks for,
    // src/map.rs:62) and, for k <= 13, every record with the orientations of its two ends in bit 31 of the positions
    // (anchors_for_query(..., only_forward = false), src/chain.rs:154-155).
    const uint32_t k = d->kmer_length;
    const uint64_t entries = 1ull << (2 * k);
    auto build = [&](bool all, std::vector<uint32_t> &table, std::vector<uint2> &pos) -> int {
        table.assign(entries, 0xFFFFFFFFu);
        pos.clear();
        pos.reserve(d->n_kmer_pos + d->n_kmers);
        for (uint64_t g = 0; g < d->n_kmers; g++) {
            const char *key = d->kmer_keys + g * k;
            uint64_t packed = 0;
            bool acgt = true;
            for (uint32_t t = 0; t < k; t++) {
                int c = vga_base_code(key[t]);
                if (c < 0) { acgt = false; break; }
                packed = (packed << 2) | (uint64_t)c;
            }
            if (!acgt)
                return vga_set_error(ctx, VGA_ERR_UNSUPPORTED,
                                     "k-mer %llu of the index holds a base outside upper-case A/C/G/T; the 2-bit probe "
                                     "table cannot represent it",
                                     (unsigned long long)g);
            uint64_t s = d->kmer_starts[g];
            if (s >= d->n_kmer_pos) return vga_set_error(ctx, VGA_ERR_ARG, "kmer_starts out of range");
            size_t header = pos.size();
            pos.push_back(make_uint2(0u, 0u));
            uint32_t cnt = 0;
            for (uint64_t e = s; e < d->n_kmer_pos; e++) {
                const vga_kmerpos &p = d->kmer_pos_table[e];
                if (p.start_orient == 1 && p.end_orient == 1 && p.start == UINT64_MAX && p.end == UINT64_MAX) break;
                if (all) {
                    pos.push_back(make_uint2((uint32_t)p.start | (p.start_orient ? 0x80000000u : 0u), (uint32_t)p.end | (p.end_orient ? 0x80000000u : 0u)));
                    cnt++;
                } else if (p.start_orient == 0 && p.end_orient == 0) {  // src/chain.rs:154
                    pos.push_back(make_uint2((uint32_t)p.start, (uint32_t)p.end));
                    cnt++;
                }
            }
            if (cnt == 0) {
                pos.pop_back();  // k-mer only on the reverse strand: a forward-only probe misses
                continue;
            }
            pos[header].x = cnt;
            if (pos.size() >= 0xFFFFFFFFull) return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "position table too large");
            table[packed] = (uint32_t)header;
        }
        if (pos.empty()) pos.push_back(make_uint2(0u, 0u));
        return VGA_OK;
    };
    std::vector<uint32_t> table;
    std::vector<uint2> pos;
    if (int rc = build(false, table, pos)) { vga_index_release(ctx->index); return rc; }
    ix.table_entries = entries;
    ix.n_pos_words = pos.size();
    VGA_HIP_CHECK(ctx, hipMalloc((void **)&ix.d_table, entries * sizeof(uint32_t)));
    VGA_HIP_CHECK(ctx, hipMalloc((void **)&ix.d_pos, pos.size() * sizeof(uint2)));
    VGA_HIP_CHECK(ctx, hipMemcpyAsync(ix.d_table, table.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    VGA_HIP_CHECK(ctx, hipMemcpyAsync(ix.d_pos, pos.data(), pos.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
    VGA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (k <= 13) {
        if (int rc = build(true, table, pos)) { vga_index_release(ctx->index); return rc; }
        VGA_HIP_CHECK(ctx, hipMalloc((void **)&ix.d_table_all, entries * sizeof(uint32_t)));
        VGA_HIP_CHECK(ctx, hipMalloc((void **)&ix.d_pos_all, pos.size() * sizeof(uint2)));
        VGA_HIP_CHECK(ctx, hipMemcpyAsync(ix.d_table_all, table.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        VGA_HIP_CHECK(ctx, hipMemcpyAsync(ix.d_pos_all, pos.data(), pos.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
        VGA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    // the graph for the device-side subgraph extraction
    const size_t nn1 = (size_t)d->n_nodes + 1;
    VGA_HIP_CHECK(ctx, hipMalloc((void **)&ix.d_seq_fwd, d->seq_length + 16));
    VGA_HIP_CHECK(ctx, hipMalloc((void **)&ix.d_node_start, nn1 * sizeof(uint32_t)));
    VGA_HIP_CHECK(ctx, hipMalloc((void **)&ix.d_edge_idx, nn1 * sizeof(uint32_t)));
    VGA_HIP_CHECK(ctx, hipMalloc((void **)&ix.d_edges_to, nn1 * sizeof(uint32_t)));
    VGA_HIP_CHECK(ctx, hipMalloc((void **)&ix.d_edges, (d->n_edges + 1) * sizeof(uint32_t)));
    VGA_HIP_CHECK(ctx, hipMemcpyAsync(ix.d_seq_fwd, ix.seq_fwd.data(), d->seq_length, hipMemcpyHostToDevice, ctx->stream));
    VGA_HIP_CHECK(ctx, hipMemcpyAsync(ix.d_node_start, ix.node_start.data(), nn1 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    VGA_HIP_CHECK(ctx, hipMemcpyAsync(ix.d_edge_idx, ix.edge_idx.data(), nn1 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    VGA_HIP_CHECK(ctx, hipMemcpyAsync(ix.d_edges_to, ix.edges_to.data(), nn1 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    if (d->n_edges)
        VGA_HIP_CHECK(ctx, hipMemcpyAsync(ix.d_edges, ix.edges.data(), d->n_edges * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    VGA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ix.loaded = true;
    return VGA_OK;
}

extern "C" int vga_index_upload(vga_ctx *ctx, const vga_index_desc *d)
{
    // nothing throws across the C ABI: an allocation failure inside becomes VGA_ERR_NOMEM
    try {
        return vga_index_upload_impl(ctx, d);
    } catch (const std::bad_alloc &) {
        return vga_set_error(ctx, VGA_ERR_NOMEM, "vga_index_upload: out of host memory");
    } catch (const std::exception &e) {
        return vga_set_error(ctx, VGA_ERR_ARG, "vga_index_upload: %s", e.what());
    }
}


// ---------------------------------------------------------------- read batches
static int vga_batch_create_impl(vga_ctx *ctx, const char *reads_concat, const uint64_t *read_off, uint64_t n_reads,
                                vga_batch **out)
{
    if (!ctx || !out || !read_off || (!reads_concat && n_reads && read_off[n_reads])) return VGA_ERR_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    for (uint64_t i = 0; i < n_reads; i++)
        if (read_off[i + 1] < read_off[i]) return vga_set_error(ctx, VGA_ERR_ARG, "read_off not monotone at %llu", (unsigned long long)i);
    vga_batch *b = new vga_batch();
    b->ctx = ctx;
    b->device = ctx->device;
    b->n_reads = n_reads;
    b->total_bases = n_reads ? read_off[n_reads] - read_off[0] : 0;
    b->read_off.resize(n_reads + 1);
    uint64_t base = n_reads ? read_off[0] : 0;
    for (uint64_t i = 0; i <= n_reads; i++) b->read_off[i] = (n_reads ? read_off[i] : 0) - base;
    b->reads.assign(reads_concat + base, reads_concat + base + b->total_bases);
    hipError_t e = hipMalloc((void **)&b->d_reads, b->total_bases + 64);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_read_off, (n_reads + 1) * sizeof(uint64_t));
    if (e == hipSuccess && b->total_bases)
        e = hipMemcpyAsync(b->d_reads, b->reads.data(), b->total_bases, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(b->d_read_off, b->read_off.data(), (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    ctx->batches.push_back(b);
    if (e != hipSuccess) {
        vga_batch_destroy(b);
        return vga_set_error(ctx, VGA_ERR_HIP, "vga_batch_create: %s", hipGetErrorString(e));
    }
    *out = b;
    return VGA_OK;
}

extern "C" int vga_batch_create(vga_ctx *ctx, const char *reads_concat, const uint64_t *read_off, uint64_t n_reads,
                                vga_batch **out)
{
    // nothing throws across the C ABI: an allocation failure inside becomes VGA_ERR_NOMEM
    try {
        return vga_batch_create_impl(ctx, reads_concat, read_off, n_reads, out);
    } catch (const std::bad_alloc &) {
        return vga_set_error(ctx, VGA_ERR_NOMEM, "vga_batch_create: out of host memory");
    } catch (const std::exception &e) {
        return vga_set_error(ctx, VGA_ERR_ARG, "vga_batch_create: %s", e.what());
    }
}


extern "C" void vga_batch_destroy(vga_batch *b)
{
    if (!b) return;
    if (b->ctx) {
        auto &v = b->ctx->batches;
        v.erase(std::remove(v.begin(), v.end(), b), v.end());
        (void)hipSetDevice(b->device);
    }
    if (b->d_reads) (void)hipFree(b->d_reads);
    if (b->d_read_off) (void)hipFree(b->d_read_off);
    delete b;
}

extern "C" void vga_map_default_params(vga_map_params *p)
{
    p->bandwidth = 50;
    p->max_gap = 1000;
    p->chain_min_n_anchors = 3;
    p->only_forward = 1;
    p->emit_dp = 1;
}

extern "C" void vga_poa_default_params(vga_poa_params *p)
{
    p->match = 2;
    p->mismatch = 4;
    p->gap_open1 = 4;
    p->gap_ext1 = 2;
    p->gap_open2 = 24;
    p->gap_ext2 = 1;
    p->wb = 10;
    p->remain_rule = VGA_REMAIN_FIRST_OUT_EDGE;  // (since round 4: include/vga_hip.h)
    p->wf = 0.01;
}
