// vga_common.hpp -- internal definitions of libvga_hip (context, device buffers, event timing).
#pragma once

#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vga_hip.h"

#define VGA_ABI_VERSION 6

struct vga_dev_index {
    uint32_t k = 0;
    uint64_t seq_length = 0;
    uint64_t n_nodes = 0, n_edges = 0;
    // k-mer probe: direct-address table on the 2-bit packed k-mer.
    // table[key] = index of the group's header word pair in pos (or 0xFFFFFFFF);
    // pos[h] = {count, 0}, pos[h+1..h+count] = {target_begin, target_end} of the forward/forward
    // records in table order (src/kmer.rs:894 order, src/chain.rs:154 filter).
    uint32_t *d_table = nullptr;
    uint64_t table_entries = 0;
    uint2 *d_pos = nullptr;
    uint64_t n_pos_words = 0;
    // the same with the records of every orientation (k <= 13): bit 31 of target_begin / target_end = reverse strand
    uint32_t *d_table_all = nullptr;
    uint2 *d_pos_all = nullptr;
    // the graph itself for the device-side subgraph extraction (vga_subgraph.hip): forward sequence, node starts
    // (n_nodes + 1), per node the first edge / the number of incoming edges, the edge lists as packed handles
    char *d_seq_fwd = nullptr;
    uint32_t *d_node_start = nullptr, *d_edge_idx = nullptr, *d_edges_to = nullptr, *d_edges = nullptr;
    // host copies (the host-thread extraction VGA_SUBGRAPH=host, path handles and cs strings of the results)
    std::vector<char> seq_fwd;
    std::vector<uint32_t> node_start;  // n_nodes+1
    std::vector<uint32_t> edge_idx;    // n_nodes+1
    std::vector<uint32_t> edges_to;    // n_nodes+1
    std::vector<uint32_t> edges;       // packed handles
    bool loaded = false;
};

struct vga_timer_entry {
    const char *name;
    hipEvent_t e0, e1;
    uint64_t bytes;
    hipStream_t stream;  // the stream both events are recorded on
};

struct vga_deferred { void *dev, *pinned, *reg; size_t reg_bytes; };

struct vga_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;     // written by vga_set_error, copied out by vga_last_error, both under err_mu (two entry points of one
    std::mutex err_mu;   // context may run on two threads: vga_chain_paths_text beside vga_align_batch)
    // settings of this context (vga_ctx_set_*): a share of the GPU's memory for contexts that share a device, and the host
    // threads its calls fan out to (0: VGA_HOST_THREADS, else the hardware's, at most 32)
    double pool_fraction = 1.0;
    unsigned host_threads = 0;
    // memory that grown buffers of THIS context gave up while launches were in flight (vga_defer_release): released by the next
    // entry point that starts on this context while it is idle, and with the context
    std::mutex deferred_mu;
    std::vector<vga_deferred> deferred;
    // buffers of this context that are being allocated right now (vga_dbuf / vga_hbuf::reserve): the thread that grows the POA pool
    // in the background holds back while this is not zero -- the runtime serialises allocations, and every small hipMalloc of a call
    // otherwise waits for one of the grower's 4 GiB segments (0.1-0.25 s each while the driver clears them)
    std::atomic<int> alloc_urgent{0};
    vga_dev_index index;
    // per-call kernel timing (events recorded on `stream`)
    std::vector<vga_timer_entry> timers;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    struct agg_t {
        std::string name;
        float ms;
        uint32_t launches;
        uint64_t bytes;
        float busy_ms;
    };
    std::vector<agg_t> last_times;
    int n_cu = 256;
    // persistent, grow-only device workspaces (owned by the modules that use them): no hipMalloc/hipFree
    // in the steady state of a call
    void *map_ws = nullptr;
    void (*map_ws_free)(void *) = nullptr;
    void *poa_ws = nullptr;
    void (*poa_ws_free)(void *) = nullptr;
    void *sg_ws = nullptr;
    void (*sg_ws_free)(void *) = nullptr;
    void *gaf_ws = nullptr;
    void (*gaf_ws_free)(void *) = nullptr;
    // live read batches: vga_ctx_destroy releases their device memory and detaches them, so a batch handle may be
    // destroyed after its context
    std::vector<struct vga_batch *> batches;
};

struct vga_batch {
    vga_ctx *ctx = nullptr;  // nullptr once the context is gone (map / align calls then fail with VGA_ERR_ARG)
    int device = 0;
    uint64_t n_reads = 0;
    uint64_t total_bases = 0;
    std::vector<uint64_t> read_off;  // host copy
    std::vector<char> reads;         // host copy (subgraph extraction + cs strings need the bases)
    char *d_reads = nullptr;
    uint64_t *d_read_off = nullptr;
};

int vga_set_error(vga_ctx *ctx, int code, const char *fmt, ...);

#define VGA_HIP_CHECK(ctx, call)                                                               \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return vga_set_error((ctx), VGA_ERR_HIP, "%s failed: %s (%s:%d)", #call,           \
                                 hipGetErrorString(e_), __FILE__, __LINE__);                   \
    } while (0)

// hipFree and hipHostFree wait for every kernel that runs on the device.  A buffer that has to grow while launches are in flight
// (a sub-batch's staging beside a DP launch that runs for a second) would stall the launch path for as long: it hands its old
// memory to the list of the context the calling thread works for (vga_current_ctx: set by the entry points, inherited by the
// threads they start), which that context's next entry point releases while it is idle (vga_release_deferred) -- never another
// context's, whose launches may be in flight -- and vga_ctx_destroy.  Outside any context the memory is released at once.
void vga_defer_release(void *device_ptr, void *pinned_ptr, void *registered_ptr, size_t registered_bytes);
// the calling thread's context (if any) has an allocation in progress: vga_ctx::alloc_urgent
struct vga_alloc_urgent {
    vga_alloc_urgent();
    ~vga_alloc_urgent();
    vga_alloc_urgent(const vga_alloc_urgent &) = delete;
    vga_alloc_urgent &operator=(const vga_alloc_urgent &) = delete;
    void *ctx_;
};
void vga_release_deferred(vga_ctx *ctx);
vga_ctx *vga_current_ctx();
struct vga_ctx_scope {  // the context the calling thread works for, for the lifetime of the object
    vga_ctx *prev;
    explicit vga_ctx_scope(vga_ctx *ctx);
    ~vga_ctx_scope();
    vga_ctx_scope(const vga_ctx_scope &) = delete;
    vga_ctx_scope &operator=(const vga_ctx_scope &) = delete;
};

// grow-only device buffer
template <typename T>
struct vga_dbuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        vga_alloc_urgent urgent;
        if (p) vga_defer_release(p, nullptr, nullptr, 0);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    ~vga_dbuf() { release(); }
};

// pinned, grow-only host staging buffer.  Large ones are anonymous huge-page memory registered with the runtime: pinning what
// the process already owns costs a quarter of a hipHostMalloc of the same size (17 against 60-90 ms per 400 MB,
// tests/microbench/pinned_time.hip) and copies run at the same rate -- a process that aligns one batch and exits (the CLI)
// spent 0.2 s of its 2 s in those allocations.
template <typename T>
struct vga_hbuf {
    T *p = nullptr;
    size_t cap = 0;
    size_t mapped = 0;  // bytes of the registered mapping (0: p came from hipHostMalloc)
    void release()
    {
        if (p && mapped) { (void)hipHostUnregister(p); (void)munmap(p, mapped); }
        else if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        mapped = 0;
    }
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        vga_alloc_urgent urgent;
        if (p) vga_defer_release(nullptr, mapped ? nullptr : p, mapped ? p : nullptr, mapped);
        p = nullptr;
        cap = 0;
        mapped = 0;
        const size_t want = n + n / 8 + 64;
        const size_t bytes = (want * sizeof(T) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        static const int pin_mode = getenv("VGA_PINNED") ? atoi(getenv("VGA_PINNED")) : 0;  // diagnostics: 1 hipHostMalloc only, 2 register without huge pages
        if (bytes >= ((size_t)8 << 20) && pin_mode != 1) {
            void *q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q != MAP_FAILED) {
                if (pin_mode != 2) (void)madvise(q, bytes, MADV_HUGEPAGE);
                if (hipHostRegister(q, bytes, hipHostRegisterDefault) == hipSuccess) {
                    p = (T *)q;
                    cap = want;
                    mapped = bytes;
                    return hipSuccess;
                }
                (void)hipGetLastError();
                (void)munmap(q, bytes);
            }
        }
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    ~vga_hbuf() { release(); }
};

// host-side phase tracing (VGA_TRACE=1): prints wall-clock deltas to stderr
struct vga_trace {
    bool on;
    std::chrono::steady_clock::time_point t;
    const char *fn;
    explicit vga_trace(const char *f) : on(getenv("VGA_TRACE") != nullptr), t(std::chrono::steady_clock::now()), fn(f) {}
    void mark(const char *what)
    {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[vga-trace] %s: %-28s %9.3f ms\n", fn, what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// host thread fan-out: the current context's setting (vga_ctx_set_host_threads), else VGA_HOST_THREADS, else the hardware's
// concurrency, at most 32
unsigned vga_host_threads(uint64_t n);
template <typename F>
void vga_parallel_for(uint64_t n, F f, unsigned max_threads = 0)  // max_threads: a cap for small jobs (starting 32 threads costs ~0.7 ms)
{
    unsigned nt = vga_host_threads(n);
    if (max_threads && nt > max_threads) nt = max_threads;
    if (nt <= 1) { for (uint64_t i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> th;
    vga_ctx *const cur = vga_current_ctx();
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t]() { vga_ctx_scope scope(cur); for (uint64_t i = t; i < n; i += nt) f(i); });
    for (auto &x : th) x.join();
}

// event timing helpers (vga_ctx.hip)
void vga_timers_reset(vga_ctx *ctx);
int vga_timer_begin(vga_ctx *ctx, const char *name, uint64_t bytes, hipStream_t stream = nullptr);  // returns timer index (stream: default ctx->stream)
void vga_timer_end(vga_ctx *ctx, int idx);
void vga_timers_collect(vga_ctx *ctx);  // requires the stream to be synchronised
float vga_timer_sum(const vga_ctx *ctx, const char *prefix);
