// vga_gaf.hip -- the path column of the chains GAF on the GPU.
//
// GAFAlignment::from_chain (src/align.rs:762-911) writes, for every anchor of a chain, the graph positions of its two ends:
//     "(>N:off,>N:off),"       N / off = node id and offset inside the node of target_begin and of the inclusive target_end
// (AnchorPosOnGraph::new, src/chain.rs:90-127: the node is the rank of the position over the node starts, the offset is
// position - start of that node).  On BASELINE's config 3 that is 4 900 anchors per read -- 405 MB of text per 10 000 reads, two
// rank look-ups and four decimal numbers per anchor -- and it was the largest piece of host work of the command line tool
// (3-4 s of one core per 10 000 reads, spread over eight threads beside the alignment call in round 2).
//
//   K6  k_gaf_chain_len    one wave per chain: length of its path field (sum over its anchors)
//       host               exclusive sum over the chains (a few thousand entries)
//   K6b k_gaf_chain_write  one wave per chain: the characters, 64 anchors at a time (wave prefix sum of the lengths)
//
// Both read the chain's anchors through chain_anchor_idx (ascending target order, as map.rs:123-133 walks them) and find a
// position's node by binary search over the index's node starts (13 probes of an L2-resident array on the HLA graphs).
#include "vga_common.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>

namespace {

struct gaf_chain {          // one chain as the kernels see it
    uint64_t a0;            // first sorted anchor of the chain's read (index into target_begin / target_end)
    uint64_t m0, m1;        // its members: chain_anchor_idx[m0 .. m1)
    uint64_t text0;         // where its text starts
};

__device__ __forceinline__ uint32_t gaf_digits(uint32_t v)
{
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
// node id (1-based) of a forward position: the last node start <= pos
__device__ __forceinline__ uint32_t gaf_node_of(const uint32_t *__restrict__ node_start, uint32_t n_nodes, uint32_t pos)
{
    uint32_t lo = 0, hi = n_nodes;  // node_start[lo] <= pos < node_start[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (node_start[mid] <= pos) lo = mid;
        else hi = mid;
    }
    return lo + 1;
}
struct gaf_anchor_txt { uint32_t fn, fo, ln, lo, len; };
__device__ __forceinline__ gaf_anchor_txt gaf_anchor(const uint32_t *__restrict__ node_start, uint32_t n_nodes, uint32_t tb, uint32_t te_incl)
{
    gaf_anchor_txt a;
    a.fn = gaf_node_of(node_start, n_nodes, tb);
    a.fo = tb - node_start[a.fn - 1];
    a.ln = gaf_node_of(node_start, n_nodes, te_incl);
    a.lo = te_incl - node_start[a.ln - 1];
    a.len = 8u + gaf_digits(a.fn) + gaf_digits(a.fo) + gaf_digits(a.ln) + gaf_digits(a.lo);  // "(>" ":" ",>" ":" "),"
    return a;
}
__device__ __forceinline__ char *gaf_put(char *w, uint32_t v)
{
    const uint32_t d = gaf_digits(v);
    for (uint32_t i = d; i-- > 0;) { w[i] = (char)('0' + v % 10u); v /= 10u; }
    return w + d;
}
__device__ __forceinline__ uint32_t gaf_wave_sum(uint32_t v)
{
    for (int s = 32; s > 0; s >>= 1) v += (uint32_t)__shfl_xor((int)v, s);
    return v;
}

__global__ __launch_bounds__(64) void k_gaf_chain_len(uint32_t n_chains, const gaf_chain *__restrict__ chains, const uint32_t *__restrict__ member,
                                                      const uint32_t *__restrict__ tb, const uint32_t *__restrict__ te,
                                                      const uint32_t *__restrict__ node_start, uint32_t n_nodes, uint64_t *__restrict__ len_out)
{
    const int lane = threadIdx.x;
    for (uint32_t c = blockIdx.x; c < n_chains; c += gridDim.x) {
        const gaf_chain ch = chains[c];
        uint32_t sum = 0;  // (a chain's path field stays far below 4 GB: 2^24 anchors at most 48 characters each would not)
        uint64_t big = 0;
        for (uint64_t t = ch.m0 + (uint64_t)lane; t < ch.m1; t += 64) {
            const uint64_t ai = ch.a0 + member[t];
            const uint32_t l = gaf_anchor(node_start, n_nodes, tb[ai], te[ai] - 1u).len;
            sum += l;
            if (sum >= (1u << 30)) { big += sum; sum = 0; }
        }
        big += sum;
        uint64_t tot = big;
        for (int s = 32; s > 0; s >>= 1) tot += (uint64_t)__shfl_xor((long long)tot, s);
        if (lane == 0) len_out[c] = tot;
    }
}

__global__ __launch_bounds__(64) void k_gaf_chain_write(uint32_t n_chains, const gaf_chain *__restrict__ chains, const uint32_t *__restrict__ member,
                                                        const uint32_t *__restrict__ tb, const uint32_t *__restrict__ te,
                                                        const uint32_t *__restrict__ node_start, uint32_t n_nodes, char *__restrict__ text)
{
    const int lane = threadIdx.x;
    for (uint32_t c = blockIdx.x; c < n_chains; c += gridDim.x) {
        const gaf_chain ch = chains[c];
        uint64_t run = ch.text0;
        for (uint64_t t0 = ch.m0; t0 < ch.m1; t0 += 64) {
            const uint64_t t = t0 + (uint64_t)lane;
            gaf_anchor_txt a = {0, 0, 0, 0, 0};
            if (t < ch.m1) {
                const uint64_t ai = ch.a0 + member[t];
                a = gaf_anchor(node_start, n_nodes, tb[ai], te[ai] - 1u);
            }
            // exclusive prefix sum of the lengths over the wave
            uint32_t incl = a.len;
            for (int s = 1; s < 64; s <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, s);
                if (lane >= s) incl += up;
            }
            const uint32_t total = (uint32_t)__shfl((int)incl, 63);
            if (t < ch.m1) {
                char *w = text + run + (incl - a.len);
                *w++ = '('; *w++ = '>';
                w = gaf_put(w, a.fn); *w++ = ':'; w = gaf_put(w, a.fo);
                *w++ = ','; *w++ = '>';
                w = gaf_put(w, a.ln); *w++ = ':'; w = gaf_put(w, a.lo);
                *w++ = ')'; *w++ = ',';
            }
            run += total;
        }
    }
}

struct gaf_ws {
    vga_dbuf<gaf_chain> d_chains;
    vga_dbuf<uint32_t> d_member, d_tb, d_te;
    vga_dbuf<uint64_t> d_len;
    vga_dbuf<char> d_text;
    vga_hbuf<gaf_chain> h_chains;
    vga_hbuf<uint64_t> h_len;
    hipStream_t st = nullptr;  // its own stream: the call may run beside vga_align_batch on the same context
    ~gaf_ws() { if (st) (void)hipStreamDestroy(st); }
};

}  // namespace

extern "C" void vga_chain_text_free(vga_chain_text *t)
{
    if (!t) return;
    free(t->text_off);
    free(t->text);
    free(t);
}

static int vga_chain_paths_text_impl(vga_ctx *ctx, const vga_map_result *m, vga_chain_text **out)
{
    if (!ctx || !m || !out) return VGA_ERR_ARG;
    *out = nullptr;
    if (!ctx->index.loaded) return vga_set_error(ctx, VGA_ERR_NO_INDEX, "vga_chain_paths_text: no index uploaded");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = nullptr;
    auto t_begin = std::chrono::steady_clock::now();
#define GAF_CHECK(call)                                                                                                                  \
    do {                                                                                                                                \
        hipError_t e_ = (call);                                                                                                         \
        if (e_ != hipSuccess) {                                                                                                         \
            if (st) (void)hipStreamSynchronize(st);                                                                                     \
            vga_chain_text_free(res);                                                                                                   \
            return vga_set_error(ctx, e_ == hipErrorOutOfMemory ? VGA_ERR_NOMEM : VGA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                 __FILE__, __LINE__);                                                                                   \
        }                                                                                                                               \
    } while (0)
    vga_chain_text *res = (vga_chain_text *)calloc(1, sizeof(vga_chain_text));
    if (!res) return vga_set_error(ctx, VGA_ERR_NOMEM, "out of host memory (chain text)");
    const uint64_t nc = m->n_chains;
    res->n_chains = nc;
    res->text_off = (uint64_t *)calloc(nc + 1, sizeof(uint64_t));
    if (!res->text_off) { vga_chain_text_free(res); return vga_set_error(ctx, VGA_ERR_NOMEM, "out of host memory (chain text offsets)"); }
    const uint64_t n_members = nc ? m->chain_anchor_off[nc] : 0;
    if (nc == 0 || n_members == 0) { *out = res; return VGA_OK; }
    if (nc >= (1ull << 31)) { vga_chain_text_free(res); return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "too many chains in one call"); }
    if (!ctx->gaf_ws) {
        ctx->gaf_ws = new gaf_ws();
        ctx->gaf_ws_free = [](void *q) { delete (gaf_ws *)q; };
    }
    gaf_ws &W = *(gaf_ws *)ctx->gaf_ws;
    if (!W.st) GAF_CHECK(hipStreamCreateWithFlags(&W.st, hipStreamNonBlocking));
    st = W.st;
    GAF_CHECK(W.h_chains.reserve(nc)); GAF_CHECK(W.d_chains.reserve(nc)); GAF_CHECK(W.h_len.reserve(nc)); GAF_CHECK(W.d_len.reserve(nc));
    GAF_CHECK(W.d_member.reserve(n_members)); GAF_CHECK(W.d_tb.reserve(m->n_anchors + 1)); GAF_CHECK(W.d_te.reserve(m->n_anchors + 1));
    {
        uint64_t c = 0;
        for (uint64_t r = 0; r < m->n_reads; r++)
            for (c = m->chain_off[r]; c < m->chain_off[r + 1]; c++)
                W.h_chains.p[c] = {m->anchor_off[r], m->chain_anchor_off[c], m->chain_placeholder[c] ? m->chain_anchor_off[c] : m->chain_anchor_off[c + 1], 0};
    }
    GAF_CHECK(hipMemcpyAsync(W.d_member.p, m->chain_anchor_idx, n_members * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GAF_CHECK(hipMemcpyAsync(W.d_tb.p, m->target_begin, m->n_anchors * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GAF_CHECK(hipMemcpyAsync(W.d_te.p, m->target_end, m->n_anchors * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GAF_CHECK(hipMemcpyAsync(W.d_chains.p, W.h_chains.p, nc * sizeof(gaf_chain), hipMemcpyHostToDevice, st));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(nc, 32ull * (uint64_t)ctx->n_cu);
    const vga_dev_index &ix = ctx->index;
    hipLaunchKernelGGL(k_gaf_chain_len, dim3(grid), dim3(64), 0, st, (uint32_t)nc, W.d_chains.p, W.d_member.p, W.d_tb.p, W.d_te.p, ix.d_node_start,
                       (uint32_t)ix.n_nodes, W.d_len.p);
    GAF_CHECK(hipGetLastError());
    GAF_CHECK(hipMemcpyAsync(W.h_len.p, W.d_len.p, nc * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    GAF_CHECK(hipStreamSynchronize(st));
    uint64_t tot = 0;
    for (uint64_t c = 0; c < nc; c++) { res->text_off[c] = tot; W.h_chains.p[c].text0 = tot; tot += W.h_len.p[c]; }
    res->text_off[nc] = tot;
    GAF_CHECK(W.d_text.reserve(tot + 64));
    // (pageable: the runtime stages the copy at ~20 GB/s; a pinned buffer of this size costs more to allocate than that)
    res->text = (char *)malloc(tot + 64);
    if (!res->text) { vga_chain_text_free(res); return vga_set_error(ctx, VGA_ERR_NOMEM, "out of host memory (chain text, %llu bytes)", (unsigned long long)tot); }
    GAF_CHECK(hipMemcpyAsync(W.d_chains.p, W.h_chains.p, nc * sizeof(gaf_chain), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gaf_chain_write, dim3(grid), dim3(64), 0, st, (uint32_t)nc, W.d_chains.p, W.d_member.p, W.d_tb.p, W.d_te.p, ix.d_node_start,
                       (uint32_t)ix.n_nodes, W.d_text.p);
    GAF_CHECK(hipGetLastError());
    GAF_CHECK(hipMemcpyAsync(res->text, W.d_text.p, tot, hipMemcpyDeviceToHost, st));
    GAF_CHECK(hipStreamSynchronize(st));
    res->ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
#undef GAF_CHECK
    *out = res;
    return VGA_OK;
}

extern "C" int vga_chain_paths_text(vga_ctx *ctx, const vga_map_result *chains, vga_chain_text **out)
{
    try {
        vga_ctx_scope scope(ctx);  // (buffers that grow here give their memory to this context's list; map / align entry points release it)
        return vga_chain_paths_text_impl(ctx, chains, out);
    } catch (const std::bad_alloc &) {
        return vga_set_error(ctx, VGA_ERR_NOMEM, "vga_chain_paths_text: out of host memory");
    }
}
