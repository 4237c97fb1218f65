// vga_map.hip -- anchors + chaining on gfx950.
//
//   K1  k_kmer_probe<false/true>  split_into_kmers + find_positions_for_query_kmer + the forward
//                                 filter and id assignment of anchors_for_query
//                                 (src/io.rs:41-56, src/index.rs:309-382, src/chain.rs:134-173)
//   K2  k_anchor_sort             the stable sort of chain_anchors (src/chain.rs:386-389)
//   K3  k_chain4                  the windowed DP (src/chain.rs:398-450, score_anchor 274-368) and
//                                 the backtracking into chains (src/chain.rs:455-558)
//
// All of it is integer / f64 latency-bound work on a few hundred KB per read; it is laid out so
// that every global access is coalesced or a wave-uniform broadcast, with one block (K1, K2) or one
// wavefront (K3) per read.  Wave width is 64 throughout.
//
// f64 parity (chain scores must match the CPU bit for bit): the gap cost
// 0.01*k*g + 0.5*log2(g) is tabulated on the host with the host libm for g in [0, max_gap]
// (g is an integer, src/chain.rs:338-344), and the remaining + - * round / are IEEE operations
// evaluated in the reference's order; this file is compiled with -ffp-contract=off.
#include "vga_common.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>

#define VGA_WAVE 64
#define VGA_PROBE_NT 256
#define VGA_SORT_NT 256
#define VGA_NONE32 0xFFFFFFFFu

// ------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ uint32_t vga_lane() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t vga_wave_incl_scan(uint32_t v)
{
    uint32_t lane = vga_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}

// block-wide exclusive scan for 256 threads; returns the exclusive prefix and the block total.
// `ws` must hold 4 words; two barriers.
__device__ __forceinline__ uint32_t vga_block_excl_scan_256(uint32_t v, uint32_t *ws, uint32_t &total)
{
    uint32_t incl = vga_wave_incl_scan(v);
    uint32_t w = threadIdx.x >> 6;
    if (vga_lane() == 63) ws[w] = incl;
    __syncthreads();
    uint32_t w0 = ws[0], w1 = ws[1], w2 = ws[2], w3 = ws[3];
    uint32_t before = (w > 0 ? w0 : 0) + (w > 1 ? w1 : 0) + (w > 2 ? w2 : 0);
    total = w0 + w1 + w2 + w3;
    __syncthreads();
    return before + incl - v;
}

__device__ __forceinline__ uint32_t vga_base_code_dev(uint8_t c)
{
    // upper-case A/C/G/T -> 0..3, everything else 4 (a query k-mer holding it misses the table,
    // exactly as its string would miss the reference's k-mer set)
    return c == 'A' ? 0u : (c == 'C' ? 1u : (c == 'G' ? 2u : (c == 'T' ? 3u : 4u)));
}

// ------------------------------------------------------------------------------------------ K1
// One block per read.  Pass 1 (EMIT=false) counts the anchors of each read; pass 2 (EMIT=true)
// writes them at anchor_off[r] in (query position, table order) order, which is the order
// anchors_for_query assigns ids in (src/chain.rs:146-166): anchor id == index within the read.
template <bool EMIT>
__global__ __launch_bounds__(VGA_PROBE_NT) void k_kmer_probe(
    const char *__restrict__ reads, const uint64_t *__restrict__ read_off, uint32_t k,
    const uint32_t *__restrict__ table, const uint2 *__restrict__ pos, uint32_t *__restrict__ cnt_out,
    const uint64_t *__restrict__ anchor_off, uint32_t *__restrict__ a_qb, uint32_t *__restrict__ a_tb,
    uint32_t *__restrict__ a_te, uint32_t *__restrict__ a_idx)
{
    __shared__ uint8_t codes[VGA_PROBE_NT + 32];
    __shared__ uint32_t ws[4];
    const uint32_t r = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint64_t off = read_off[r];
    const uint64_t L = read_off[r + 1] - off;
    if (L < k) {  // src/io.rs:47: no k-mers
        if (!EMIT && tid == 0) cnt_out[r] = 0;
        return;
    }
    const uint64_t nk = L - k + 1;
    const uint64_t abase = EMIT ? anchor_off[r] : 0;
    uint32_t running = 0;
    for (uint64_t c0 = 0; c0 < nk; c0 += VGA_PROBE_NT) {
        for (uint32_t t = tid; t < VGA_PROBE_NT + k - 1; t += VGA_PROBE_NT) {
            uint64_t p = c0 + t;
            codes[t] = p < L ? (uint8_t)vga_base_code_dev((uint8_t)reads[off + p]) : (uint8_t)4;
        }
        __syncthreads();
        const uint64_t i = c0 + tid;
        uint32_t cnt = 0, hdr = VGA_NONE32;
        if (i < nk) {
            uint32_t key = 0, bad = 0;
            for (uint32_t t = 0; t < k; t++) {
                uint32_t c = codes[tid + t];
                bad |= c >> 2;
                key = (key << 2) | (c & 3u);
            }
            if (!bad) {
                hdr = table[key];
                if (hdr != VGA_NONE32) cnt = pos[hdr].x;
            }
        }
        uint32_t total;
        uint32_t excl = vga_block_excl_scan_256(cnt, ws, total);  // two barriers: also protects `codes`
        if (EMIT && cnt) {
            uint64_t o = abase + running + excl;
            for (uint32_t t = 0; t < cnt; t++) {
                uint2 rec = pos[hdr + 1 + t];
                a_qb[o + t] = (uint32_t)i;
                a_tb[o + t] = rec.x;
                a_te[o + t] = rec.y;
                a_idx[o + t] = running + excl + t;
            }
        }
        running += total;
    }
    if (!EMIT && tid == 0) cnt_out[r] = running;
}

// ------------------------------------------------------------------------------------------ K2
// One block per read: stable LSD radix sort (8-bit digits) of (key = target_end, value = anchor id)
// through global ping-pong buffers.  Stability + ascending original ids reproduce Vec::sort_by.
__global__ __launch_bounds__(VGA_SORT_NT) void k_anchor_sort(const uint64_t *__restrict__ anchor_off,
                                                             uint32_t n_pass, uint32_t *key_a, uint32_t *val_a,
                                                             uint32_t *key_b, uint32_t *val_b)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t base[256];
    __shared__ uint32_t wave_cnt[4][256];
    __shared__ uint32_t ws[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint64_t a0 = anchor_off[blockIdx.x];
    const uint32_t A = (uint32_t)(anchor_off[blockIdx.x + 1] - a0);
    if (A < 2) {
        // nothing to move; make the buffer the host will read (a or b by pass parity) valid
        if ((n_pass & 1u) && A == 1 && tid == 0) { key_b[a0] = key_a[a0]; val_b[a0] = val_a[a0]; }
        return;
    }
    uint32_t *kin = key_a + a0, *vin = val_a + a0, *kout = key_b + a0, *vout = val_b + a0;
    for (uint32_t pass = 0; pass < n_pass; pass++) {
        const uint32_t shift = pass * 8;
        hist[tid] = 0;
        for (int q = 0; q < 4; q++) wave_cnt[q][tid] = 0;
        __syncthreads();
        for (uint32_t e = tid; e < A; e += VGA_SORT_NT) atomicAdd(&hist[(kin[e] >> shift) & 255u], 1u);
        __syncthreads();
        uint32_t total;
        uint32_t ex = vga_block_excl_scan_256(hist[tid], ws, total);
        base[tid] = ex;
        __syncthreads();
        for (uint32_t t0 = 0; t0 < A; t0 += VGA_SORT_NT) {
            const uint32_t e = t0 + tid;
            const bool valid = e < A;
            uint32_t key = 0, val = 0, d = 0;
            if (valid) { key = kin[e]; val = vin[e]; d = (key >> shift) & 255u; }
            // lanes of this wave holding the same digit
            uint64_t m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                uint64_t bal = __ballot((d >> b) & 1u);
                m &= ((d >> b) & 1u) ? bal : ~bal;
            }
            const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            const uint32_t rank = __popcll(m & lt);
            if (valid && rank == 0) wave_cnt[w][d] = __popcll(m);
            __syncthreads();
            if (valid) {
                uint32_t o = base[d] + rank;
                for (uint32_t q = 0; q < w; q++) o += wave_cnt[q][d];
                kout[o] = key;
                vout[o] = val;
            }
            __syncthreads();
            base[tid] += wave_cnt[0][tid] + wave_cnt[1][tid] + wave_cnt[2][tid] + wave_cnt[3][tid];
            for (int q = 0; q < 4; q++) wave_cnt[q][tid] = 0;
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        uint32_t *t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
}

// sort keys of the both-orientations order: reverse-strand ends (bit 31 set) come first
__global__ __launch_bounds__(256) void k_flip_orient_bit(uint32_t *key, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) key[i] ^= 0x80000000u;
}

// one block per read: gather sorted fields (perm holds read-local anchor ids)
__global__ __launch_bounds__(256) void k_anchor_gather_seg(const uint64_t *__restrict__ anchor_off,
                                                           const uint32_t *__restrict__ perm,
                                                           const uint32_t *__restrict__ a_qb,
                                                           const uint32_t *__restrict__ a_tb,
                                                           const uint32_t *__restrict__ a_te,
                                                           uint32_t *__restrict__ s_qb, uint32_t *__restrict__ s_tb,
                                                           uint32_t *__restrict__ s_te)
{
    const uint64_t a0 = anchor_off[blockIdx.x];
    const uint32_t A = (uint32_t)(anchor_off[blockIdx.x + 1] - a0);
    for (uint32_t i = threadIdx.x; i < A; i += blockDim.x) {
        uint32_t p = perm[a0 + i];
        s_qb[a0 + i] = a_qb[a0 + p];
        s_tb[a0 + i] = a_tb[a0 + p];
        s_te[a0 + i] = a_te[a0 + p];
    }
}


// K3.  What sets the latency of one step of the serial chain:
//   * the argmax over the window is a DPP max-reduction of the f64 scores (18 VALU instructions, result in lane 63;
//     round 1's first form used a 6-stage butterfly of 18 ds_bpermute); the winning j follows from the ballot of the lanes that hold
//     the maximum: lane l keeps the most recent anchor j = l (mod 64), so the largest j is the set lane cyclically
//     nearest below (i - 1) & 63 -- a rotate and a count-leading-zeros on the scalar unit;
//   * the gap-cost table sits in LDS (shared by the four reads of a workgroup) instead of a per-lane gather from HBM;
//   * anchors are loaded 64 at a time (lane l holds anchor base + l) and read per step with v_readlane, and f(i) /
//     predecessors go out 64 at a time, coalesced, with the predecessor ids gathered in parallel -- no global load or
//     store on the per-anchor chain.
// One wavefront per read, four reads per workgroup.
__device__ __forceinline__ double vga_wave_max_f64_to_lane63(double v)
{
    const int ident_lo = (int)0xffffffffu, ident_hi = (int)0xffefffffu;  // -f64::MAX
#define VGA_STAGE(ctrl, rmask)                                                                                    \
    {                                                                                                             \
        const int lo = __builtin_amdgcn_update_dpp(ident_lo, __double2loint(v), ctrl, rmask, 0xf, false);         \
        const int hi = __builtin_amdgcn_update_dpp(ident_hi, __double2hiint(v), ctrl, rmask, 0xf, false);         \
        const double t = __hiloint2double(hi, lo);                                                                \
        v = t > v ? t : v;                                                                                        \
    }
    VGA_STAGE(0x111, 0xf)  // row_shr:1
    VGA_STAGE(0x112, 0xf)  // row_shr:2
    VGA_STAGE(0x114, 0xf)  // row_shr:4
    VGA_STAGE(0x118, 0xf)  // row_shr:8
    VGA_STAGE(0x142, 0xa)  // row_bcast:15 -> rows 1, 3
    VGA_STAGE(0x143, 0xc)  // row_bcast:31 -> rows 2, 3
#undef VGA_STAGE
    return v;
}

__device__ __forceinline__ int vga_wave_max_i32_to_lane63(int v)
{
#define VGA_STAGE(ctrl, rmask)                                                                                    \
    {                                                                                                             \
        const int t = __builtin_amdgcn_update_dpp(INT32_MIN, v, ctrl, rmask, 0xf, false);                         \
        v = t > v ? t : v;                                                                                        \
    }
    VGA_STAGE(0x111, 0xf)
    VGA_STAGE(0x112, 0xf)
    VGA_STAGE(0x114, 0xf)
    VGA_STAGE(0x118, 0xf)
    VGA_STAGE(0x142, 0xa)
    VGA_STAGE(0x143, 0xc)
#undef VGA_STAGE
    return v;
}

// The scores of a step's candidates are round(1000 s) / 1000 (src/chain.rs:362-366): the argmax runs on the integers round(1000 s)
// when every score of the read is known to fit 32 bits (IKEY: k (A + 2) + gap_cost[max_gap] < 2^31 / 1000, decided per read from
// key_anchors) -- six v_max_i32_dpp instead of six stages of two DPP moves, an f64 compare and two selects, and the division by
// 1000 once per step, for the winner.  x -> x / 1000 is strictly increasing on these integers (neighbours are 10^-3 apart, an ulp
// is below 10^-9), so the maximum, the lanes that hold it and f(i) = max / 1000 + 0.0 are what the f64 reduction gives, bit for bit.
template <bool GAP_LDS, bool IKEY>
__device__ __forceinline__ void vga_chain_dp(uint32_t lane, uint64_t a0, uint32_t A, const uint32_t *__restrict__ s_id,
                                             const uint32_t *__restrict__ s_qb, const uint32_t *__restrict__ s_tb,
                                             const uint32_t *__restrict__ s_te, uint32_t k, uint32_t bandwidth, uint64_t max_gap,
                                             const double *__restrict__ gap_cost, const double *s_gap, double *__restrict__ f_out,
                                             int32_t *__restrict__ pred_id_out, int32_t *pred_pos, double &curr_max)
{
    const double kd = (double)k;
    const double NEGMAX = -1.7976931348623157e308;  // -f64::MAX
    // gaps are differences of 32-bit coordinates: compared with min(max_gap, 2^32 - 1) in 32 bits
    const uint32_t max_gap32 = max_gap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)max_gap;
    const int k1000 = (int)(k * 1000u);  // (IKEY) f(i) = k is beaten exactly by the integers above 1000 k
    int ckey = 0;                        // (IKEY) the running maximum as an integer: curr_max starts at 0.0 (src/chain.rs:400)
    double f_l = 0.0;
    uint32_t qb_l = 0, tb_l = 0, te_l = 0;
    int j_l = -1, pj_l = -1;
    // x / 1000.0 without the division sequence (v_rcp_f64 and ten more): q = x c, r = fma(-q, 1000, x), q + r c with c = 1 / 1000 is
    // the correctly rounded quotient for EVERY 32-bit integer x (tests/test_chain_arith_cpu.py checks all of them on the host)
    auto div1000 = [](int x) -> double {
        const double xd = (double)x, c = 1.0 / 1000.0;
        const double q = xd * c;
        const double r = __builtin_fma(-q, 1000.0, xd);
        return __builtin_fma(r, c, q) + 0.0;
    };

    for (uint32_t base = 0; base < A; base += 64) {
        const uint32_t nb = A - base < 64 ? A - base : 64;
        uint32_t cq = 0, ct = 0, ce = 0;  // anchor base + lane
        if (lane < nb) { cq = s_qb[a0 + base + lane]; ct = s_tb[a0 + base + lane]; ce = s_te[a0 + base + lane]; }
        for (uint32_t ii = 0; ii < nb; ii++) {
            const uint32_t i = base + ii;
            const uint32_t qbi = (uint32_t)__builtin_amdgcn_readlane((int)cq, (int)ii);
            const uint32_t tbi = (uint32_t)__builtin_amdgcn_readlane((int)ct, (int)ii);
            const uint32_t tei = (uint32_t)__builtin_amdgcn_readlane((int)ce, (int)ii);
            double p = NEGMAX;
            int j = -1;
            bool take = false;  // (IKEY) the step's best candidate beats f(i) = k
            int kbest = INT32_MIN;
            if (i > 0) {
                const int min_j = (bandwidth > i) ? 0 : (int)(i - bandwidth);  // src/chain.rs:404-407
                // score_anchor(a = j_l, b = i), src/chain.rs:274-368, as ONE predicate over cheap integer work (the nested form
                // cost four exec-mask round trips per step).  Bit 31 of a target coordinate is its orientation (always 0 with
                // only_forward): the four ends must agree (chain.rs:280-283), then positions compare
                const uint32_t ob = tei >> 31;
                const uint32_t ql = qbi - qb_l;
                const uint32_t tbd = tbi > tb_l ? tbi - tb_l : tb_l - tbi;
                const uint32_t ted = tei - te_l;
                const uint32_t tl = tbd < ted ? tbd : ted;
                const uint32_t g = ql > tl ? ql - tl : tl - ql;
                const bool ok = j_l >= min_j && (tb_l >> 31) == ob && (te_l >> 31) == ob && (tbi >> 31) == ob && qb_l < qbi && te_l < tei &&
                                g <= max_gap32;
                int key = INT32_MIN;  // (IKEY) round(1000 s) of this lane's candidate; no real score reaches INT32_MIN
                if (ok) {
                    const double gc = GAP_LDS ? s_gap[g] : gap_cost[g];
                    uint32_t ml = ql < tl ? ql : tl;
                    if (k < ml) ml = k;
                    double s = f_l + (double)ml;
                    s = s - gc;
                    s = s * 1000.0;
                    s = round(s);
                    if constexpr (IKEY) key = (int)s;
                    else {
                        s = s / 1000.0;
                        s = s + 0.0;
                        p = s;
                    }
                }
                // the maximum, then the largest j among the lanes that hold it (src/chain.rs:417,430: the scan runs from
                // i-1 downwards with a strict '>')
                uint64_t mask;
                if constexpr (IKEY) {
                    kbest = __builtin_amdgcn_readlane(vga_wave_max_i32_to_lane63(key), 63);
                    mask = kbest != INT32_MIN ? __ballot(key == kbest) : 0ull;
                } else {
                    const double red = vga_wave_max_f64_to_lane63(p);
                    const double pmax = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(red), 63),
                                                         __builtin_amdgcn_readlane(__double2loint(red), 63));
                    mask = __ballot(ok && p == pmax);
                    p = pmax;
                }
                if (mask) {
                    const uint32_t sft = 63u - ((i - 1u) & 63u);  // lane (i-1) & 63 -> bit 63
                    const uint64_t rot = sft ? ((mask << sft) | (mask >> (64u - sft))) : mask;
                    const int d = __builtin_clzll(rot);  // 0 for the lane of anchor i-1
                    j = (int)i - 1 - d;
                    if constexpr (IKEY) {
                        // p > k and p > curr_max are decided on the integers (x -> x / 1000 is strictly increasing, 1000 k / 1000 = k):
                        // scalar compares instead of f64 compares and selects in every lane
                        take = kbest > k1000;
                        ckey = kbest > ckey ? kbest : ckey;
                    }
                } else p = NEGMAX;
            }
            double fi = kd;  // src/chain.rs:163: initial f(i) = k
            int pj = -1;
            if constexpr (IKEY) {
                if (take) { fi = div1000(kbest); pj = j; }
            } else {
                if (p > fi) { fi = p; pj = j; }
                if (p > curr_max) curr_max = p;
            }
            if (lane == ii) { f_l = fi; qb_l = cq; tb_l = ct; te_l = ce; j_l = (int)i; pj_l = pj; }
        }
        // lane l now holds f and the predecessor of anchor base + l
        if (lane < nb) {
            f_out[a0 + base + lane] = f_l;
            pred_pos[a0 + base + lane] = pj_l;
            pred_id_out[a0 + base + lane] = pj_l >= 0 ? (int32_t)s_id[a0 + pj_l] : -1;
        }
    }
    if constexpr (IKEY) curr_max = ckey > 0 ? div1000(ckey) : 0.0;
}

template <bool GAP_LDS>
__global__ __launch_bounds__(256) void k_chain4(
    uint32_t R, const uint64_t *__restrict__ anchor_off, const uint32_t *__restrict__ s_id, const uint32_t *__restrict__ s_qb,
    const uint32_t *__restrict__ s_tb, const uint32_t *__restrict__ s_te, uint32_t k, uint32_t bandwidth,
    uint64_t max_gap, uint32_t min_anchors, const double *__restrict__ gap_cost, double *__restrict__ f_out,
    int32_t *__restrict__ pred_id_out, int32_t *pred_pos, double *__restrict__ curr_max_out,
    uint32_t *__restrict__ chain_buf, uint32_t *__restrict__ chain_cnt, uint32_t *__restrict__ chain_words, uint32_t key_anchors)
{
    extern __shared__ __attribute__((aligned(16))) double s_gap[];
    if constexpr (GAP_LDS) {
        for (uint64_t g = threadIdx.x; g <= max_gap; g += blockDim.x) s_gap[g] = gap_cost[g];
        __syncthreads();
    }
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t a0 = anchor_off[r];
    const uint32_t A = (uint32_t)(anchor_off[r + 1] - a0);
    double curr_max = 0.0;
    if (A <= key_anchors)
        vga_chain_dp<GAP_LDS, true>(lane, a0, A, s_id, s_qb, s_tb, s_te, k, bandwidth, max_gap, gap_cost, s_gap, f_out, pred_id_out, pred_pos, curr_max);
    else
        vga_chain_dp<GAP_LDS, false>(lane, a0, A, s_id, s_qb, s_tb, s_te, k, bandwidth, max_gap, gap_cost, s_gap, f_out, pred_id_out, pred_pos, curr_max);
    if (lane == 0) curr_max_out[r] = curr_max;
    __threadfence_block();

    // ---- backtracking, src/chain.rs:455-558
    uint32_t *buf = chain_buf + 3 * a0 + 2 * (uint64_t)r;
    volatile int32_t *vpred = pred_pos + a0;
    const double *fr = f_out + a0;
    uint32_t nch = 0, wpos = 0;
    for (int top = (int)A; top > 0; top -= 64) {
        const int i = top - 1 - (int)lane;
        bool cand = false;
        if (i >= 0) cand = vpred[i] >= 0 && fr[i] == curr_max;
        uint64_t mask = __ballot(cand);
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            int cur = top - 1 - l;
            if (vpred[cur] < 0) continue;  // consumed by an earlier chain (src/chain.rs:469,478)
            const uint32_t hdr = wpos++;
            uint32_t len = 0;
            int pnext;
            while ((pnext = vpred[cur]) >= 0) {
                if (lane == 0) { vpred[cur] = -1; buf[wpos] = (uint32_t)cur; }
                __threadfence_block();
                wpos++;
                len++;
                cur = pnext;
            }
            if (lane == 0) buf[wpos] = (uint32_t)cur;
            wpos++;
            len++;
            if (len >= min_anchors) {
                if (lane == 0) buf[hdr] = len;
                nch++;
            } else {
                wpos = hdr;
            }
        }
    }
    if (lane == 0) { chain_cnt[r] = nch; chain_words[r] = wpos; }
}

// the chain buffer is sized for the worst case (3 A + 2 words per read); a read uses chain_words[r] of them.  Only the
// used prefix of every read crosses PCIe: block r copies it to its slot of the compact buffer.
__global__ __launch_bounds__(256) void k_chain_compact(const uint64_t *__restrict__ anchor_off, const uint32_t *__restrict__ chain_words,
                                                       const uint64_t *__restrict__ woff, const uint32_t *__restrict__ chain_buf,
                                                       uint32_t *__restrict__ comp)
{
    const uint64_t r = blockIdx.x;
    const uint32_t *src = chain_buf + 3 * anchor_off[r] + 2 * r;
    uint32_t *dst = comp + woff[r];
    const uint32_t n = chain_words[r];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------ host
namespace {

struct map_ws {
    vga_dbuf<uint32_t> cnt;
    vga_dbuf<uint64_t> anchor_off;
    vga_dbuf<uint32_t> a_qb, a_tb, a_te, a_idx, key_b, val_b, s_qb, s_tb, s_te;
    vga_dbuf<double> f, curr_max, gap_cost;
    vga_dbuf<int32_t> pred_id, pred_pos;
    vga_dbuf<uint32_t> chain_buf, chain_cnt, chain_words, key_a, chain_comp;
    vga_dbuf<uint64_t> chain_woff;
    vga_hbuf<uint64_t> h_chain_woff;
    // pinned staging for the result copies (pageable D2H runs at a fraction of the PCIe rate)
    vga_hbuf<uint32_t> h_chain_cnt, h_chain_words, h_cnt;
    vga_hbuf<double> h_curr_max;
    // the sorted anchor coordinates go back to the host on a second stream while the chaining kernel runs
    hipStream_t st_copy = nullptr;
    hipEvent_t ev_sorted = nullptr;
    ~map_ws()
    {
        if (st_copy) (void)hipStreamDestroy(st_copy);
        if (ev_sorted) (void)hipEventDestroy(ev_sorted);
    }
};

template <typename T>
T *xmalloc(size_t n)
{
    return (T *)malloc((n ? n : 1) * sizeof(T));
}

}  // namespace

extern "C" void vga_map_result_free(vga_map_result *r)
{
    if (!r) return;
    free(r->anchor_off);
    free(r->anchor_id);
    free(r->query_begin);
    free(r->target_begin);
    free(r->target_end);
    free(r->max_chain_score);
    free(r->best_pred_id);
    free(r->curr_max);
    free(r->chain_off);
    free(r->chain_placeholder);
    free(r->chain_anchor_off);
    free(r->chain_anchor_idx);
    free(r);
}

static int vga_map_batch_impl(vga_batch *b, const vga_map_params *params, vga_map_result **out)
{
    if (!b || !params || !out || !b->ctx) return VGA_ERR_ARG;  // b->ctx == nullptr: the context was destroyed
    vga_ctx *ctx = b->ctx;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    vga_ctx_scope scope(ctx);
    vga_release_deferred(ctx);  // (buffers of this context that grew during an earlier call: freed now, while it has nothing in flight)
    if (!ctx->index.loaded) return vga_set_error(ctx, VGA_ERR_NO_INDEX, "vga_map_batch: no index uploaded");
    // only_forward = 0 (anchors_for_query(..., false), src/chain.rs:154-155): every k-mer record becomes an anchor; the
    // orientation of each end travels in bit 31 of target_begin / target_end.  vga_align_batch accepts forward chains only.
    const bool all_orients = !params->only_forward;
    if (all_orients && !ctx->index.d_table_all)
        return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "only_forward=0 needs the all-orientation probe table, which is built for k <= 13");
    if (params->bandwidth == 0 || params->bandwidth > 64)
        return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "bandwidth %u: the wavefront chaining kernel supports 1..64", params->bandwidth);
    if (params->max_gap > (1u << 22))
        return vga_set_error(ctx, VGA_ERR_UNSUPPORTED, "max_gap too large for the tabulated gap cost");
    const vga_dev_index &ix = ctx->index;
    const uint64_t R = b->n_reads;
    hipStream_t st = ctx->stream;
    if (!ctx->map_ws) {
        ctx->map_ws = new map_ws();
        ctx->map_ws_free = [](void *p) { delete (map_ws *)p; };
    }
    map_ws &ws = *(map_ws *)ctx->map_ws;
    vga_timers_reset(ctx);
    auto t_host0 = std::chrono::steady_clock::now();
    vga_trace tr("map");

    vga_map_result *res = (vga_map_result *)calloc(1, sizeof(vga_map_result));
    if (!res) return vga_set_error(ctx, VGA_ERR_NOMEM, "out of host memory (map result)");
    // every host allocation of the result is checked before it is written to: VGA_ERR_NOMEM instead of a crash
    auto nomem = [&]() { vga_map_result_free(res); return vga_set_error(ctx, VGA_ERR_NOMEM, "out of host memory (map result of %llu reads)", (unsigned long long)R); };
    res->n_reads = R;
    res->anchor_off = xmalloc<uint64_t>(R + 1);
    res->curr_max = xmalloc<double>(R);
    res->chain_off = xmalloc<uint64_t>(R + 1);
    if (!res->anchor_off || !res->curr_max || !res->chain_off) return nomem();
    res->anchor_off[0] = 0;
    res->chain_off[0] = 0;
    if (R == 0) {
        res->chain_anchor_off = xmalloc<uint64_t>(1);
        if (!res->chain_anchor_off) return nomem();
        res->chain_anchor_off[0] = 0;
        *out = res;
        return VGA_OK;
    }

#define MAP_CHECK(call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (void)hipStreamSynchronize(st); /* (copies into the result's arrays may be in flight) */ \
            if (ws.st_copy) (void)hipStreamSynchronize(ws.st_copy);                                  \
            vga_map_result_free(res);                                                                \
            return vga_set_error(ctx, VGA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                                 __LINE__);                                                          \
        }                                                                                            \
    } while (0)

    // ---- K1 pass 1: count
    MAP_CHECK(ws.cnt.reserve(R));
    MAP_CHECK(ws.anchor_off.reserve(R + 1));
    int t_total = vga_timer_begin(ctx, "map_total", 0);
    int t1 = vga_timer_begin(ctx, "kmer_probe_count", 0);
    const uint32_t *probe_table = all_orients ? ix.d_table_all : ix.d_table;
    const uint2 *probe_pos = all_orients ? ix.d_pos_all : ix.d_pos;
    hipLaunchKernelGGL(k_kmer_probe<false>, dim3((unsigned)R), dim3(VGA_PROBE_NT), 0, st, b->d_reads, b->d_read_off, ix.k,
                       probe_table, probe_pos, ws.cnt.p, (const uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                       (uint32_t *)nullptr, (uint32_t *)nullptr);
    vga_timer_end(ctx, t1);
    MAP_CHECK(ws.h_cnt.reserve(R));
    uint32_t *h_cnt = ws.h_cnt.p;
    MAP_CHECK(hipMemcpyAsync(h_cnt, ws.cnt.p, R * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MAP_CHECK(hipStreamSynchronize(st));
    tr.mark("count kernel + sync");
    uint64_t total = 0;
    for (uint64_t r = 0; r < R; r++) {
        res->anchor_off[r] = total;
        total += h_cnt[r];
    }
    res->anchor_off[R] = total;
    res->n_anchors = total;
    res->n_hits = total;
    MAP_CHECK(hipMemcpyAsync(ws.anchor_off.p, res->anchor_off, (R + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));

    const size_t An = (size_t)total;
    MAP_CHECK(ws.a_qb.reserve(An)); MAP_CHECK(ws.a_tb.reserve(An)); MAP_CHECK(ws.a_te.reserve(An));
    MAP_CHECK(ws.a_idx.reserve(An)); MAP_CHECK(ws.key_b.reserve(An)); MAP_CHECK(ws.val_b.reserve(An));
    MAP_CHECK(ws.s_qb.reserve(An)); MAP_CHECK(ws.s_tb.reserve(An)); MAP_CHECK(ws.s_te.reserve(An));
    MAP_CHECK(ws.f.reserve(An)); MAP_CHECK(ws.pred_id.reserve(An)); MAP_CHECK(ws.pred_pos.reserve(An));
    MAP_CHECK(ws.curr_max.reserve(R));
    MAP_CHECK(ws.chain_buf.reserve(3 * An + 2 * R + 2));
    MAP_CHECK(ws.chain_cnt.reserve(R)); MAP_CHECK(ws.chain_words.reserve(R));

    tr.mark("workspace reserve");
    // gap cost table (src/chain.rs:348-354), host libm
    const uint64_t mg = params->max_gap;
    std::vector<double> gc(mg + 1);
    gc[0] = 0.0;
    for (uint64_t g = 1; g <= mg; g++) gc[g] = 0.01 * (double)ix.k * (double)g + 0.5 * log2((double)g);
    MAP_CHECK(ws.gap_cost.reserve(mg + 1));
    MAP_CHECK(hipMemcpyAsync(ws.gap_cost.p, gc.data(), (mg + 1) * sizeof(double), hipMemcpyHostToDevice, st));

    // ---- K1 pass 2: emit
    // byte model B_map: L + 4(L-k+1) read+table, 8H positions, 16A anchor write
    uint64_t nkm = 0;
    for (uint64_t r = 0; r < R; r++) {
        uint64_t L = b->read_off[r + 1] - b->read_off[r];
        if (L >= ix.k) nkm += L - ix.k + 1;
    }
    int t2 = vga_timer_begin(ctx, "kmer_probe_emit", b->total_bases + 4 * nkm + 8 * total + 16 * total);
    hipLaunchKernelGGL(k_kmer_probe<true>, dim3((unsigned)R), dim3(VGA_PROBE_NT), 0, st, b->d_reads, b->d_read_off, ix.k,
                       probe_table, probe_pos, (uint32_t *)nullptr, ws.anchor_off.p, ws.a_qb.p, ws.a_tb.p, ws.a_te.p, ws.a_idx.p);
    vga_timer_end(ctx, t2);

    // ---- K2: sort by target_end
    uint32_t nbits = 1;
    while ((1ull << nbits) <= ix.seq_length && nbits < 32) nbits++;
    // with both orientations the order is (orientation descending, position ascending), src/chain.rs:386-389: the key is
    // the end position with its orientation bit flipped, all 32 bits sorted
    const uint32_t n_pass = all_orients ? 4 : (nbits + 7) / 8;
    // the keys are sorted in place of a_te (ping) / key_b (pong); the original te is re-gathered from a copy
    vga_dbuf<uint32_t> &key_a = ws.key_a;
    MAP_CHECK(key_a.reserve(An));
    if (An) MAP_CHECK(hipMemcpyAsync(key_a.p, ws.a_te.p, An * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    if (An && all_orients) hipLaunchKernelGGL(k_flip_orient_bit, dim3((unsigned)((An + 255) / 256)), dim3(256), 0, st, key_a.p, (uint64_t)An);
    int t3 = vga_timer_begin(ctx, "anchor_sort", (uint64_t)n_pass * 16 * total + 24 * total);
    hipLaunchKernelGGL(k_anchor_sort, dim3((unsigned)R), dim3(VGA_SORT_NT), 0, st, ws.anchor_off.p, n_pass, key_a.p,
                       ws.a_idx.p, ws.key_b.p, ws.val_b.p);
    const uint32_t *perm = (n_pass & 1u) ? ws.val_b.p : ws.a_idx.p;
    hipLaunchKernelGGL(k_anchor_gather_seg, dim3((unsigned)R), dim3(256), 0, st, ws.anchor_off.p, perm, ws.a_qb.p, ws.a_tb.p,
                       ws.a_te.p, ws.s_qb.p, ws.s_tb.p, ws.s_te.p);
    vga_timer_end(ctx, t3);
    // the sorted coordinates are final: they go back beside the chaining kernel (below, once it is launched)
    if (!ws.st_copy) {
        MAP_CHECK(hipStreamCreateWithFlags(&ws.st_copy, hipStreamNonBlocking));
        MAP_CHECK(hipEventCreateWithFlags(&ws.ev_sorted, hipEventDisableTiming));
    }
    MAP_CHECK(hipEventRecord(ws.ev_sorted, st));
    MAP_CHECK(hipStreamWaitEvent(ws.st_copy, ws.ev_sorted, 0));
    tr.mark("probe + sort launches");

    // ---- K3: chain DP + backtracking
    int t4 = vga_timer_begin(ctx, "chain_dp", 16 * total + 12 * total);
    {
        const size_t gap_bytes = (size_t)(params->max_gap + 1) * sizeof(double);
        // reads with at most this many anchors take their argmax on round(1000 score) as a 32-bit integer (vga_chain_dp):
        // |score| <= k (A + 1) + gap_cost[max_gap], with room to spare
        uint32_t key_anchors = 0;
        {
            const double room = 2147483647.0 / 1000.0 - gc[mg] - 4.0 * (double)ix.k - 16.0;
            if (room > 0.0 && !getenv("VGA_CHAIN_F64")) key_anchors = (uint32_t)std::min(room / (double)ix.k, 4.0e9);
        }
#define CHAIN_ARGS ws.anchor_off.p, perm, ws.s_qb.p, ws.s_tb.p, ws.s_te.p, ix.k, params->bandwidth, params->max_gap,           \
                   params->chain_min_n_anchors, ws.gap_cost.p, ws.f.p, ws.pred_id.p, ws.pred_pos.p, ws.curr_max.p,         \
                   ws.chain_buf.p, ws.chain_cnt.p, ws.chain_words.p, key_anchors
        if (gap_bytes <= 16 * 1024)  // (8 KB at the default max_gap; bigger tables stay in HBM)
            hipLaunchKernelGGL(k_chain4<true>, dim3((unsigned)((R + 3) / 4)), dim3(256), gap_bytes, st, (uint32_t)R, CHAIN_ARGS);
        else
            hipLaunchKernelGGL(k_chain4<false>, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, (uint32_t)R, CHAIN_ARGS);
#undef CHAIN_ARGS
    }
    vga_timer_end(ctx, t4);
    vga_timer_end(ctx, t_total);

    tr.mark("chain launch");
    // ---- results to host.  The per-anchor arrays go straight into the result's own (pageable) arrays: the runtime stages
    // such a copy through its own pinned buffers at ~20 GB/s, where pinned staging of our own costs a hipHostMalloc of the
    // same size first (60-90 ms per 400 MB, tests/microbench/pinned_time.hip) and a fan-out copy afterwards.  The small
    // per-read arrays keep their pinned staging.
    const bool emit_dp = params->emit_dp != 0;
    if (emit_dp) {
        res->anchor_id = xmalloc<uint32_t>(An);
        res->max_chain_score = xmalloc<double>(An);
        res->best_pred_id = xmalloc<int32_t>(An);
    }
    res->query_begin = xmalloc<uint32_t>(An);
    res->target_begin = xmalloc<uint32_t>(An);
    res->target_end = xmalloc<uint32_t>(An);
    if ((emit_dp && (!res->anchor_id || !res->max_chain_score || !res->best_pred_id)) || !res->query_begin || !res->target_begin || !res->target_end) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(ws.st_copy);
        return nomem();
    }
    if (An) {  // (beside the chaining kernel: these wait for the sort only)
        MAP_CHECK(hipMemcpyAsync(res->query_begin, ws.s_qb.p, An * 4, hipMemcpyDeviceToHost, ws.st_copy));
        MAP_CHECK(hipMemcpyAsync(res->target_begin, ws.s_tb.p, An * 4, hipMemcpyDeviceToHost, ws.st_copy));
        MAP_CHECK(hipMemcpyAsync(res->target_end, ws.s_te.p, An * 4, hipMemcpyDeviceToHost, ws.st_copy));
    }
    MAP_CHECK(ws.h_curr_max.reserve(R)); MAP_CHECK(ws.h_chain_cnt.reserve(R)); MAP_CHECK(ws.h_chain_words.reserve(R));
    MAP_CHECK(ws.h_chain_woff.reserve(R + 1)); MAP_CHECK(ws.chain_woff.reserve(R + 1));
    // the per-read counts first: they say how much of the chain buffer is in use
    MAP_CHECK(hipMemcpyAsync(ws.h_curr_max.p, ws.curr_max.p, R * 8, hipMemcpyDeviceToHost, st));
    MAP_CHECK(hipMemcpyAsync(ws.h_chain_cnt.p, ws.chain_cnt.p, R * 4, hipMemcpyDeviceToHost, st));
    MAP_CHECK(hipMemcpyAsync(ws.h_chain_words.p, ws.chain_words.p, R * 4, hipMemcpyDeviceToHost, st));
    MAP_CHECK(hipStreamSynchronize(st));
    uint64_t chain_total_words = 0;
    for (uint64_t r = 0; r < R; r++) { ws.h_chain_woff.p[r] = chain_total_words; chain_total_words += ws.h_chain_words.p[r]; }
    ws.h_chain_woff.p[R] = chain_total_words;
    std::unique_ptr<uint32_t, void (*)(void *)> chain_words_host(xmalloc<uint32_t>(chain_total_words + 2), free);
    if (!chain_words_host) { (void)hipStreamSynchronize(ws.st_copy); return nomem(); }
    if (chain_total_words) {
        MAP_CHECK(ws.chain_comp.reserve(chain_total_words + 2));
        MAP_CHECK(hipMemcpyAsync(ws.chain_woff.p, ws.h_chain_woff.p, (R + 1) * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_chain_compact, dim3((unsigned)R), dim3(256), 0, st, ws.anchor_off.p, ws.chain_words.p, ws.chain_woff.p,
                           ws.chain_buf.p, ws.chain_comp.p);
        MAP_CHECK(hipMemcpyAsync(chain_words_host.get(), ws.chain_comp.p, chain_total_words * 4, hipMemcpyDeviceToHost, st));
    }
    if (An && emit_dp) {
        MAP_CHECK(hipMemcpyAsync(res->anchor_id, perm, An * 4, hipMemcpyDeviceToHost, st));
        MAP_CHECK(hipMemcpyAsync(res->max_chain_score, ws.f.p, An * 8, hipMemcpyDeviceToHost, st));
        MAP_CHECK(hipMemcpyAsync(res->best_pred_id, ws.pred_id.p, An * 4, hipMemcpyDeviceToHost, st));
    }
    MAP_CHECK(hipStreamSynchronize(st));
    MAP_CHECK(hipStreamSynchronize(ws.st_copy));
    tr.mark("kernels + D2H");
    vga_timers_collect(ctx);
    memcpy(res->curr_max, ws.h_curr_max.p, R * 8);
    const uint32_t *h_chain_cnt = ws.h_chain_cnt.p, *h_chain_words = ws.h_chain_words.p, *h_chain_buf = chain_words_host.get();

    // ---- chains: discovery order per read, members reversed to ascending (src/chain.rs:546);
    // a read without chains gets one placeholder (src/chain.rs:644-649)
    uint64_t n_chains = 0, n_members = 0;
    std::vector<uint64_t> mem0(R);  // first member slot of each read
    for (uint64_t r = 0; r < R; r++) {
        uint32_t c = h_chain_cnt[r];
        res->chain_off[r] = n_chains;
        mem0[r] = n_members;
        n_chains += c ? c : 1;
        n_members += h_chain_words[r] - c;
    }
    res->chain_off[R] = n_chains;
    res->n_chains = n_chains;
    res->chain_placeholder = xmalloc<uint8_t>(n_chains);
    res->chain_anchor_off = xmalloc<uint64_t>(n_chains + 1);
    res->chain_anchor_idx = xmalloc<uint32_t>(n_members);
    if (!res->chain_placeholder || !res->chain_anchor_off || !res->chain_anchor_idx) return nomem();
    res->chain_anchor_off[n_chains] = n_members;
    vga_parallel_for(R, [&](uint64_t r) {
        uint64_t ci = res->chain_off[r], mi = mem0[r];
        const uint32_t *buf = h_chain_buf + ws.h_chain_woff.p[r];
        uint32_t c = h_chain_cnt[r], wp = 0;
        if (c == 0) {
            res->chain_placeholder[ci] = 1;
            res->chain_anchor_off[ci] = mi;
            return;
        }
        for (uint32_t q = 0; q < c; q++) {
            uint32_t len = buf[wp++];
            res->chain_placeholder[ci] = 0;
            res->chain_anchor_off[ci] = mi;
            for (uint32_t t = 0; t < len; t++) res->chain_anchor_idx[mi + t] = buf[wp + len - 1 - t];
            wp += len;
            mi += len;
            ci++;
        }
    }, (unsigned)std::max<uint64_t>(1, (n_members + 64 * R) / 1000000));
    tr.mark("chain assembly");
    res->ms_probe = vga_timer_sum(ctx, "kmer_probe");
    res->ms_sort = vga_timer_sum(ctx, "anchor_sort");
    res->ms_chain = vga_timer_sum(ctx, "chain_dp");
    res->ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
#undef MAP_CHECK
    *out = res;
    return VGA_OK;
}

extern "C" int vga_map_batch(vga_batch *b, const vga_map_params *params, vga_map_result **out)
{
    // nothing throws across the C ABI: an allocation failure inside becomes VGA_ERR_NOMEM
    try {
        return vga_map_batch_impl(b, params, out);
    } catch (const std::bad_alloc &) {
        return vga_set_error((b ? b->ctx : nullptr), VGA_ERR_NOMEM, "vga_map_batch: out of host memory");
    } catch (const std::exception &e) {
        return vga_set_error((b ? b->ctx : nullptr), VGA_ERR_ARG, "vga_map_batch: %s", e.what());
    }
}

