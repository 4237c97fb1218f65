// vga_poa_kernels.hpp -- device side of the POA engine shared by every DP kernel: problem / row / result records, the
// fallback kernel k_poa_dp_lds<NT,4> (any gap penalties that fit a byte per gap state), the traceback (K4b) and the LDS
// sizing helpers.  Included by vga_poa.hip only; the default kernels live in vga_poa_t4.hpp / vga_poa_t5.hpp.
#pragma once

#define POA_NEG (-(1 << 21))  // "minus infinity"; H stays inside 23 signed bits (see k_poa_dp_pk)
#define POA_IDENT (INT32_MIN / 2)
#define POA_CHUNK (1ull << 20)
#define POA_RING_SPAN 32  // value rows read within this many nodes live in the per-problem ring, the others are kept
#define POA_SLOTS 4  // most sub-batches in flight (VGA_POA_SLOTS; default 2): own stream, pool segment and staging buffers each

#define POA_ST_OK 0
#define POA_ST_POOL 1
#define POA_ST_NOALN 2
#define POA_ST_TRACE 3
#define POA_ST_RETRY 5  // a specialised DP kernel hands the problem back: it is re-run by the general one (k_poa_dp_t4)

struct poa_prob {
    uint64_t node0;  // first entry of the node table (entry 0 of a problem is the virtual source)
    uint64_t pred0;  // first entry of the predecessor list (row ids)
    uint64_t sink0;  // first entry of the sink predecessor list
    uint64_t q0;     // first query byte
    uint64_t ops0;   // first entry of the traceback output
    uint64_t row0;   // first entry of the per-row arrays (rows 0..N)
    uint64_t seq0;   // first byte of the node sequences (row r is byte r-1)
    uint32_t n_sink;
    uint32_t qlen;
    uint32_t N;
    uint32_t w;      // adaptive band half-width: wb + floor(wf * qlen), computed on the host in double
    uint32_t n_nodes;  // node-table entries incl. the source
    uint32_t ring_rows;  // value rows of node-end rows live in a ring of this many worst-case rows (k_poa_dp_pk)
    uint32_t flags;      // bit 0: too large for an arena (k_poa_dp_pk in arena mode reports POA_ST_POOL at once)
    uint32_t pad;
};

struct poa_row {          // per DP row, 48 B
    int32_t beg, end;     // band
    uint64_t doff, voff;  // direction row / value row in the pool
    int32_t lmax, rmax;   // leftmost / rightmost column of the row maximum
    // the last four words form one aligned 16-byte group: k_poa_rowprep fills them for k_poa_dp_w1, which reads them with a
    // single scalar load per row
    uint32_t pred, npred; // predecessor row or predecessor-list slice; npred != 0 only on the first row of a node
    int32_t base, hmax;   // k_poa_dp_pk<.., H16>: the row's values are stored relative to `base`, hmax = the row maximum.
                          // k_poa_dp_w1: base = graph bases after this row on the longest path to the sink ("remain"),
                          // hmax = static flags of the row (POA_RF_*), both written by k_poa_rowprep
};
static_assert(sizeof(poa_row) == 48 && offsetof(poa_row, pred) == 32, "poa_row layout");
#define POA_RF_FIRST 1u    // first base of a node (other than the source)
#define POA_RF_LAST 2u     // last base of a node
#define POA_RF_SINK 4u     // ... of a node without successors: the row feeds the sink
#define POA_RF_KEEP 8u     // its value row is read more than POA_RING_SPAN nodes ahead: kept outside the ring
#define POA_RF_FAR 16u     // a predecessor is not the row directly above
                           // bits 8..10: code of the row's base (A C G T other), bits 16..23: number of predecessors

struct poa_out {          // per problem, 56 B
    int32_t score;
    uint32_t row;         // sink predecessor the traceback starts from
    int32_t status;
    uint32_t maxw;        // widest row (storage columns)
    uint64_t cells, vcells;
    uint32_t nops, pad;
    uint64_t t_begin, t_end;  // s_memrealtime (100 MHz) when the DP workgroup started / finished: occupancy diagnostics
};

struct poa_dev_params {
    int32_t match, mismatch, o1, e1, o2, e2, banded;
};

// ---------------------------------------------------------------------------------------------------------
// K4.  * The row that was just filled stays in LDS, indexed by ABSOLUTE query column and overwritten in place
//        by the next row:  Hs[j] int32 (H), Ds[j] uint16 (d1 | d2 << 8); the common predecessor (the row directly
//        above) costs three vector LDS reads per four cells instead of an L2 round trip.
//      * Every lane owns four adjacent, 4-aligned columns: the insertion recurrence
//        Fk[j] = max_{j'<j} Ht[j'] - Ok - Ek (j - j')  runs serially inside the lane and only the per-lane
//        aggregates go through the wave scan (DPP row_shr / row_bcast, no LDS permutes).
//      * Workgroup barriers wait for LDS only (s_waitcnt lgkmcnt(0); s_barrier): direction bytes and node-end
//        value rows are fire-and-forget dword / dwordx4 global stores; the query is staged in LDS and the node
//        table / node bases come through the scalar cache, so the row loop issues no vector loads at all
//        (gfx950 retires vector memory operations in order: a load behind those stores would wait for them to
//        reach HBM).
//      * In-place hazard: inside a step every lane reads Hs[j0-1 .. j0+3] before the step's barrier and writes
//        after it; the first lane of the NEXT step needs the old Hs of this step's last column, which the last
//        lane parks in `edge` before the barrier.
//      * Rows with a predecessor that is not the row directly above (bubble arms, multi-predecessor rows) read
//        that predecessor's value row from HBM; such a row starts with a full __syncthreads() (vmcnt(0)).
// -DPOA_MARKERS puts region markers into the ISA (tests/isa_regions.py counts instructions between them)
#ifdef POA_MARKERS
#define POA_MARK(name) asm volatile("; MARK " name)
#else
#define POA_MARK(name)
#endif
#define POA_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ uint64_t poa_uniform_u64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
// The traceback pool of k_poa_dp_t5 (vga_poa.hip: poa_ws owns it).  Direction rows -- nine tenths of a problem's footprint, and
// unknown in size until the rows have been computed, because the band is adaptive -- come out of 1 MiB CHUNKS that a
// workgroup pops from a device-wide lock-free free list when it needs one and pushes back, all at once, when its traceback
// is done: what the pool has to hold is what the resident workgroups have written SO FAR, not a worst-case arena for each
// (rounds 1-2: 2 000 arenas of 128 MB = 257 GB of HBM for problems that use 20-55 MB; allocating and freeing that much
// dominated a 10 000-read run of the command line tool).  The chunks live in SEGMENTS that the host allocates on a thread
// of its own while launches already run (a chunk's address: seg_base[chunk >> cps_log2] + ((chunk & mask) << 20), the table
// in device memory, an entry written before its chunks are listed).  What must be contiguous -- the value-row ring, the two wide-row scratch rows, kept value rows --
// sits in a small fixed STATE region per resident workgroup, taken like an arena before (flag 0 -> 1).
// Offsets stored in the row records are absolute device addresses in this mode (pool base 0).
#define POA_NIL 0xFFFFFFFFu
#define POA_LISTS 64        // the free list is sharded: workgroups of a launch start together and run in step, so they ask for
#define POA_LIST_STRIDE 16  // chunks at the same moments -- one list head would serialise 2 000 compare-and-swap loops
struct poa_chunk_pool {
    unsigned long long *head;    // [POA_LISTS * POA_LIST_STRIDE] free lists: change counter << 32 | first free chunk (POA_NIL: none)
    uint32_t *next;              // per chunk: the next chunk of the list it is in (a free list, or its owner's)
    const uint64_t *seg_base;    // device address of every segment
    uint32_t cps_log2;           // chunks per segment, log2
    uint32_t n_slots;            // state regions (0: classic mode, no chunk pool)
    uint8_t *state_base;         // n_slots regions of state_size bytes
    uint64_t state_size;
    uint32_t *slot_flag;         // 0 free / 1 taken
    unsigned long long *stats;   // [0] requests that found every list empty (the host adds segments when it grows)
    uint32_t *owner;             // VGA_POOL_CHECK=1 (diagnostics, else null): per chunk, who holds it (0: a free list) -- a chunk popped while
                                 // held, or pushed by somebody else, counts in stats[4] / stats[5] and fails the call
    uint32_t *short_flag;        // pinned host memory: set to 1 with stats[0] -- the host's keeper thread reads (and clears) it
                                 // without any GPU work of its own (a copy of stats[0] waited 0.1-0.2 s for a slot on a full GPU)
};
__device__ __forceinline__ uint64_t poa_chunk_addr(const poa_chunk_pool &C, uint32_t idx)
{
    // (an agent-scope load: the table grows while this kernel runs -- a new segment's entry is written by a copy on another stream
    // before its chunks are listed -- and a plain load through the const pointer may become a scalar load, whose cache the
    // acquire of poa_chunk_pop does not invalidate)
    return __hip_atomic_load(C.seg_base + (idx >> C.cps_log2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + ((uint64_t)(idx & ((1u << C.cps_log2) - 1u)) << 20);
}
// one thread: pop a chunk, starting at this workgroup's home list and going round; waits (bounded, ~0.1 s) while all lists
// are empty -- chunks come back as other workgroups finish, and new segments arrive from the host
__device__ __forceinline__ uint32_t poa_chunk_pop(const poa_chunk_pool &C, uint32_t home)
{
    for (uint32_t round = 0; round < (1u << 15); round++) {
        for (uint32_t k = 0; k < POA_LISTS; k++) {
            unsigned long long *hd = C.head + (size_t)((home + k) % POA_LISTS) * POA_LIST_STRIDE;
            for (;;) {
                unsigned long long h = __hip_atomic_load(hd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t idx = (uint32_t)h;
                if (idx == POA_NIL) break;
                const uint32_t nx = __hip_atomic_load(C.next + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (atomicCAS(hd, h, ((h >> 32) + 1ull) << 32 | nx) == h) {
                    if (C.owner && atomicExch(&C.owner[idx], home + 1u) != 0u) (void)atomicAdd(C.stats + 4, 1ull);
                    return idx;
                }
            }
        }
        if (round == 0) {
            (void)atomicAdd(C.stats, 1ull);
            __hip_atomic_store(C.short_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __builtin_amdgcn_s_sleep(127);
    }
    return POA_NIL;
}
// one thread: push the list first -> ... -> last (linked through C.next) onto free list `home` in one step
__device__ __forceinline__ void poa_chunk_push(const poa_chunk_pool &C, uint32_t home, uint32_t first, uint32_t last, bool held = true)
{
    if (C.owner && held)
        for (uint32_t c = first, guard = 0; guard < (1u << 20); guard++) {
            const uint32_t was = atomicExch(&C.owner[c], 0u);
            if (was != home + 1u) {
                (void)atomicAdd(C.stats + 5, 1ull);
                if (atomicCAS(C.stats + 6, 0ull, (unsigned long long)was << 32 | c) == 0ull)
                    (void)atomicExch(C.stats + 7, (unsigned long long)blockDim.x << 48 | (unsigned long long)guard << 32 | (home + 1u));
            }
            if (c == last) break;
            c = __hip_atomic_load(C.next + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c == POA_NIL) { (void)atomicAdd(C.stats + 5, 1ull << 32); break; }
        }
    unsigned long long *hd = C.head + (size_t)(home % POA_LISTS) * POA_LIST_STRIDE;
    for (;;) {
        unsigned long long h = __hip_atomic_load(hd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(C.next + last, (uint32_t)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (atomicCAS(hd, h, ((h >> 32) + 1ull) << 32 | first) == h) return;
    }
}
// a new segment's chunks [first, first + count): dealt round the free lists (thread l builds and pushes list l's share)
__global__ void k_poa_chunks_add(poa_chunk_pool C, uint32_t first, uint32_t count)
{
    const uint32_t l = threadIdx.x;
    if (blockIdx.x != 0 || l >= POA_LISTS || l >= count) return;
    uint32_t last = first + l;
    for (uint32_t i = first + l; i + POA_LISTS < first + count; i += POA_LISTS) { C.next[i] = i + POA_LISTS; last = i + POA_LISTS; }
    poa_chunk_push(C, l, first + l, last, false);
}
__device__ __forceinline__ int poa_slot_acquire(uint32_t *flag, uint32_t n, uint32_t block)
{
    uint32_t a = (uint32_t)(((uint64_t)block * 2654435761ull) % n);
    for (uint32_t tries = 0; tries < (1u << 24); tries++) {
        if (atomicCAS(&flag[a], 0u, 1u) == 0u) return (int)a;
        a = a + 1 == n ? 0 : a + 1;
        if ((tries & 15u) == 15u) __builtin_amdgcn_s_sleep(64);
    }
    return -1;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int poa_dpp(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
// wave64 inclusive max-scan (the sequence LLVM's atomic optimizer emits for gfx9).  `old` is the identity of signed
// max so that the DPP combiner folds each stage into a single v_max_i32_dpp.
__device__ __forceinline__ int poa_wave_scan_max(int v)
{
    int t;
    t = poa_dpp<0x111, 0xf>(INT32_MIN, v); v = t > v ? t : v;  // row_shr:1
    t = poa_dpp<0x112, 0xf>(INT32_MIN, v); v = t > v ? t : v;  // row_shr:2
    t = poa_dpp<0x114, 0xf>(INT32_MIN, v); v = t > v ? t : v;  // row_shr:4
    t = poa_dpp<0x118, 0xf>(INT32_MIN, v); v = t > v ? t : v;  // row_shr:8
    t = poa_dpp<0x142, 0xa>(INT32_MIN, v); v = t > v ? t : v;  // row_bcast:15 -> rows 1,3
    t = poa_dpp<0x143, 0xc>(INT32_MIN, v); v = t > v ? t : v;  // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ int poa_wave_scan_min(int v)
{
    int t;
    t = poa_dpp<0x111, 0xf>(INT32_MAX, v); v = t < v ? t : v;
    t = poa_dpp<0x112, 0xf>(INT32_MAX, v); v = t < v ? t : v;
    t = poa_dpp<0x114, 0xf>(INT32_MAX, v); v = t < v ? t : v;
    t = poa_dpp<0x118, 0xf>(INT32_MAX, v); v = t < v ? t : v;
    t = poa_dpp<0x142, 0xa>(INT32_MAX, v); v = t < v ? t : v;
    t = poa_dpp<0x143, 0xc>(INT32_MAX, v); v = t < v ? t : v;
    return v;
}
__device__ __forceinline__ int poa_wave_shr1(int v) { return poa_dpp<0x138, 0xf>(POA_IDENT, v); }  // wave_shr:1

template <int NT, int CPT, bool STAMP = false>
__global__ __launch_bounds__(NT) void k_poa_dp_lds(
    const poa_prob *__restrict__ probs, const char *__restrict__ queries, const uint4 *__restrict__ node_tab,
    const uint32_t *__restrict__ seq32, const uint32_t *__restrict__ preds, const uint32_t *__restrict__ sink_preds,
    poa_dev_params P, poa_row *rows, uint8_t *pool, unsigned long long *pool_next, uint64_t pool_size,
    poa_out *__restrict__ outs, uint32_t lds_cols, unsigned long long *stamps = nullptr)
{
    static_assert(CPT == 4, "row storage is 4-column aligned");
    constexpr int NW = NT / 64;
    constexpr int STEP = NT * CPT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int32_t *Hs = (int32_t *)smem;                           // [lds_cols]
    uint16_t *Ds = (uint16_t *)(smem + 4ull * lds_cols);     // [lds_cols]
    uint8_t *Qs = smem + 6ull * lds_cols;                    // [lds_cols] the query
    int32_t *scr = (int32_t *)(smem + 7ull * lds_cols);      // lds_cols is a multiple of 16
    int32_t *sW1 = scr;               // [2][NW] inclusive wave maxima of a1
    int32_t *sW2 = sW1 + 2 * NW;      // [2][NW]
    int32_t *sL1 = sW2 + 2 * NW;      // [2][NW] a1 of each wave's last cell
    int32_t *sL2 = sL1 + 2 * NW;      // [2][NW]
    int32_t *sRed = sL2 + 2 * NW;     // [NW][3]
    int32_t *edgeH = sRed + 3 * NW;   // [2]
    unsigned long long *s_alloc = (unsigned long long *)(edgeH + 2);

    const poa_prob pb = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qlen = (int)pb.qlen;
    const char *query = queries + pb.q0;
    const uint4 *ntab = node_tab + pb.node0;  // {first_row, len | npred << 24, remain_last, pred_start}: one scalar load per node
    const uint32_t *plist = preds + pb.pred0;
    // Plain (non-volatile) accesses: a volatile store makes hipcc wait vmcnt(0) first, i.e. for every direction-byte
    // store still in flight.  Cross-wave visibility of these arrays is only needed by "far" rows and by the sink
    // evaluation, both of which sit behind a full __syncthreads().
    poa_row *R = rows + pb.row0;

    const int o1 = P.o1, e1 = P.e1, o2 = P.o2, e2 = P.e2;
    const int bw = (int)pb.w;
    // diagnostic build only (STAMP): cycles per row segment, summed over the rows of this workgroup's wave 0 / last wave
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, n_far = 0, n_multi = 0, n_rows = 0, n_steps = 0;
    auto stamp = [&](int seg) {
        if constexpr (STAMP) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (seg >= 0) tacc[seg] += t - tprev;
            tprev = t;
        }
    };

    uint64_t dcur = 0, dend = 0, vcur = 0, vendp = 0;
    bool failed = false;
    auto take_chunk = [&](uint64_t &cur, uint64_t &end) {
        __syncthreads();
        if (tid == 0) *s_alloc = atomicAdd(pool_next, (unsigned long long)POA_CHUNK);
        __syncthreads();
        uint64_t b = *s_alloc;
        if (b + POA_CHUNK > pool_size) failed = true;
        cur = b;
        end = b + POA_CHUNK;
    };
    auto alloc = [&](uint64_t &cur, uint64_t &end, uint64_t bytes) -> uint64_t {
        bytes = (bytes + 15ull) & ~15ull;
        if (cur + bytes > end) take_chunk(cur, end);
        uint64_t r = cur;
        cur += bytes;
        return r;
    };

    for (int t = tid; t < (int)lds_cols; t += NT) {
        uint8_t code = 4;
        if (t < qlen) {
            const char ch = query[t];
            code = ch == 'A' ? 0 : (ch == 'C' ? 1 : (ch == 'G' ? 2 : (ch == 'T' ? 3 : 4)));
        }
        Qs[t] = code;  // 0..3 = ACGT, 4 = anything else (scores 0 against everything)
    }
    __syncthreads();

    int prev_beg = 0, prev_end = -1, prev_lmax = 0, prev_rmax = 0;
    uint64_t cells = 0, vcells = 0;
    uint32_t seq_word = 0, seq_word_idx = 0xFFFFFFFFu;

    // Rows are generated from the node table: node 0 is the virtual source (one row, no predecessor), node v
    // (v >= 1) contributes len rows whose first has the node's predecessor list and whose others follow the row
    // above.  Everything here is wave-uniform and comes through the scalar cache.
    for (uint32_t v = 0; v < pb.n_nodes && !failed; v++) {
    const uint4 nt = ntab[v];
    const uint32_t nlen = nt.y & 0xFFFFFFu;
    for (uint32_t tn = 0; tn < nlen && !failed; tn++) {
        const uint32_t r = nt.x + tn;
        const bool first = tn == 0 && v > 0;
        const bool last = tn + 1 == nlen;
        const int np = v == 0 ? 0 : (tn == 0 ? (int)(nt.y >> 24) : 1);
        const uint32_t ps = nt.w;
        const int remain = (int)(nt.z & 0x3fffffffu) + (int)(nlen - 1 - tn);
        uint8_t gb = 0;
        if (v > 0) {
            // row r is base r-1 of the problem's node sequences (seq0 is 4-aligned): one scalar dword per 4 rows
            const uint32_t bi = r - 1;
            if ((bi & 3u) == 0 || (bi >> 2) != seq_word_idx) { seq_word_idx = bi >> 2; seq_word = seq32[(pb.seq0 >> 2) + seq_word_idx]; }
            gb = (uint8_t)(seq_word >> (8u * (bi & 3u)));
        }
        // nt.w is the predecessor ROW itself for a node with one predecessor, the slice start in `preds` otherwise
        bool far = false;
        if (first) {
            if (np == 1) far = ps != r - 1;
            else
                for (int t = 0; t < np; t++) far |= plist[ps + t] != r - 1;
        }
        stamp(-1);
        if (far) __syncthreads();  // vmcnt(0) + barrier: the value rows / row arrays of far predecessors have landed
        int mpl, mpr;
        if (r == 0) { mpl = 0; mpr = 0; }
        else if (!first) { mpl = prev_lmax + 1; mpr = prev_rmax + 1; }
        else {
            mpl = INT32_MAX; mpr = 0;
            for (int t = 0; t < np; t++) {
                const uint32_t p = np == 1 ? ps : plist[ps + t];
                int lm, rm;
                if (p == r - 1) { lm = prev_lmax + 1; rm = prev_rmax + 1; }
                else { lm = R[p].lmax + 1; rm = R[p].rmax + 1; }
                mpl = lm < mpl ? lm : mpl;
                mpr = rm > mpr ? rm : mpr;
            }
        }
        int beg, end;
        if (!P.banded) { beg = 0; end = qlen; }
        else {
            const int diag = qlen - remain;
            const int lo = mpl < diag ? mpl : diag;
            const int hi = mpr > diag ? mpr : diag;
            beg = lo - bw; if (beg < 0) beg = 0;
            end = hi + bw; if (end > qlen) end = qlen;
        }
        const int bal = beg & ~3;
        const int W = (end - bal + 1 + 3) & ~3;  // storage width / plane stride
        if (r > 0) cells += (uint64_t)(end - beg + 1);
        if (last) vcells += (uint64_t)(end - beg + 1);
        const uint64_t doff = alloc(dcur, dend, (uint64_t)W * (np > 1 ? 4u : 1u));
        if (failed) break;
        uint64_t voff = 0;
        if (last) { voff = alloc(vcur, vendp, 6ull * (uint64_t)W); if (failed) break; }
        if (tid == 0) {
            R[r].beg = beg; R[r].end = end;
            R[r].doff = doff; R[r].voff = voff;
            R[r].pred = ps; R[r].npred = first ? (uint32_t)np : 0u;
        }
        int32_t *Hrow = (int32_t *)(pool + voff);                       // value row: int32 H[W] then uint16 D[W]
        uint16_t *Drow = (uint16_t *)(pool + voff + 4ull * (uint64_t)W);
        uint8_t *drow = pool + doff;
        // substitution score of this row's base against a query CODE (0..3 = ACGT, 4 = anything else)
        const int gcode = gb == 'A' ? 0 : (gb == 'C' ? 1 : (gb == 'G' ? 2 : (gb == 'T' ? 3 : 4)));
        const int sc_eq = gcode == 4 ? 0 : P.match, sc_ne = gcode == 4 ? 0 : -P.mismatch;
        // single predecessor = the row directly above (in LDS): the branch-free fast path applies
        const bool single_lds = r > 0 && np == 1 && !far;
        stamp(0);  // row prologue (band, allocation, metadata)
        if constexpr (STAMP) { n_far += far; n_multi += np > 1; n_rows++; n_steps += (W + STEP - 1) / STEP; }

        int carry1 = POA_IDENT, carry2 = POA_IDENT, left1 = POA_IDENT, left2 = POA_IDENT;
        int best = INT32_MIN, lpos = beg, rpos = beg;
        int buf = 0;
        for (int c0 = 0; c0 < W; c0 += STEP, buf ^= 1) {
            const int c = c0 + CPT * tid;   // storage index of this lane's first cell (multiple of 4)
            const int j0 = bal + c;         // absolute column of this lane's first cell (multiple of 4)
            const bool lane_act = j0 <= end;  // j0 + CPT - 1 >= beg always holds (bal > beg - 4)
            const int nw_step = (W - c0 + 64 * CPT - 1) / (64 * CPT) < NW ? (W - c0 + 64 * CPT - 1) / (64 * CPT) : NW;
            const bool wave_act = wv < nw_step;
            const bool fast = single_lds && wave_act;  // lean branch-free path (band edges handled by masks)
            int ht[CPT], hts[CPT], ev1[CPT], ev2[CPT], pmv[CPT], p1v[CPT], p2v[CPT], ofl[CPT];
            int a1[CPT], a2[CPT];
            int agg1 = POA_IDENT, agg2 = POA_IDENT;
            if (fast) {
                // ---------------- lean path, phase 1: one predecessor, the row above (LDS); masks instead of branches
                const int4 hv = *(const int4 *)(Hs + j0);
                const uint2 dv = *(const uint2 *)(Ds + j0);
                const uint32_t qw = *(const uint32_t *)(Qs + j0);
                int hm[CPT], hj[CPT], g1[CPT], g2[CPT], qc[CPT];
                hj[0] = hv.x; hj[1] = hv.y; hj[2] = hv.z; hj[3] = hv.w;
                if (tid == NT - 1) edgeH[buf] = hv.w;
                const int jm1 = j0 > 0 ? j0 - 1 : 0;
                hm[0] = (tid == 0 && c0 > 0) ? edgeH[buf ^ 1] : Hs[jm1];
                hm[1] = hv.x; hm[2] = hv.y; hm[3] = hv.z;
                g1[0] = dv.x & 255; g2[0] = (dv.x >> 8) & 255; g1[1] = (dv.x >> 16) & 255; g2[1] = dv.x >> 24;
                g1[2] = dv.y & 255; g2[2] = (dv.y >> 8) & 255; g1[3] = (dv.y >> 16) & 255; g2[3] = dv.y >> 24;
                qc[0] = Qs[jm1]; qc[1] = qw & 255; qc[2] = (qw >> 8) & 255; qc[3] = (qw >> 16) & 255;
                const unsigned pspan = (unsigned)(prev_end - prev_beg), span = (unsigned)(end - beg);
                bool inprev = j0 >= 1 && (unsigned)(j0 - 1 - prev_beg) <= pspan;  // column j-1 inside the predecessor's band
#pragma unroll
                for (int k = 0; k < CPT; k++) {
                    const int j = j0 + k;
                    const bool inj = (unsigned)(j - prev_beg) <= pspan;
                    const bool actk = (unsigned)(j - beg) <= span;
                    const int s = qc[k] == gcode ? sc_eq : (qc[k] == 4 ? 0 : sc_ne);
                    const int m = inprev ? hm[k] + s : POA_NEG;
                    ev1[k] = inj ? hj[k] - g1[k] : POA_NEG;
                    ev2[k] = inj ? hj[k] - g2[k] : POA_NEG;
                    const int me = m > ev1[k] ? m : ev1[k];
                    ht[k] = me > ev2[k] ? me : ev2[k];
                    hts[k] = ev2[k] > me ? 2 : (ev1[k] > m ? 1 : 0);
                    ofl[k] = (g1[k] == o1 + e1 ? 1 : 0) | (g2[k] == o2 + e2 ? 2 : 0);
                    pmv[k] = 0; p1v[k] = 0; p2v[k] = 0;
                    a1[k] = actk ? ht[k] + e1 * j : POA_IDENT;
                    a2[k] = actk ? ht[k] + e2 * j : POA_IDENT;
                    agg1 = a1[k] > agg1 ? a1[k] : agg1;
                    agg2 = a2[k] > agg2 ? a2[k] : agg2;
                    inprev = inj;
                }
            } else if (wave_act) {
                // ---------------- general path, phase 1: band edges, source row, several / far predecessors
                bool act[CPT];
#pragma unroll
                for (int k = 0; k < CPT; k++) {
                    const int j = j0 + k;
                    act[k] = j >= beg && j <= end;
                    ht[k] = POA_NEG; hts[k] = 0; ev1[k] = POA_NEG; ev2[k] = POA_NEG; pmv[k] = 0; p1v[k] = 0; p2v[k] = 0; ofl[k] = 0;
                }
                if (r == 0) {
#pragma unroll
                    for (int k = 0; k < CPT; k++) ht[k] = (j0 + k == 0) ? 0 : POA_NEG;
                } else if (lane_act) {
                    int sub[CPT], m[CPT];
                    {
                        const uint32_t qw = *(const uint32_t *)(Qs + j0);  // codes of query[j0 .. j0+3]
                        const int qm1 = j0 >= 1 ? (int)Qs[j0 - 1] : 4;
#pragma unroll
                        for (int k = 0; k < CPT; k++) {
                            const int qc = k == 0 ? qm1 : (int)((qw >> (8 * (k - 1))) & 255u);
                            sub[k] = qc == gcode ? sc_eq : (qc == 4 ? 0 : sc_ne);
                            m[k] = POA_NEG;
                        }
                    }
                    for (int t = 0; t < np; t++) {
                        const uint32_t p = first ? (np == 1 ? ps : plist[ps + t]) : r - 1;
                        int hj[CPT], dj[CPT], hm0;
                        int bp, ep;
                        if (p == r - 1) {
                            bp = prev_beg; ep = prev_end;
                            const int4 hv = *(const int4 *)(Hs + j0);
                            const uint2 dv = *(const uint2 *)(Ds + j0);
                            hj[0] = hv.x; hj[1] = hv.y; hj[2] = hv.z; hj[3] = hv.w;
                            dj[0] = dv.x & 0xffff; dj[1] = dv.x >> 16; dj[2] = dv.y & 0xffff; dj[3] = dv.y >> 16;
                            if (tid == NT - 1) edgeH[buf] = hv.w;   // old value of this step's last column
                            hm0 = POA_NEG;
                            if (j0 >= 1) hm0 = (tid == 0 && c0 > 0) ? edgeH[buf ^ 1] : Hs[j0 - 1];
                        } else {
                            bp = R[p].beg; ep = R[p].end;
                            const uint64_t pv = R[p].voff;
                            const int balp = bp & ~3;
                            const int Wp = (ep - balp + 1 + 3) & ~3;
                            const int32_t *Hp = (const int32_t *)(pool + pv);
                            const uint16_t *Dp = (const uint16_t *)(pool + pv + 4ull * (uint64_t)Wp);
#pragma unroll
                            for (int k = 0; k < CPT; k++) {
                                const int j = j0 + k;
                                hj[k] = POA_NEG; dj[k] = 0;
                                if (j >= bp && j <= ep) { hj[k] = Hp[j - balp]; dj[k] = (int)Dp[j - balp]; }
                            }
                            hm0 = POA_NEG;
                            if (j0 - 1 >= bp && j0 - 1 <= ep) hm0 = Hp[j0 - 1 - balp];
                        }
#pragma unroll
                        for (int k = 0; k < CPT; k++) {
                            const int j = j0 + k;
                            const int hm = k == 0 ? hm0 : hj[k - 1];
                            if (act[k] && j >= 1 && j - 1 >= bp && j - 1 <= ep) {
                                const int cnd = hm + sub[k];
                                if (cnd > m[k]) { m[k] = cnd; pmv[k] = t; }
                            }
                            if (act[k] && j >= bp && j <= ep) {
                                const int gg1 = dj[k] & 255, gg2 = dj[k] >> 8;  // E + d
                                const int c1 = hj[k] - gg1;
                                if (c1 > ev1[k]) { ev1[k] = c1; p1v[k] = t; ofl[k] = (ofl[k] & 2) | (gg1 == o1 + e1 ? 1 : 0); }
                                const int c2 = hj[k] - gg2;
                                if (c2 > ev2[k]) { ev2[k] = c2; p2v[k] = t; ofl[k] = (ofl[k] & 1) | (gg2 == o2 + e2 ? 2 : 0); }
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < CPT; k++) {
                        ht[k] = m[k];
                        if (ev1[k] > ht[k]) { ht[k] = ev1[k]; hts[k] = 1; }
                        if (ev2[k] > ht[k]) { ht[k] = ev2[k]; hts[k] = 2; }
                    }
                }
#pragma unroll
                for (int k = 0; k < CPT; k++) {
                    a1[k] = act[k] ? ht[k] + e1 * (j0 + k) : POA_IDENT;
                    a2[k] = act[k] ? ht[k] + e2 * (j0 + k) : POA_IDENT;
                    agg1 = a1[k] > agg1 ? a1[k] : agg1;
                    agg2 = a2[k] > agg2 ? a2[k] : agg2;
                }
            } else {
#pragma unroll
                for (int k = 0; k < CPT; k++) { a1[k] = POA_IDENT; a2[k] = POA_IDENT; ht[k] = POA_NEG; hts[k] = 0; ev1[k] = ev2[k] = POA_NEG; pmv[k] = p1v[k] = p2v[k] = ofl[k] = 0; }
            }
            stamp(1);  // phase 1
            // ---- insertion recurrence: serial inside the lane, scan of the lane aggregates across the wave
            int i1 = POA_IDENT, i2 = POA_IDENT;
            if (wave_act) {
                i1 = poa_wave_scan_max(agg1);
                i2 = poa_wave_scan_max(agg2);
            }
            if (lane == 63) {  // inactive waves publish the identity, so readers need no masks
                sW1[buf * NW + wv] = i1; sW2[buf * NW + wv] = i2;
                sL1[buf * NW + wv] = a1[CPT - 1]; sL2[buf * NW + wv] = a2[CPT - 1];
            }
            stamp(2);  // scans
            POA_LDS_BARRIER();
            stamp(3);  // step barrier
            int tw1[NW], tw2[NW];
#pragma unroll
            for (int q = 0; q < NW; q++) { tw1[q] = sW1[buf * NW + q]; tw2[q] = sW2[buf * NW + q]; }
            int all1 = carry1, all2 = carry2;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                all1 = tw1[q] > all1 ? tw1[q] : all1;
                all2 = tw2[q] > all2 ? tw2[q] : all2;
            }
            if (wave_act) {
                int x1 = poa_wave_shr1(i1), x2 = poa_wave_shr1(i2);
                int la1 = poa_wave_shr1(a1[CPT - 1]), la2 = poa_wave_shr1(a2[CPT - 1]);
                if (lane == 0) {
                    la1 = wv == 0 ? left1 : sL1[buf * NW + wv - 1];
                    la2 = wv == 0 ? left2 : sL2[buf * NW + wv - 1];
                }
                int pre1 = carry1, pre2 = carry2;
#pragma unroll
                for (int q = 0; q < NW; q++) {
                    if (q < wv) { pre1 = tw1[q] > pre1 ? tw1[q] : pre1; pre2 = tw2[q] > pre2 ? tw2[q] : pre2; }
                }
                int run1 = pre1 > x1 ? pre1 : x1;   // max of a1 over every column before this lane's first cell
                int run2 = pre2 > x2 ? pre2 : x2;
                int hv[CPT], codev[CPT], dpk[CPT];
                if (fast) {
                    // ---------------- lean path, phase 2
                    const unsigned span = (unsigned)(end - beg);
#pragma unroll
                    for (int k = 0; k < CPT; k++) {
                        const int j = j0 + k;
                        const bool actk = (unsigned)(j - beg) <= span;
                        const int f1 = run1 - (o1 + e1 * j), f2 = run2 - (o2 + e2 * j);  // run = POA_IDENT at the first column
                        const int fo = (run1 == la1 ? 64 : 0) | (run2 == la2 ? 128 : 0);
                        const int hf = ht[k] > f1 ? ht[k] : f1;
                        const int h = hf > f2 ? hf : f2;
                        const int fsel = f2 > hf ? 32 : (f1 > ht[k] ? 16 : 0);
                        codev[k] = hts[k] | (ofl[k] << 2) | fsel | fo;
                        int dd1 = h - ev1[k]; dd1 = (dd1 < o1 ? dd1 : o1) + e1;
                        int dd2 = h - ev2[k]; dd2 = (dd2 < o2 ? dd2 : o2) + e2;
                        hv[k] = h;
                        dpk[k] = dd1 | (dd2 << 8);
                        const int hb = actk ? h : INT32_MIN;  // inactive cells never take part in the row maximum
                        if (hb > best) { best = hb; lpos = j; rpos = j; }
                        else if (actk && hb == best) rpos = j;
                        run1 = a1[k] > run1 ? a1[k] : run1;   // a1 is IDENT on inactive cells
                        run2 = a2[k] > run2 ? a2[k] : run2;
                        la1 = actk ? a1[k] : la1; la2 = actk ? a2[k] : la2;
                    }
                } else if (lane_act) {
                    // ---------------- general path, phase 2
#pragma unroll
                    for (int k = 0; k < CPT; k++) {
                        const int j = j0 + k;
                        const bool actk = j >= beg && j <= end;
                        const int f1 = run1 - o1 - e1 * j, f2 = run2 - o2 - e2 * j;  // run = POA_IDENT at the first column
                        const int fo1 = run1 == la1, fo2 = run2 == la2;
                        int h = ht[k], hs = hts[k];
                        if (f1 > h) { h = f1; hs = 3; }
                        if (f2 > h) { h = f2; hs = 4; }
                        codev[k] = hts[k] | (ofl[k] << 2) | (hs >= 3 ? (hs - 2) << 4 : 0) | (fo1 << 6) | (fo2 << 7);
                        int dd1 = h - ev1[k]; dd1 = (dd1 < o1 ? dd1 : o1) + e1;
                        int dd2 = h - ev2[k]; dd2 = (dd2 < o2 ? dd2 : o2) + e2;
                        hv[k] = h;
                        dpk[k] = dd1 | (dd2 << 8);
                        if (actk) {
                            if (h > best) { best = h; lpos = j; rpos = j; }
                            else if (h == best) rpos = j;
                            run1 = a1[k] > run1 ? a1[k] : run1;
                            run2 = a2[k] > run2 ? a2[k] : run2;
                            la1 = a1[k]; la2 = a2[k];
                        }
                    }
                }
                if (lane_act) {
                    const uint2 dq = make_uint2((uint32_t)dpk[0] | ((uint32_t)dpk[1] << 16), (uint32_t)dpk[2] | ((uint32_t)dpk[3] << 16));
                    *(int4 *)(Hs + j0) = make_int4(hv[0], hv[1], hv[2], hv[3]);
                    *(uint2 *)(Ds + j0) = dq;
                    *(uint32_t *)(drow + c) = (uint32_t)codev[0] | ((uint32_t)codev[1] << 8) | ((uint32_t)codev[2] << 16) | ((uint32_t)codev[3] << 24);
                    if (last) {
                        *(int4 *)(Hrow + c) = make_int4(hv[0], hv[1], hv[2], hv[3]);
                        *(uint2 *)(Drow + c) = dq;
                    }
                    if (np > 1) {
                        *(uint32_t *)(drow + (uint64_t)W + c) = (uint32_t)pmv[0] | ((uint32_t)pmv[1] << 8) | ((uint32_t)pmv[2] << 16) | ((uint32_t)pmv[3] << 24);
                        *(uint32_t *)(drow + 2ull * W + c) = (uint32_t)p1v[0] | ((uint32_t)p1v[1] << 8) | ((uint32_t)p1v[2] << 16) | ((uint32_t)p1v[3] << 24);
                        *(uint32_t *)(drow + 3ull * W + c) = (uint32_t)p2v[0] | ((uint32_t)p2v[1] << 8) | ((uint32_t)p2v[2] << 16) | ((uint32_t)p2v[3] << 24);
                    }
                }
            }
            carry1 = all1; carry2 = all2;
            left1 = sL1[buf * NW + nw_step - 1]; left2 = sL2[buf * NW + nw_step - 1];
        }
        stamp(4);  // phase 2 + stores
        {
            // wave-level: maximum of best, then the leftmost / rightmost column among the lanes that hold it
            int wb = poa_wave_scan_max(best);                       // lane 63 holds the wave maximum
            wb = __builtin_amdgcn_readlane(wb, 63);
            int lm = best == wb ? -lpos : POA_IDENT;                 // min(lpos) = -max(-lpos)
            int rm = best == wb ? rpos : POA_IDENT;
            lm = poa_wave_scan_max(lm);
            rm = poa_wave_scan_max(rm);
            if (lane == 63) { sRed[wv * 3 + 0] = wb; sRed[wv * 3 + 1] = -lm; sRed[wv * 3 + 2] = rm; }
        }
        POA_LDS_BARRIER();  // row complete in LDS; also fences the scratch buffers between rows
        {
            int rb[NW], rl[NW], rr[NW];
#pragma unroll
            for (int q = 0; q < NW; q++) { rb[q] = sRed[q * 3]; rl[q] = sRed[q * 3 + 1]; rr[q] = sRed[q * 3 + 2]; }
            best = rb[0]; lpos = rl[0]; rpos = rr[0];
#pragma unroll
            for (int q = 1; q < NW; q++) {
                if (rb[q] > best) { best = rb[q]; lpos = rl[q]; rpos = rr[q]; }
                else if (rb[q] == best) { lpos = rl[q] < lpos ? rl[q] : lpos; rpos = rr[q] > rpos ? rr[q] : rpos; }
            }
        }
        // uniform values: keep them in scalar registers so the next row's band arithmetic runs on the scalar unit
        lpos = __builtin_amdgcn_readfirstlane(lpos);
        rpos = __builtin_amdgcn_readfirstlane(rpos);
        if (tid == 0) { R[r].lmax = lpos; R[r].rmax = rpos; }
        prev_beg = beg; prev_end = end; prev_lmax = lpos; prev_rmax = rpos;
        stamp(5);  // row reduce + row barrier
    }
    }
    __syncthreads();
    if (tid == 0) {
        poa_out &O = outs[blockIdx.x];
        O.cells = cells;
        O.vcells = vcells;
        O.maxw = 0;
        if constexpr (STAMP) {
            if (stamps && blockIdx.x < 64)
                for (int s = 0; s < 6; s++) stamps[blockIdx.x * 6 + s] = tacc[s];
            if (stamps && blockIdx.x == 0) { stamps[384] = n_far; stamps[385] = n_multi; stamps[386] = n_rows; stamps[387] = n_steps; }
        }
        if (failed) {
            O.status = POA_ST_POOL;
            O.score = POA_NEG;
            O.row = 0;
        } else {
            int bestv = INT32_MIN;
            uint32_t brow = 0;
            bool have = false;
            for (uint32_t t = 0; t < pb.n_sink; t++) {
                const uint32_t p = sink_preds[pb.sink0 + t];
                const int bp = R[p].beg, ep = R[p].end;
                int val = POA_NEG;
                if (qlen >= bp && qlen <= ep) val = ((const int32_t *)(pool + R[p].voff))[qlen - (bp & ~3)];
                if (!have || val > bestv) { bestv = val; brow = p; have = true; }
            }
            O.score = bestv;
            O.row = brow;
            O.status = (have && bestv > POA_NEG / 2) ? POA_ST_OK : POA_ST_NOALN;
        }
    }
}

// K4b: one lane per problem.  ops are written in reverse (sink -> source) order.
// ENC 0: direction bytes of k_poa_dp_lds; ENC 1: of k_poa_dp_t4 (vga_poa_t4.hpp); ENC 2: direction dwords of k_poa_dp_t5
struct tb_code { int hts, fsel, eo1, eo2, fo1, fo2; };
template <int ENC>
__device__ __forceinline__ tb_code tb_decode(int code)
{
    tb_code c;
    if constexpr (ENC == 0) {
        // [1:0] source of Ht (M, E1, E2), [3:2] E1/E2 opened here, [4] F1 > Ht, [5] F2 > max(Ht, F1), [7:6] F1/F2 opened here
        c.hts = code & 3;
        c.fsel = (code & 32) ? 2 : ((code >> 4) & 1);
        c.eo1 = (code >> 2) & 1; c.eo2 = (code >> 3) & 1;
        c.fo1 = (code >> 6) & 1; c.fo2 = (code >> 7) & 1;
    } else {
        // [7:6] tag of H (3 Ht, 1 F1, 0 F2), [5:4] tag of Ht (2 M, 1 E1, 0 E2), [3] / [2] E1 / E2 of a successor opens from
        // this cell, [1] / [0] F1 / F2 of this cell did not open
        const int th = (code >> 6) & 3;
        c.hts = 2 - ((code >> 4) & 3);
        c.fsel = th == 3 ? 0 : (th == 1 ? 1 : 2);
        c.eo1 = (code >> 3) & 1; c.eo2 = (code >> 2) & 1;
        c.fo1 = ((code >> 1) & 1) ^ 1; c.fo2 = (code & 1) ^ 1;
    }
    return c;
}
// ENC 2 (k_poa_dp_t5): the dword of four cells.  Byte k: [5:4] tag of H (3 Ht, 1 F1, 0 F2), [3:2] tag of Ht (2 M, 1 E1, 0 E2),
// [1] / [0] F1 / F2 of the cell did not open.  "E1 / E2 of a successor opens from cell k": bit 7 (k < 2) or 6 (k >= 2) of bytes
// 2 (k & 1) and 2 (k & 1) + 1.
__device__ __forceinline__ tb_code tb_decode2(uint32_t d, int k)
{
    tb_code c;
    const int b = (int)(d >> (8 * k)) & 0xff;
    const int th = (b >> 4) & 3;
    c.hts = 2 - ((b >> 2) & 3);
    c.fsel = th == 3 ? 0 : (th == 1 ? 1 : 2);
    c.fo1 = ((b >> 1) & 1) ^ 1; c.fo2 = (b & 1) ^ 1;
    const int eb = 16 * (k & 1) + (k < 2 ? 7 : 6);
    c.eo1 = (int)(d >> eb) & 1; c.eo2 = (int)(d >> (eb + 8)) & 1;
    return c;
}

// K4b, cooperative form (the default): one wave per problem.  The walk itself is a chain of dependent reads (row
// record -> direction byte -> predecessor), two HBM round trips per operation when one lane does it alone.  Here the
// 64 lanes stage, in two round trips, what the next stretch of the walk can need -- the records of rows i .. i-63
// and, for each of them, a TB_WIN-byte window of its direction row around the column the path would reach it at if
// every row in between lay on the path (fewer columns are consumed when rows are skipped, so the window starts just
// below that column and extends above it); rows with several predecessors also stage the window of their predecessor
// choice plane and their first four predecessors -- and then all lanes walk in lock step out of LDS until the path
// leaves the staged rows or a window (30-60 operations on the HLA graphs).  Same outputs as k_poa_traceback.
#define TB_WIN 32
struct tb_lds {  // 6 912 B: what one wave stages per stretch
    int beg[64], end[64], ws[64];
    uint64_t doff[64];
    uint32_t pred[64], np[64];
    uint32_t dir[64][TB_WIN / 4], pl1[64][TB_WIN / 4], pr4[64][4];
};
// LDS traffic of one wave is processed in program order: between the staging stores and the walk's loads (other
// lanes' data) the wave only has to wait for its own stores to be issued -- no s_barrier, so the function can run in
// one wave of a larger workgroup whose other waves have finished
__device__ __forceinline__ void tb_wave_sync()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// every lane of the wave calls this with the same arguments (lane = its index); status / start_row are the DP's result
template <int ENC = 0>
__device__ __forceinline__ void poa_traceback_wave(
    tb_lds &T, const int lane, const poa_prob &pb, const poa_row *__restrict__ rows, const uint32_t *__restrict__ preds,
    const uint8_t *__restrict__ pool, poa_out &O, uint8_t *__restrict__ ops, uint32_t *__restrict__ orow, const int code_xor,
    const int status, const uint32_t start_row)
{
    if (status != POA_ST_OK) {
        if (lane == 0) O.nops = 0;
        return;
    }
    const uint32_t cap = (uint32_t)std::min<uint64_t>((uint64_t)pb.N + pb.qlen + 2, 0xffffffffu);
    uint8_t *po = ops + pb.ops0;
    uint32_t *pr = orow + pb.ops0;
    uint32_t i = start_row;
    int j = (int)pb.qlen;
    int st = 0;  // 0 H, 1 E1, 2 E2, 3 F1, 4 F2, 5 Ht
    int pend_e = 0;
    uint32_t nops = 0;
    bool bad = false;
    uint32_t my_op = 0, my_row = 0;  // lane l keeps operation number (64 m + l) until the wave stores 64 of them together
    auto emit = [&](uint32_t op, uint32_t row) {
        if ((uint32_t)lane == (nops & 63u)) { my_op = op; my_row = row; }
        nops++;
        if ((nops & 63u) == 0) {
            po[nops - 64 + lane] = (uint8_t)my_op;
            pr[nops - 64 + lane] = my_row;
        }
    };
    while (i > 0 && !bad) {
        // ---- stage rows i0 .. i0 - 63
        const uint32_t i0 = i;
        tb_wave_sync();  // the previous stretch is done with the LDS arrays
        if ((uint32_t)lane < i0) {
            const uint32_t r = i0 - (uint32_t)lane;
            const poa_row rw = rows[pb.row0 + r];
            const int bal = rw.beg & ~3;
            const int W = (rw.end - bal + 1 + 3) & ~3;
            int ws = ((j - lane - 8) - bal) & ~3;  // window start relative to bal
            if (ws > W - TB_WIN) ws = W - TB_WIN;
            if (ws < 0) ws = 0;
            const uint8_t *drow = pool + rw.doff;
#pragma unroll
            for (int d = 0; d < TB_WIN / 4; d++) {
                const int cc = ws + 4 * d;
                T.dir[lane][d] = cc < W ? *(const uint32_t *)(drow + cc) : 0u;
            }
            if (rw.npred > 1) {
#pragma unroll
                for (int d = 0; d < TB_WIN / 4; d++) {
                    const int cc = ws + 4 * d;
                    T.pl1[lane][d] = cc < W ? *(const uint32_t *)(drow + (uint64_t)W + cc) : 0u;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) T.pr4[lane][q] = (uint32_t)q < rw.npred ? preds[pb.pred0 + rw.pred + q] : 0u;
            }
            T.beg[lane] = rw.beg;
            T.end[lane] = rw.end;
            T.ws[lane] = bal + ws;
            T.doff[lane] = rw.doff;
            T.pred[lane] = rw.pred;
            T.np[lane] = rw.npred;
        }
        tb_wave_sync();
        // ---- walk out of LDS, every lane the same steps
        while (i > 0) {
            const uint32_t t = i0 - i;
            if (t >= 64) break;
            const int beg = T.beg[t], end = T.end[t];
            if (j < beg || j > end) { bad = true; break; }
            const int off = j - T.ws[t];
            if (off < 0 || off >= TB_WIN) break;  // t > 0 here: the window of row i0 was placed around j
            const uint32_t npred = T.np[t];
            const bool first = npred != 0;
            const int np = first ? (int)npred : 1;
            tb_code dc;
            if constexpr (ENC == 2) dc = tb_decode2(T.dir[t][off >> 2], off & 3);
            else dc = tb_decode<ENC>((int)((T.dir[t][off >> 2] >> (8 * (off & 3))) & 0xffu) ^ code_xor);
            if (ENC >= 1 && pend_e) {  // arrived through a deletion: this cell says whether that gap opened from it
                if (pend_e == 1 ? dc.eo1 : dc.eo2) st = 0;
                pend_e = 0;
            }
            const int hts = dc.hts;
            const int fsel = dc.fsel;
            const int hs = fsel ? 2 + fsel : hts;
            const int src = st == 0 ? hs : (st == 5 ? hts : st);
            if (nops + 1 >= cap) { bad = true; break; }
            if (src <= 2) {
                uint32_t p = i - 1;
                if (first) {
                    p = T.pred[t];
                    if (np > 1) {
                        int tt;
                        if (src == 0) tt = (int)((T.pl1[t][off >> 2] >> (8 * (off & 3))) & 0xffu);
                        else {
                            const int bal = beg & ~3;
                            const uint64_t W = (uint64_t)((end - bal + 1 + 3) & ~3);
                            tt = pool[T.doff[t] + (src == 1 ? 2 : 3) * W + (uint64_t)(j - bal)];
                        }
                        p = tt < 4 ? T.pr4[t][tt] : preds[pb.pred0 + p + tt];
                    }
                }
                p = (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
                if (src == 0) {
                    if (j < 1) { bad = true; break; }
                    emit(0, i);
                    i = p; j -= 1; st = 0;
                } else {
                    const int open = src == 1 ? dc.eo1 : dc.eo2;
                    emit(2, i);
                    if (ENC >= 1) { st = src; pend_e = src; }
                    else st = open ? 0 : src;
                    i = p;
                }
            } else {
                const int open = src == 3 ? dc.fo1 : dc.fo2;
                if (j - 1 < beg) { bad = true; break; }
                emit(1, 0);
                st = open ? 5 : src;
                j -= 1;
            }
        }
    }
    // the rest of the query is an insertion before the first aligned row
    if (!bad && j > 0 && (uint64_t)nops + (uint64_t)j >= cap) bad = true;
    if (bad) {
        if (lane == 0) { O.status = POA_ST_TRACE; O.nops = 0; }
        return;
    }
    const uint32_t base = nops & ~63u;
    if ((uint32_t)lane < (nops & 63u)) { po[base + lane] = (uint8_t)my_op; pr[base + lane] = my_row; }
    for (uint32_t x = (uint32_t)lane; x < (uint32_t)j; x += 64) { po[nops + x] = 1; pr[nops + x] = 0; }
    if (lane == 0) O.nops = nops + (uint32_t)j;
}

template <int ENC>
__global__ __launch_bounds__(64) void k_poa_traceback_wave(
    uint32_t n, const poa_prob *__restrict__ probs, const poa_row *__restrict__ rows,
    const uint32_t *__restrict__ preds, const uint8_t *__restrict__ pool, poa_out *__restrict__ outs,
    uint8_t *__restrict__ ops, uint32_t *__restrict__ orow, int code_xor)
{
    __shared__ tb_lds T;
    const uint32_t pi = blockIdx.x;
    if (pi >= n) return;
    const poa_prob pb = probs[pi];
    poa_traceback_wave<ENC>(T, (int)threadIdx.x, pb, rows, preds, pool, outs[pi], ops, orow, code_xor, outs[pi].status, outs[pi].row);
}

static inline size_t poa_pk_lds_bytes(uint32_t hg_cols, uint32_t lds_cols, int nt, bool h16 = false)
{
    const int nw = nt / 64;
    // (the fused traceback reuses the row-state area: at least sizeof(tb_lds) behind the header)
    return std::max<size_t>((h16 ? 3ull : 4ull) * hg_cols + ((lds_cols / 2 + 15u) & ~15u), sizeof(tb_lds)) + (size_t)(3 * nw + 1 + 4 + 6 + 1) * 16 + 16;
}

static inline uint32_t poa_lds_cols(uint32_t max_q) { return ((max_q + 1 + 15u) & ~15u) + 16u; }

static inline size_t poa_lds_bytes(uint32_t lds_cols, int nt)
{
    const int nw = nt / 64;
    return 7ull * lds_cols + (size_t)(8 * nw + 3 * nw + 2) * 4 + 16;
}

