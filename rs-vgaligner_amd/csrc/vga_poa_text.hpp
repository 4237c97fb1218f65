// vga_poa_text.hpp -- K4c: the variable fields of an alignment record on the device.  generate_alignment (src/align.rs:1096-1168)
// needs, of every aligned read, the cs string, the CIGAR and the node path with consecutive duplicates removed
// (graph_nodes.dedup(), align.rs:1114); rounds 1-3 shipped the traceback's raw operations to the host (5 bytes per alignment
// column: 100 KB per 10 kbp read, 1 GB per 10 000 reads over PCIe) and ran the run-length encoding on host threads.  Here one wave
// per problem reads the operations the traceback left in HBM, forward, 64 at a time, and writes the three fields as text /
// node indices into a compact arena (claimed with one atomic add per problem: ~10-20 KB instead of 100):
//   cs     ":N" for a run of N equal bases, "*gq" per mismatch, "+q.." / "-g.." per insertion / deletion run (lower case);
//   CIGAR  "<run length><M|I|D>" per run of operations (M covers matches and mismatches);
//   path   the 0-based index of every node the alignment enters, in order -- a row opens a node exactly when its row record
//          has predecessors of its own (npred != 0), the node index is a binary search over the node table's first rows.
// A run that crosses a block of 64 operations is carried in scalars; the text of a finished run is written by the lane of the
// first operation AFTER it (one more virtual operation closes the last runs).  Two passes over the operations: lengths first
// (the claim needs them), then the bytes.  The host path (poa_run: post_one) remains for callers that want the per-base rows
// (vga_poa_batch) and as the fallback when the arena is full.
#pragma once

struct poa_text_out {  // per problem, 48 B
    uint32_t cs_off, cs_len;      // bytes in the arena ("cs:Z:" included)
    uint32_t cg_off, cg_len;
    uint32_t runs_off, n_runs;    // byte offset (4-aligned) of n_runs node indices
    uint32_t n_path;              // graph-consuming alignment columns (abpoa_nodes.len())
    uint32_t start_off, end_off;  // aln_start_offset / aln_end_offset (align.rs:1155-1156)
    uint32_t aligned;             // n_aligned_bases
    uint32_t flags;               // 1: written; 2: no room in the arena (the host falls back to the operations)
    uint32_t pad;
};

__device__ __forceinline__ int ptx_ndigits(uint32_t v)
{
    return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 : v < 10000000u ? 7 : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}
__device__ __forceinline__ void ptx_put_u(char *w, uint32_t v, int nd)
{
    for (int i = nd - 1; i >= 0; i--) { w[i] = (char)('0' + v % 10u); v /= 10u; }
}
__device__ __forceinline__ char ptx_lower(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c + 32) : c; }
// inclusive scans over the wave (DPP: poa_wave_scan_max's stages with an add / a max)
__device__ __forceinline__ int ptx_scan_add(int v)
{
    int t;
    t = poa_dpp<0x111, 0xf>(0, v); v += t;
    t = poa_dpp<0x112, 0xf>(0, v); v += t;
    t = poa_dpp<0x114, 0xf>(0, v); v += t;
    t = poa_dpp<0x118, 0xf>(0, v); v += t;
    t = poa_dpp<0x142, 0xa>(0, v); v += t;
    t = poa_dpp<0x143, 0xc>(0, v); v += t;
    return v;
}

__global__ __launch_bounds__(64) void k_poa_text(uint32_t n, const poa_prob *__restrict__ probs, const poa_out *__restrict__ outs,
                                                  const uint8_t *__restrict__ ops, const uint32_t *__restrict__ orow, const poa_row *__restrict__ rows,
                                                  const uint4 *__restrict__ node_tab, const char *__restrict__ seq, const char *__restrict__ queries,
                                                  char *__restrict__ arena, uint32_t arena_bytes, unsigned long long *__restrict__ cursor,
                                                  poa_text_out *__restrict__ touts)
{
    const uint32_t pi = blockIdx.x;
    if (pi >= n) return;
    const int lane = threadIdx.x;
    const poa_prob pb = probs[pi];
    const poa_out po = outs[pi];
    poa_text_out T = {};
    if (po.status != POA_ST_OK) {
        if (lane == 0) touts[pi] = T;
        return;
    }
    const uint32_t nops = po.nops;
    const uint8_t *op_p = ops + pb.ops0;
    const uint32_t *or_p = orow + pb.ops0;
    const poa_row *R = rows + pb.row0;
    const char *bases = seq + pb.seq0;  // row r is bases[r - 1]
    const char *q = queries + pb.q0;
    const uint4 *ntab = node_tab + pb.node0;
    const uint32_t nv = pb.n_nodes;  // entries incl. the virtual source (entry 0)
    auto node_of = [&](uint32_t row) -> uint32_t {  // 0-based index of the real node that holds `row`
        uint32_t lo = 1, hi = nv;  // the last entry in [1, nv) whose first row is <= row
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (ntab[mid].x <= row) lo = mid; else hi = mid;
        }
        return lo - 1;
    };
    char *cs_w = nullptr, *cg_w = nullptr;
    uint32_t *runs_w = nullptr;
    uint32_t first_row = 0, last_row = 0, v_first = 0, v_last = 0;
    for (int pass = 0; pass < 2; pass++) {
        const bool wr = pass == 1;
        // state carried from block to block
        int c_cls = 5;       // cs class of the last operation seen (0 E, 1 X, 2 I, 3 D; 5: none yet)
        uint32_t c_erun = 0; // equal bases at the end of what has been seen
        int c_cop = 3;       // CIGAR operation of the last operation seen (3: none yet)
        uint32_t c_crun = 0; // length so far of the CIGAR run that is open
        uint32_t qi0 = 0, cs_n = wr ? 5u : 5u, cg_n = 0, n_runs = 0, n_path = 0, aligned = 0;
        if (wr && lane < 5) cs_w[lane] = "cs:Z:"[lane];
        bool seen_path = false;
        for (uint32_t base = 0; base <= nops; base += 64) {
            const uint32_t f = base + (uint32_t)lane;
            const bool valid = f < nops;
            const bool closing = f == nops;  // the virtual operation behind the last one: closes the runs that are open
            uint32_t op = 3, row = 0;
            if (valid) { op = op_p[nops - 1 - f]; row = or_p[nops - 1 - f]; }
            const bool cq = valid && (op == 0 || op == 1), cg = valid && (op == 0 || op == 2);
            const uint64_t qmask = __builtin_amdgcn_ballot_w64(cq);
            const uint32_t qi = qi0 + (uint32_t)__builtin_popcountll(qmask & ((1ull << lane) - 1ull));
            const char gb = cg ? bases[row - 1] : 0, qb = cq ? q[qi] : 0;
            // ---- cs
            int cls = closing ? 4 : 5;
            if (valid) cls = op == 0 ? (gb == qb ? 0 : 1) : (op == 1 ? 2 : 3);
            const int pcls = t4_shr1_mov(cls, c_cls);
            const bool non_e = cls != 0 && (valid || closing);
            const int m = poa_wave_scan_max(non_e ? lane : -1);
            const int ex = t4_shr1_mov(m, -1);
            const uint32_t erun = ex >= 0 ? (uint32_t)(lane - 1 - ex) : (uint32_t)lane + c_erun;
            const bool flush = non_e && pcls == 0;
            const int nd = flush ? ptx_ndigits(erun) : 0;
            const int own = cls == 1 ? 3 : (cls == 2 ? 1 + (pcls != 2) : (cls == 3 ? 1 + (pcls != 3) : 0));
            const int len = (flush ? 1 + nd : 0) + own;
            const int inc = ptx_scan_add(len);
            if (wr && len) {
                char *w = cs_w + cs_n + (uint32_t)(inc - len);
                if (flush) { *w++ = ':'; ptx_put_u(w, erun, nd); w += nd; }
                if (cls == 1) { w[0] = '*'; w[1] = ptx_lower(gb); w[2] = ptx_lower(qb); }
                else if (cls == 2) { if (pcls != 2) *w++ = '+'; *w = ptx_lower(qb); }
                else if (cls == 3) { if (pcls != 3) *w++ = '-'; *w = ptx_lower(gb); }
            }
            cs_n += (uint32_t)__builtin_amdgcn_readlane(inc, 63);
            {
                const int m63 = __builtin_amdgcn_readlane(m, 63);
                c_erun = m63 >= 0 ? (uint32_t)(63 - m63) : c_erun + 64u;
                c_cls = __builtin_amdgcn_readlane(cls, 63);
            }
            // ---- CIGAR
            const int cop = valid ? (int)op : (closing ? 4 : 3);
            const int pcop = t4_shr1_mov(cop, c_cop);
            const bool start = (valid || closing) && cop != pcop;
            const int m2 = poa_wave_scan_max(start ? lane : -1);
            const int ex2 = t4_shr1_mov(m2, -1);
            const uint32_t crun = ex2 >= 0 ? (uint32_t)(lane - ex2) : (uint32_t)lane + c_crun;
            const bool cflush = start && pcop != 3;
            const int nd2 = cflush ? ptx_ndigits(crun) : 0;
            const int len2 = cflush ? nd2 + 1 : 0;
            const int inc2 = ptx_scan_add(len2);
            if (wr && len2) {
                char *w = cg_w + cg_n + (uint32_t)(inc2 - len2);
                ptx_put_u(w, crun, nd2);
                w[nd2] = pcop == 0 ? 'M' : (pcop == 1 ? 'I' : 'D');
            }
            cg_n += (uint32_t)__builtin_amdgcn_readlane(inc2, 63);
            {
                const int m63 = __builtin_amdgcn_readlane(m2, 63);
                c_crun = m63 >= 0 ? (uint32_t)(64 - m63) : c_crun + 64u;
                c_cop = __builtin_amdgcn_readlane(cop, 63);
            }
            // ---- node path
            const bool opens = cg && R[row].npred != 0u;
            const uint64_t omask = __builtin_amdgcn_ballot_w64(opens), gmask = __builtin_amdgcn_ballot_w64(cg);
            if (wr && opens) runs_w[n_runs + (uint32_t)__builtin_popcountll(omask & ((1ull << lane) - 1ull))] = node_of(row);
            n_runs += (uint32_t)__builtin_popcountll(omask);
            n_path += (uint32_t)__builtin_popcountll(gmask);
            aligned += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(valid && op == 0));
            if (!wr && gmask) {
                if (!seen_path) { first_row = (uint32_t)__builtin_amdgcn_readlane((int)row, __builtin_ctzll(gmask)); seen_path = true; }
                last_row = (uint32_t)__builtin_amdgcn_readlane((int)row, 63 - __builtin_clzll(gmask));
            }
            qi0 += (uint32_t)__builtin_popcountll(qmask);
        }
        if (!wr) {
            // ---- claim: cs | cigar | (4-aligned) node indices, one atomic add
            const uint32_t bytes = ((cs_n + cg_n + 3u) & ~3u) + 4u * n_runs;
            unsigned long long at = 0;
            if (lane == 0) at = atomicAdd(cursor, (unsigned long long)((bytes + 15u) & ~15u));
            at = poa_uniform_u64(at);
            T.cs_len = cs_n; T.cg_len = cg_n; T.n_runs = n_runs; T.n_path = n_path; T.aligned = aligned;
            if (at + bytes > (unsigned long long)arena_bytes) {
                T.flags = 2u;
                if (lane == 0) touts[pi] = T;
                return;
            }
            T.cs_off = (uint32_t)at; T.cg_off = T.cs_off + cs_n; T.runs_off = T.cs_off + ((cs_n + cg_n + 3u) & ~3u);
            cs_w = arena + T.cs_off; cg_w = arena + T.cg_off; runs_w = (uint32_t *)(arena + T.runs_off);
            if (n_path) {
                v_first = node_of(first_row); v_last = node_of(last_row);
                T.start_off = first_row - ntab[v_first + 1].x;
                T.end_off = last_row - ntab[v_last + 1].x + 1u;
            }
        }
    }
    T.flags = 1u;
    if (lane == 0) touts[pi] = T;
}

// The second half of K4c's hand-over: the text (its length is known to the host by now) goes to pinned host memory through a
// KERNEL, not through hipMemcpyAsync.  A copy issued while another sub-batch is in flight lands in a DMA queue behind that
// sub-batch's own result copies, which wait for its DP kernel -- 35 ms for 18 MB on config 5 (measured in the command line
// tool); 16-byte loads and stores from the compute units cross PCIe at once.
__global__ __launch_bounds__(256) void k_poa_text_to_host(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n16)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u) dst[i] = src[i];
}
