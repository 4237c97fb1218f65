// vgh_map.cpp -- reads, GAF records and the map_reads driver.
//   read_seqs_from_file ........ src/io.rs:74-162
//   GAFAlignment::from_chain ... src/align.rs:762-911 (+ AnchorPosOnGraph::new, src/chain.rs:90-127)
//   from_placeholder_chain ..... src/align.rs:913-930
//   generate_alignment ......... src/align.rs:1096-1168
//   GAFAlignment::to_string .... src/align.rs:971-1027
//   map_reads .................. src/map.rs:27-216, write_gaf_to_file 219-226
// The reference's debug printing inside the hot loops is not part of the contract and is not reproduced.
#include "vgh.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <memory>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace vgh {

void trace_mark(const char *what)
{
    static const bool on = getenv("VGA_TRACE") != nullptr;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    static auto prev = t0;
    const auto t = std::chrono::steady_clock::now();
    long rss_pages = 0;  // (resident host memory: what the exit of the process will have to give back)
    if (FILE *f = fopen("/proc/self/statm", "r")) { long sz = 0; if (fscanf(f, "%ld %ld", &sz, &rss_pages) != 2) rss_pages = 0; fclose(f); }
    fprintf(stderr, "[vgh-trace] %-44s +%9.3f ms  (at %9.3f ms, %5.2f GB resident)\n", what, std::chrono::duration<double, std::milli>(t - prev).count(),
            std::chrono::duration<double, std::milli>(t - t0).count(), (double)rss_pages * 4096.0 / 1e9);
    prev = t;
}

std::vector<QuerySequence> read_seqs_from_file(const std::string &filename)
{
    std::ifstream in(filename);
    if (!in) throw Error("cannot open " + filename);
    size_t dot = filename.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : filename.substr(dot + 1);
    bool fasta = ext == "fasta" || ext == "fa", fastq = ext == "fastq" || ext == "fq";
    if (!fasta && !fastq) throw Error("Unrecognized file type");  // io.rs:86
    std::vector<QuerySequence> seqs;
    std::string line;
    auto chomp = [](std::string &s) { if (!s.empty() && s.back() == '\r') s.pop_back(); };
    if (fasta) {
        // io.rs:100-122: every non-empty sequence line is its own read; the 2nd+ line under one header gets
        // the name suffixed with 1, 2, ...
        std::string name;
        int same = 0;
        while (std::getline(in, line)) {
            chomp(line);
            if (!line.empty() && line[0] == '>') { name = line.substr(1); same = 0; }
            else if (!line.empty()) {
                seqs.push_back({same == 0 ? name : name + std::to_string(same), line});
                same++;
            }
        }
    } else {
        std::string l1, l2, l3, l4;  // io.rs:124-130: strict 4-line records
        while (std::getline(in, l1) && std::getline(in, l2) && std::getline(in, l3) && std::getline(in, l4)) {
            chomp(l1); chomp(l2);
            seqs.push_back({l1.empty() ? "" : l1.substr(1), l2});
        }
    }
    return seqs;
}

std::string gaf_placeholder(const QuerySequence &q)
{
    return q.name + "\t" + std::to_string(q.seq.size()) + "\t*\t*\t*\t*\t*\t*\t*\t*\t*\t0\t*\n";
}

namespace {
inline void put_u64(std::string &s, uint64_t v)
{
    char t[24];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s.push_back(t[--n]);
}
}  // namespace

std::string gaf_from_chain(const Index &ix, const QuerySequence &q, const vga_map_result *m, uint64_t read, uint64_t chain)
{
    if (m->chain_placeholder[chain]) return gaf_placeholder(q);
    const uint64_t a0 = m->anchor_off[read];
    const uint64_t c0 = m->chain_anchor_off[chain], c1 = m->chain_anchor_off[chain + 1];
    const uint64_t k = ix.kmer_length;
    std::string path;
    path.reserve((c1 - c0) * 28);
    // node of a forward position (get_bv_rank): the anchors of a chain ascend on the target, so the node is found by
    // walking on from the previous one; a position before it falls back to the binary search
    const size_t n_ref = ix.node_ref.size();
    uint64_t cur = 0;  // = rank of the last position looked up (0: none yet)
    auto node_of = [&](uint64_t pos) -> uint64_t {
        if (cur == 0 || ix.node_ref[cur - 1].seq_idx > pos) cur = ix.node_id_from_fwd_pos(pos);
        else {
            uint64_t steps = 0;
            while (cur < n_ref && ix.node_ref[cur].seq_idx <= pos) {
                cur++;
                if (++steps == 64) { cur = ix.node_id_from_fwd_pos(pos); break; }
            }
        }
        return cur;
    };
    for (uint64_t t = c0; t < c1; t++) {
        const uint64_t ai = a0 + m->chain_anchor_idx[t];
        const uint64_t tb = m->target_begin[ai], te_incl = (uint64_t)m->target_end[ai] - 1;
        const uint64_t fn = node_of(tb);
        const uint64_t fo = tb - ix.get_bv_select(fn);
        const uint64_t ln = node_of(te_incl);
        const uint64_t lo = te_incl - ix.get_bv_select(ln);
        path += "(>"; put_u64(path, fn); path.push_back(':'); put_u64(path, fo);
        path += ",>"; put_u64(path, ln); path.push_back(':'); put_u64(path, lo);
        path += "),";
    }
    const uint64_t first = a0 + m->chain_anchor_idx[c0], last = a0 + m->chain_anchor_idx[c1 - 1];
    // plen pstart pend residue block = 0; mapq = min(f64::MIN as u64, 254) = 0 (align.rs:904, chain.rs:203)
    std::string out;
    out.reserve(q.name.size() + path.size() + 96);
    out += q.name; out.push_back('\t'); put_u64(out, q.seq.size()); out.push_back('\t'); put_u64(out, m->query_begin[first]);
    out.push_back('\t'); put_u64(out, m->query_begin[last] + k); out += "\t+\t"; out += path;
    out += "\t0\t0\t0\t0\t0\t0\tta:Z:chain,n_anchors: "; put_u64(out, c1 - c0); out.push_back('\n');
    return out;
}

// the same record with the path column already written (vga_chain_paths_text: the GPU formats it)
void gaf_from_chain_text(std::string &out, const Index &ix, const QuerySequence &q, const vga_map_result *m, uint64_t read, uint64_t chain,
                         const char *path, uint64_t path_len)
{
    if (m->chain_placeholder[chain]) { out += gaf_placeholder(q); return; }
    const uint64_t a0 = m->anchor_off[read];
    const uint64_t c0 = m->chain_anchor_off[chain], c1 = m->chain_anchor_off[chain + 1];
    const uint64_t first = a0 + m->chain_anchor_idx[c0], last = a0 + m->chain_anchor_idx[c1 - 1];
    out += q.name; out.push_back('\t'); put_u64(out, q.seq.size()); out.push_back('\t'); put_u64(out, m->query_begin[first]);
    out.push_back('\t'); put_u64(out, m->query_begin[last] + ix.kmer_length); out += "\t+\t"; out.append(path, path_len);
    out += "\t0\t0\t0\t0\t0\t0\tta:Z:chain,n_anchors: "; put_u64(out, c1 - c0); out.push_back('\n');
}

namespace {
// decimal digits through a raw pointer (the records are tens of kilobytes of small numbers: std::string::push_back per
// character, with its capacity check, is what the text threads would spend their time in)
inline char *put_u64_raw(char *p, uint64_t v)
{
    char t[24];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = t[--n];
    return p;
}
inline char *put_str_raw(char *p, const char *s, size_t n) { memcpy(p, s, n); return p + n; }
}  // namespace

void gaf_from_alignment(std::string &out, const QuerySequence &q, const vga_align_result *a, uint64_t r)
{
    if (!a->aligned[r]) { out += gaf_placeholder(q); return; }
    const char *cs = a->cs + a->cs_off[r], *cg = a->cigar + a->cigar_off[r];
    const size_t n_cs = strlen(cs), n_cg = strlen(cg);
    const size_t n_path = (size_t)(a->path_off[r + 1] - a->path_off[r]);
    // an upper bound of the record (a path step is '>' and the digits of at most the largest id), written in place and cut to
    // what was used
    uint64_t max_id = 0;
    for (uint64_t t = a->path_off[r]; t < a->path_off[r + 1]; t++) max_id = std::max<uint64_t>(max_id, id_of(a->path_handles[t]));
    size_t step = 2;
    for (uint64_t v = max_id; v >= 10; v /= 10) step++;
    const size_t old = out.size(), bound = q.name.size() + n_path * step + n_cs + n_cg + 256;
    if (out.capacity() < old + bound) out.reserve(std::max(old + bound, 2 * out.capacity()));  // (records are appended: grow geometrically)
    out.resize(old + bound);
    char *p = &out[old];
    // align.rs:1145-1167: qstart 0, qend len, '+', residue 0, mapq 255, literal "as:i:-30"
    p = put_str_raw(p, q.name.data(), q.name.size()); *p++ = '\t'; p = put_u64_raw(p, q.seq.size()); p = put_str_raw(p, "\t0\t", 3);
    p = put_u64_raw(p, q.seq.size()); p = put_str_raw(p, "\t+\t", 3);
    for (uint64_t t = a->path_off[r]; t < a->path_off[r + 1]; t++) {
        const Handle h = a->path_handles[t];
        *p++ = is_rev(h) ? '<' : '>';
        p = put_u64_raw(p, id_of(h));
    }
    *p++ = '\t'; p = put_u64_raw(p, a->path_length[r]); *p++ = '\t'; p = put_u64_raw(p, a->path_start[r]);
    *p++ = '\t'; p = put_u64_raw(p, a->path_end[r]); p = put_str_raw(p, "\t0\t", 3); p = put_u64_raw(p, a->block_length[r]);
    p = put_str_raw(p, "\t255\tas:i:-30 ", 14); p = put_str_raw(p, cs, n_cs); p = put_str_raw(p, ",cg:Z:", 6); p = put_str_raw(p, cg, n_cg); *p++ = '\n';
    out.resize((size_t)(p - out.data()));
}

std::string gaf_from_alignment(const QuerySequence &q, const vga_align_result *a, uint64_t r)
{
    std::string out;
    gaf_from_alignment(out, q, a, r);
    return out;
}

// src/validate.rs:36-102.  The record is derived from the GAF record alone, as in the reference: read name, the last
// comma-separated piece of the notes (cg:Z:<CIGAR>), the sequence of the FIRST read with that name (validate.rs:133-136),
// the node ids of the path (both orientations parse to the bare id, validate.rs:104-111) and their sequences --
// reverse-complemented when the last id is smaller than the first (validate.rs:50-53,113-124).  Debug formatting of
// Vec<u64> / Vec<String>: [1, 2] and ["AC", "G"].
std::string validation_record(const Index &ix, const std::string &gaf_line, const std::vector<QuerySequence> &reads)
{
    std::vector<std::string> f;
    size_t a = 0;
    std::string ln = gaf_line;
    if (!ln.empty() && ln.back() == '\n') ln.pop_back();
    for (;;) {
        size_t b = ln.find('\t', a);
        f.push_back(ln.substr(a, b == std::string::npos ? std::string::npos : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
    }
    if (f.size() < 13) throw Error("validation: malformed GAF record");
    const std::string &name = f[0], &path = f[5];
    const QuerySequence *read = nullptr;
    for (const QuerySequence &q : reads)
        if (q.name == name) { read = &q; break; }
    if (!read) throw Error("validation: no read named " + name);  // the reference unwraps
    if (path == "*") return name + "\nNOT ALIGNED\n" + read->seq + "\n[]\n[]\n\n";
    std::vector<uint64_t> ids;
    for (size_t i = 0; i < path.size(); i++)
        if ((path[i] == '>' || path[i] == '<') && i + 1 < path.size() && isdigit((unsigned char)path[i + 1])) {
            uint64_t v = 0;
            size_t j = i + 1;
            while (j < path.size() && isdigit((unsigned char)path[j])) v = v * 10 + (uint64_t)(path[j++] - '0');
            ids.push_back(v);
            i = j - 1;
        }
    const bool rev = ids.size() >= 2 && ids.back() < ids.front();
    const std::string &notes = f[12];
    const size_t comma = notes.rfind(',');
    std::string out = name + "\n" + (comma == std::string::npos ? notes : notes.substr(comma + 1)) + "\n" + read->seq + "\n[";
    for (size_t i = 0; i < ids.size(); i++) out += (i ? ", " : "") + std::to_string(ids[i]);
    out += "]\n[";
    for (size_t i = 0; i < ids.size(); i++) {
        if (ids[i] == 0 || ids[i] > ix.n_nodes) throw Error("validation: node " + std::to_string(ids[i]) + " is not in the graph");
        std::string sq = ix.seq_fwd.substr(ix.node_ref[ids[i] - 1].seq_idx, ix.node_ref[ids[i]].seq_idx - ix.node_ref[ids[i] - 1].seq_idx);
        if (rev) {  // graph.sequence(Handle::pack(id, true)): reverse complement, anything but ACGT kept
            std::string rc(sq.rbegin(), sq.rend());
            for (char &c : rc) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == 'a' ? 't' : c == 'c' ? 'g' : c == 'g' ? 'c' : c == 't' ? 'a' : c;
            sq = rc;
        }
        out += std::string(i ? ", " : "") + "\"" + sq + "\"";
    }
    return out + "]\n\n";
}

static void write_file(const std::string &name, const std::string &body)
{
    std::ofstream o(name, std::ios::binary);
    if (!o) throw Error("Couldn't create file " + name);
    o.write(body.data(), body.size());
    if (!o) throw Error("Couldn't write to file " + name);
}

std::vector<Shard> plan_shards(const std::vector<uint64_t> &len, uint32_t n_slots, uint64_t chunk_reads)
{
    std::vector<Shard> out;
    if (n_slots == 0) n_slots = 1;
    const uint64_t n = len.size();
    uint64_t total = 0;
    for (uint64_t l : len) total += l;
    uint64_t start = 0, acc = 0;
    for (uint32_t slot = 0; slot < n_slots; slot++) {
        uint64_t end = start;
        if (slot == n_slots - 1) end = n;
        else {
            // (the arithmetic of sharding.py::split_by_bases, in doubles like Python's)
            const double target = (double)(total * (uint64_t)(slot + 1)) / (double)n_slots;
            while (end < n && (double)acc + (double)len[end] / 2.0 <= target) { acc += len[end]; end++; }
        }
        // chunks of equal size, as few as chunk_reads allows: a slice of 100 000 reads in chunks of at most 32 768 is four chunks of
        // 25 000, not three full ones and a rest of 1 696 -- every chunk ends with the tail of its longest alignments, a cost the
        // rest would pay for almost nothing
        const uint64_t len_slice = end - start;
        const uint64_t n_chunks = chunk_reads ? (len_slice + chunk_reads - 1) / chunk_reads : (len_slice ? 1 : 0);
        for (uint64_t q = 0; q < n_chunks; q++) {
            const uint64_t c = start + len_slice * q / n_chunks, e = start + len_slice * (q + 1) / n_chunks;
            if (e > c) out.push_back({c, e, slot});
        }
        start = end;
    }
    return out;
}

namespace {

// GAF text of reads [0, n) built by several threads over contiguous ranges: the pieces, in read order
// (size_of(r): an upper estimate of read r's text, so that a piece is allocated once; nullptr_t: none)
template <typename F, typename S>
std::vector<std::string> text_of_reads(uint64_t n, unsigned n_threads, F per_read, S size_of)
{
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_threads, n / 64 + 1));
    std::vector<std::string> parts(T);
    std::vector<std::string> errs(T);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            try {
                const uint64_t a = n * t / T, b = n * (t + 1) / T;
                if constexpr (!std::is_same<S, std::nullptr_t>::value) {
                    size_t tot = 0;
                    for (uint64_t r = a; r < b; r++) tot += size_of(r);
                    parts[t].reserve(tot);
                }
                for (uint64_t r = a; r < b; r++) per_read(r, parts[t]);
            } catch (const std::exception &e) { errs[t] = e.what(); if (errs[t].empty()) errs[t] = "error"; }
        });
    for (auto &x : th) x.join();
    for (auto &e : errs)
        if (!e.empty()) throw Error(e);
    return parts;
}

void append_pieces(std::string &dst, std::vector<std::string> &pieces)
{
    size_t tot = dst.size();
    for (auto &q : pieces) tot += q.size();
    dst.reserve(tot);
    for (auto &q : pieces) { dst += q; std::string().swap(q); }
}

// host threads that put GAF text together (and copy it into the files) for one context: the cores of the host shared between the
// device slots -- 8 ranks on one host must not each start 32 threads -- with VGA_HOST_THREADS as the diagnostic override
unsigned text_threads(uint32_t n_slots)
{
    if (const char *e = getenv("VGA_HOST_THREADS")) return (unsigned)std::max(1, atoi(e));
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return std::max(2u, std::min(32u, hw / std::max(1u, n_slots)));
}

// A GAF file that grows by whole chunks whose text already exists in pieces: the file is extended and the pieces are written at
// their offsets (known once a chunk's text exists: the order is fixed by the plan) by several threads at once, in slices of
// 8 MiB -- pwrite by default; VGA_GAF_MMAP=1 copies into a shared mapping of the new range instead.  Measured on the GPU box
// (overlay file system, tests/prof_textpath.py, config 5, 973 MB per round): one writer thread 1.1 GB/s (round 3), the mapping
// 1.6-1.7 GB/s whatever the number of threads (page faults), pwrite from 16 threads 4.5 GB/s.
class AppendFile {
public:
    explicit AppendFile(const std::string &path) : path_(path)
    {
        fd_ = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd_ < 0) throw Error("Couldn't create file " + path);
    }
    ~AppendFile() { if (fd_ >= 0) ::close(fd_); }
    AppendFile(const AppendFile &) = delete;
    AppendFile &operator=(const AppendFile &) = delete;
    uint64_t size() const { return size_; }
    // appends the pieces in order (and frees them); up to n_threads copy
    void append(std::vector<std::string> &pieces, unsigned n_threads)
    {
        uint64_t total = 0;
        for (const std::string &q : pieces) total += q.size();
        if (total == 0) { pieces.clear(); return; }
        const uint64_t old = size_;
        if (::ftruncate(fd_, (off_t)(old + total)) != 0) throw Error("Couldn't write to file " + path_);
        // slices of at most 8 MiB, dealt to the threads round robin
        struct Slice { const char *src; uint64_t off, len; };
        std::vector<Slice> slices;
        {
            uint64_t off = old;
            for (const std::string &q : pieces) {
                for (uint64_t a = 0; a < q.size(); a += (8u << 20)) {
                    const uint64_t len = std::min<uint64_t>(8u << 20, q.size() - a);
                    slices.push_back({q.data() + a, off + a, len});
                }
                off += q.size();
            }
        }
        const long page = sysconf(_SC_PAGESIZE);
        const uint64_t map_off = old & ~((uint64_t)page - 1);
        const size_t map_len = (size_t)(old + total - map_off);
        char *base = use_mmap_ ? (char *)::mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, (off_t)map_off) : (char *)MAP_FAILED;
        if (base == (char *)MAP_FAILED) use_mmap_ = false;
        const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(n_threads, slices.size()));
        std::vector<std::string> errs(T);
        auto work = [&](unsigned t) {
            for (size_t i = t; i < slices.size(); i += T) {
                const Slice &sl = slices[i];
                if (use_mmap_) memcpy(base + (sl.off - map_off), sl.src, sl.len);
                else {
                    uint64_t done = 0;
                    while (done < sl.len) {
                        const ssize_t w = ::pwrite(fd_, sl.src + done, sl.len - done, (off_t)(sl.off + done));
                        if (w <= 0) { errs[t] = "Couldn't write to file " + path_; return; }
                        done += (uint64_t)w;
                    }
                }
            }
        };
        if (T == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; t++) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        if (use_mmap_) ::munmap(base, map_len);
        for (const std::string &e : errs)
            if (!e.empty()) throw Error(e);
        size_ = old + total;
        for (std::string &q : pieces) std::string().swap(q);
        pieces.clear();
    }

private:
    std::string path_;
    int fd_ = -1;
    uint64_t size_ = 0;
    bool use_mmap_ = getenv("VGA_GAF_MMAP") != nullptr;
};

struct ChunkOut {
    std::vector<std::string> chains, aligns;  // GAF text in read order, in pieces
    uint64_t n_aligned = 0, n_anchors = 0, poa_cells = 0;
    double ms_map = 0, ms_align = 0;
};

// what the caller may do before map_chunk returns
struct ChunkHooks {
    std::function<void()> chains_ready;  // ChunkOut::chains is final (called from the chains thread)
    std::function<void()> gpu_done;      // the chunk needs its context no longer (the alignments' text is still to be written)
};

// anchors -> chains -> (alignments) of reads [b, e) on one context; the GAF text of exactly those reads
void map_chunk(vga_ctx *ctx, const Index &ix, const std::vector<QuerySequence> &inputs, uint64_t b0, uint64_t e0, const MapOptions &opt, ChunkOut &out,
               const ChunkHooks &hooks, unsigned T)
{
    const uint64_t n = e0 - b0;
    const bool trace = getenv("VGA_TRACE") != nullptr;
    const auto t_first = std::chrono::steady_clock::now();
    auto t_prev = t_first;
    auto mark = [&](const char *what) {
        if (!trace) return;
        const auto t = std::chrono::steady_clock::now();
        long rss_pages = 0;
        if (FILE *f = fopen("/proc/self/statm", "r")) { long sz = 0; if (fscanf(f, "%ld %ld", &sz, &rss_pages) != 2) rss_pages = 0; fclose(f); }
        fprintf(stderr, "[vgh-trace] reads [%llu, %llu): %-28s %9.3f ms  (%5.2f GB resident)\n", (unsigned long long)b0, (unsigned long long)e0, what,
                std::chrono::duration<double, std::milli>(t - t_prev).count(), (double)rss_pages * 4096.0 / 1e9);
        t_prev = t;
    };
    std::string concat;
    std::vector<uint64_t> off(n + 1, 0);
    uint64_t tot = 0;
    for (uint64_t i = 0; i < n; i++) tot += inputs[b0 + i].seq.size();
    concat.reserve(tot);
    for (uint64_t i = 0; i < n; i++) { concat += inputs[b0 + i].seq; off[i + 1] = concat.size(); }
    vga_batch *b = nullptr;
    if (vga_batch_create(ctx, concat.data(), off.data(), n, &b) != VGA_OK) throw Error(vga_last_error(ctx));
    std::unique_ptr<vga_batch, void (*)(vga_batch *)> b_owner(b, vga_batch_destroy);
    if (opt.also_align) {  // the traceback memory of the alignment call begins to be allocated beside the chaining
        uint64_t longest = 0;
        for (uint64_t i = 0; i < n; i++) longest = std::max<uint64_t>(longest, inputs[b0 + i].seq.size());
        if (longest < (1u << 24)) (void)vga_align_prepare(ctx, n, (uint32_t)longest);
    }
    vga_map_params mp;
    vga_map_default_params(&mp);
    mp.bandwidth = (uint32_t)opt.bandwidth;
    mp.max_gap = opt.max_gap;
    mp.chain_min_n_anchors = (uint32_t)opt.chain_min_n_anchors;
    mp.emit_dp = 0;  // the GAF writers read anchor coordinates and chain membership only
    vga_map_result *m = nullptr;
    if (vga_map_batch(b, &mp, &m) != VGA_OK) throw Error(vga_last_error(ctx));
    std::unique_ptr<vga_map_result, void (*)(vga_map_result *)> m_owner(m, vga_map_result_free);
    out.n_anchors = m->n_anchors;
    out.ms_map = m->ms_total;
    mark("batch + vga_map_batch");
    // chains GAF (map.rs:123-145): every chain of every read, in order.  The path column of every chain is written by the GPU
    // (K6) and the records are put together around it, on a thread beside the alignment call: K6 works on a stream of its own
    // (include/vga_hip.h) and the rest only reads the chains
    std::string chain_err;
    std::mutex k6_mu;
    std::condition_variable k6_cv;
    bool k6_done = false;
    std::thread chains_thread([&]() {
        auto k6_finished = [&]() { { std::lock_guard<std::mutex> lk(k6_mu); k6_done = true; } k6_cv.notify_all(); };
        try {
            const auto t0 = std::chrono::steady_clock::now();
            vga_chain_text *ct = nullptr;
            const int rc = vga_chain_paths_text(ctx, m, &ct);
            const std::string e = rc != VGA_OK ? vga_last_error(ctx) : "";
            k6_finished();
            if (rc != VGA_OK) throw Error(e);
            std::unique_ptr<vga_chain_text, void (*)(vga_chain_text *)> ct_owner(ct, vga_chain_text_free);
            const auto t1 = std::chrono::steady_clock::now();
            out.chains = text_of_reads(n, opt.also_align ? std::max(1u, T / 2) : T, [&](uint64_t r, std::string &dst) {
                for (uint64_t c = m->chain_off[r]; c < m->chain_off[r + 1]; c++)
                    gaf_from_chain_text(dst, ix, inputs[b0 + r], m, r, c, ct->text + ct->text_off[c], ct->text_off[c + 1] - ct->text_off[c]);
            }, [&](uint64_t r) {
                const uint64_t c0 = m->chain_off[r], c1 = m->chain_off[r + 1];
                return (size_t)(ct->text_off[c1] - ct->text_off[c0]) + (size_t)(c1 - c0) * (inputs[b0 + r].name.size() + 128);
            });
            if (trace)
                fprintf(stderr, "[vgh-trace] reads [%llu, %llu): (beside) vga_chain_paths_text %.3f ms, chains GAF text %.3f ms, ready %.3f ms after the chunk began\n",
                        (unsigned long long)b0, (unsigned long long)e0, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_first).count());
            if (hooks.chains_ready) hooks.chains_ready();
        } catch (const std::exception &e) {
            chain_err = e.what();
            if (chain_err.empty()) chain_err = "chains GAF text failed";
            k6_finished();
        }
    });
    struct joiner {  // (an exception below must not leave the thread joinable: std::terminate)
        std::thread &t;
        ~joiner() { if (t.joinable()) t.join(); }
    } chains_joiner{chains_thread};
    auto wait_k6 = [&]() { std::unique_lock<std::mutex> lk(k6_mu); k6_cv.wait(lk, [&]() { return k6_done; }); };
    if (opt.also_align) {
        vga_poa_params pp;
        vga_poa_default_params(&pp);
        if (opt.poa_remain_rule >= 0) pp.remain_rule = opt.poa_remain_rule;
        vga_align_result *a = nullptr;
        if (vga_align_batch(b, m, (uint32_t)opt.align_best_n, &pp, &a) != VGA_OK) {
            const std::string e = vga_last_error(ctx);
            wait_k6();
            throw Error(e);
        }
        std::unique_ptr<vga_align_result, void (*)(vga_align_result *)> a_owner(a, vga_align_result_free);
        out.ms_align = a->ms_total;
        out.poa_cells = a->poa_cells;
        mark("vga_align_batch");
        // everything the text needs is in host memory now: the batch and, if the caller says so, the context can go
        wait_k6();
        b_owner.reset();
        if (hooks.gpu_done) hooks.gpu_done();
        out.aligns = text_of_reads(n, T, [&](uint64_t r, std::string &dst) { gaf_from_alignment(dst, inputs[b0 + r], a, r); },
                                   [&](uint64_t r) {
                                       return (size_t)(a->cs_off[r + 1] - a->cs_off[r]) + (size_t)(a->cigar_off[r + 1] - a->cigar_off[r]) +
                                              (size_t)(a->path_off[r + 1] - a->path_off[r]) * 8 + inputs[b0 + r].name.size() + 160;
                                   });
        for (uint64_t r = 0; r < n; r++) out.n_aligned += a->aligned[r];
        mark("alignments GAF text");
    } else {
        wait_k6();
        b_owner.reset();
        if (hooks.gpu_done) hooks.gpu_done();
    }
    chains_thread.join();
    mark("chains GAF text (joined)");
    if (!chain_err.empty()) throw Error(chain_err);
}

}  // namespace

// Diagnostics (tests/prof_textpath.py): the text half of the product path on its own.  One chunk is mapped and aligned once; then,
// `repeat` times, its GAF text is put together (chains and alignments side by side, as in map_chunk) and appended to two files
// (AppendFile), and the rate of GAF bytes is what comes back.  Nothing of this is on the path of map_reads / map_reads_multi.
TextReplay textpath_replay(vga_ctx *ctx, const Index &ix, const std::vector<QuerySequence> &inputs, const MapOptions &opt, const std::string &out_prefix,
                           unsigned repeat, unsigned n_threads)
{
    const uint64_t n = inputs.size();
    std::string concat;
    std::vector<uint64_t> off(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) { concat += inputs[i].seq; off[i + 1] = concat.size(); }
    vga_batch *b = nullptr;
    if (vga_batch_create(ctx, concat.data(), off.data(), n, &b) != VGA_OK) throw Error(vga_last_error(ctx));
    std::unique_ptr<vga_batch, void (*)(vga_batch *)> b_owner(b, vga_batch_destroy);
    vga_map_params mp;
    vga_map_default_params(&mp);
    mp.bandwidth = (uint32_t)opt.bandwidth; mp.max_gap = opt.max_gap; mp.chain_min_n_anchors = (uint32_t)opt.chain_min_n_anchors; mp.emit_dp = 0;
    vga_map_result *m = nullptr;
    if (vga_map_batch(b, &mp, &m) != VGA_OK) throw Error(vga_last_error(ctx));
    std::unique_ptr<vga_map_result, void (*)(vga_map_result *)> m_owner(m, vga_map_result_free);
    vga_chain_text *ct = nullptr;
    if (vga_chain_paths_text(ctx, m, &ct) != VGA_OK) throw Error(vga_last_error(ctx));
    std::unique_ptr<vga_chain_text, void (*)(vga_chain_text *)> ct_owner(ct, vga_chain_text_free);
    vga_poa_params pp;
    vga_poa_default_params(&pp);
    vga_align_result *a = nullptr;
    if (vga_align_batch(b, m, (uint32_t)opt.align_best_n, &pp, &a) != VGA_OK) throw Error(vga_last_error(ctx));
    std::unique_ptr<vga_align_result, void (*)(vga_align_result *)> a_owner(a, vga_align_result_free);
    const unsigned T = n_threads ? n_threads : text_threads(1);
    TextReplay res;
    AppendFile fc(out_prefix + "-chains.gaf"), fa(out_prefix + "-alignments.gaf");
    const auto t0 = std::chrono::steady_clock::now();
    double text_s = 0, write_s = 0;
    for (unsigned k = 0; k < repeat; k++) {
        std::string err[2];
        double ts[2] = {0, 0}, ws[2] = {0, 0};
        std::thread tc([&]() {
            try {
                const auto a0 = std::chrono::steady_clock::now();
                std::vector<std::string> pieces = text_of_reads(n, std::max(1u, T / 2), [&](uint64_t r, std::string &dst) {
                    for (uint64_t c = m->chain_off[r]; c < m->chain_off[r + 1]; c++)
                        gaf_from_chain_text(dst, ix, inputs[r], m, r, c, ct->text + ct->text_off[c], ct->text_off[c + 1] - ct->text_off[c]);
                }, [&](uint64_t r) {
                    const uint64_t c0 = m->chain_off[r], c1 = m->chain_off[r + 1];
                    return (size_t)(ct->text_off[c1] - ct->text_off[c0]) + (size_t)(c1 - c0) * (inputs[r].name.size() + 128);
                });
                const auto a1 = std::chrono::steady_clock::now();
                fc.append(pieces, std::max(2u, T));
                ts[0] = std::chrono::duration<double>(a1 - a0).count();
                ws[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - a1).count();
            } catch (const std::exception &e) { err[0] = e.what(); }
        });
        try {
            const auto a0 = std::chrono::steady_clock::now();
            std::vector<std::string> pieces = text_of_reads(n, T, [&](uint64_t r, std::string &dst) { gaf_from_alignment(dst, inputs[r], a, r); },
                                                            [&](uint64_t r) {
                                                                return (size_t)(a->cs_off[r + 1] - a->cs_off[r]) + (size_t)(a->cigar_off[r + 1] - a->cigar_off[r]) +
                                                                       (size_t)(a->path_off[r + 1] - a->path_off[r]) * 8 + inputs[r].name.size() + 160;
                                                            });
            const auto a1 = std::chrono::steady_clock::now();
            fa.append(pieces, std::max(2u, T));
            ts[1] = std::chrono::duration<double>(a1 - a0).count();
            ws[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - a1).count();
        } catch (const std::exception &e) { err[1] = e.what(); }
        tc.join();
        for (const std::string &e : err)
            if (!e.empty()) throw Error(e);
        text_s += std::max(ts[0], ts[1]);
        write_s += std::max(ws[0], ws[1]);
    }
    res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    res.bytes = fc.size() + fa.size();
    res.chains_bytes = fc.size();
    res.text_seconds = text_s;
    res.write_seconds = write_s;
    res.threads = T;
    return res;
}

namespace {

void check_aligner(const MapOptions &opt)
{
    if (opt.poa_aligner != "abpoa") {
        if (opt.poa_aligner == "rspoa") throw Error("the rspoa aligner is not available in the MI355X build yet; use -p abpoa");
        throw Error("POA Aligner not recognized");  // map_main.rs:67
    }
}

// file output and validation records of the assembled GAF texts (map.rs:135-139, 174-178, 186-208)
void finish(MapOutput &out, const Index &ix, const std::vector<QuerySequence> &inputs, const MapOptions &opt, const std::string &out_prefix)
{
    const bool same_file = out_prefix.size() >= 4 && out_prefix.compare(out_prefix.size() - 4, 4, ".gaf") == 0;
    if (!out_prefix.empty()) write_file(same_file ? out_prefix : out_prefix + "-chains.gaf", out.chains_gaf);
    if (!opt.also_align) return;
    // map.rs:174-178: a prefix ending in .gaf makes the alignments overwrite the chains file
    if (!out_prefix.empty()) write_file(same_file ? out_prefix : out_prefix + "-alignments.gaf", out.alignments_gaf);
    if (opt.also_validate) {  // map.rs:186-208
        size_t p0 = 0;
        while (p0 < out.alignments_gaf.size()) {
            const size_t p1 = out.alignments_gaf.find('\n', p0);
            out.validation += validation_record(ix, out.alignments_gaf.substr(p0, p1 - p0), inputs);
            p0 = p1 + 1;
        }
        if (!opt.validation_path.empty()) write_file(opt.validation_path, out.validation);
    }
}

}  // namespace

MapOutput map_reads(vga_ctx *ctx, const Index &ix, const std::vector<QuerySequence> &inputs, const MapOptions &opt,
                    const std::string &out_prefix)
{
    check_aligner(opt);
    MapOutput out;
    out.n_reads = inputs.size();
    out.n_devices = 1;
    std::vector<uint64_t> len(inputs.size());
    for (size_t i = 0; i < inputs.size(); i++) len[i] = inputs[i].seq.size();
    for (const Shard &s : plan_shards(len, 1, opt.chunk_reads)) {
        ChunkOut c;
        map_chunk(ctx, ix, inputs, s.begin, s.end, opt, c, ChunkHooks(), text_threads(1));
        append_pieces(out.chains_gaf, c.chains);
        append_pieces(out.alignments_gaf, c.aligns);
        out.n_aligned += c.n_aligned; out.n_anchors += c.n_anchors; out.poa_cells += c.poa_cells;
        out.ms_map += c.ms_map; out.ms_align += c.ms_align;
        out.n_chunks++;
    }
    finish(out, ix, inputs, opt, out_prefix);
    return out;
}

namespace {

// one context per device slot; with no list, every GPU vga_ctx_create accepts
std::vector<vga_ctx *> create_contexts(const std::vector<int> &devs)
{
    std::vector<vga_ctx *> ctxs;
    if (devs.empty()) {
        for (int d = 0; d < 64; d++) {
            vga_ctx *c = nullptr;
            if (vga_ctx_create(d, &c) != VGA_OK) break;
            ctxs.push_back(c);
        }
        if (ctxs.empty()) throw Error("no MI355X device available (this build has no CPU path)");
    } else {
        for (int d : devs) {
            vga_ctx *c = nullptr;
            if (vga_ctx_create(d, &c) != VGA_OK) {
                for (vga_ctx *x : ctxs) vga_ctx_destroy(x);
                throw Error("no MI355X device available: cannot create a context on device " + std::to_string(d) + " (this build has no CPU path)");
            }
            ctxs.push_back(c);
        }
    }
    return ctxs;
}

std::vector<int> device_list(const MapOptions &opt)
{
    std::vector<int> devs = opt.devices;
    if (devs.empty() && !opt.all_devices) devs.push_back(opt.device);
    return devs;
}

std::mutex g_prewarm_mu;
std::future<std::vector<vga_ctx *>> g_prewarm;
std::vector<int> g_prewarm_devs;

}  // namespace

void prewarm_contexts(const MapOptions &opt)
{
    std::lock_guard<std::mutex> lk(g_prewarm_mu);
    if (g_prewarm.valid()) return;
    g_prewarm_devs = device_list(opt);
    g_prewarm = std::async(std::launch::async, [devs = g_prewarm_devs]() { return create_contexts(devs); });
}

MapOutput map_reads_multi(const Index &ix, const std::vector<QuerySequence> &inputs, const MapOptions &opt, const std::string &out_prefix)
{
    check_aligner(opt);
    trace_mark("map_reads_multi: start");
    std::vector<vga_ctx *> ctxs;
    auto release = [&]() { for (vga_ctx *c : ctxs) if (c) vga_ctx_destroy(c); ctxs.clear(); };
    const std::vector<int> devs = device_list(opt);
    {
        std::unique_lock<std::mutex> lk(g_prewarm_mu);
        if (g_prewarm.valid()) {  // contexts prewarm_contexts began to create (its error, if any, is thrown here)
            std::future<std::vector<vga_ctx *>> f = std::move(g_prewarm);
            const bool same = g_prewarm_devs == devs;
            lk.unlock();
            std::vector<vga_ctx *> got = f.get();
            if (same) ctxs = std::move(got);
            else for (vga_ctx *c : got) vga_ctx_destroy(c);
        }
    }
    if (ctxs.empty()) ctxs = create_contexts(devs);
    const uint32_t n_slots = (uint32_t)ctxs.size();
    trace_mark("contexts created");
    // contexts that share a GPU share its memory: each takes its part of the traceback pool (a per-context setting of the
    // library; VGA_POOL_FRACTION in the environment still overrides it for diagnostics)
    if (!devs.empty()) {
        size_t most = 1;
        for (int d : devs) most = std::max<size_t>(most, (size_t)std::count(devs.begin(), devs.end(), d));
        if (most > 1)
            for (vga_ctx *c : ctxs) (void)vga_ctx_set_pool_fraction(c, 1.0 / (double)most);
    }
    {
        vga_index_desc d;
        Index::DescScratch sc;
        ix.describe(d, sc);
        for (vga_ctx *c : ctxs)
            if (vga_index_upload(c, &d) != VGA_OK) { const std::string e = vga_last_error(c); release(); throw Error(e); }
    }
    trace_mark("index uploaded");
    // the library's worker threads (CIGAR strings, result copies) are per call: the slots share the cores
    if (n_slots > 1 && !getenv("VGA_HOST_THREADS")) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        for (vga_ctx *c : ctxs) (void)vga_ctx_set_host_threads(c, std::max(2u, std::min(32u, hw / n_slots)));
    }
    std::vector<uint64_t> len(inputs.size());
    for (size_t i = 0; i < inputs.size(); i++) len[i] = inputs[i].seq.size();
    const std::vector<Shard> plan = plan_shards(len, n_slots, opt.chunk_reads);
    std::vector<ChunkOut> parts(plan.size());
    std::vector<std::string> errors(n_slots);
    // Streaming output: when the caller does not need the GAF text back (the CLI without -C / -v), a writer thread appends
    // every finished chunk to the files in read order and drops its text, so that host memory holds a few chunks, not the
    // run.  (A prefix ending in .gaf makes the alignments overwrite the chains file, map.rs:174-178: that quirk is kept
    // by the non-streaming path.)
    const bool same_file = out_prefix.size() >= 4 && out_prefix.compare(out_prefix.size() - 4, 4, ".gaf") == 0;
    const bool stream = !opt.keep_text && !out_prefix.empty() && !same_file && !opt.also_validate;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<uint8_t> done(plan.size(), 0), chains_done(plan.size(), 0);
    bool abort_writer = false, abort_workers = false;
    const unsigned T_text = text_threads(n_slots);
    // one writer per file (the chains of a chunk are ready long before its alignments, and neither should wait for the other's
    // copy), each appending the chunks in read order with several copying threads (AppendFile)
    std::string writer_err[2];
    std::thread writer[2];
    auto start_writer = [&](int which) {
        writer[which] = std::thread([&, which]() {
            try {
                AppendFile f(out_prefix + (which == 0 ? "-chains.gaf" : "-alignments.gaf"));
                for (size_t i = 0; i < plan.size(); i++) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&]() { return (which == 0 && chains_done[i]) || done[i] || abort_writer; });
                        if (!(which == 0 && chains_done[i]) && !done[i]) return;
                    }
                    const auto t0 = std::chrono::steady_clock::now();
                    const uint64_t before = f.size();
                    f.append(which == 0 ? parts[i].chains : parts[i].aligns, std::max(2u, T_text));
                    if (getenv("VGA_TRACE"))
                        fprintf(stderr, "[vgh-trace] %s of chunk %zu: %.1f MB appended in %.1f ms\n", which == 0 ? "chains GAF" : "alignments GAF", i,
                                (double)(f.size() - before) / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                }
            } catch (const std::exception &e) {
                writer_err[which] = e.what();
                std::lock_guard<std::mutex> lk(mu);
                abort_workers = true;  // (nothing that is still to be mapped could be written)
            }
        });
    };
    if (stream) {
        start_writer(0);
        if (opt.also_align) start_writer(1);
    }
    // the last chunk of a slot: once its GPU work is done nothing needs the context any more.  A caller that is about to leave
    // the process (leave_contexts) has it torn down right then, beside the text and file work that is left -- the driver
    // takes about 5 ms per GB of pool to take the memory back, at hipFree or at exit alike
    std::vector<size_t> last_of_slot(n_slots, plan.size());
    for (size_t i = 0; i < plan.size(); i++) last_of_slot[plan[i].slot] = i;
    std::vector<std::thread> destroyers;
    std::mutex destroyers_mu;
    std::vector<std::thread> workers;
    for (uint32_t slot = 0; slot < n_slots; slot++)
        workers.emplace_back([&, slot]() {
            try {
                for (size_t i = 0; i < plan.size(); i++)
                    if (plan[i].slot == slot) {
                        { std::lock_guard<std::mutex> lk(mu); if (abort_workers) break; }
                        ChunkHooks hooks;
                        if (stream) hooks.chains_ready = [&, i]() { { std::lock_guard<std::mutex> lk(mu); chains_done[i] = 1; } cv.notify_all(); };
                        if (opt.leave_contexts && i == last_of_slot[slot])
                            hooks.gpu_done = [&, slot]() {
                                vga_ctx *c = ctxs[slot];
                                ctxs[slot] = nullptr;
                                std::lock_guard<std::mutex> lk(destroyers_mu);
                                destroyers.emplace_back([c]() { vga_ctx_destroy(c); });
                            };
                        map_chunk(ctxs[slot], ix, inputs, plan[i].begin, plan[i].end, opt, parts[i], hooks, T_text);
                        { std::lock_guard<std::mutex> lk(mu); done[i] = 1; }
                        cv.notify_all();
                    }
            } catch (const std::exception &e) {
                errors[slot] = e.what();
                { std::lock_guard<std::mutex> lk(mu); abort_writer = true; }
                cv.notify_all();
            }
        });
    for (std::thread &t : workers) t.join();
    trace_mark("chunks mapped and aligned");
    for (std::thread &w : writer)
        if (w.joinable()) w.join();
    trace_mark("GAF files written");
    bool any_error = false;
    for (const std::string &e : errors) any_error = any_error || !e.empty();
    for (const std::string &e : writer_err) any_error = any_error || !e.empty();
    if (opt.leave_contexts && !any_error) {
        // the caller leaves the process next (vgaligner: _exit once the files are closed): whatever the tear-down threads have not
        // finished, the exit finishes
        for (std::thread &t : destroyers) t.detach();
        ctxs.clear();
    } else {
        // an error is reported by a normal return / exit path: no detached thread may still be inside the HIP runtime when its
        // static destructors run, and no context may leak
        for (std::thread &t : destroyers) t.join();
        release();
    }
    trace_mark("contexts destroyed");
    for (const std::string &e : errors)
        if (!e.empty()) throw Error(e);
    for (const std::string &e : writer_err)
        if (!e.empty()) throw Error(e);
    MapOutput out;
    out.n_reads = inputs.size();
    out.n_devices = n_slots;
    out.n_chunks = plan.size();
    std::vector<double> ms_map(n_slots, 0.0), ms_align(n_slots, 0.0);
    for (size_t i = 0; i < plan.size(); i++) {  // read order
        if (!stream) {
            append_pieces(out.chains_gaf, parts[i].chains);
            append_pieces(out.alignments_gaf, parts[i].aligns);
        }
        out.n_aligned += parts[i].n_aligned; out.n_anchors += parts[i].n_anchors; out.poa_cells += parts[i].poa_cells;
        ms_map[plan[i].slot] += parts[i].ms_map; ms_align[plan[i].slot] += parts[i].ms_align;
    }
    for (uint32_t s = 0; s < n_slots; s++) { out.ms_map = std::max(out.ms_map, ms_map[s]); out.ms_align = std::max(out.ms_align, ms_align[s]); }
    if (stream) return out;
    finish(out, ix, inputs, opt, out_prefix);
    return out;
}

}  // namespace vgh
