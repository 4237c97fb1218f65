// vgh_index.cpp -- HashGraph stand-in, GFA1 reader and Index::build (host prerequisite of the GPU path:
// it runs once per graph and defines the contents and ORDER of the k-mer position table, which fixes
// anchor multiplicity and order on the device).
//
//   HashGraph::create_edge / neighbors ... handlegraph 0.5.0 (order pinned by src/index.rs:1261-1367)
//   find_forward_sequence ................. src/utils.rs:81-146
//   reverse_complement .................... src/dna.rs:5-33
//   generate_kmers_parallel ............... src/kmer.rs:277-505
//   generate_pos_on_ref_2 ................. src/kmer.rs:752-770, 816-928
//   Index::build .......................... src/index.rs:109-281
#include "vgh.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>

namespace vgh {

// ------------------------------------------------------------------------------------------ graph
Handle HashGraph::create_handle(const std::string &seq, uint64_t id)
{
    if (id == 0) throw Error("node id 0");
    if (id >= nodes.size()) nodes.resize(std::max<uint64_t>(id + 1, nodes.size() * 2));
    if (nodes[id].present) throw Error("duplicate node id " + std::to_string(id));
    nodes[id].seq = seq;
    nodes[id].present = true;
    n_nodes++;
    min_id = std::min(min_id, id);
    max_id = std::max(max_id, id);
    return pack(id, false);
}

void HashGraph::create_edge(Handle l, Handle r)
{
    if (id_of(l) >= nodes.size() || id_of(r) >= nodes.size() || !nodes[id_of(l)].present || !nodes[id_of(r)].present)
        throw Error("edge references a missing node");
    Node &ln = nodes[id_of(l)];
    // the crate's duplicate test looks at the left node's right list only
    if (std::find(ln.right.begin(), ln.right.end(), r) != ln.right.end()) return;
    if (is_rev(l)) ln.left.push_back(flip(r)); else ln.right.push_back(r);
    if (l != flip(r)) {
        Node &rn = nodes[id_of(r)];
        if (is_rev(r)) rn.right.push_back(flip(l)); else rn.left.push_back(l);
    }
}

static char complement(char c)
{
    switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    default: return c;
    }
}

std::string HashGraph::sequence(Handle h) const
{
    const std::string &s = nodes[id_of(h)].seq;
    if (!is_rev(h)) return s;
    std::string r(s.rbegin(), s.rend());
    for (char &c : r) c = complement(c);
    return r;
}

std::vector<Handle> HashGraph::neighbors(Handle h, bool left) const
{
    const Node &nd = nodes[id_of(h)];
    const bool use_left = left != is_rev(h);
    std::vector<Handle> out(use_left ? nd.left : nd.right);
    if (is_rev(h)) for (Handle &x : out) x = flip(x);
    return out;
}

static std::vector<std::string> split_tabs(const std::string &line)
{
    std::vector<std::string> f;
    size_t s = 0;
    while (true) {
        size_t t = line.find('\t', s);
        if (t == std::string::npos) { f.push_back(line.substr(s)); break; }
        f.push_back(line.substr(s, t - s));
        s = t + 1;
    }
    return f;
}

HashGraph HashGraph::from_gfa(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw Error("cannot open " + path);
    std::vector<std::vector<std::string>> S, L, P;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.size() < 2 || line[1] != '\t') continue;
        if (line[0] == 'S') S.push_back(split_tabs(line));
        else if (line[0] == 'L') L.push_back(split_tabs(line));
        else if (line[0] == 'P') P.push_back(split_tabs(line));
    }
    HashGraph g;
    auto num = [](const std::string &s) -> uint64_t {
        if (s.empty() || s.find_first_not_of("0123456789") != std::string::npos) throw Error("non-numeric segment name " + s);
        return std::stoull(s);
    };
    for (auto &f : S) {
        if (f.size() < 3) throw Error("malformed S line");
        g.create_handle(f[2], num(f[1]));
    }
    for (auto &f : L) {
        if (f.size() < 5 || (f[2] != "+" && f[2] != "-") || (f[4] != "+" && f[4] != "-")) throw Error("malformed L line");
        g.create_edge(pack(num(f[1]), f[2] == "-"), pack(num(f[3]), f[4] == "-"));
    }
    for (auto &f : P) {
        if (f.size() < 3) throw Error("malformed P line");
        Path p;
        p.name = f[1];
        std::stringstream ss(f[2]);
        std::string step;
        while (std::getline(ss, step, ',')) {
            if (step.size() < 2) throw Error("malformed P step");
            char o = step.back();
            step.pop_back();
            p.steps.push_back(pack(num(step), o == '-'));
        }
        g.paths.push_back(std::move(p));
    }
    return g;
}

// ------------------------------------------------------------------------------------------ k-mers
namespace {

struct SeqPosT {
    uint8_t orient;
    uint64_t position;
    bool operator==(const SeqPosT &o) const { return orient == o.orient && position == o.position; }
};

struct GraphKmer {  // src/kmer.rs:47-65
    std::string seq;
    SeqPosT begin_offset, end_offset;
    Handle first_handle, last_handle;
    bool handle_orient;
    uint64_t forks;
    bool operator==(const GraphKmer &o) const
    {
        return seq == o.seq && begin_offset == o.begin_offset && end_offset == o.end_offset &&
               first_handle == o.first_handle && last_handle == o.last_handle && handle_orient == o.handle_orient &&
               forks == o.forks;
    }
};

// src/kmer.rs:347-505.  Returns false when the call is abandoned because of an 'N' (-> no k-mers).
bool kmers_from_handle(const HashGraph &g, Handle start, bool orient, uint64_t k, uint64_t edge_max, uint64_t degree_max,
                       std::vector<GraphKmer> &complete)
{
    Handle handle = start;
    if (g.neighbors(handle, false).size() > degree_max) return true;  // kmer.rs:361-372
    std::string hs = g.sequence(handle);
    std::vector<GraphKmer> pending;
    for (uint64_t i = 0; i < hs.size(); i++) {
        uint64_t end = std::min<uint64_t>(i + k, hs.size());
        GraphKmer km{hs.substr(i, end - i), {(uint8_t)is_rev(handle), i}, {(uint8_t)is_rev(handle), end}, handle, handle, orient, 0};
        if (km.seq.find('N') != std::string::npos) return false;  // kmer.rs:401-403
        if (km.seq.size() == k) {
            complete.push_back(std::move(km));
            continue;
        }
        std::vector<Handle> nb = g.neighbors(handle, false);
        if (nb.size() < degree_max || km.forks < edge_max)
            for (Handle n : nb) {
                GraphKmer c = km;
                c.last_handle = n;
                if (nb.size() > 1) c.forks += 1;
                pending.push_back(std::move(c));
            }
    }
    while (!pending.empty()) {  // Vec::pop(): depth first, last pushed first
        GraphKmer km = std::move(pending.back());
        pending.pop_back();
        handle = km.last_handle;
        hs = g.sequence(handle);
        uint64_t take = std::min<uint64_t>(k - km.seq.size(), hs.size());
        km.seq.append(hs, 0, take);  // extend_kmer, kmer.rs:80-84
        km.end_offset = {(uint8_t)is_rev(handle), take};
        km.last_handle = handle;
        if (km.seq.find('N') != std::string::npos) return false;  // kmer.rs:459-461
        if (km.seq.size() == k) {
            complete.push_back(std::move(km));
            continue;
        }
        std::vector<Handle> nb = g.neighbors(handle, false);
        for (Handle n : nb)
            if (nb.size() < degree_max || km.forks < edge_max) {
                GraphKmer c = km;
                c.last_handle = n;
                if (nb.size() > 1) c.forks += 1;
                pending.push_back(std::move(c));
            }
    }
    return true;
}

}  // namespace

// ------------------------------------------------------------------------------------------ Index
Index Index::build(const HashGraph &g, uint64_t k, uint64_t max_furcations, uint64_t max_degree)
{
    if (k == 0) throw Error("kmer_length must be > 0");
    // NodeRef is addressed by id-1 (src/index.rs:489-491): ids must be exactly 1..n
    if (g.n_nodes == 0 || g.min_id != 1 || g.max_id != g.n_nodes)
        throw Error("node ids must be the contiguous range 1..n (sort the graph first, as the reference requires)");
    Index ix;
    ix.kmer_length = k;
    ix.n_nodes = g.n_nodes;
    // src/utils.rs:81-146
    for (uint64_t id = 1; id <= g.max_id; id++) {
        const Handle h = pack(id, false);
        std::vector<Handle> l = g.neighbors(h, true), r = g.neighbors(h, false);
        ix.node_ref.push_back({(uint64_t)ix.seq_fwd.size(), (uint64_t)ix.edges.size(), (uint64_t)l.size()});
        ix.edges.insert(ix.edges.end(), l.begin(), l.end());
        ix.edges.insert(ix.edges.end(), r.begin(), r.end());
        ix.seq_fwd += g.nodes[id].seq;
    }
    ix.seq_length = ix.seq_fwd.size();
    ix.n_edges = ix.edges.size();
    ix.node_ref.push_back({ix.seq_length, ix.n_edges, 0});
    ix.seq_bv.assign(ix.seq_length + 1, 0);
    for (const NodeRef &nr : ix.node_ref) ix.seq_bv[nr.seq_idx] = 1;
    // src/dna.rs:5-33
    ix.seq_rev.resize(ix.seq_length);
    for (uint64_t i = 0; i < ix.seq_length; i++) {
        char b = ix.seq_fwd[ix.seq_length - 1 - i], c;
        switch (b) {
        case 'a': c = 't'; break; case 'c': c = 'g'; break; case 't': c = 'a'; break; case 'g': c = 'c'; break;
        case 'u': c = 'a'; break; case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'T': c = 'A'; break;
        case 'G': c = 'C'; break; case 'U': c = 'A'; break; case 'N': c = 'N'; break;
        default: throw Error(std::string("Input sequence base is not DNA: ") + b);
        }
        ix.seq_rev[i] = c;
    }
    // src/kmer.rs:277-304
    std::vector<GraphKmer> kmers;
    for (uint64_t id = 1; id <= g.max_id; id++)
        for (bool orient : {true, false}) {
            std::vector<GraphKmer> part;
            Handle h = orient ? pack(id, false) : pack(id, true);
            if (kmers_from_handle(g, h, orient, k, max_furcations, max_degree, part))
                for (auto &x : part) kmers.push_back(std::move(x));
        }
    std::stable_sort(kmers.begin(), kmers.end(), [](const GraphKmer &a, const GraphKmer &b) { return a.seq < b.seq; });
    kmers.erase(std::unique(kmers.begin(), kmers.end()), kmers.end());
    if (kmers.empty()) throw Error("the graph has no k-mer of this length");  // kmer.rs:828 unwrap()
    // src/kmer.rs:816-928
    auto seq_pos = [&](Handle h) -> uint64_t {  // kmer.rs:752-770
        uint64_t start = ix.node_ref[id_of(h) - 1].seq_idx;
        return is_rev(h) ? ix.seq_length - start - g.node_len(id_of(h)) : start;
    };
    const vga_kmerpos delim = {UINT64_MAX, UINT64_MAX, 1, 1};
    size_t i = 0;
    while (i < kmers.size()) {
        size_t j = i;
        std::vector<vga_kmerpos> group;
        while (j < kmers.size() && kmers[j].seq == kmers[i].seq) {
            const GraphKmer &km = kmers[j];
            group.push_back({seq_pos(km.first_handle) + km.begin_offset.position, seq_pos(km.last_handle) + km.end_offset.position,
                             km.begin_offset.orient, km.end_offset.orient});
            j++;
        }
        std::sort(group.begin(), group.end(), [](const vga_kmerpos &a, const vga_kmerpos &b) {  // derive(Ord): kmer.rs:732
            if (a.start_orient != b.start_orient) return a.start_orient < b.start_orient;
            if (a.start != b.start) return a.start < b.start;
            if (a.end_orient != b.end_orient) return a.end_orient < b.end_orient;
            return a.end < b.end;
        });
        ix.kmer_keys += kmers[i].seq;
        ix.kmer_starts.push_back(ix.kmer_pos_table.size());
        ix.kmer_pos_table.insert(ix.kmer_pos_table.end(), group.begin(), group.end());
        ix.kmer_pos_table.push_back(delim);
        i = j;
    }
    ix.n_kmers = ix.kmer_starts.size();
    ix.n_kmer_pos = ix.kmer_pos_table.size();
    return ix;
}

void Index::describe(vga_index_desc &d, DescScratch &s) const
{
    s.seq_idx.clear(); s.edge_idx.clear(); s.edges_to.clear();
    for (const NodeRef &nr : node_ref) { s.seq_idx.push_back(nr.seq_idx); s.edge_idx.push_back(nr.edge_idx); s.edges_to.push_back(nr.edges_to_node); }
    d.kmer_length = (uint32_t)kmer_length;
    d.seq_length = seq_length;
    d.seq_fwd = seq_fwd.data();
    d.n_nodes = n_nodes;
    d.node_seq_idx = s.seq_idx.data();
    d.node_edge_idx = s.edge_idx.data();
    d.node_edges_to = s.edges_to.data();
    d.n_edges = n_edges;
    d.edges = edges.data();
    d.n_kmers = n_kmers;
    d.kmer_keys = kmer_keys.data();
    d.kmer_starts = kmer_starts.data();
    d.n_kmer_pos = n_kmer_pos;
    d.kmer_pos_table = kmer_pos_table.data();
}

uint64_t Index::get_bv_select(uint64_t element_no) const
{  // src/index.rs:461-480
    if (element_no == 0) throw Error("Element_no should be > 0");
    return element_no <= n_nodes + 1 ? node_ref[element_no - 1].seq_idx : 0;
}

uint64_t Index::node_id_from_fwd_pos(uint64_t pos) const
{  // get_bv_rank, src/index.rs:427-439: node starts <= pos
    size_t lo = 0, hi = node_ref.size();
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (node_ref[mid].seq_idx <= pos) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------ .idx file
namespace {
template <typename T>
void put(std::ofstream &o, const T &v) { o.write((const char *)&v, sizeof(T)); }
template <typename T>
void get(std::ifstream &i, T &v) { i.read((char *)&v, sizeof(T)); if (!i) throw Error("truncated index file"); }
template <typename T>
void put_vec(std::ofstream &o, const std::vector<T> &v) { put<uint64_t>(o, v.size()); if (!v.empty()) o.write((const char *)v.data(), v.size() * sizeof(T)); }
template <typename T>
void get_vec(std::ifstream &i, std::vector<T> &v) { uint64_t n; get(i, n); v.resize(n); if (n) { i.read((char *)v.data(), n * sizeof(T)); if (!i) throw Error("truncated index file"); } }
void put_str(std::ofstream &o, const std::string &s) { put<uint64_t>(o, s.size()); o.write(s.data(), s.size()); }
void get_str(std::ifstream &i, std::string &s) { uint64_t n; get(i, n); s.resize(n); if (n) { i.read(&s[0], n); if (!i) throw Error("truncated index file"); } }
}  // namespace

void Index::store(const std::string &path) const
{
    std::ofstream o(path, std::ios::binary);
    if (!o) throw Error("Couldn't create file " + path);
    o.write("VGAIDX1\0", 8);
    put(o, kmer_length); put(o, seq_length); put(o, n_edges); put(o, n_nodes); put(o, n_kmers); put(o, n_kmer_pos);
    put_str(o, seq_fwd); put_str(o, seq_rev); put_vec(o, seq_bv); put_vec(o, edges); put_vec(o, node_ref);
    put_str(o, kmer_keys); put_vec(o, kmer_starts); put_vec(o, kmer_pos_table);
}

Index Index::load(const std::string &path)
{
    std::ifstream i(path, std::ios::binary);
    if (!i) throw Error("cannot open index " + path);
    char magic[8];
    i.read(magic, 8);
    if (!i || memcmp(magic, "VGAIDX1\0", 8) != 0) throw Error(path + " is not a VGAIDX1 index (reference .idx files are not readable yet)");
    Index ix;
    get(i, ix.kmer_length); get(i, ix.seq_length); get(i, ix.n_edges); get(i, ix.n_nodes); get(i, ix.n_kmers); get(i, ix.n_kmer_pos);
    get_str(i, ix.seq_fwd); get_str(i, ix.seq_rev); get_vec(i, ix.seq_bv); get_vec(i, ix.edges); get_vec(i, ix.node_ref);
    get_str(i, ix.kmer_keys); get_vec(i, ix.kmer_starts); get_vec(i, ix.kmer_pos_table);
    return ix;
}

}  // namespace vgh
