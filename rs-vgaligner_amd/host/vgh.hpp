// vgh.hpp -- C++ host side of the MI355X vgaligner: everything around the C ABI of libvga_hip.so.
//
// Mirrors the reference's public interface for the map -> chain -> align path (same names, argument
// meaning and error behaviour; panics become vgh::Error):
//   HashGraph / from_gfa ........ handlegraph 0.5.0 + gfa 0.8.0 as used at src/subcommands/index_main.rs:72-74
//   Index::build ................ src/index.rs:109-281 (+ src/utils.rs:81-146, src/kmer.rs:277-505, 816-928,
//                                 src/dna.rs:5-33)
//   read_seqs_from_file ......... src/io.rs:74-162
//   map_reads ................... src/map.rs:27-216   (anchors/chains/alignments come from the GPU)
//   GAFAlignment ................ src/align.rs:746-1028, 1096-1168
//   CLI ......................... src/subcommands/cli.yml, index_main.rs, map_main.rs
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vga_hip.h"

namespace vgh {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

typedef uint64_t Handle;  // (id << 1) | is_reverse
inline Handle pack(uint64_t id, bool rev) { return (id << 1) | (rev ? 1u : 0u); }
inline uint64_t id_of(Handle h) { return h >> 1; }
inline bool is_rev(Handle h) { return (h & 1) != 0; }
inline Handle flip(Handle h) { return h ^ 1; }

// ---- graph ------------------------------------------------------------------------------------
struct Node {
    std::string seq;
    std::vector<Handle> left, right;  // insertion order, relative to the forward orientation
    bool present = false;
};

struct Path {
    std::string name;
    std::vector<Handle> steps;
};

class HashGraph {
  public:
    std::vector<Node> nodes;  // indexed by id
    uint64_t min_id = UINT64_MAX, max_id = 0, n_nodes = 0;
    std::vector<Path> paths;

    Handle create_handle(const std::string &seq, uint64_t id);
    void create_edge(Handle left, Handle right);
    std::string sequence(Handle h) const;
    size_t node_len(uint64_t id) const { return nodes[id].seq.size(); }
    // handle_edges_iter(h, Direction::Left | Right)
    std::vector<Handle> neighbors(Handle h, bool left) const;
    // GFAParser::parse_file + HashGraph::from_gfa
    static HashGraph from_gfa(const std::string &path);
};

// ---- index ------------------------------------------------------------------------------------
struct NodeRef {
    uint64_t seq_idx, edge_idx, edges_to_node;
};

struct Index {
    uint64_t kmer_length = 0, seq_length = 0, n_edges = 0, n_nodes = 0, n_kmers = 0, n_kmer_pos = 0;
    std::string seq_fwd, seq_rev;
    std::vector<uint8_t> seq_bv;
    std::vector<Handle> edges;
    std::vector<NodeRef> node_ref;
    std::string kmer_keys;              // n_kmers * k, sorted
    std::vector<uint64_t> kmer_starts;  // the MPHF's values
    std::vector<vga_kmerpos> kmer_pos_table;

    static Index build(const HashGraph &g, uint64_t kmer_length, uint64_t max_furcations, uint64_t max_degree);
    // own container format ("VGAIDX1"); the reference's bincode .idx needs boomphf/bv/ahash internals
    void store(const std::string &path) const;
    static Index load(const std::string &path);
    // fills a vga_index_desc whose pointers stay valid while *this and `scratch` live
    struct DescScratch {
        std::vector<uint64_t> seq_idx, edge_idx, edges_to;
    };
    void describe(vga_index_desc &d, DescScratch &scratch) const;

    uint64_t get_bv_select(uint64_t element_no) const;
    uint64_t node_id_from_fwd_pos(uint64_t pos) const;
};

// ---- reads ------------------------------------------------------------------------------------
struct QuerySequence {
    std::string name, seq;
};
std::vector<QuerySequence> read_seqs_from_file(const std::string &filename);

// ---- GAF --------------------------------------------------------------------------------------
std::string gaf_placeholder(const QuerySequence &q);
std::string gaf_from_chain(const Index &ix, const QuerySequence &q, const vga_map_result *m, uint64_t read, uint64_t chain);
void gaf_from_chain_text(std::string &out, const Index &ix, const QuerySequence &q, const vga_map_result *m, uint64_t read, uint64_t chain,
                         const char *path, uint64_t path_len);
std::string gaf_from_alignment(const QuerySequence &q, const vga_align_result *a, uint64_t read);
void gaf_from_alignment(std::string &out, const QuerySequence &q, const vga_align_result *a, uint64_t read);  // (appends)
// ValidationRecord::from_graph_and_alignment + to_string (src/validate.rs:36-102) from one alignments-GAF line
std::string validation_record(const Index &ix, const std::string &gaf_line, const std::vector<QuerySequence> &reads);

// ---- map_reads --------------------------------------------------------------------------------
struct MapOptions {
    uint64_t bandwidth = 50;             // map_main.rs:103
    uint64_t max_gap = 1000;             // -g
    uint64_t chain_min_n_anchors = 3;    // -a
    double secondary_chain_threshold = 0.5;
    double max_mismatch_rate = 0.1;      // -r (parsed, unused by the reference's live code)
    double max_mapq = 60.0;
    bool write_console = false;          // -C
    bool also_align = false;             // -D
    uint64_t align_best_n = 1;           // -b
    std::string poa_aligner = "abpoa";   // -p
    int poa_remain_rule = -1;            // --poa-remain longest|first-edge: vga_poa_params.remain_rule (-1: the library's default)
    int device = 0;                      // used when `devices` is empty and map_reads is handed a context
    // Multi-GPU / streaming (not in the reference, which is single-threaded): one context and one host thread per entry of
    // `devices` (an id may repeat: two contexts on one GPU), each mapping a contiguous slice of the reads balanced by bases, in
    // chunks of at most `chunk_reads` reads (0 = the whole slice at once) so that host and device memory stay bounded.
    std::vector<int> devices;            // empty: `device` alone (--devices all: every visible GPU)
    bool all_devices = false;
    uint64_t chunk_reads = 32768;
    // the caller is about to leave the process with _exit: the contexts are not torn down (freeing tens of GB of HBM takes the
    // driver hundreds of milliseconds that a command line tool would only spend waiting), tear-down threads may still be running
    // when map_reads_multi returns.  ONLY for callers that leave the process right after a successful return (the contexts leak
    // otherwise); when anything fails, map_reads_multi joins those threads and destroys every context before it throws.
    bool leave_contexts = false;
    // false: map_reads_multi writes the GAF files chunk by chunk and returns no text (the CLI without -C / -v); true: the
    // whole GAF text comes back in MapOutput
    bool keep_text = true;
    bool also_validate = false;          // -v: write validation records (src/validate.rs:18-102, map.rs:186-208)
    std::string validation_path;         // -P
};

// VGA_TRACE=1: wall-clock marks of the driver's phases on stderr (since the first call)
void trace_mark(const char *what);

struct MapOutput {
    std::string chains_gaf, alignments_gaf, validation;
    uint64_t n_reads = 0, n_aligned = 0, n_anchors = 0, poa_cells = 0;
    double ms_map = 0, ms_align = 0;     // summed over chunks (per device: the maximum over devices)
    uint64_t n_chunks = 0, n_devices = 0;
};

// One [begin, end) range of the read list, the device slot (index into MapOptions::devices) that maps it.
struct Shard {
    uint64_t begin, end;
    uint32_t slot;
};
// Contiguous slices balanced by bases, one per device slot (the rule of sharding.py::split_by_bases), each cut into chunks of
// at most chunk_reads reads.  Shards come in read order; concatenating their outputs reproduces the single-batch output.
std::vector<Shard> plan_shards(const std::vector<uint64_t> &read_lengths, uint32_t n_slots, uint64_t chunk_reads);

// diagnostics: the text half of the product path replayed (vgh_map.cpp: textpath_replay)
struct TextReplay { uint64_t bytes = 0, chains_bytes = 0; double seconds = 0, text_seconds = 0, write_seconds = 0; unsigned threads = 0; };
TextReplay textpath_replay(vga_ctx *ctx, const Index &ix, const std::vector<QuerySequence> &inputs, const MapOptions &opt, const std::string &out_prefix,
                           unsigned repeat, unsigned n_threads);
// ctx must already hold the uploaded index: maps everything on that one context, in chunks of opt.chunk_reads.
// out_prefix == "" writes no files.
MapOutput map_reads(vga_ctx *ctx, const Index &ix, const std::vector<QuerySequence> &inputs, const MapOptions &opt,
                    const std::string &out_prefix);
// Creates one context per entry of opt.devices (all visible GPUs when empty), uploads the index to each and maps the slices
// concurrently; GAF order = read order (src/map.rs:123-133, 174-184).
MapOutput map_reads_multi(const Index &ix, const std::vector<QuerySequence> &inputs, const MapOptions &opt,
                          const std::string &out_prefix);
// Begins to create the contexts map_reads_multi(.., opt, ..) will use, on a thread of its own: starting HIP takes 0.1-0.3 s,
// which a caller can spend reading its index and reads.  The next map_reads_multi call picks them up (and reports any error).
void prewarm_contexts(const MapOptions &opt);

}  // namespace vgh
