// vgaligner_main.cpp -- `vgaligner index` / `vgaligner map`: same flags, defaults and output naming as the
// reference CLI (src/subcommands/cli.yml:1-175, index_main.rs:11-87, map_main.rs:12-118).
#include "vgh.hpp"

#include <cstdio>
#include <malloc.h>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unistd.h>

using namespace vgh;

namespace {

// short flag, long flag (src/subcommands/cli.yml), takes a value, key (the argument's name in cli.yml)
struct Flag { const char *shortf, *longf; bool takes_value; const char *key; };

const Flag INDEX_FLAGS[] = {{"-i", "--input", true, "input"}, {"-o", "--output", true, "out-prefix"}, {"-k", "--kmer-length", true, "kmer-length"},
                            {"-e", "--max-furcations", true, "max-furcations"}, {"-m", "--max-degree", true, "max-degree"},
                            {"-r", "--sampling-rate", true, "sampling-rate"}, {"-g", "--generate-mappings", false, "generate-mappings"},
                            {"-p", "--mappings-path", true, "mappings-path"}, {"-t", "--threads", true, "n-threads"},
                            // the argument names, accepted as long flags too
                            {"", "--out-prefix", true, "out-prefix"}, {"", "--n-threads", true, "n-threads"}};
const Flag MAP_FLAGS[] = {{"-i", "--index", true, "index"}, {"-f", "--input-file", true, "input-file"}, {"-o", "--out", true, "out-prefix"},
                          {"-g", "--max-gap-length", true, "max-gap-length"}, {"-r", "--max-mismatch-rate", true, "max-mismatch-rate"},
                          {"-c", "--chain-overlap-max", true, "chain-overlap-max"}, {"-a", "--chain-min-anchors", true, "chain-min-anchors"},
                          {"-b", "--align-best-n", true, "align-best-n"}, {"-C", "--write-console", false, "write-console"},
                          {"-D", "--also-align", false, "also-align"}, {"-t", "--threads", true, "n-threads"},
                          {"-v", "--also-validate", false, "also-validate"}, {"-G", "--graph", true, "graph"},
                          {"-P", "--validation-path", true, "validation-path"}, {"-p", "--poa-aligner", true, "poa-aligner"},
                          {"", "--out-prefix", true, "out-prefix"}, {"", "--n-threads", true, "n-threads"},
                          // not in the reference: --device one GPU, --devices a list (one context + host thread each; an id may
                          // repeat; --devices all: every visible GPU; default: --device 0), --chunk-reads reads per batch (bounded memory; 0 = one batch)
                          {"-d", "--device", true, "device"}, {"", "--devices", true, "devices"}, {"", "--chunk-reads", true, "chunk-reads"},
                          {"", "--poa-remain", true, "poa-remain"}};

template <size_t N>
std::map<std::string, std::string> parse(const Flag (&flags)[N], int argc, char **argv, int first)
{
    std::map<std::string, std::string> m;
    for (int i = first; i < argc; i++) {
        const Flag *f = nullptr;
        for (const Flag &c : flags)
            if ((c.shortf[0] && !strcmp(argv[i], c.shortf)) || !strcmp(argv[i], c.longf)) { f = &c; break; }
        if (!f) throw Error(std::string("unknown argument ") + argv[i]);
        if (f->takes_value) {
            if (i + 1 >= argc) throw Error(std::string("missing value for ") + argv[i]);
            m[f->key] = argv[++i];
        } else m[f->key] = "1";
    }
    return m;
}

std::string need(const std::map<std::string, std::string> &m, const char *key)
{
    auto it = m.find(key);
    if (it == m.end()) throw Error(std::string("the required argument --") + key + " was not provided");
    return it->second;
}
std::string opt(const std::map<std::string, std::string> &m, const char *key, const std::string &dflt)
{
    auto it = m.find(key);
    return it == m.end() ? dflt : it->second;
}

int index_main(int argc, char **argv)
{
    auto m = parse(INDEX_FLAGS, argc, argv, 2);
    std::string in = need(m, "input");
    std::string prefix = opt(m, "out-prefix", in.size() > 4 ? in.substr(0, in.size() - 4) : in);  // index_main.rs:16-18
    uint64_t k = std::stoull(need(m, "kmer-length"));
    uint64_t furc = std::stoull(opt(m, "max-furcations", "100")), deg = std::stoull(opt(m, "max-degree", "100"));
    if (m.count("sampling-rate")) throw Error("--sampling-rate depends on ahash's hash values and is not supported");
    if (m.count("generate-mappings")) fprintf(stderr, "[vgaligner] --generate-mappings (debug JSON) is not produced by this build\n");
    Index ix = Index::build(HashGraph::from_gfa(in), k, furc, deg);
    fprintf(stderr, "[vgaligner] Index with k=%llu built: %llu different kmers, %llu positions\n", (unsigned long long)k,
            (unsigned long long)ix.n_kmers, (unsigned long long)ix.n_kmer_pos);
    bool exact = prefix.size() >= 4 && prefix.compare(prefix.size() - 4, 4, ".idx") == 0;  // index.rs:267-278
    ix.store(exact ? prefix : prefix + ".idx");
    return 0;
}

int map_main(int argc, char **argv)
{
    auto m = parse(MAP_FLAGS, argc, argv, 2);
    std::string idx = need(m, "index"), in = need(m, "input-file");
    std::string dflt_prefix;  // map_main.rs:19-28 (strips 3 chars for ...fa / ...fasta, else 4)
    auto ends = [&](const char *s) { size_t n = strlen(s); return in.size() >= n && in.compare(in.size() - n, n, s) == 0; };
    dflt_prefix = (ends("fa") || ends("fasta")) ? in.substr(0, in.size() - 3) : in.substr(0, in.size() > 4 ? in.size() - 4 : 0);
    MapOptions o;
    std::string prefix = opt(m, "out-prefix", dflt_prefix);
    o.max_gap = std::stoull(opt(m, "max-gap-length", "1000"));
    o.max_mismatch_rate = std::stod(opt(m, "max-mismatch-rate", "0.1"));
    o.chain_min_n_anchors = std::stoull(opt(m, "chain-min-anchors", "3"));
    o.align_best_n = std::stoull(opt(m, "align-best-n", "1"));
    o.write_console = m.count("write-console") > 0;
    o.also_align = m.count("also-align") > 0;
    o.poa_aligner = need(m, "poa-aligner");  // cli.yml:169-175: required
    o.device = std::stoi(opt(m, "device", "0"));
    if (m.count("poa-remain")) {  // which path the adaptive band's `remain` follows (include/vga_hip.h: VGA_REMAIN_*)
        const std::string r = m["poa-remain"];
        if (r == "longest") o.poa_remain_rule = VGA_REMAIN_LONGEST_PATH;
        else if (r == "first-edge") o.poa_remain_rule = VGA_REMAIN_FIRST_OUT_EDGE;
        else throw Error("--poa-remain takes longest or first-edge");
    }
    o.also_validate = m.count("also-validate") > 0;
    if (o.also_validate) {
        if (!o.also_align) fprintf(stderr, "[vgaligner] --also-validate has no effect without --also-align (map.rs:150-186)\n");
        o.validation_path = need(m, "validation-path");  // map.rs:201 unwraps it
    }
    if (o.also_align && !m.count("graph")) throw Error("--also-align needs --graph (the reference unwraps it, map.rs:157)");
    bool exact = idx.size() >= 4 && idx.compare(idx.size() - 4, 4, ".idx") == 0;
    trace_mark("start");
    if (m.count("devices") && m["devices"] == "all") o.all_devices = true;
    else if (m.count("devices")) {
        const std::string l = m["devices"];
        size_t p0 = 0;
        while (p0 <= l.size()) {
            const size_t p1 = l.find(',', p0);
            const std::string tok = l.substr(p0, p1 == std::string::npos ? std::string::npos : p1 - p0);
            if (!tok.empty()) o.devices.push_back(std::stoi(tok));
            if (p1 == std::string::npos) break;
            p0 = p1 + 1;
        }
        if (o.devices.empty()) throw Error("--devices needs a comma-separated list of GPU ids");
    }
    prewarm_contexts(o);  // (HIP starts beside the reading of the index and the reads)
    Index ix = Index::load(exact ? idx : idx + ".idx");
    trace_mark("index loaded");
    std::vector<QuerySequence> reads = read_seqs_from_file(in);
    trace_mark("reads parsed");
    fprintf(stderr, "[vgaligner] Found %zu reads!\n", reads.size());
    o.leave_contexts = !getenv("VGA_NO_FAST_EXIT");
    o.chunk_reads = std::stoull(opt(m, "chunk-reads", "32768"));
    o.keep_text = o.write_console || o.also_validate;
    MapOutput out = map_reads_multi(ix, reads, o, prefix);
    fprintf(stderr, "[vgaligner] %llu GPU context(s), %llu batch(es)\n", (unsigned long long)out.n_devices, (unsigned long long)out.n_chunks);
    fprintf(stderr, "[vgaligner] Chaining took: %.0f ms\n", out.ms_map);
    if (o.also_align) fprintf(stderr, "[vgaligner] Alignment took: %.0f ms; Found %llu alignments!\n", out.ms_align, (unsigned long long)out.n_reads);
    if (o.write_console) fputs(o.also_align ? out.alignments_gaf.c_str() : out.chains_gaf.c_str(), stdout);
    trace_mark("done");
    if (getenv("VGA_TRACE"))  // (what the exit has to give back: resident host memory)
        if (FILE *f = fopen("/proc/self/status", "r")) {
            char ln[256];
            while (fgets(ln, sizeof ln, f))
                if (!strncmp(ln, "VmRSS", 5) || !strncmp(ln, "VmHWM", 5) || !strncmp(ln, "RssAnon", 7) || !strncmp(ln, "RssShmem", 8) || !strncmp(ln, "RssFile", 7))
                    fprintf(stderr, "[vgh-trace] %s", ln);
            fclose(f);
        }
    if (o.leave_contexts) {  // the GAF files are closed; what is left is HBM the driver reclaims by itself
        fflush(stdout);
        fflush(stderr);
        _exit(0);
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    // This process only maps reads, and its big blocks (anchor arrays, GAF text: 0.1-1.8 GB each, one set per chunk) come and go on
    // worker threads.  glibc gives every thread an arena of 64 MB heaps, so such blocks are mmap'ed and munmap'ed whatever
    // M_MMAP_THRESHOLD says: 1.8 GB of page faults per array set, and a munmap that holds the address-space lock for 0.1-0.2 s
    // while every other thread's page faults and mmaps wait (the DP launches of the next sub-batch among them).  One arena (the
    // brk heap), nothing mmap'ed below 2 GB, nothing trimmed: blocks are recycled from chunk to chunk.  Config 5, 100 000 reads end
    // to end: 6.4 -> 4.3 s (vga_map_batch of a 25 000-read chunk 316 -> 115 ms).  Before the first thread exists.
    if (!getenv("VGA_NO_TUNE_MALLOC")) {
        mallopt(M_ARENA_MAX, 1);
        mallopt(M_MMAP_THRESHOLD, INT32_MAX);
        mallopt(M_TRIM_THRESHOLD, -1);  // (never)
        mallopt(M_TOP_PAD, 256 << 20);
    }
    try {
        if (argc >= 2 && !strcmp(argv[1], "index")) return index_main(argc, argv);
        if (argc >= 2 && !strcmp(argv[1], "map")) return map_main(argc, argv);
        fprintf(stderr, "vgaligner 0.7 (MI355X build)\nUSAGE:\n  vgaligner index -i <graph.gfa> -k <K> [-o prefix] [-e 100] [-m 100]\n"
                        "  vgaligner map -i <index> -f <reads.fa|fq> -p abpoa [-o prefix] [-g 1000] [-a 3] [-b 1] [-D -G <graph.gfa>] [-C]\n"
                        "                [--device 0 | --devices 0,1,... | --devices all] [--chunk-reads 32768] [--poa-remain longest|first-edge]\n");
        return 2;
    } catch (const std::exception &e) {
        fprintf(stderr, "vgaligner: %s\n", e.what());
        return 101;  // a Rust panic exits with 101
    }
}
