// juliet — command-line front end over libjuliet_hip.so: aligned CCS BAM in, JSON (and a plain HTML
// rendering of it) out.  Keeps the documented surface of the reference tool:
//   juliet [--config/-c CFG] [--mode-phasing/-p] [--region/-r B-E] [--min-perc X] [--max-perc X] [--drm-only]
//          in.align.bam out.{json,html} [out2.{json,html}]
// (doc/JULIET.md:62-66, 121, 160-163, 195, 270-271, 342-344, 352-354, 370).  Everything the reference text
// leaves open is an explicit flag with the docs/SPEC.md default.  All compute happens on the GPU through
// the C ABI; without a gfx950 device the tool exits with status 3.
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <future>
#include <cstdlib>
#include <ctime>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <memory>

#include "config.hpp"
#include "decode.hpp"
#include "format.hpp"
#include "fuse.hpp"
#include "html.hpp"
#include "msa_builder.hpp"

using namespace jlhost;

namespace {

const char *kVersion = "0.1.0 (minorseq_amd, MI355X)";

struct Options {
    std::string bam, config;
    std::vector<std::string> outputs;
    bool phasing = false, drm_only = false;
    bool have_region = false;
    uint32_t region_b = 0, region_e = 0;
    double min_perc = -1.0, max_perc = -1.0;
    double alpha = 0.01, n_tests = 0.0;
    std::string chemistry = "auto";
    double match = -1.0, substitution = -1.0;
    int expected_round = 0;
    int fisher_tail = 0;             // --fisher-tail greater|two-sided (doc/JULIET.md:38-42 leaves the sidedness open)
    uint32_t min_reads = 10, min_qv = 0;
    double min_rq = 0.0;
    int device = 0;
    uint32_t windows = 1;            // column windows the reference is cut into (doc/JULIET.md:261-264: the split never shows)
    std::vector<int> devices;        // --devices a,b,...: one rank (thread) per device, consecutive windows each
    std::string dump_msa, dump_config, consensus;
    bool fuse_only = false;        // invoked as `fuse in.bam out.fasta` (doc/FUSE.md:26-31): the consensus and nothing else
    double ins_min_frac = 0.5;     // an insertion enters the consensus when more than this share of the covering reads carries it
    uint32_t ins_min_distance = 10;  // ... and the previous included insertion lies at least this many columns back (UNPINNED)
    bool timing = false;
    std::string exchange;            // --exchange rccl|inproc: how the rank threads exchange (default: rccl, inproc when a device repeats)
};

[[noreturn]] void usage(int code)
{
    std::cerr <<
        "juliet " << kVersion << "\n"
        "usage: juliet [options] in.align.bam out.json|out.html [second output]\n"
        "  -c, --config <HIV|ABL1|file.json>   target config (doc/JULIET.md:109-180)\n"
        "  -p, --mode-phasing                  cluster reads into haplotypes (doc/JULIET.md:192-211)\n"
        "  -r, --region <begin-end>            1-based [begin,end) window of the config to call\n"
        "      --min-perc <x> / --max-perc <x> only calls above / below x percent\n"
        "  -k, --drm-only                      only known DRM positions of the config\n"
        "  parameters the reference text leaves open (docs/SPEC.md):\n"
        "      --alpha 0.01  --n-tests <auto>  --chemistry auto|sequel|permissive\n"
        "      --match-rate <r> --substitution-rate <r> --expected-round ceil|floor|nearest\n"
        "      --fisher-tail greater|two-sided  sidedness of Fisher's exact test (default greater: an excess of observed codons)\n"
        "      --min-reads 10  --min-qv 0  --min-rq 0  --device 0\n"
        "      --windows K [--devices a,b,...] cut the reference into K column windows (2-column overlap, global Bonferroni\n"
        "                                      factor), consecutive windows per device; phasing runs across the windows with\n"
        "                                      the reads sharded over the devices.  The output is that of one window.\n"
        "      --exchange rccl|inproc          how the rank threads exchange: RCCL (default), or device copies between the ranks'\n"
        "                                      buffers (peer copies over xGMI; the default when a device is named twice,\n"
        "                                      which RCCL refuses)\n"
        "      --consensus <out.fasta>         also write the window's consensus as `fuse` would (doc/FUSE.md:17-20):\n"
        "                                      majority base, major deletions removed, in-frame majority insertions kept\n"
        "      --ins-min-frac 0.5  --ins-min-distance 10   when an insertion enters the consensus\n"
        "      --timing                        wall time of each stage on stderr\n"
        "  diagnostics (no GPU needed): --dump-msa <file>  --dump-config <file>\n";
    std::exit(code);
}

Options parse(int argc, char **argv)
{
    Options o;
    auto need = [&](int &i) -> std::string {
        if (i + 1 >= argc) { std::cerr << "juliet: " << argv[i] << " needs a value\n"; usage(1); }
        return argv[++i];
    };
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") usage(0);
        else if (a == "--version") { std::cout << kVersion << "\n"; std::exit(0); }
        else if (a == "-c" || a == "--config") o.config = need(i);
        else if (a == "-p" || a == "--mode-phasing") o.phasing = true;
        else if (a == "-k" || a == "--drm-only") o.drm_only = true;
        else if (a == "-r" || a == "--region") {
            const std::string v = need(i);
            const size_t d = v.find('-');
            if (d == std::string::npos) { std::cerr << "juliet: --region wants begin-end\n"; usage(1); }
            o.region_b = (uint32_t)std::stoul(v.substr(0, d));
            o.region_e = (uint32_t)std::stoul(v.substr(d + 1));
            o.have_region = true;
        }
        else if (a == "--min-perc") o.min_perc = std::stod(need(i));
        else if (a == "--max-perc") o.max_perc = std::stod(need(i));
        else if (a == "--alpha") o.alpha = std::stod(need(i));
        else if (a == "--n-tests") o.n_tests = std::stod(need(i));
        else if (a == "--chemistry") o.chemistry = need(i);
        else if (a == "--match-rate") o.match = std::stod(need(i));
        else if (a == "--substitution-rate") o.substitution = std::stod(need(i));
        else if (a == "--expected-round") {
            const std::string v = need(i);
            o.expected_round = v == "floor" ? 1 : v == "nearest" ? 2 : 0;
        }
        else if (a == "--fisher-tail") {
            const std::string v = need(i);
            if (v == "greater") o.fisher_tail = 0;
            else if (v == "two-sided") o.fisher_tail = 1;
            else { std::cerr << "juliet: --fisher-tail takes greater or two-sided\n"; usage(1); }
        }
        else if (a == "--min-reads") o.min_reads = (uint32_t)std::stoul(need(i));
        else if (a == "--min-qv") o.min_qv = (uint32_t)std::stoul(need(i));
        else if (a == "--min-rq") o.min_rq = std::stod(need(i));
        else if (a == "--device") o.device = std::stoi(need(i));
        else if (a == "--windows") o.windows = (uint32_t)std::stoul(need(i));
        else if (a == "--exchange") {
            o.exchange = need(i);
            if (o.exchange != "rccl" && o.exchange != "inproc") { std::cerr << "juliet: --exchange wants rccl or inproc\n"; usage(1); }
        }
        else if (a == "--devices") {
            const std::string v = need(i);
            size_t b = 0;
            while (b <= v.size()) {
                const size_t e = std::min(v.find(',', b), v.size());
                if (e > b) o.devices.push_back(std::stoi(v.substr(b, e - b)));
                b = e + 1;
            }
        }
        else if (a == "--consensus") o.consensus = need(i);
        else if (a == "--ins-min-frac") o.ins_min_frac = std::stod(need(i));
        else if (a == "--ins-min-distance") o.ins_min_distance = (uint32_t)std::stoul(need(i));
        else if (a == "--dump-msa") o.dump_msa = need(i);
        else if (a == "--dump-config") o.dump_config = need(i);
        else if (a == "--timing") o.timing = true;
        else if (!a.empty() && a[0] == '-') { std::cerr << "juliet: unknown option " << a << "\n"; usage(1); }
        else pos.push_back(a);
    }
    if (!o.dump_config.empty() && pos.empty()) { if (o.devices.empty()) o.devices.push_back(o.device); return o; }
    {   // `fuse in.bam out.fasta` (doc/FUSE.md:26-31): the same front end, asked for the consensus only
        const std::string prog = argv[0];
        const size_t slash = prog.find_last_of('/');
        if ((slash == std::string::npos ? prog : prog.substr(slash + 1)) == "fuse") {
            if (pos.size() != 2) { std::cerr << "fuse: usage: fuse in.align.bam out.fasta\n"; std::exit(1); }
            o.bam = pos[0];
            o.consensus = pos[1];
            o.fuse_only = true;
            o.devices.assign(1, o.device);
            o.windows = 1;
            return o;
        }
    }
    if (o.devices.empty()) o.devices.push_back(o.device);
    if (o.windows == 0 || o.windows > 32u * o.devices.size()) { std::cerr << "juliet: --windows wants 1 .. 32 per device\n"; usage(1); }
    if (o.windows < o.devices.size()) { std::cerr << "juliet: fewer windows than devices\n"; usage(1); }
    if ((o.windows > 1 || o.devices.size() > 1) && !o.consensus.empty()) {
        std::cerr << "juliet: --consensus works on one window (drop --windows / --devices)\n";
        usage(1);
    }
    if (pos.size() < 2 && o.dump_msa.empty()) { std::cerr << "juliet: need an input BAM and at least one output\n"; usage(1); }
    if (pos.empty()) usage(1);
    o.bam = pos[0];
    o.outputs.assign(pos.begin() + 1, pos.end());
    for (const std::string &out : o.outputs) {
        const bool ok = (out.size() > 5 && out.substr(out.size() - 5) == ".json") || (out.size() > 5 && out.substr(out.size() - 5) == ".html");
        if (!ok) { std::cerr << "juliet: output '" << out << "' must end in .json or .html (doc/JULIET.md:61-66)\n"; usage(1); }
    }
    return o;
}

std::string iso_now()
{
    using namespace std::chrono;
    const auto now = system_clock::now();
    const std::time_t t = system_clock::to_time_t(now);
    const int ms = (int)(duration_cast<milliseconds>(now.time_since_epoch()).count() % 1000);
    std::tm tm;
    gmtime_r(&t, &tm);
    char buf[80];
    snprintf(buf, sizeof buf, "%04d-%02d-%02dT%02d:%02d:%02d.%03dZ", tm.tm_year + 1900, tm.tm_mon + 1, tm.tm_mday,
             tm.tm_hour, tm.tm_min, tm.tm_sec, ms);
    return buf;
}

std::string haplotype_name(uint32_t h)  // [A-Z]{1}[a-z]?  (doc/JULIET.md:198)
{
    if (h < 26) return std::string(1, (char)('A' + h));
    h -= 26;
    return std::string{(char)('A' + h / 26), (char)('a' + h % 26)};
}

void die_jl(jl_ctx *ctx, const char *what)
{
    std::cerr << "juliet: " << what << ": " << jl_last_error(ctx) << "\n";
    std::exit(3);
}

// A few threads that copy: the uploader's gather is 0.45 GB into pages nobody has touched yet (1.35 GB of a 100k-read rich-QV BAM's
// records become 0.45 GB of arrays).  As range inserts on the uploader thread it was 120-140 ms — a vector with an allocator of its own
// inserts element by element — more than the whole decode takes since the quality tracks are folded sixteen bases an instruction;
// as memcpy in 1 MB pieces by these threads and the uploader 18-34 ms.  add() splits a copy; wait() helps until every piece is done.
class CopyCrew {
public:
    explicit CopyCrew(unsigned n)
    {
        for (unsigned i = 0; i < n; ++i) th_.emplace_back([this] { work(false); });
    }
    ~CopyCrew()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void add(void *dst, const void *src, size_t bytes)
    {
        const size_t piece = (size_t)1 << 20;
        {
            std::lock_guard<std::mutex> lk(m_);
            for (size_t o = 0; o < bytes; o += piece) {
                q_.push_back({(uint8_t *)dst + o, (const uint8_t *)src + o, std::min(piece, bytes - o)});
                ++pending_;
            }
        }
        cv_.notify_all();
    }
    void wait() { work(true); }

private:
    struct Job { uint8_t *dst; const uint8_t *src; size_t n; };
    void work(bool until_idle)
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (!q_.empty()) {
                const Job j = q_.front();
                q_.pop_front();
                lk.unlock();
                memcpy(j.dst, j.src, j.n);
                lk.lock();
                if (--pending_ == 0) done_.notify_all();
                continue;
            }
            if (until_idle) {
                done_.wait(lk, [this] { return pending_ == 0; });
                return;
            }
            if (stop_) return;
            cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::deque<Job> q_;
    size_t pending_ = 0;
    bool stop_ = false;
    std::vector<std::thread> th_;
};

// Hands decoded records to the device chunk by chunk while the parser works on the next chunk: the upload (0.03 s for
// 100k reads) hides under the decode whenever the GPU context is up before the file ends; chunks that arrive earlier
// simply wait.  One consumer thread: chunks stay in file order.
class RecordUploader {
public:
    // one records context per device: every chunk goes to each of them (one rank per device reads its windows out of it)
    RecordUploader(std::vector<std::shared_future<std::pair<int, jl_ctx *>>> ctx_up, uint64_t file_bytes, bool want_qual)
        : ctx_up_(std::move(ctx_up)), file_bytes_(file_bytes), want_qual_(want_qual), th_([this] { run(); })
    {
    }
    ~RecordUploader() { finish(); }
    RecordUploader(const RecordUploader &) = delete;
    RecordUploader &operator=(const RecordUploader &) = delete;

    // parser side: trade the full chunk for an empty one
    void give(RecordArrays &chunk)
    {
        RecordArrays fresh;
        const size_t want_seq = chunk.seq4.size() + chunk.seq4.size() / 4, want_cig = chunk.cigar.size() + chunk.cigar.size() / 4,
                     want_qual = chunk.qual.size() + chunk.qual.size() / 4, want_reads = chunk.pos.size() + 1;
        {
            std::lock_guard<std::mutex> lk(m_);
            if (!pool_.empty()) {
                fresh = std::move(pool_.back());
                pool_.pop_back();
            }
            q_.push_back(std::move(chunk));
        }
        cv_.notify_one();
        chunk = std::move(fresh);
        chunk.clear();
        // a new chunk starts at the size of the one before it instead of growing by doubling
        chunk.seq4.reserve(want_seq);
        chunk.cigar.reserve(want_cig);
        chunk.qual.reserve(want_qual);
        chunk.pos.reserve(want_reads);
        chunk.cig_off.reserve(want_reads);
        chunk.seq_off.reserve(want_reads);
        if (want_qual) chunk.qual_off.reserve(want_reads);
        chunk.names.reserve(want_reads);
    }
    // no more chunks: waits for the uploads; the records are on the device when this returns JL_OK
    int finish()
    {
        if (th_.joinable()) {
            {
                std::lock_guard<std::mutex> lk(m_);
                done_ = true;
            }
            cv_.notify_one();
            th_.join();
        }
        return rc_;
    }
    jl_ctx *ctx(size_t k = 0) const { return k < ctxs_.size() ? ctxs_[k] : nullptr; }
    jl_ctx *failed() const { return failed_; }
    std::vector<std::string> names;
    uint64_t n_reads = 0;
    double ms_begin = 0, ms_append = 0, ms_append_max = 0, ms_names = 0, ms_gather = 0;   // --timing
    unsigned n_appends = 0;

private:
    static double ms_since(std::chrono::steady_clock::time_point t)
    {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    }
    // Chunk after chunk (offsets relative to the chunk) behind each other in `big_`: what the decoder hands over while the
    // GPU runtime is still starting goes to the device as a few LARGE copies once the contexts exist — a pageable copy pins its
    // source range first, and sixty-one chunks of a few MB, each a buffer the runtime has not seen, cost 16-25 ms where the same
    // 200 MB out of five arrays cost 5-6 (tools_tuning/h2d_threads.cpp: 21 against 36 GB/s on first touch).  The gathering
    // itself runs beside the decode, on this thread.
    // (the large arrays — bases, qualities, cigar words — by the copy crew: the chunk and `big_` must stay as they are until crew_.wait())
    template <typename V, typename W> void gather_array(V &dst, const W &src)
    {
        const size_t at = dst.size();
        if (at + src.size() > dst.capacity()) crew_.wait();      // (it moves: nobody may be copying into the old place)
        dst.resize(at + src.size());
        crew_.add(dst.data() + at, src.data(), src.size() * sizeof(src[0]));
    }
    void gather(const RecordArrays &c)
    {
        const size_t n = c.pos.size();
        const uint64_t cb = big_.cigar.size(), sb = big_.seq4.size(), qb = big_.qual.size();
        big_.pos.insert(big_.pos.end(), c.pos.begin(), c.pos.end());
        gather_array(big_.cigar, c.cigar);
        gather_array(big_.seq4, c.seq4);
        for (size_t i = 1; i <= n; ++i) {
            big_.cig_off.push_back(cb + c.cig_off[i] - c.cig_off[0]);
            big_.seq_off.push_back(sb + c.seq_off[i] - c.seq_off[0]);
        }
        if (want_qual_) {
            gather_array(big_.qual, c.qual);
            for (size_t i = 1; i <= n; ++i) big_.qual_off.push_back(qb + c.qual_off[i] - c.qual_off[0]);
        }
    }
    size_t gathered_bytes() const { return big_.seq4.size() + big_.qual.size() + 4 * big_.cigar.size(); }
    bool contexts_ready() const
    {
        for (const auto &f : ctx_up_)
            if (f.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return false;
        return true;
    }
    void open()   // waits for the contexts
    {
        for (auto &f : ctx_up_) {
            const auto up = f.get();
            ctxs_.push_back(up.second);
            if (up.first != JL_OK && rc_ == JL_OK) rc_ = up.first;
        }
        for (jl_ctx *c : ctxs_) {
            if (rc_ != JL_OK) break;
            // CCS BAMs inflate 5-10x; the packed bases are about a third of that, qualities twice the bases, a cigar word per
            // dozen bases when every filtered base is an X of its own (the arrays grow if not — each growth is an allocation, a
            // device copy and a free behind a synchronisation, so the hints err on the large side: memory is not the constraint)
            const uint64_t seq_hint = std::min<uint64_t>(file_bytes_ * 7 / 2, (uint64_t)4 << 30);
            const auto t = std::chrono::steady_clock::now();
            rc_ = jl_records_begin(c, seq_hint / 512 + 1024, seq_hint / 8 + 1024, seq_hint, want_qual_ ? seq_hint * 2 : 0);
            if (rc_ != JL_OK) failed_ = c;
            ms_begin += ms_since(t);
        }
        ready_ = true;
    }
    void flush()
    {
        if (big_.pos.empty()) return;
        const auto t = std::chrono::steady_clock::now();
        for (jl_ctx *dst : ctxs_) {
            if (rc_ != JL_OK) break;
            rc_ = jl_records_append(dst, big_.pos.size(), big_.pos.data(), big_.cigar.data(), big_.cig_off.data(), big_.seq4.data(),
                                    big_.seq_off.data(), want_qual_ ? big_.qual.data() : nullptr,
                                    want_qual_ ? big_.qual_off.data() : nullptr);
            if (rc_ != JL_OK) failed_ = dst;
        }
        const double ms = ms_since(t);
        ms_append += ms;
        ms_append_max = std::max(ms_append_max, ms);
        ++n_appends;
        big_.clear();
    }
    void run()
    {
        // the gathered arrays at about the size the device arrays get (virtual until touched), at most kGatherCap at a time
        const size_t kGatherCap = (size_t)512 << 20;
        {
            const size_t seq_hint = (size_t)std::min<uint64_t>(file_bytes_ * 7 / 2, kGatherCap);
            big_.seq4.reserve(seq_hint);
            big_.cigar.reserve(seq_hint / 8);
            if (want_qual_) big_.qual.reserve(2 * seq_hint);
        }
        for (;;) {
            std::deque<RecordArrays> got;
            bool finished = false;
            {
                std::unique_lock<std::mutex> lk(m_);
                if (ready_) cv_.wait(lk, [this] { return done_ || !q_.empty(); });
                else cv_.wait_for(lk, std::chrono::microseconds(250), [this] { return done_ || !q_.empty(); });   // (the contexts too)
                got.swap(q_);
                finished = done_ && got.empty();
            }
            auto t = std::chrono::steady_clock::now();
            for (RecordArrays &c : got) gather(c);
            ms_gather += ms_since(t);
            t = std::chrono::steady_clock::now();
            for (RecordArrays &c : got) {      // (beside the crew's copies)
                n_reads += c.pos.size();
                for (std::string &nm : c.names) names.push_back(std::move(nm));
            }
            ms_names += ms_since(t);
            t = std::chrono::steady_clock::now();
            crew_.wait();
            ms_gather += ms_since(t);
            for (RecordArrays &c : got) {
                c.clear();
                std::lock_guard<std::mutex> lk(m_);
                if (pool_.size() < 8) pool_.push_back(std::move(c));
            }
            if (!ready_ && (finished || gathered_bytes() >= kGatherCap || contexts_ready())) open();
            // on the device as soon as nothing more is waiting to be gathered (while the decode still runs: chunk by chunk,
            // hidden under it, as before)
            if (ready_) {
                bool idle;
                {
                    std::lock_guard<std::mutex> lk(m_);
                    idle = q_.empty();
                }
                if (idle || finished || gathered_bytes() >= kGatherCap / 2) flush();
            }
            if (finished) return;
        }
    }
    std::vector<std::shared_future<std::pair<int, jl_ctx *>>> ctx_up_;
    uint64_t file_bytes_;
    bool want_qual_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<RecordArrays> q_;
    std::vector<RecordArrays> pool_;
    bool done_ = false, ready_ = false;
    RecordArrays big_;
    static unsigned crew_size()
    {
        if (const char *e = getenv("JL_COPY_THREADS")) return std::max(1, atoi(e));     // (tuning)
        return 3;      // (1, 3, 8 on the 16-thread box: 23-34, 18-25, 19-28 ms for the 0.45 GB — the uploader thread copies too)
    }
    CopyCrew crew_{crew_size()};
    int rc_ = JL_OK;
    std::vector<jl_ctx *> ctxs_;
    jl_ctx *failed_ = nullptr;
    std::thread th_;   // last: starts in the constructor's initialiser list
};


// What the device stage hands to the writers, whichever way it ran (one window, or K windows over R devices).
struct Results {
    std::vector<jl_variant> var;          // (gene, codon_pos, codon) order; col relative to the overall window
    std::vector<uint32_t> col_counts;     // [n_cols][6] of the overall window
    jl_phase_summary ps = {};
    std::vector<uint32_t> pos_cols, hap_count;   // pos_cols relative to the overall window
    std::vector<uint8_t> hap_pattern, hit;
    size_t pat_stride = 0, hit_stride = 0;       // hap_pattern[h * pat_stride + p], hit[v * hit_stride + h]
    std::vector<uint16_t> read_hap;
};

struct WindowPlan {
    uint32_t begin = 0, ncols = 0;   // reference columns [begin, begin + ncols)
    uint32_t own_begin = 0, own_end = 0;   // the columns whose pileup counts this window contributes (no overlap)
    int rank = 0;
};

// K windows with a 2-column overlap, so that every codon is evaluated by exactly one window whatever its frame
// (minorseq_amd/sharding.py window_bounds); consecutive windows per rank.
std::vector<WindowPlan> plan_windows(uint32_t win_begin, uint32_t n_cols, uint32_t k_windows, uint32_t n_ranks)
{
    std::vector<WindowPlan> w(k_windows);
    for (uint32_t k = 0; k < k_windows; ++k) {
        const uint32_t c0 = (uint32_t)((uint64_t)n_cols * k / k_windows), c1 = (uint32_t)((uint64_t)n_cols * (k + 1) / k_windows);
        w[k].begin = win_begin + c0;
        w[k].ncols = std::min(n_cols, c1 + (k + 1 < k_windows ? 2u : 0u)) - c0;
        w[k].own_begin = c0;
        w[k].own_end = c1;
        w[k].rank = (int)((uint64_t)k * n_ranks / k_windows);
    }
    return w;
}

struct DeviceStageInput {
    const Options *opt;
    const TargetConfig *cfg;
    const std::vector<jl_gene> *genes;
    const std::vector<uint8_t> *refcodes;
    jl_params prm;
    uint32_t win_begin, n_cols;
    uint64_t n_reads;
};

// --drm-only: the codons of the config's DRMs per evaluated position of one window (doc/JULIET.md:370)
int drm_masks_of(jl_ctx *ctx, const DeviceStageInput &in, std::vector<uint64_t> &masks)
{
    const uint8_t *refp = in.refcodes->empty() ? nullptr : in.refcodes->data();
    if (jl_pileup_async(ctx, in.genes->data(), (uint32_t)in.genes->size(), refp, (uint32_t)in.refcodes->size()) != JL_OK) return 1;
    const uint32_t P = jl_n_positions(ctx);
    std::vector<uint32_t> pg(P), pk(P);
    if (jl_pileup_fetch(ctx, nullptr, pg.data(), pk.data(), nullptr, nullptr, nullptr) != JL_OK) return 1;
    masks.assign(P, 0);
    for (uint32_t p = 0; p < P; ++p) {
        const GeneCfg &g = in.cfg->genes[pg[p]];
        for (unsigned cod = 0; cod < 64; ++cod)
            if (!in.cfg->known_drms(pg[p], pk[p] + g.first_codon, translate(cod)).empty()) masks[p] |= 1ull << cod;
    }
    return 0;
}

// One rank = one device: its windows out of the records uploaded to it, the call stage per window with the GLOBAL
// Bonferroni factor, then — with phasing — its share of the cross-window sequence (jl_xwin_phase_sharded: the ranks'
// collectives meet inside).  Every rank ends with the whole result; rank 0's is written.
// The rank threads of one process agree before they enter anything collective: a rank that failed on its own (context,
// ingest, call stage) must not leave its peers waiting inside the communicator's bootstrap or an exchange.  Every rank
// calls vote() exactly once; all of them learn whether all of them are fine.
struct RankVote {
    explicit RankVote(int n) : n_(n) {}
    bool vote(bool ok)
    {
        std::unique_lock<std::mutex> lk(m_);
        all_ok_ = all_ok_ && ok;
        if (++arrived_ == n_) cv_.notify_all();
        else cv_.wait(lk, [this] { return arrived_ == n_; });
        return all_ok_;
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, arrived_ = 0;
    bool all_ok_ = true;
};

struct RankJob {
    int rank = 0, world = 1, device = 0;
    jl_ctx *records = nullptr;
    std::vector<uint32_t> widx;          // this rank's windows (indices into the plan)
    std::vector<jl_ctx *> wins;
    jl_comm *comm = nullptr;
    bool inproc = false;                 // the ranks exchange by device copies, not over RCCL
    std::string error;                   // empty: fine
    std::vector<std::pair<const char *, double>> laps;   // --timing: milliseconds by stage of this rank (rank 0's are printed)
    // outputs
    std::vector<std::vector<jl_variant>> tables;   // per window (window-relative columns), call only
    Results res;                         // with phasing: the merged table and the haplotypes (rank 0's is used)
    uint64_t slice_begin = 0, slice_reads = 0;
    std::vector<uint16_t> ids;           // this rank's slice
};

// the stages of a rank that involve no other rank: window contexts, ingest, call stage, column counts
static void run_rank_local(RankJob &job, const DeviceStageInput &in, const std::vector<WindowPlan> &plan, std::vector<uint32_t> &col_counts,
                           std::chrono::steady_clock::time_point &t_last);

void run_rank(RankJob &job, const DeviceStageInput &in, const std::vector<WindowPlan> &plan, const uint8_t *comm_id,
              std::vector<uint32_t> &col_counts, const std::vector<uint64_t> &slice_begin, RankVote *vote)
{
    auto t_last = std::chrono::steady_clock::now();
    run_rank_local(job, in, plan, col_counts, t_last);
    const Options &opt = *in.opt;
    if (opt.phasing && job.world > 1 && vote) {
        // nothing collective has been touched yet: either every rank goes on, or none does
        if (!vote->vote(job.error.empty())) {
            if (job.error.empty()) job.error = "stopped: another rank failed before the exchange";
            return;
        }
    } else if (!job.error.empty()) {
        return;
    }
    if (!opt.phasing) return;
    auto lap = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        job.laps.emplace_back(what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    auto fail = [&](const char *what, jl_ctx *c) { job.error = std::string(what) + ": " + (c ? jl_last_error(c) : "failed"); };
    // The communicator's bootstrap is collective too: a rank that fails in it leaves the others to RCCL's own time-out.
    if (job.world > 1 && (job.inproc ? jl_comm_create_inproc(job.wins[0], comm_id, job.rank, job.world, &job.comm)
                                     : jl_comm_create(job.wins[0], comm_id, job.rank, job.world, &job.comm)) != JL_OK)
        return fail("communicator", job.wins[0]);
    std::vector<uint32_t> wb, wn;
    std::vector<int32_t> wr;
    for (const WindowPlan &wp : plan) { wb.push_back(wp.begin); wn.push_back(wp.ncols); wr.push_back(wp.rank); }
    jl_xwin *x = nullptr;
    if (jl_xwin_create(job.wins.data(), (uint32_t)job.wins.size(), job.comm, wb.data(), wn.data(), wr.data(), (uint32_t)plan.size(),
                       slice_begin.data(), &x) != JL_OK)
        return fail("cross-window session", nullptr);
    lap("communicator + session");
    jl_xwin_result r;
    if (jl_xwin_phase_sharded(x, opt.min_reads, &r) != JL_OK) {
        job.error = std::string("cross-window phasing: ") + jl_xwin_last_error(x);
        jl_xwin_destroy(x);
        return;
    }
    lap("cross-window phasing");
    Results &R = job.res;
    R.var.assign(r.merged, r.merged + r.n_variants);
    for (jl_variant &v : R.var) v.col -= in.win_begin;
    R.ps = r.summary;
    R.ps.n_positions = r.n_positions;
    R.ps.n_haplotypes = r.n_haplotypes;
    R.pos_cols.resize(r.n_positions);
    for (uint32_t p = 0; p < r.n_positions; ++p) R.pos_cols[p] = r.pos_global[p] - in.win_begin;
    if (r.n_positions) {
        R.hap_count.assign(r.hap_count, r.hap_count + r.n_haplotypes);
        R.hap_pattern.assign(r.hap_pattern, r.hap_pattern + (size_t)r.n_haplotypes * r.n_positions);
        R.hit.assign(r.hit, r.hit + (size_t)r.n_variants * r.n_haplotypes);
    }
    R.pat_stride = r.n_positions;
    R.hit_stride = r.n_haplotypes;
    job.slice_begin = r.slice_begin;
    job.slice_reads = r.slice_reads;
    job.ids.resize(r.slice_reads ? r.slice_reads : 1);
    if (jl_xwin_read_hap_fetch(x, job.ids.data()) != JL_OK) job.error = std::string("per-read ids: ") + jl_xwin_last_error(x);
    job.ids.resize(r.slice_reads);
    lap("per-read ids");
    jl_xwin_destroy(x);
    lap("session closed");
}

static void run_rank_local(RankJob &job, const DeviceStageInput &in, const std::vector<WindowPlan> &plan, std::vector<uint32_t> &col_counts,
                           std::chrono::steady_clock::time_point &t_last)
{
    auto fail = [&](const char *what, jl_ctx *c) { job.error = std::string(what) + ": " + (c ? jl_last_error(c) : "failed"); };
    auto lap = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        job.laps.emplace_back(what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const Options &opt = *in.opt;
    const uint8_t *refp = in.refcodes->empty() ? nullptr : in.refcodes->data();
    for (uint32_t k : job.widx) {
        // a window's context orders its work on the stream of this rank's records context: a stream of its own is a hardware
        // queue the runtime takes 8 ms to create (tools_tuning/ctx_startup.cpp), eight windows 70 ms — and one rank drives its
        // windows one after the other anyway
        jl_ctx *w = nullptr;
        if (jl_ctx_create(job.device, jl_ctx_stream(job.records), &w) != JL_OK) return fail("context", nullptr);
        job.wins.push_back(w);
        if (jl_records_window(job.records, w, plan[k].ncols, plan[k].begin, opt.min_qv) != JL_OK) return fail("ingest", w);
    }
    lap("window contexts + device ingest");
    jl_records_drop(job.records);
    lap("records dropped");
    // the call stage of every window: enqueued one after the other on the windows' own streams (they overlap on the device)
    std::vector<std::vector<uint64_t>> masks(job.wins.size());
    for (size_t i = 0; i < job.wins.size(); ++i) {
        if (opt.drm_only && drm_masks_of(job.wins[i], in, masks[i])) return fail("pileup", job.wins[i]);
        if (jl_run_async(job.wins[i], in.genes->data(), (uint32_t)in.genes->size(), refp, (uint32_t)in.refcodes->size(), &in.prm,
                         opt.drm_only ? masks[i].data() : nullptr, 0, opt.min_reads, 0) != JL_OK)
            return fail("run", job.wins[i]);
    }
    lap("call stage enqueued");
    // column counts of the columns each window owns (the MSA context of the output, doc/JULIET.md:99-100)
    for (size_t i = 0; i < job.wins.size(); ++i) {
        const WindowPlan &wp = plan[job.widx[i]];
        std::vector<uint32_t> cc((size_t)wp.ncols * 6);
        if (jl_pileup_fetch(job.wins[i], cc.data(), nullptr, nullptr, nullptr, nullptr, nullptr) != JL_OK) return fail("pileup fetch", job.wins[i]);
        const uint32_t off = wp.own_begin - (wp.begin - in.win_begin);   // 0: a window starts where its own columns start
        std::copy(cc.begin() + (size_t)off * 6, cc.begin() + (size_t)(off + wp.own_end - wp.own_begin) * 6,
                  col_counts.begin() + (size_t)wp.own_begin * 6);
    }
    lap("column counts");
    if (!opt.phasing) {
        for (jl_ctx *w : job.wins) {
            std::vector<jl_variant> t(4096);
            uint32_t n = 0;
            if (jl_call_fetch(w, t.data(), 4096, &n) != JL_OK) return fail("call fetch", w);
            t.resize(n);
            job.tables.push_back(std::move(t));
        }
        return;
    }
}

}  // namespace

int main(int argc, char **argv)
{
    // eight hardware queues instead of the runtime's four (the rank threads' streams beside the exchange): read by the HIP
    // runtime at the process's first HIP call, so set here, before any thread exists and before that call
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    Options opt;
    std::string cmdline;
    for (int i = 0; i < argc; ++i) cmdline += (i ? " " : "") + std::string(argv[i]);
    try {
        opt = parse(argc, argv);
        const auto t_start = std::chrono::steady_clock::now();
        auto t_last = t_start;
        auto tick = [&](const char *what) {
            if (!opt.timing) return;
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "juliet: timing %-26s %9.1f ms  (at %9.1f ms)\n", what,
                    std::chrono::duration<double, std::milli>(now - t_last).count(),
                    std::chrono::duration<double, std::milli>(now - t_start).count());
            t_last = now;
        };
        // ---------------------------------------------------------------- target config
        TargetConfig cfg;
        if (!opt.config.empty()) cfg = TargetConfig::load(opt.config);
        if (!opt.dump_config.empty() && opt.bam.empty()) {
            if (opt.have_region) cfg.apply_region(opt.region_b, opt.region_e);
            Json j = cfg.echo();
            Json eff = Json::array();
            for (const GeneCfg &g : cfg.genes)
                eff.push(Json::object().set("name", Json::of(g.name)).set("begin", Json::of(g.begin_eff)).set("end", Json::of(g.end_eff)).set("first_codon", Json::of(g.first_codon)));
            j.set("effective_genes", eff);
            std::string s;
            j.write(s);
            std::ofstream(opt.dump_config) << s << "\n";
            return 0;
        }
        // ---------------------------------------------------------------- ingest
        IngestOptions io;
        io.min_qv = opt.min_qv;
        io.min_rq = opt.min_rq;
        // the GPU context comes up (runtime start, stream, pinned blocks) while the host reads the BAM
        const bool need_gpu = !opt.outputs.empty() || opt.fuse_only;
        std::vector<std::shared_future<std::pair<int, jl_ctx *>>> ctx_ups;
        std::unique_ptr<RecordUploader> uploader;
        RecordSink sink;
        if (need_gpu) {
            for (int dev : opt.devices)
                ctx_ups.push_back(std::async(std::launch::async, [dev]() {
                    jl_ctx *c = nullptr;
                    const int rc = jl_ctx_create(dev, nullptr, &c);
                    return std::make_pair(rc, c);
                }).share());
            std::error_code ec;
            const uintmax_t fsz = std::filesystem::file_size(opt.bam, ec);
            uploader.reset(new RecordUploader(ctx_ups, ec ? 0 : (uint64_t)fsz, opt.min_qv > 0));
            sink.give = [&uploader](RecordArrays &c) { uploader->give(c); };
        }
        // ONE pass over the file: records as decoded from BAM (cigar expansion, QV masking and the transpose run on
        // the device) and the extent they cover
        RecordArrays rec;
        std::vector<BamRef> bam_refs;
        std::string header_text;
        // (with a device behind it: the pipelined reader — inflate and record parsing on every core, chunks to the uploader
        // in file order; the GPU-free diagnostics and non-BGZF files take the sequential one)
        const ReadExtent ext = (uploader && PipelinedBamReader::is_bgzf(opt.bam))
                                   ? PipelinedBamReader::run(opt.bam, io, io.ref_id, opt.min_qv > 0, sink, &bam_refs, &header_text)
                                   : collect_records(opt.bam, io, io.ref_id, opt.min_qv > 0, rec, &bam_refs, &header_text,
                                                     uploader ? &sink : nullptr);
        tick("bam decode");
        if (ext.n_reads == 0) { std::cerr << "juliet: no primary or supplementary alignments in " << opt.bam << "\n"; return 2; }
        int64_t ref_len = std::numeric_limits<int64_t>::max();
        if (ext.ref_id >= 0 && (size_t)ext.ref_id < bam_refs.size()) ref_len = bam_refs[(size_t)ext.ref_id].length;

        const bool have_cfg = !cfg.genes.empty();
        if (!have_cfg) {
            // no target config: one ORF over the covered window, labelled "unknown" (doc/JULIET.md:182-188);
            // --region marks the reading frame
            GeneCfg g;
            g.name = "unknown";
            g.begin = g.begin_eff = opt.have_region ? opt.region_b : (uint32_t)ext.min_pos + 1;
            g.end = g.end_eff = opt.have_region ? opt.region_e : (uint32_t)ext.max_end + 1;
            cfg.genes.push_back(g);
        } else if (opt.have_region) {
            cfg.apply_region(opt.region_b, opt.region_e);
            if (cfg.genes.empty()) { std::cerr << "juliet: --region leaves no gene of the config\n"; return 1; }
        }
        // window: the called genes plus the -3..+5 context columns (doc/JULIET.md:99-100), inside the reference
        int64_t gb = std::numeric_limits<int64_t>::max(), ge = 0;
        for (const GeneCfg &g : cfg.genes) { gb = std::min<int64_t>(gb, (int64_t)g.begin_eff - 1); ge = std::max<int64_t>(ge, (int64_t)g.end_eff - 1); }
        const int64_t wb = std::max<int64_t>(0, gb - 3);
        const int64_t we = std::max<int64_t>(wb + 1, std::min<int64_t>(ref_len, ge + 5));
        const uint32_t win_begin = (uint32_t)wb, n_cols = (uint32_t)(we - wb);

        std::vector<std::string> names;
        uint64_t n_reads = 0;
        if (!opt.dump_msa.empty()) {  // host-side ingest check, no GPU involved
            std::vector<uint8_t> rows;
            n_reads = build_rows(opt.bam, io, ext.ref_id, win_begin, n_cols, ext.n_reads, rows, nullptr);
            std::ofstream f(opt.dump_msa, std::ios::binary);
            const uint64_t hdr[3] = {n_reads, n_cols, win_begin};
            f.write((const char *)hdr, sizeof hdr);
            f.write((const char *)rows.data(), (std::streamsize)((size_t)n_reads * n_cols));
            if (opt.outputs.empty()) return 0;
        }
        n_reads = ext.n_reads;

        // ---------------------------------------------------------------- parameters
        std::string chem = opt.chemistry;
        if (chem == "auto") {
            // chemistry-keyed rates with a permissive fallback (doc/JULIET.md:221-225); the key here is the
            // platform model in the @RG line
            chem = (header_text.find("SEQUEL") != std::string::npos || header_text.find("S/P") != std::string::npos) ? "sequel" : "permissive";
            if (chem == "permissive") std::cerr << "juliet: chemistry not recognised, permissive mode is active (doc/JULIET.md:221-225)\n";
        }
        jl_params prm;
        prm.alpha = opt.alpha;
        prm.n_tests = opt.n_tests;
        if (chem == "sequel") prm.err = {0.998826, 5.8e-5, 1.0e-3};
        else prm.err = {0.99764, 1.2e-4, 2.0e-3};
        if (opt.match > 0) prm.err.match = opt.match;
        if (opt.substitution >= 0) prm.err.substitution = opt.substitution;
        prm.expected_round = opt.expected_round;
        prm.tail = opt.fisher_tail;
        prm.min_perc = opt.min_perc;
        prm.max_perc = opt.max_perc;

        std::vector<jl_gene> genes;
        for (const GeneCfg &g : cfg.genes) genes.push_back({g.begin_eff, g.end_eff});
        std::vector<uint8_t> refcodes;
        if (!cfg.reference_sequence.empty())
            for (char ch : cfg.reference_sequence) refcodes.push_back(base_code(ch));

        // ---------------------------------------------------------------- device
        jl_ctx *ctx = nullptr;
        for (auto &f : ctx_ups) {
            const auto up = f.get();
            if (up.first != JL_OK) die_jl(nullptr, "no usable GPU (this tool has no CPU fallback)");
            if (!ctx) ctx = up.second;
        }
        tick("context ready");
        if (uploader->finish() != JL_OK) die_jl(uploader->failed() ? uploader->failed() : ctx, "record upload");
        if (uploader->n_reads != n_reads) die_jl(nullptr, "record upload lost reads");
        names.swap(uploader->names);
        tick("rest of the upload");
        if (opt.timing)
            fprintf(stderr, "juliet: timing   uploader thread: gather %.1f ms, begin %.1f ms, %u appends %.1f ms (longest %.1f), names %.1f ms\n",
                    uploader->ms_gather, uploader->ms_begin, uploader->n_appends, uploader->ms_append, uploader->ms_append_max, uploader->ms_names);
        const uint8_t *refp = refcodes.empty() ? nullptr : refcodes.data();
        Results R;
        R.col_counts.assign((size_t)n_cols * 6, 0);
        DeviceStageInput in{&opt, &cfg, &genes, &refcodes, prm, win_begin, n_cols, n_reads};
        const size_t n_ranks = opt.devices.size();
        if (opt.windows > 1 || n_ranks > 1) {
            // ---- K column windows over R devices (doc/JULIET.md:261-264: each gene is treated separately, so the split
            // never shows): one rank (thread) per device; the Bonferroni factor counts the codons of ALL genes in every window
            const uint32_t K = std::min<uint32_t>(opt.windows, std::max<uint32_t>(1, n_cols / 8));
            if (K < n_ranks) { std::cerr << "juliet: the window is too narrow for " << n_ranks << " devices\n"; return 1; }
            const std::vector<WindowPlan> plan = plan_windows(win_begin, n_cols, K, (uint32_t)n_ranks);
            // read slices for phasing: starts on multiples of 256 reads (a 128-byte line of every column)
            std::vector<uint64_t> slices(n_ranks + 1, n_reads);
            {
                uint64_t per = (n_reads + n_ranks - 1) / n_ranks;
                per = (per + 255) / 256 * 256;
                for (size_t r = 0; r < n_ranks; ++r) slices[r] = std::min<uint64_t>(n_reads, r * per);
            }
            uint8_t comm_id[128] = {0};
            if (opt.phasing && n_ranks > 1 && jl_comm_unique_id(comm_id) != JL_OK) die_jl(nullptr, "communicator id");
            // RCCL refuses two ranks on one device; ranks that are threads of one process can exchange by device copies
            bool inproc = opt.exchange == "inproc";
            if (opt.exchange.empty())
                for (size_t a = 0; a < n_ranks; ++a)
                    for (size_t b = a + 1; b < n_ranks; ++b) inproc = inproc || opt.devices[a] == opt.devices[b];
            {   // distinct devices: the exchanges between them (RCCL, or peer copies in process) have never run on hardware
                bool distinct = false;
                for (size_t a = 0; a < n_ranks; ++a)
                    for (size_t b = a + 1; b < n_ranks; ++b) distinct = distinct || opt.devices[a] != opt.devices[b];
                if (distinct)
                    fprintf(stderr, "juliet: warning: --devices with more than one distinct device is experimental: the exchange between devices is "
                                    "covered by one-device tests only (in-process ranks, one-rank RCCL)\n");
            }
            std::vector<RankJob> jobs(n_ranks);
            for (size_t r = 0; r < n_ranks; ++r) {
                jobs[r].inproc = inproc;
                jobs[r].rank = (int)r;
                jobs[r].world = (int)n_ranks;
                jobs[r].device = opt.devices[r];
                jobs[r].records = uploader->ctx(r);
                for (uint32_t k = 0; k < K; ++k)
                    if (plan[k].rank == (int)r) jobs[r].widx.push_back(k);
            }
            RankVote vote((int)n_ranks);
            std::vector<std::thread> threads;
            for (size_t r = 1; r < n_ranks; ++r)
                threads.emplace_back([&, r] { run_rank(jobs[r], in, plan, comm_id, R.col_counts, slices, &vote); });
            run_rank(jobs[0], in, plan, comm_id, R.col_counts, slices, &vote);
            for (std::thread &t : threads) t.join();
            for (const RankJob &j : jobs)
                if (!j.error.empty()) { std::cerr << "juliet: rank " << j.rank << " (device " << j.device << "): " << j.error << "\n"; return 3; }
            tick("windows: ingest + call + phase");
            if (opt.timing)
                for (const auto &l : jobs[0].laps) fprintf(stderr, "juliet: timing   rank 0: %-34s %6.1f ms\n", l.first, l.second);
            std::vector<uint32_t> cc;
            cc.swap(R.col_counts);
            if (opt.phasing) {
                R = std::move(jobs[0].res);
                R.read_hap.assign(n_reads, (uint16_t)JL_HAP_DAMAGED);
                for (const RankJob &j : jobs) std::copy(j.ids.begin(), j.ids.end(), R.read_hap.begin() + (ptrdiff_t)j.slice_begin);
            } else {
                std::vector<const jl_variant *> tabs;
                std::vector<uint32_t> cnt, begins;
                for (const RankJob &j : jobs)
                    for (size_t i = 0; i < j.tables.size(); ++i) {
                        tabs.push_back(j.tables[i].data());
                        cnt.push_back((uint32_t)j.tables[i].size());
                        begins.push_back(plan[j.widx[i]].begin - win_begin);
                    }
                uint64_t total = 0;
                for (uint32_t c : cnt) total += c;
                R.var.resize(total ? total : 1);
                uint32_t n = 0;
                if (jl_merge_tables(tabs.data(), cnt.data(), begins.data(), (uint32_t)tabs.size(), R.var.data(), (uint32_t)R.var.size(), &n) != JL_OK)
                    die_jl(nullptr, "merge of the windows' tables");
                R.var.resize(n);
            }
            R.col_counts.swap(cc);
            for (RankJob &j : jobs)
                if (j.comm) jl_comm_destroy(j.comm);   // (RCCL wants its communicators closed; contexts end with the process)
            tick("kernels + fetch");
        } else {
        if (!opt.consensus.empty()) jl_msa_track_insertions(ctx, 1);   // fuse keeps in-frame insertions (doc/FUSE.md:19)
        if (jl_records_finish(ctx, n_cols, win_begin, opt.min_qv) != JL_OK) die_jl(ctx, "ingest");
        tick("device ingest");

        // --drm-only needs the position list, which the plan of a first pileup provides
        std::vector<uint64_t> drm_masks;
        if (opt.drm_only && drm_masks_of(ctx, in, drm_masks)) die_jl(ctx, "pileup");
        if (opt.fuse_only) {   // the column pileup is all a consensus needs
            if (jl_pileup_async(ctx, genes.data(), (uint32_t)genes.size(), refp, (uint32_t)refcodes.size()) != JL_OK) die_jl(ctx, "pileup");
        } else if (jl_run_async(ctx, genes.data(), (uint32_t)genes.size(), refp, (uint32_t)refcodes.size(), &prm,
                                opt.drm_only ? drm_masks.data() : nullptr, opt.phasing, opt.min_reads, opt.phasing) != JL_OK)
            die_jl(ctx, "run");
        tick("plan + enqueue");

        R.var.resize(4096);
        uint32_t nv = 0;
        if (!opt.fuse_only && jl_call_fetch(ctx, R.var.data(), 4096, &nv) != JL_OK) die_jl(ctx, "call fetch");
        R.var.resize(nv);
        tick("  wait for the run + table");
        if (jl_pileup_fetch(ctx, R.col_counts.data(), nullptr, nullptr, nullptr, nullptr, nullptr) != JL_OK) die_jl(ctx, "pileup fetch");
        tick("  column counts");

        if (!opt.consensus.empty()) {  // what `fuse` writes for this window (doc/FUSE.md:17-24)
            std::vector<uint32_t> len_hist((size_t)n_cols * 32), base_counts((size_t)n_cols * 120);
            if (jl_insertions_fetch(ctx, len_hist.data(), base_counts.data()) != JL_OK) die_jl(ctx, "insertions");
            const std::string seq = fuse_consensus(n_cols, R.col_counts, len_hist, base_counts, opt.ins_min_frac, opt.ins_min_distance);
            std::ofstream f(opt.consensus);
            if (!f) { std::cerr << "juliet: cannot write " << opt.consensus << "\n"; return 2; }
            f << ">consensus window=" << (win_begin + 1) << "-" << (win_begin + n_cols) << " source=" << opt.bam << "\n";
            for (size_t i = 0; i < seq.size(); i += 70) f << seq.substr(i, 70) << "\n";
        }
        if (opt.fuse_only) {
            tick("pileup + consensus");
            jl_ctx_destroy(ctx);
            return 0;
        }
        const uint32_t cap_var = std::max<uint32_t>(1, nv);
        if (opt.phasing) {
            R.pos_cols.resize(cap_var);
            R.hap_count.resize(JL_MAX_HAPLOTYPES);
            R.hap_pattern.resize((size_t)JL_MAX_HAPLOTYPES * cap_var);
            R.hit.resize((size_t)cap_var * JL_MAX_HAPLOTYPES);
            R.read_hap.resize(n_reads);
            R.pat_stride = cap_var;
            R.hit_stride = JL_MAX_HAPLOTYPES;
            if (jl_phase_fetch(ctx, &R.ps, R.pos_cols.data(), R.hap_count.data(), R.hap_pattern.data(), R.hit.data(), R.read_hap.data(), nullptr, cap_var) != JL_OK)
                die_jl(ctx, "phase fetch");
        }
        tick("  haplotypes + ids");
        // (the context is not torn down: the process is about to end, and freeing two dozen device buffers one by one took
        // 4-6 ms of a 0.1 s run)
        }
        const std::vector<jl_variant> &var = R.var;
        const std::vector<uint32_t> &col_counts = R.col_counts;
        const jl_phase_summary &ps = R.ps;
        const std::vector<uint32_t> &pos_cols = R.pos_cols, &hap_count = R.hap_count;
        const std::vector<uint8_t> &hap_pattern = R.hap_pattern, &hit = R.hit;
        const std::vector<uint16_t> &read_hap = R.read_hap;

        // ---------------------------------------------------------------- JSON (doc/JULIET.md:61-107, 207-211)
        Json root = Json::object();
        root.set("input", Json::object()
                              .set("timestamp", Json::of(iso_now()))
                              .set("input_file", Json::of(opt.bam))
                              .set("command_line", Json::of(cmdline))
                              .set("juliet_version", Json::of(kVersion)));
        Json tc = cfg.echo();
        tc.set("n_reads", Json::of((int64_t)n_reads));
        tc.set("window_begin", Json::of(win_begin + 1)).set("window_end", Json::of(win_begin + n_cols + 1));
        tc.set("chemistry_model", Json::of(chem));
        root.set("target_config", tc);

        Json genes_json = Json::array();
        const uint32_t H = ps.n_haplotypes;
        for (size_t g = 0; g < cfg.genes.size(); ++g) {
            Json gj = Json::object();
            gj.set("name", Json::of(cfg.genes[g].name));
            Json vps = Json::array();
            size_t v = 0;
            while (v < var.size()) {
                if (var[v].gene != g) { ++v; continue; }
                size_t e = v;
                while (e < var.size() && var[e].gene == g && var[e].codon_pos == var[v].codon_pos) ++e;
                const jl_variant &f = var[v];
                Json vp = Json::object();
                vp.set("ref_codon", Json::of(codon_string(f.ref_codon)));
                vp.set("ref_amino_acid", Json::of(std::string(1, translate(f.ref_codon))));
                const uint32_t aa_pos = f.codon_pos + cfg.genes[g].first_codon;
                vp.set("ref_position", Json::of(aa_pos));
                vp.set("coverage", Json::of(f.coverage));
                // variant codons grouped by amino acid (position 223 with two rows)
                Json aas = Json::array();
                std::vector<char> order;
                for (size_t k = v; k < e; ++k) {
                    const char aa = translate(var[k].codon);
                    if (std::find(order.begin(), order.end(), aa) == order.end()) order.push_back(aa);
                }
                // amino acids in alphabetical order: juliet_abl-nohaplotype.png prints "A GCC" above "P CCA" at ABL1 223
                std::sort(order.begin(), order.end());
                for (char aa : order) {
                    Json aj = Json::object();
                    aj.set("amino_acid", Json::of(std::string(1, aa)));
                    Json cods = Json::array();
                    for (size_t k = v; k < e; ++k) {
                        if (translate(var[k].codon) != aa) continue;
                        Json cj = Json::object();
                        cj.set("codon", Json::of(codon_string(var[k].codon)));
                        cj.set("frequency", Json::of((double)var[k].count / (double)var[k].coverage));
                        cj.set("count", Json::of(var[k].count));
                        cj.set("expected", Json::of(var[k].expected));
                        cj.set("pValue", Json::of(var[k].p_value));
                        cj.set("log_pValue", Json::of(var[k].log_p));
                        cj.set("known_drm", Json::of(cfg.known_drms(g, aa_pos, aa)));
                        if (opt.phasing) {
                            Json hh = Json::array();
                            for (uint32_t h = 0; h < H; ++h) hh.push(Json::of(hit[(size_t)k * R.hit_stride + h] != 0));
                            cj.set("haplotype_hit", hh);  // doc/JULIET.md:207-209
                        }
                        cods.push(cj);
                    }
                    aj.set("variant_codons", cods);
                    aas.push(aj);
                }
                vp.set("variant_amino_acids", aas);
                // MSA context: -3 .. +5 around the codon's first base (doc/JULIET.md:99-100)
                Json msa = Json::array();
                for (int rel = -3; rel <= 5; ++rel) {
                    const int64_t c = (int64_t)f.col + rel;
                    if (c < 0 || c >= (int64_t)n_cols) continue;
                    const uint32_t *cc = &col_counts[(size_t)c * 6];
                    Json mj = Json::object();
                    mj.set("rel_pos", Json::of((int64_t)rel)).set("abs_pos", Json::of((int64_t)(win_begin + c + 1)));
                    static const char *sym[6] = {"A", "C", "G", "T", "-", "N"};
                    for (int s = 0; s < 6; ++s) mj.set(sym[s], Json::of(cc[s]));
                    const size_t r = (size_t)win_begin + (size_t)c;
                    if (r < cfg.reference_sequence.size()) mj.set("wt", Json::of(std::string(1, (char)std::toupper((unsigned char)cfg.reference_sequence[r]))));
                    msa.push(mj);
                }
                vp.set("msa", msa);
                vps.push(vp);
                v = e;
            }
            gj.set("variant_positions", vps);
            genes_json.push(gj);
        }
        root.set("genes", genes_json);

        // Section 4, drug summaries: variants grouped by annotated drug (doc/JULIET.md:104-107)
        {
            std::vector<std::pair<std::string, Json>> by_drug;
            for (const jl_variant &f : var) {
                const GeneCfg &g = cfg.genes[f.gene];
                const uint32_t aa_pos = f.codon_pos + g.first_codon;
                const char aa = translate(f.codon);
                for (const Drm &d : g.drms) {
                    bool hit_drm = false;
                    for (const DrmPosition &dp : d.positions) hit_drm = hit_drm || dp.matches(aa_pos, aa);
                    if (!hit_drm) continue;
                    Json e = Json::object();
                    e.set("gene", Json::of(g.name));
                    e.set("mutation", Json::of(std::string(1, translate(f.ref_codon)) + std::to_string(aa_pos) + std::string(1, aa)));
                    e.set("codon", Json::of(codon_string(f.codon)));
                    e.set("frequency", Json::of((double)f.count / (double)f.coverage));
                    auto it = std::find_if(by_drug.begin(), by_drug.end(), [&](const std::pair<std::string, Json> &kv) { return kv.first == d.name; });
                    if (it == by_drug.end()) { by_drug.emplace_back(d.name, Json::array()); it = by_drug.end() - 1; }
                    it->second.push(e);
                }
            }
            Json ds = Json::array();
            for (auto &kv : by_drug) ds.push(Json::object().set("drug", Json::of(kv.first)).set("variants", kv.second));
            root.set("drug_summaries", ds);
        }

        if (opt.phasing) {  // root `haplotype` block: counts and read names, same order as haplotype_hit (doc/JULIET.md:209-211)
            Json hb = Json::object();
            hb.set("reported_reads", Json::of(ps.reported_reads)).set("insufficient_coverage_reads", Json::of(ps.insufficient_reads));
            hb.set("damaged_reads", Json::of(ps.damaged_reads)).set("marginal_gaps", Json::of(ps.marginal_gap));
            hb.set("marginal_heteroduplexes", Json::of(ps.marginal_heteroduplex)).set("marginal_partial", Json::of(ps.marginal_partial));
            std::vector<std::vector<uint32_t>> members(H);
            for (uint64_t i = 0; i < n_reads; ++i)
                if (read_hap[i] < H) members[read_hap[i]].push_back((uint32_t)i);
            Json hs = Json::array();
            for (uint32_t h = 0; h < H; ++h) {
                Json hj = Json::object();
                hj.set("name", Json::of(haplotype_name(h))).set("reads", Json::of(hap_count[h]));
                hj.set("frequency", Json::of(ps.reported_reads ? (double)hap_count[h] / (double)ps.reported_reads : 0.0));
                Json cods = Json::array();
                for (uint32_t p = 0; p < ps.n_positions; ++p) cods.push(Json::of(codon_string(hap_pattern[(size_t)h * R.pat_stride + p])));
                hj.set("codons", std::move(cods));
                Json rn = Json::array();      // (moved on, level by level: a copy of this list per level was most of the stage at a million reads)
                rn.arr.reserve(members[h].size());
                for (uint32_t i : members[h]) rn.push(Json::of(names[i]));
                hj.set("read_names", std::move(rn));
                hs.push(std::move(hj));
            }
            hb.set("haplotypes", std::move(hs));
            Json pc = Json::array();
            for (uint32_t p = 0; p < ps.n_positions; ++p) pc.push(Json::of(win_begin + pos_cols[p] + 1));
            hb.set("variant_positions_abs", std::move(pc));
            root.set("haplotype", std::move(hb));
        }

        std::string text;
        root.write(text);
        text += "\n";
        for (const std::string &out : opt.outputs) {
            std::ofstream f(out);
            if (!f) { std::cerr << "juliet: cannot write " << out << "\n"; return 2; }
            if (out.substr(out.size() - 5) == ".json") f << text;
            else f << render_html(root);
        }
        tick("json / html");
        // Everything is written and closed.  What a `return` would still do — free a gigabyte of record arrays page by page, take down
        // the uploader and the decode pool, destroy the GPU contexts and the HIP runtime's own state — the operating system does at
        // once when the process ends: 40-60 ms of the wall time of a 100k-read run (JL_SLOW_EXIT=1: the long way, for leak checkers).
        if (!getenv("JL_SLOW_EXIT")) {
            std::cout.flush();
            std::cerr.flush();
            fflush(nullptr);
            _exit(0);
        }
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "juliet: " << e.what() << "\n";
        return 2;
    }
}
