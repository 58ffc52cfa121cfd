// filegrep.h -- host-side drop-in for the reference's FileGrep class.
//
// Public interface = the one /root/reference/src/grab.h:55-85 declares (what the CLI,
// the nftw callback and the worker threads call): why / recurse / show_path / prepare /
// config / find(path) / find(path, st, typeflag) / find_recursive, all returning 0 or
// -1 with the reason in why().  Behind it the libpcre match loop is gone: prepare()
// compiles the pattern for the gfx950 scan engine and opens a device context, find()
// streams the file's chunks through HBM and prints from the engine's candidate list.
// There is no CPU scanning path: prepare() fails if the pattern is outside the engine's
// subset or no HIP device can be opened.
#pragma once

#include <sys/stat.h>
#include <sys/types.h>

#include <cstddef>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gscan.h"

// Output switches of one scan, in the bit layout grab_report_chunk() takes.
enum : unsigned {
    GRAB_OFFSETS = 1u, // -O  "Match at offset N"
    GRAB_NOLINE = 2u,  // -l  do not print the line
    GRAB_SINGLE = 4u,  // -s  stop after the first match
    GRAB_PREFIX = 8u,  // "path:" in front of every record (-r, or several paths)
    GRAB_COLOR = 16u,  // -I  inverse video around the match
};

class FileGrep {
public:
    FileGrep();
    ~FileGrep();
    FileGrep(const FileGrep &) = delete;
    FileGrep &operator=(const FileGrep &) = delete;

    const char *why() { return err_.c_str(); }
    // a DEVICE error has happened (not a per-file one): nothing can be scanned any more and files handed over since may not
    // have been; why() says which.  Callers that ignore find()'s per-file results (the -n workers) ask this when they are done.
    bool failed() const { return failed_; }
    void recurse() { recursive_ = true; }
    void show_path(bool on) { show_path_ = on; }

    // keys the reference understands (grab.cc:83-98): color noline offsets single low_mem
    // chunk_size; extensions: literal (-S), device (HIP device index), out_fd, batch (largest
    // file size that is batched with others, 0 = none), devices (how many HIP devices, starting
    // at `device`, the windows of ONE multi-window file are spread over: contexts on the further
    // devices are opened when the first such file arrives; output stays in window order)
    void config(const std::map<std::string, size_t> &kv);
    int prepare(const std::string &regex);
    // prepare() without the device: 0 the pattern is fine; -1 PCRE itself rejects it (why: the reference's message);
    // -2 valid PCRE outside the engine's subset (why says what).  `grab -n` uses it to tell the two apart before any
    // worker exists (the reference ignores prepare()'s result there, src/main.cc:198).
    static int validate(const std::string &regex, bool literal, std::string &why, int *minlen = nullptr, gscan_db **db = nullptr);
    int find(const std::string &path);
    int find(const char *path, const struct stat *st, int typeflag);
    int find_recursive(const std::string &path);

    // Work handed over by find(path, st, typeflag) stays in flight across calls (the next file is read
    // while this one is on the GPU; small files wait for their batch to fill).  flush() scans and prints
    // whatever is pending; find(string), find_recursive() and the destructor call it themselves.
    int flush();
    // The caller has nothing to hand over right now (the work queue is empty): small files waiting for their batch to fill are
    // handed to the device if it has nothing of this instance's in flight, or if they are a fair launch's worth (8 MiB).
    // Nothing is waited for.
    void idle();
    // GRAB_TIMING=1: this instance's time and byte totals on stderr, once (the destructor's job; a caller that leaves
    // without destroying the instance -- the command line does -- asks for them itself)
    void report_timing();

    // engine knob pass-through (gscan_set_option) for A/B runs
    int engine_option(const char *name, long value);

private:
    struct FileRef;
    struct Job;
    int retire_oldest(bool print);
    int make_room(int ctx);       // retire until context `ctx` has a free slot
    int want_contexts(size_t n);  // open contexts on further devices up to n (lazily: a multi-window file asks)
    void deferred_error();        // a window of an EARLIER file failed while this one was being handed over
    int submit_batch();
    int batch_add(const char *path, const struct stat *st, int oflags);
    int batch_add_read(const char *path, int fd, size_t size);
    bool report_needs_text(unsigned rflags, size_t n, const uint32_t *ext, const uint32_t *ends, const uint8_t *gather) const;
    void file_error(const char *path); // an error of ONE file of a batch, found when the batch retires
    unsigned report_flags() const;
    int read_chunk(int fd, void *dst, size_t len, off_t at);
    void emit(std::string &text);

    std::string err_, deferred_;
    int minlen_ = 1;               // PCRE_INFO_MINLENGTH of the pattern (grab.h:44)
    size_t chunk_size_ = 1u << 30; // grab.h:48
    bool offsets_ = false, noline_ = false, single_ = false, color_ = false, low_mem_ = false;
    bool recursive_ = false, show_path_ = false, literal_ = false;
    bool anchored_ = false; // the pattern can only match at a restart position / chunk end: nothing goes to the GPU
    bool never_ = false;    // ... and not even there: the pattern's assertions contradict each other
    bool context_ = false;  // the pattern looks at the byte before / after its match
    bool resolve_ = false;  // the device settles the matches and their ends (gscan_info.resolve); reach_: how far in front of a match the pattern looks
    size_t reach_ = 0;
    bool lines_ = false;    // the device's line-extent pass applies to the pattern
    bool textfree_ = false; // fixed length, every candidate listed: without line printing the walk never reads the chunk (gscan_info.textfree)
    bool ends_ = false;     // the device's match-end pass applies to the pattern (-O -l is then walked without the text)
    uid_t uid_;
    int device_ = 0, out_fd_ = 1;
    // GRAB_TIMING=1 in the environment: per-instance wall-clock split, printed to stderr by the destructor
    bool timing_ = false, timing_reported_ = false;
    size_t t_files_ = 0, t_chunks_ = 0, t_bytes_ = 0;
    double t_map_ = 0, t_read_ = 0, t_submit_ = 0, t_wait_ = 0, t_report_ = 0, t_unmap_ = 0, t_emit_ = 0, t_text_ = 0;
    size_t t_text_files_ = 0;
    // pipeline state
    std::deque<Job> flight_;
    std::vector<gscan_ctx *> ctxs_;   // [0] == ctx_; further devices for the windows of one big file ("devices")
    std::vector<size_t> inflight_;    // jobs in flight per context
    std::vector<int> ctx_dev_;        // HIP device of each context
    std::vector<size_t> ctx_bytes_;   // bytes handed to each context (GRAB_TIMING prints them per device)
    size_t devices_ = 1;              // config "devices"
    size_t next_ctx_ = 0;             // round-robin cursor over the contexts a big file uses
    bool failed_ = false;             // a device error: every later find() fails at once with why() saying so
    std::string report_buf_;
    size_t batch_max_ = size_t(2) << 20; // files up to this size are batched ("batch" config key; 0 = never)
    size_t batch_bytes_ = size_t(32) << 20; // a batch is handed over when it holds this much (GRAB_BATCH_MIB), or kBatchMaxFiles files
    bool batch_by_worker_ = false;       // GRAB_BATCH_READ=worker (A/B): the worker reads the files into a pinned block itself
    bool silent_errors_ = false;         // config "silent_errors": per-file errors found after find() returned are not printed (the -n workers: main.cc:97)
    void *batch_buf_ = nullptr;          // ... that block, being filled
    size_t batch_used_ = 0;
    std::vector<std::shared_ptr<FileRef>> batch_files_;
    std::vector<gscan_seg> batch_segs_;
    gscan_db *db_ = nullptr;   // replaces pcre *d_pcreh
    gscan_ctx *ctx_ = nullptr; // replaces pcre_extra *d_extra (+ owns streams and buffers)
};

// What the reference prints for ONE chunk (grab.cc:171-213), driven by the engine's
// ascending candidate list instead of repeated pcre_exec calls.  Pure host function.
void grab_report_chunk(const gscan_db *db, int minlen, unsigned flags, const char *path,
                       const char *content, size_t clen, long long off, const uint32_t *starts,
                       size_t nstarts, std::string &out, const uint32_t *ext = nullptr, const uint32_t *ends = nullptr,
                       const uint8_t *gather = nullptr);
