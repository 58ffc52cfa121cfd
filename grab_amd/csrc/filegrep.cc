// filegrep.cc -- FileGrep over the gscan engine (see filegrep.h).
//
// Behavioural contract = /root/reference/src/grab.cc, cited where a rule comes from it;
// the implementation is organised around the device pipeline instead of the mmap +
// pcre_exec loop: chunks are read(2) into pinned memory, copied to HBM and scanned while
// the previous chunk is being printed.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include "filegrep.h"

#include <fcntl.h>
#include <ftw.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <deque>
#include <iostream>
#include <memory>
#include <mutex>

#ifdef GRAB_PCRE_VALIDATE
// libpcre is consulted ONLY about the pattern text: so that prepare() rejects exactly what
// the reference rejects, with the reference's messages, and to cross-check minlen.  It
// never sees file data.
#include <pcre.h>
#ifndef PCRE_STUDY_JIT_COMPILE
#define PCRE_STUDY_JIT_COMPILE 0
#endif
#endif

namespace {

std::mutex g_out_lock; // one writer at a time, whole chunks only (grab.cc:56,217-226)
// (Round 6, measured and taken out again: stdout that is a plain file written by OFFSET -- the lock held only to reserve the
// chunk's range, the bytes going out with pwrite beside the other workers' chunks.  On tmpfs the writers then queue on the
// file's inode lock instead, with more system time: 0.83 s per worker "writing it out" against 0.13 s, 8.7 s of system time
// against 3.3 -- profiles/r06_c_dense_timing_positional_writes_rejected.txt.)
// GRAB_TIMING: how the printed lines came about (device line pass with gathered text / with text from the window / the host's loop)
std::atomic<unsigned long long> g_lines_gathered{0}, g_lines_window{0}, g_lines_loop{0};

constexpr size_t kContext = 511; // bytes of line context kept on each side (grab.cc:173: char[512])
constexpr off_t kOverlap = 0x1000; // consecutive chunks share 4 KiB (grab.cc:151)

const char kInvOn[] = "\33[7m", kInvOff[] = "\33[27m"; // grab.cc:66-67

// Files up to batch_max_ bytes (config key "batch", default 2 MiB, 0 = off) are scanned many to a launch over a segment
// table, instead of paying one copy + launch + readback each: the worker only QUEUES them (path + size from the walk's
// stat), the device's reader threads open, read and close them (gscan_submit_files).
constexpr size_t kBatchMaxFiles = 4096;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

// ------------------------------------------------------------------------------------
// One chunk's output.
//
// The reference repeats: m = leftmost match in [s, clen); print; s = m.end (+ rest of the
// printed line).  For the engine's pattern subset "a match starts at p" (p is a candidate)
// depends only on the minlen bytes at p.  The engine reports the start of every group of
// consecutive candidates, so the leftmost match from s is s itself when the window matches
// there, else the first reported start after s (a candidate that follows a non-candidate
// begins a group, so it is in the list); its end is the greedy tail extension.
// Rules kept from grab.cc:
//   :175      the loop runs while s + minlen < clen (strict)
//   :171,179  ovector holds ONE pair, so a match that sets a capturing group returns 0: the chunk ends there
//   :186      printed offset = file offset of the chunk + match start
//   :190-196  line context: back to a newline, to s, or 511 bytes; forward to a newline,
//             the chunk end, or 511 bytes
//   :204-207  -l without -O prints "matches" once per chunk
//   :209      s = match end + bytes printed after the match
//   :211      -s stops after one match
// ------------------------------------------------------------------------------------
void grab_report_chunk(const gscan_db *db, int minlen, unsigned flags, const char *path, const char *content,
                       size_t clen, long long off, const uint32_t *starts, size_t nstarts, std::string &out, const uint32_t *ext,
                       const uint32_t *ends, const uint8_t *gather)
{
    if (minlen < 0) return;
    gscan_cursor cur;
    cur.ready = 0;
    // "Match at offset N\n": the digits are written backwards into the tail of this buffer (dense outputs print millions
    // of these lines; snprintf was a third of the walk's time)
    char line[48];
    static const char kHead[] = "Match at offset ";
    const size_t plen = (flags & GRAB_PREFIX) ? strlen(path) : 0;
    auto put_head = [&](size_t m0) { // "path:" and "Match at offset N\n" (grab.cc:182-186)
        if (flags & GRAB_PREFIX) {
            out.append(path, plen);
            out += ':';
        }
        if (flags & GRAB_OFFSETS) {
            char *q = line + sizeof line;
            *--q = '\n';
            unsigned long long v = (unsigned long long)(off + (long long)m0);
            do {
                *--q = (char)('0' + v % 10);
                v /= 10;
            } while (v);
            q -= sizeof kHead - 1;
            memcpy(q, kHead, sizeof kHead - 1);
            out.append(q, (size_t)(line + sizeof line - q));
        }
    };
    // (dense outputs: one allocation instead of the doublings -- 21 bytes of text per offset line, a line of context otherwise)
    if (nstarts > 1024) out.reserve(out.size() + nstarts * (plen + ((flags & GRAB_NOLINE) ? 28 : 96)));
    size_t s = 0;
    unsigned long long n_gathered = 0, n_window = 0, n_loop = 0;
    struct Tally {
        unsigned long long &a, &b, &c;
        ~Tally()
        {
            if (a) g_lines_gathered += a;
            if (b) g_lines_window += b;
            if (c) g_lines_loop += c;
        }
    } tally{n_gathered, n_window, n_loop};
    // Line-printing modes with the device's line pass (k_lines): it has already decided which candidates the loop prints and
    // where their lines begin and end, and copied the text of those lines into one buffer -- for those records there is
    // nothing to search and nothing to read from the window, only to format.  Its verdicts hold while the restart position
    // is 0 or the newline that ended the last printed line.  A record it marks "ask the host" (a line that runs on past the
    // 511 bytes of printed context, a line start or a tail further than 4 KiB away) is handed to the reference's loop
    // below, which keeps going until a printed line has ended at its newline again: from there the device's verdicts apply
    // once more.  (A corpus with one line in ten thousand longer than 511 bytes has a hundred such records per 64 MiB
    // window: falling back for the REST of the window at the first of them, as round 2 did, left the pass 5 % of the lines.)
    // -O -l with every match's end from the device and nothing else asked for: the walk is two array reads per match and the
    // output one fixed-format line -- written through a local buffer, digits two at a time, without a call per match (a worker
    // formats BASELINE configs[2]'s 21 M lines per 8 GiB in this loop; at N GPUs that is N x 123 M lines per second)
    gscan_info info;
    if (gscan_db_info(db, &info) != GSCAN_OK) return;
    const bool resolve = info.resolve != 0; // the list is the device's list of MATCHES, each with its end (k_resolve)
    if (ends && (flags & GRAB_NOLINE) && (flags & GRAB_OFFSETS) && !(flags & GRAB_SINGLE) && (info.ends_ok || resolve)) {
        static const char kPairs[] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
                                     "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
        constexpr size_t kBuf = 64u << 10;
        char buf[kBuf];
        const size_t need = plen + 1 + (sizeof kHead - 1) + 20 + 1; // one line at most
        size_t w = 0, i = 0;
        bool fell_back = false;
        // a resolved list is good from `reach` bytes behind the restart position on; what could begin a match closer than that
        // is put to the host's matcher (gscan_next_resolved) -- after a match that is the byte or two behind its end, and
        // almost never one a match can begin with: one look at the table of first bytes
        uint8_t first[256];
        const size_t reach = resolve ? (size_t)info.reach : 0;
        memset(first, 1, sizeof first);
        const bool first_ok = reach && gscan_db_first(db, first) == 1;
        if (need < kBuf / 2) {
            // Does the restart at s have to LOOK at the text?  At the chunk's start only when offset 0 cannot be listed (windows with a
            // leading context byte); behind a match of the list the device has said so in bit 31 of its end (GSCAN_END_LOOK: it
            // looked at the two bytes around the end itself); behind a match the host's matcher found: yes.  Without a look the
            // walk reads two arrays and nothing of the chunk.
            bool look = reach && (info.has_context & 1);
            while (s + (size_t)minlen < clen) {
                size_t m0, m1;
                bool ask = false;
                size_t near_end = s;
                if (look) {
                    near_end = reach == 1 && s > 0 && (first[(unsigned char)content[s - 1]] & 2) ? s : std::min(clen, s + reach);
                    for (size_t q = s; q < near_end && !ask; q++) ask = !first_ok || (first[(unsigned char)content[q]] & 1);
                }
                if (!ask) {
                    while (i < nstarts && starts[i] < near_end) i++;
                    if (i >= nstarts) break;
                    if (resolve && ends[i] == GSCAN_END_CAPTURES) break; // a match that sets a capturing group: rc == 0, the chunk ends (grab.cc:179)
                    if (ends[i] == 0) { // this match is the host's to find (a tail longer than the device follows; the VM gave up)
                        if (!resolve) {
                            fell_back = true; // the loop below takes over from s
                            break;
                        }
                        ask = true;
                    } else {
                        m0 = starts[i];
                        m1 = ends[i] & ~GSCAN_END_LOOK;
                        look = resolve && (ends[i] & GSCAN_END_LOOK);
                        // (a look touches the byte or two around the restart position: a cache miss, the list says where the next ones are)
                        if (look && i + 8 < nstarts) __builtin_prefetch(content + (ends[i + 8] & ~GSCAN_END_LOOK) - 1);
                        i++;
                    }
                }
                if (ask) {
                    uint32_t b0 = 0, b1 = 0;
                    if (gscan_next_resolved(db, content, clen, starts, ends, nstarts, &cur, (uint32_t)s, &b0, &b1) != 1) break;
                    m0 = b0, m1 = b1;
                    look = reach != 0;
                }
                if (w + need > kBuf) {
                    out.append(buf, w);
                    w = 0;
                }
                if (flags & GRAB_PREFIX) {
                    memcpy(buf + w, path, plen);
                    w += plen;
                    buf[w++] = ':';
                }
                memcpy(buf + w, kHead, sizeof kHead - 1);
                w += sizeof kHead - 1;
                char dig[24];
                char *q = dig + sizeof dig;
                unsigned long long v = (unsigned long long)(off + (long long)m0);
                while (v >= 100) {
                    const unsigned r = (unsigned)(v % 100);
                    v /= 100;
                    q -= 2;
                    memcpy(q, kPairs + 2 * r, 2);
                }
                if (v >= 10) {
                    q -= 2;
                    memcpy(q, kPairs + 2 * (unsigned)v, 2);
                } else {
                    *--q = (char)('0' + v);
                }
                const size_t nd = (size_t)(dig + sizeof dig - q);
                memcpy(buf + w, q, nd);
                w += nd;
                buf[w++] = '\n';
                n_loop++;
                s = m1; // grab.cc:209 with a == 0
            }
            if (w) out.append(buf, w);
            if (!fell_back) return;
        }
    }
    size_t di = 0;                                         // next record of the device's pass
    bool device = ext && !(flags & GRAB_NOLINE);           // its verdicts apply at s
    // -O -l with the match ends from the device (k_ends): s is always 0 or a match end, the next match is the next listed
    // start and its end is in the list -- `content` is not looked at (a window that was never read stays unmapped pages)
    const bool listed = !resolve && ends && (flags & GRAB_NOLINE) && (flags & GRAB_OFFSETS);
    while (s + (size_t)minlen < clen) {
        if (device) {
            while (di < nstarts && (starts[di] < s || ext[4 * di] == 0)) di++; // behind s, or not printed (an earlier candidate in its line)
            if (di >= nstarts) return; // whatever follows the last listed start belongs to its group, i.e. to its line
            const uint32_t m1 = ext[4 * di], lb = ext[4 * di + 1], le = ext[4 * di + 2], goff = ext[4 * di + 3];
            if (lb != 0xffffffffu) {
                const size_t m0 = starts[di];
                put_head(m0);
                // the line [lb, le): from the gathered text when the device put it there, else from the window itself
                const bool got = gather && goff != 0xffffffffu;
                const char *line_text = got ? (const char *)gather + goff : content + lb;
                (got ? n_gathered : n_window)++;
                if (flags & GRAB_COLOR) {
                    out.append(line_text, m0 - lb);
                    out += kInvOn;
                    out.append(line_text + (m0 - lb), m1 - m0);
                    out += kInvOff;
                    out.append(line_text + (m1 - lb), le - m1);
                } else {
                    out.append(line_text, le - lb);
                }
                out += '\n';
                s = le; // grab.cc:209
                di++;
                if (flags & GRAB_SINGLE) return;
                continue;
            }
            device = false; // this one is the loop's; back to the device's verdicts when a printed line has ended at its newline
        }
        // rc = pcre_exec(d_pcreh, d_extra, start, end - start, 0, 0, ovector, 3)            (grab.cc:178)
        uint32_t b0 = 0, b1 = 0;
        int rc = resolve ? gscan_next_resolved(db, content, clen, starts, ends, nstarts, &cur, (uint32_t)s, &b0, &b1)
                 : listed ? gscan_next_listed(db, clen, starts, ends, nstarts, &cur, (uint32_t)s, &b0, &b1)
                          : -1;
        if (rc < 0) rc = gscan_next_match(db, content, clen, starts, nstarts, &cur, (uint32_t)s, &b0, &b1);
        if (rc != 1) break; // no match -- or one that sets a capturing group: 0 with ovector[3], same exit (grab.cc:179)
        const size_t m0 = b0, m1 = b1;

        put_head(m0);
        n_loop++;

        size_t tail = 0;
        if (!(flags & GRAB_NOLINE)) {
            const size_t lo = m0 - std::min(m0 - s, kContext);
            const void *nl = memrchr(content + lo, '\n', m0 - lo);
            const size_t line_begin = nl ? (size_t)((const char *)nl - content) + 1 : lo;
            const size_t hi = std::min(clen, m1 + kContext);
            const void *nr = memchr(content + m1, '\n', hi - m1);
            const size_t line_end = nr ? (size_t)((const char *)nr - content) : hi;
            out.append(content + line_begin, m0 - line_begin);
            if (flags & GRAB_COLOR) out += kInvOn;
            out.append(content + m0, m1 - m0);
            if (flags & GRAB_COLOR) out += kInvOff;
            out.append(content + m1, line_end - m1);
            out += '\n';
            tail = line_end - m1;
            if (ext && nr) device = true; // the printed line ended at its newline: every later record lies in a later line
        } else if (!(flags & GRAB_OFFSETS)) {
            out += "matches\n";
            break;
        }
        s = m1 + tail;
        if (flags & GRAB_SINGLE) break;
    }
}

// ------------------------------------------------------------------------------------

FileGrep::FileGrep() : uid_(geteuid())
{
    timing_ = getenv("GRAB_TIMING") != nullptr;
    // GRAB_BATCH_READ=worker (A/B runs): round 3's small-file path -- the worker itself read(2)s every file into a pinned
    // block and hands the block over (gscan_acquire + gscan_submit_segs)
    const char *br = getenv("GRAB_BATCH_READ");
    batch_by_worker_ = br && !strcmp(br, "worker");
    if (const char *bm = getenv("GRAB_BATCH_MIB")) batch_bytes_ = (size_t)std::max(1, atoi(bm)) << 20;
}

FileGrep::~FileGrep()
{
    flush();
    report_timing();
    for (gscan_ctx *c : ctxs_) gscan_close(c);
    if (db_) gscan_free(db_);
}

void FileGrep::report_timing()
{
    if (timing_ && !timing_reported_) {
        timing_reported_ = true;
        fprintf(stderr, "[grab timing] device %d: files %zu launches %zu bytes %zu | open %.3f s  read(batch) %.3f s  submit %.3f s  wait %.3f s  map+report %.3f s (of which writing it out, lock included: %.3f s; mapping small files that had something to print: %.3f s, %zu files)  close %.3f s\n",
                device_, t_files_, t_chunks_, t_bytes_, t_map_, t_read_, t_submit_, t_wait_, t_report_, t_emit_, t_text_, t_text_files_, t_unmap_);
        fprintf(stderr, "[grab timing] printed so far (all workers): %llu via the device's line pass + gathered text, %llu via the pass with text from the window, %llu by the host's loop\n",
                g_lines_gathered.load(), g_lines_window.load(), g_lines_loop.load());
        for (size_t k = 0; k < ctxs_.size(); k++) // what each device was handed: the work queue's balance (bench.py --mode e2e sums these)
            fprintf(stderr, "[grab bytes] device %d: %zu\n", ctx_dev_[k], ctx_bytes_[k]);
    }
}

void FileGrep::config(const std::map<std::string, size_t> &kv)
{
    auto has = [&](const char *k) { return kv.find(k) != kv.end(); };
    if (has("color")) color_ = true;
    if (has("noline")) noline_ = true;
    if (has("offsets")) offsets_ = true;
    if (has("single")) single_ = true;
    if (has("low_mem")) low_mem_ = true;
    if (has("literal")) literal_ = true;
    if (has("chunk_size")) chunk_size_ = kv.at("chunk_size");
    if (has("device")) device_ = (int)kv.at("device");
    if (has("out_fd")) out_fd_ = (int)kv.at("out_fd");
    if (has("batch")) batch_max_ = kv.at("batch");
    if (has("devices")) devices_ = std::max<size_t>(1, kv.at("devices"));
    if (has("silent_errors")) silent_errors_ = kv.at("silent_errors") != 0;
}

unsigned FileGrep::report_flags() const
{
    return (offsets_ ? GRAB_OFFSETS : 0u) | (noline_ ? GRAB_NOLINE : 0u) | (single_ ? GRAB_SINGLE : 0u) |
           ((recursive_ || show_path_) ? GRAB_PREFIX : 0u) | (color_ ? GRAB_COLOR : 0u);
}

int FileGrep::engine_option(const char *name, long value) { return ctx_ ? gscan_set_option(ctx_, name, value) : -1; }

// Replaces grab.cc:101-123.  Error strings for patterns PCRE itself rejects are the
// reference's; a valid pattern the engine cannot scan is a distinct, loud error.
int FileGrep::validate(const std::string &regex, bool literal, std::string &why, int *minlen, gscan_db **db_out)
{
    int want_minlen = 0;
    bool cross_check = false;
#ifdef GRAB_PCRE_VALIDATE
    if (!literal) {
        const char *msg = nullptr;
        int at = 0;
        pcre *re = pcre_compile(regex.c_str(), 0, &msg, &at, pcre_maketables());
        if (!re) {
            why = "FileGrep::prepare::pcre_compile error";
            return -1;
        }
        pcre_extra *study = pcre_study(re, PCRE_STUDY_JIT_COMPILE, &msg);
        if (!study) { // no JIT / no study data counts as failure in the reference (Q12)
            pcre_free(re);
            why = "FileGrep::prepare::pcre_study error";
            return -1;
        }
        want_minlen = 1;
        pcre_fullinfo(re, study, PCRE_INFO_MINLENGTH, &want_minlen);
        cross_check = true;
        pcre_free_study(study);
        pcre_free(re);
    }
#endif
    gscan_db *db = nullptr;
    char reason[160] = {0};
    int got = 1;
    const int rc = gscan_compile(regex.data(), regex.size(), (literal ? GSCAN_LITERAL : 0u) | (cross_check ? GSCAN_PCRE_CHECKED : 0u), &db, &got, reason, sizeof reason);
    if (rc == GSCAN_UNSUPPORTED) {
        why = std::string("FileGrep::prepare: pattern is outside the GPU engine's subset (") + reason + ")";
        return -2;
    }
    if (rc != GSCAN_OK && cross_check) {
        // libpcre has just compiled this text: whatever the engine's parser makes of it, it is not the reference's
        // "pcre_compile error" (and in -n mode that answer would mean a silent exit 0)
        why = std::string("FileGrep::prepare: pattern is outside the GPU engine's subset (the engine's parser: ") + reason + ")";
        return -2;
    }
    if (rc != GSCAN_OK) {
        why = "FileGrep::prepare::pcre_compile error";
        return -1;
    }
    if (cross_check && got != want_minlen) {
        gscan_free(db);
        why = "FileGrep::prepare: engine minlen " + std::to_string(got) + " disagrees with PCRE's " + std::to_string(want_minlen);
        return -2;
    }
    if (minlen) *minlen = got;
    if (db_out) *db_out = db;
    else gscan_free(db);
    return 0;
}

int FileGrep::prepare(const std::string &regex)
{
    if (db_) gscan_free(db_);
    db_ = nullptr;
    int got = 1;
    if (validate(regex, literal_, err_, &got, &db_) != 0) return -1;
    minlen_ = got;
    if (minlen_ < 0) return 0; // can match "": every file is skipped (Q2), nothing to open
    {
        gscan_info info;
        gscan_db_info(db_, &info);
        anchored_ = info.tier == GSCAN_TIER_ANCHORED;
        never_ = anchored_ && info.n_alts == 0; // assertions that can never hold (a\Ab): nothing matches anywhere
        // (a database the device resolves: the list is complete from the first offset that HAS a byte in front of it -- offset 0
        // of a chunk is the host's to test when the windows carry a leading context position, whatever the list says)
        resolve_ = info.resolve != 0;
        reach_ = (size_t)info.reach;
        context_ = resolve_ ? (info.has_context & 1) != 0 : info.has_context != 0;
        lines_ = info.lines_ok != 0;
        ends_ = info.ends_ok != 0;
        textfree_ = info.textfree != 0;
    }

    flush();
    for (gscan_ctx *c : ctxs_) gscan_close(c);
    ctxs_.clear();
    inflight_.clear();
    ctx_dev_.clear();
    ctx_bytes_.clear();
    ctx_ = nullptr;
    failed_ = false;
    return want_contexts(1);
}

// Contexts 1.. sit on the devices after `device` (mod the number there is): the windows of one multi-window file are
// dealt out to them round robin (the unit is (file, chunk index); the per-file reorder buffer is flight_,
// which retires in submission order).  Opened on demand -- a walk over small files never touches a second GPU.
int FileGrep::want_contexts(size_t n)
{
    while (ctxs_.size() < n) {
        const int ndev = std::max(1, gscan_device_count());
        const int dev = ctxs_.empty() ? device_ : (device_ + (int)ctxs_.size()) % ndev;
        gscan_ctx *c = nullptr;
        const int orc = gscan_open(dev, chunk_size_, &c);
        if (orc != GSCAN_OK) {
            if (!ctxs_.empty()) { // a further device that cannot be opened (busy, out of HBM, permissions): go on with the ones there are
                fprintf(stderr, "grab: HIP device %d cannot be opened (rc %d): scanning on %zu device(s)\n", dev, orc, ctxs_.size());
                devices_ = ctxs_.size();
                break;
            }
            err_ = "FileGrep::prepare::gscan_open: no usable HIP device " + std::to_string(dev) + " (rc " + std::to_string(orc) + ")";
            return -1;
        }
        // Line-printing modes: where the pattern allows it (one plain alternative, no newline in its classes) the device picks the
        // printed matches, finds their line extents and gathers the text of the printed lines (k_lines); the
        // host formats what comes back and never reads the window -- no page of a mapped file is faulted in for a line the
        // pass settled.  Measured end to end (16 GiB, -n 8, identifier regex, -O): the report stage 0.57 -> 0.24 s per worker,
        // wall clock -3 .. -24 % (profiles/r03_m_*); sparse outputs: neutral.  GRAB_LINE_PASS=0 keeps the host walk (A/B runs).
        {
            const char *lp = getenv("GRAB_LINE_PASS");
            if (lines_ && !noline_ && !(lp && atoi(lp) == 0)) gscan_set_option(c, "line_extents", 1);
        }
        // -O -l: the device measures every listed match (k_ends) and the walk prints offsets without touching the window
        // (grab.cc:175-213 with a == 0).  GRAB_NO_ENDS=1 keeps the host walk over the text (A/B runs).
        if (ends_ && noline_ && offsets_ && !getenv("GRAB_NO_ENDS")) gscan_set_option(c, "match_ends", 1);
        ctxs_.push_back(c);
        inflight_.push_back(0);
        ctx_dev_.push_back(dev);
        ctx_bytes_.push_back(0);
    }
    ctx_ = ctxs_[0];
    return 0;
}

int FileGrep::read_chunk(int fd, void *dst, size_t len, off_t at)
{
    size_t got = 0;
    while (got < len) {
        const ssize_t r = pread(fd, (char *)dst + got, len - got, at + (off_t)got);
        if (r > 0) {
            got += (size_t)r;
        } else if (r == 0) {
            err_ = "FileGrep::find::read: file shrank while reading";
            return -1;
        } else if (errno != EINTR) {
            err_ = std::string("FileGrep::find::read: ") + strerror(errno);
            return -1;
        }
    }
    return 0;
}

void FileGrep::emit(std::string &text)
{
    const double t_emit0 = timing_ ? now_s() : 0;
    struct Stop {
        FileGrep *g;
        double t0;
        ~Stop()
        {
            if (g->timing_) g->t_emit_ += now_s() - t0;
        }
    } stop{this, t_emit0};
    std::lock_guard<std::mutex> hold(g_out_lock);
    if (out_fd_ == 1) {
        std::cout << text; // same stream the reference prints to
    } else {
        for (size_t done = 0; done < text.size();) {
            const ssize_t w = write(out_fd_, text.data() + done, text.size() - done);
            if (w < 0 && errno != EINTR) break;
            if (w > 0) done += (size_t)w;
        }
    }
    text.clear();
}

// ------------------------------------------------------------------------------------
// In-flight work.  A Job is one engine chunk: a window of a big file (gscan_submit_fd), or a batch
// of small files packed into one pinned block (gscan_submit_segs).  Jobs retire in submission
// order, which is walk order, so the serial modes print exactly in the reference's order; they
// stay in flight ACROSS find() calls, so the host opens and reads file k+1 while file k is on the
// GPU.  flush() drains everything (end of a walk, explicit paths, destructor).
// ------------------------------------------------------------------------------------
struct FileGrep::FileRef { // what the report needs to know about a file after find() has returned
    std::string path;
    int oflags = 0;    // batched small files: how the readers open it (and the report, if the file has something to print)
    int fd = -1;       // big files: kept open until the last window has been printed (the report maps from it)
    bool done = false; // -s: a chunk of this file has printed, the rest of it stays silent (grab.cc:232-233)
    ~FileRef()
    {
        if (fd >= 0) close(fd);
    }
};

struct FileGrep::Job {
    int ctx = 0; // index into ctxs_
    // big-file window
    std::shared_ptr<FileRef> file;
    off_t off = 0;
    size_t len = 0;
    // batch: segment i is file i
    std::vector<std::shared_ptr<FileRef>> files;
    std::vector<gscan_seg> segs;
};

int FileGrep::retire_oldest(bool print)
{
    Job job = std::move(flight_.front());
    flight_.pop_front();
    gscan_ctx *const ctx = ctxs_[(size_t)job.ctx];
    inflight_[(size_t)job.ctx]--;
    const uint32_t *starts = nullptr;
    const size_t *first = nullptr;
    size_t nseg = 0;
    const void *bytes = nullptr;
    double t = timing_ ? now_s() : 0;
    const int rc = gscan_wait_segs(ctx, nullptr, &starts, &first, &nseg, &bytes);
    if (timing_) t_wait_ += now_s() - t, t = now_s();
    if (rc != GSCAN_OK) {
        // the error belongs to the file the job came from, which need not be the one find() is busy with: name it.  The
        // engine has dropped the chunk and freed its slot; a read error leaves the context usable, a device error does not.
        const std::string &whose = job.file ? job.file->path : (job.files.empty() ? std::string("?") : job.files.front()->path + " (+ the rest of its batch)");
        err_ = std::string(rc == GSCAN_EIO ? "FileGrep::find::read: " : "FileGrep::find::gscan_wait: ") + gscan_strerror(ctx) + " [" + whose + "]";
        if (rc != GSCAN_EIO) failed_ = true;
        return -1;
    }
    if (!print) return 0;
    const unsigned rflags = report_flags();
    std::string &text = report_buf_; // kept across jobs: dense outputs are tens of MB per window, no point in growing it anew each time
    text.clear();
    int status = 0;
    if (job.files.empty()) { // one window of a big file
        FileRef &f = *job.file;
        // no candidate start at all -> nothing is printed (a match at s = 0 would head a group and be in the list),
        // and the file's bytes are never touched by the host.  Not so for patterns with context (\b ^ $ ...): a match
        // at offset 0 or at the very end of the window is the host's to find
        if (!f.done && (first[nseg] > 0 || context_)) {
            void *map = mmap(nullptr, job.len, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, f.fd, job.off); // grab.cc:161 (MAP_POPULATE: no gain, measured)
            if (map == MAP_FAILED) {
                err_ = std::string("FileGrep::find::mmap: ") + strerror(errno);
                status = -1;
            } else {
                grab_report_chunk(db_, minlen_, rflags, f.path.c_str(), (const char *)map, job.len, (long long)job.off, starts, first[nseg], text,
                                  gscan_last_ext(ctx), gscan_last_ends(ctx), gscan_last_gather(ctx, nullptr));
                munmap(map, job.len); // grab.cc:215
                if (!text.empty()) {
                    emit(text);
                    if (single_) f.done = true;
                }
            }
        }
    } else { // a batch: every segment is a whole small file, i.e. its one and only chunk
        size_t nerr = 0;
        const int *ferr = gscan_last_file_errors(ctx, &nerr);
        const uint32_t *ext = gscan_last_ext(ctx), *ends = gscan_last_ends(ctx);
        const uint8_t *gather = gscan_last_gather(ctx, nullptr);
        for (size_t i = 0; i < job.files.size(); i++) {
            const char *path = job.files[i]->path.c_str();
            if (ferr && i < nerr && ferr[i]) { // the readers could not open or read it: this file's error, the batch goes on (grab.cc:267-268)
                err_ = ferr[i] == -1 ? std::string("FileGrep::find::read: file shrank while reading") : std::string("FileGrep::find::open: ") + strerror(ferr[i]);
                file_error(path);
                status = recursive_ ? status : -1;
                continue;
            }
            const size_t n_i = first[i + 1] - first[i];
            if (n_i == 0 && !context_) continue;
            const uint32_t *ext_i = ext ? ext + 4 * first[i] : nullptr, *ends_i = ends ? ends + first[i] : nullptr;
            const size_t len = job.segs[i].len;
            const char *chunk = bytes ? (const char *)bytes + job.segs[i].offset : nullptr;
            void *map = nullptr;
            int fd = -1;
            if (!chunk && report_needs_text(rflags, n_i, ext_i, ends_i, gather)) {
                // the bytes went through the readers' blocks and are gone: a file that has something to print is mapped like the
                // reference maps it (grab.cc:137-169) -- only the pages the report looks at are touched
                const double tm = timing_ ? now_s() : 0;
                fd = open(path, job.files[i]->oflags);
                map = fd >= 0 && len ? mmap(nullptr, len, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, fd, 0) : MAP_FAILED;
                if (timing_) t_text_ += now_s() - tm, t_text_files_++;
                if (map == MAP_FAILED) {
                    err_ = std::string(fd < 0 ? "FileGrep::find::open: " : "FileGrep::find::mmap: ") + strerror(errno);
                    if (fd >= 0) close(fd);
                    file_error(path);
                    status = recursive_ ? status : -1;
                    continue;
                }
                chunk = (const char *)map;
            }
            grab_report_chunk(db_, minlen_, rflags, path, chunk, len, 0, starts + first[i], n_i, text, ext_i, ends_i, gather);
            if (map) {
                munmap(map, len);
                close(fd);
            }
        }
        if (!text.empty()) emit(text); // one lock per batch; per-file output stays contiguous and in order
    }
    if (timing_) t_report_ += now_s() - t;
    return status;
}

int FileGrep::submit_batch()
{
    if (batch_files_.empty()) return 0;
    Job job;
    job.files.swap(batch_files_);
    job.segs.swap(batch_segs_);
    const void *buf = batch_buf_;
    batch_buf_ = nullptr;
    batch_used_ = 0;
    double t = timing_ ? now_s() : 0;
    // A batch that cannot be handed over is not one file's error: its files have left batch_files_ and would never be scanned
    // or reported.  The instance is marked failed -- every later find() fails, and the -n workers (which ignore per-file
    // errors, main.cc:97) report it when they are done: failed() / why().
    if (batch_by_worker_) {
        if (gscan_submit_segs(ctx_, db_, buf, job.segs.data(), job.segs.size(), 0) != GSCAN_OK) {
            err_ = std::string("FileGrep::find::gscan_submit_segs: ") + gscan_strerror(ctx_) + " [" + job.files.front()->path + " (+ the rest of its batch)]";
            failed_ = true;
            return -1;
        }
    } else {
        if (make_room(0) < 0) { // a slot has to be free for the batch (make_room fails on a device error only: failed_ is set)
            err_ += " [then " + job.files.front()->path + " (+ the rest of its batch) could not be handed over]";
            return -1;
        }
        std::vector<gscan_file> list(job.files.size());
        for (size_t i = 0; i < job.files.size(); i++) list[i] = gscan_file{job.files[i]->path.c_str(), -1, job.files[i]->oflags, job.segs[i].len};
        t = timing_ ? now_s() : 0;
        if (gscan_submit_files(ctx_, db_, list.data(), list.size(), 0) != GSCAN_OK) {
            err_ = std::string("FileGrep::find::gscan_submit_files: ") + gscan_strerror(ctx_) + " [" + job.files.front()->path + " (+ the rest of its batch)]";
            failed_ = true;
            return -1;
        }
    }
    if (timing_) t_submit_ += now_s() - t, t_chunks_++;
    inflight_[0]++;
    flight_.push_back(std::move(job));
    return 0;
}

// Jobs retire in submission order, so "a free slot in context k" means: retire from the front until k has one.
int FileGrep::make_room(int ctx)
{
    while (inflight_[(size_t)ctx] >= GSCAN_SLOTS)
        if (retire_oldest(true) < 0) deferred_error();
    return failed_ ? -1 : 0;
}

// A window of an earlier file failed while a later one was being handed over: say so against the file it belongs to (err_
// names it) and carry on with the current one, as the reference reports per file and keeps walking (grab.cc:267-268).
void FileGrep::deferred_error()
{
    if (silent_errors_) return;
    if (recursive_) std::cerr << err_ << std::endl;
    else deferred_ = err_; // explicit paths: find(path) returns it
}

void FileGrep::idle()
{
    if (batch_by_worker_ || batch_files_.empty() || !ctx_) return;
    if (inflight_[0] == 0 || batch_used_ >= (size_t(8) << 20)) (void)submit_batch();
}

// Everything handed over so far is scanned and printed when this returns.
int FileGrep::flush()
{
    int status = 0;
    if (!ctx_) return 0;
    if (submit_batch() < 0) status = -1;
    while (!flight_.empty())
        if (retire_oldest(status == 0) < 0) status = -1;
    return status;
}

// A small file joins the open batch.  Nothing is opened or read here: the batch is a list of names that the device's reader
// threads work off (gscan_submit_files) while this thread is already queueing the next one -- the reference's workers open,
// map and scan each file themselves (main.cc:86-100, grab.cc:137-169); here that is the readers' job, eight to a device.
int FileGrep::batch_add(const char *path, const struct stat *st, int oflags)
{
    const size_t size = (size_t)st->st_size;
    const size_t cap = std::min(batch_bytes_, chunk_size_);
    const size_t at = (batch_used_ + 15) & ~size_t(15);
    if (!batch_files_.empty() && (at + size > cap || batch_files_.size() >= kBatchMaxFiles) && submit_batch() < 0) return -1;
    const size_t off = (batch_used_ + 15) & ~size_t(15);
    auto ref = std::make_shared<FileRef>();
    ref->path = path;
    ref->oflags = oflags;
    batch_files_.push_back(std::move(ref));
    batch_segs_.push_back({(uint64_t)off, (uint32_t)size, 0});
    batch_used_ = off + size;
    ctx_bytes_[0] += size;
    if (timing_) t_bytes_ += size, t_files_++;
    return 0;
}

// GRAB_BATCH_READ=worker: the worker read(2)s the file into the engine's pinned block itself (round 3's path, kept for A/B runs)
int FileGrep::batch_add_read(const char *path, int fd, size_t size)
{
    const size_t cap = gscan_block_size();
    const size_t at = (batch_used_ + 15) & ~size_t(15);
    if (batch_buf_ && (at + size > cap || batch_files_.size() >= kBatchMaxFiles) && submit_batch() < 0) return -1;
    if (!batch_buf_) {
        // a slot has to be free for the new batch
        if (make_room(0) < 0) return -1;
        void *buf = nullptr;
        if (gscan_acquire(ctx_, cap, &buf) != GSCAN_OK) {
            err_ = std::string("FileGrep::find::gscan_acquire: ") + gscan_strerror(ctx_);
            return -1;
        }
        batch_buf_ = buf;
        batch_used_ = 0;
    }
    const size_t off = (batch_used_ + 15) & ~size_t(15);
    double t = timing_ ? now_s() : 0;
    if (read_chunk(fd, (char *)batch_buf_ + off, size, 0) < 0) return -1;
    if (timing_) t_read_ += now_s() - t, t_bytes_ += size;
    ctx_bytes_[0] += size;
    auto ref = std::make_shared<FileRef>();
    ref->path = path;
    batch_files_.push_back(std::move(ref));
    batch_segs_.push_back({(uint64_t)off, (uint32_t)size, 0});
    batch_used_ = off + size;
    return 0;
}

// Does the walk over one chunk's records look at the chunk's bytes?  Not when every record it will print was settled on the
// device: -O -l with the match ends (k_ends), the line modes with every printed line gathered (k_lines).
bool FileGrep::report_needs_text(unsigned rflags, size_t n, const uint32_t *ext, const uint32_t *ends, const uint8_t *gather) const
{
    if (context_) return true; // (matches at the restart position / chunk end are the host's to find, list or no list)
    if (n == 0) return false;
    if (resolve_) { // the list holds the matches and their ends: the text is needed for printed lines, for what lies within `reach` of a restart position, and for the records the device left to the host
        if (!(rflags & GRAB_NOLINE) || !ends) return true;
        for (size_t i = 0; i < n; i++) // (a record the host has to decide, or a match behind whose end it has to look: GSCAN_END_LOOK)
            if (ends[i] == GSCAN_END_ASK || (ends[i] != GSCAN_END_CAPTURES && (ends[i] & GSCAN_END_LOOK))) return true;
        return false;
    }
    if (textfree_ && (rflags & GRAB_NOLINE)) return false; // a fixed-length pattern all of whose candidates are listed: gscan_next_match walks the list alone
    if (ends && (rflags & GRAB_NOLINE) && (rflags & GRAB_OFFSETS)) {
        for (size_t i = 0; i < n; i++)
            if (ends[i] == 0) return true;
        return false;
    }
    if (ext && gather && !(rflags & GRAB_NOLINE)) {
        for (size_t i = 0; i < n; i++)
            if (ext[4 * i] != 0 && (ext[4 * i + 1] == 0xffffffffu || ext[4 * i + 3] == 0xffffffffu)) return true;
        return false;
    }
    return true;
}

// An error that belongs to ONE file of a batch, found long after find() returned for it: said the way the walk says it
// (grab.cc:267-268 prints "path: why" and goes on; the -n workers ignore per-file errors, main.cc:97), or kept for
// find(path) to return.
void FileGrep::file_error(const char *path)
{
    if (silent_errors_) return;
    if (recursive_) std::cerr << path << ": " << err_ << std::endl;
    else deferred_ = err_;
}

// Replaces grab.cc:131-239.  Geometry is the reference's: windows of chunk_size bytes that
// advance by chunk_size - 4 KiB, files shorter than minlen skipped unopened, per-chunk output
// flushed atomically and in file order, -s ends the file after the first chunk that printed.
// Where the reference maps a window and runs pcre_exec over it, a window goes to the engine by
// file descriptor (reader threads -> pinned blocks -> HBM) and is mapped only if it has matches to
// print; small files are packed into batches.  Work stays in flight when find() returns.
int FileGrep::find(const char *path, const struct stat *st, int /*typeflag*/)
{
    const off_t size = st->st_size;
    if ((size_t)minlen_ > (size_t)size) return 0;
    if (failed_) return -1; // a device error earlier on (why() still says which): nothing can be scanned any more, and no file may pass for clean

    double t0 = timing_ ? now_s() : 0;
    int oflags = O_RDONLY | O_NOCTTY;
#ifdef __linux__
    if (st->st_uid == uid_ || uid_ == 0) oflags |= O_NOATIME; // do not dirty the inode (grab.cc:139-143)
#endif
    const bool small = !anchored_ && (size_t)size <= batch_max_ && (size_t)size <= gscan_block_size() && (size_t)size <= chunk_size_;
    if (small && !batch_by_worker_) return batch_add(path, st, oflags); // queued by name: the device's readers open and read it
    const int fd = open(path, oflags);
    if (fd < 0) {
        err_ = std::string("FileGrep::find::open: ") + strerror(errno);
        return -1;
    }
    if (timing_) t_map_ += now_s() - t0, t_files_++;

    int status = 0;
    bool keep_fd = false;
    if (anchored_) {
        // ^foo, foo$ ...: a match can only start at a restart position or end at the chunk end; there is nothing to
        // scan (pcre_exec does not scan for an anchored pattern either): map each window and walk it with an empty list
        if (flush() < 0) status = -1;
        const unsigned rflags = report_flags();
        std::string text;
        const off_t stride = (off_t)chunk_size_ - kOverlap;
        for (off_t off = 0; off < size && status == 0 && !never_; off += stride) {
            const size_t len = (size_t)std::min<off_t>(size - off, (off_t)chunk_size_);
            void *map = mmap(nullptr, len, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, fd, off);
            if (map == MAP_FAILED) {
                err_ = std::string("FileGrep::find::mmap: ") + strerror(errno);
                status = -1;
                break;
            }
            grab_report_chunk(db_, minlen_, rflags, path, (const char *)map, len, (long long)off, nullptr, 0, text);
            munmap(map, len);
            if (!text.empty()) {
                emit(text);
                if (single_) break;
            }
        }
    } else if (small) {
        status = batch_add_read(path, fd, (size_t)size);
    } else {
        if (submit_batch() < 0) status = -1; // keeps the submission order == walk order
        auto ref = std::make_shared<FileRef>();
        ref->path = path;
        ref->fd = fd;
        keep_fd = true;
        const off_t stride = (off_t)chunk_size_ - kOverlap;
        // -s needs to know whether a window printed before the next one is worth reading: everything older is retired
        // first, then one window at a time (grab.cc:232-233)
        if (single_)
            while (!flight_.empty())
                if (retire_oldest(true) < 0) deferred_error();
        // a file of several windows is dealt out over the configured devices (contexts opened on first need)
        const size_t nwin = (size_t)((size + stride - 1) / stride);
        size_t use = single_ ? 1 : std::min(devices_, nwin);
        if (status == 0 && use > ctxs_.size() && want_contexts(use) < 0) status = -1;
        use = std::max<size_t>(1, std::min(use, ctxs_.size())); // (a further device that could not be opened is left out: want_contexts)
        for (off_t off = 0; off < size && status == 0 && !ref->done; off += stride) {
            const size_t len = (size_t)std::min<off_t>(size - off, (off_t)chunk_size_);
            const int k = use > 1 ? (int)(next_ctx_++ % use) : 0;
            if (make_room(k) < 0) {
                status = -1;
                break;
            }
            double t = timing_ ? now_s() : 0;
            // (mapping the window, registering the mapping and DMA-ing it in place -- no copy on the host at all -- was built and
            // measured in round 2: registration serialises inside the runtime, 55 ms per GiB per thread; taken out again)
            const int src = gscan_submit_fd(ctxs_[(size_t)k], db_, fd, (long long)off, len, (uint64_t)off);
            if (src != GSCAN_OK) {
                err_ = std::string("FileGrep::find::read: ") + gscan_strerror(ctxs_[(size_t)k]);
                status = -1;
                break;
            }
            if (timing_) t_submit_ += now_s() - t, t_chunks_++, t_bytes_ += len;
            ctx_bytes_[(size_t)k] += len;
            Job job;
            job.ctx = k;
            job.file = ref;
            job.off = off;
            job.len = len;
            inflight_[(size_t)k]++;
            flight_.push_back(std::move(job));
            if (single_ && retire_oldest(true) < 0) status = -1; // (its own window: the error is this file's)
        }
    }
    t0 = timing_ ? now_s() : 0;
    if (!keep_fd) close(fd);
    if (timing_) t_unmap_ += now_s() - t0;
    return status;
}

int FileGrep::find(const std::string &path)
{
    struct stat st;
    if (stat(path.c_str(), &st) < 0) {
        err_ = std::string("FileGrep::find::stat: ") + strerror(errno);
        return -1;
    }
    if (S_ISREG(st.st_mode)) {
        const int rc = find(path.c_str(), &st, FTW_F);
        if (flush() < 0) { // an explicit path is done when this returns, like the reference's
            deferred_.clear(); // (err_ says what failed: a per-file error noted on the way must not be handed to the NEXT find())
            return -1;
        }
        if (!deferred_.empty()) { // a window of an earlier explicit path failed while this one was handed over
            err_.swap(deferred_);
            deferred_.clear();
            return -1;
        }
        return rc;
    }
    if (S_ISDIR(st.st_mode)) std::cerr << "Clever boy! Want recursion? Add -R!\n"; // grab.cc:253-254, rc stays 0
    return 0;
}

namespace {
thread_local FileGrep *t_walker = nullptr; // the instance nftw()'s callback reports to

int on_entry(const char *path, const struct stat *st, int type, struct FTW *)
{
    // regular files only; symlinks arrive as FTW_SL under FTW_PHYS and are skipped (grab.cc:265-266)
    if (type == FTW_F && S_ISREG(st->st_mode) && t_walker->find(path, st, type) < 0)
        std::cerr << path << ": " << t_walker->why() << std::endl; // report and keep walking (grab.cc:267-268)
    return 0;
}
} // namespace

int FileGrep::find_recursive(const std::string &path)
{
    recursive_ = true;
    t_walker = this;
    const int rc = nftw(path.c_str(), on_entry, 1024, FTW_PHYS); // grab.cc:278
    return flush() < 0 ? -1 : rc;
}
