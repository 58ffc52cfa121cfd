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
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <deque>
#include <iostream>
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

constexpr size_t kContext = 511; // bytes of line context kept on each side (grab.cc:173: char[512])
constexpr off_t kOverlap = 0x1000; // consecutive chunks share 4 KiB (grab.cc:151)

const char kInvOn[] = "\33[7m", kInvOff[] = "\33[27m"; // grab.cc:66-67

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
//   :186      printed offset = file offset of the chunk + match start
//   :190-196  line context: back to a newline, to s, or 511 bytes; forward to a newline,
//             the chunk end, or 511 bytes
//   :204-207  -l without -O prints "matches" once per chunk
//   :209      s = match end + bytes printed after the match
//   :211      -s stops after one match
// ------------------------------------------------------------------------------------
void grab_report_chunk(const gscan_db *db, int minlen, unsigned flags, const char *path, const char *content,
                       size_t clen, long long off, const uint32_t *starts, size_t nstarts, std::string &out)
{
    if (minlen < 0) return;
    const uint32_t *cur = starts, *const last = starts + nstarts;
    char line[64];
    size_t s = 0;
    while (s + (size_t)minlen < clen) {
        size_t m0 = s;
        if (!gscan_match_at(db, content, clen, (uint32_t)s)) {
            cur = std::upper_bound(cur, last, s, [](size_t key, uint32_t v) { return key < (size_t)v; });
            if (cur == last) break;
            m0 = *cur;
        }
        const size_t m1 = gscan_match_end(db, content, clen, (uint32_t)m0);

        if (flags & GRAB_PREFIX) {
            out += path;
            out += ':';
        }
        if (flags & GRAB_OFFSETS) out.append(line, (size_t)snprintf(line, sizeof line, "Match at offset %lld\n", off + (long long)m0));

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
        } else if (!(flags & GRAB_OFFSETS)) {
            out += "matches\n";
            break;
        }
        s = m1 + tail;
        if (flags & GRAB_SINGLE) break;
    }
}

// ------------------------------------------------------------------------------------

FileGrep::FileGrep() : uid_(geteuid()) {}

FileGrep::~FileGrep()
{
    if (ctx_) gscan_close(ctx_);
    if (db_) gscan_free(db_);
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
}

unsigned FileGrep::report_flags() const
{
    return (offsets_ ? GRAB_OFFSETS : 0u) | (noline_ ? GRAB_NOLINE : 0u) | (single_ ? GRAB_SINGLE : 0u) |
           ((recursive_ || show_path_) ? GRAB_PREFIX : 0u) | (color_ ? GRAB_COLOR : 0u);
}

int FileGrep::engine_option(const char *name, long value) { return ctx_ ? gscan_set_option(ctx_, name, value) : -1; }

// Replaces grab.cc:101-123.  Error strings for patterns PCRE itself rejects are the
// reference's; a valid pattern the engine cannot scan is a distinct, loud error.
int FileGrep::prepare(const std::string &regex)
{
    int want_minlen = 0;
    bool cross_check = false;
#ifdef GRAB_PCRE_VALIDATE
    if (!literal_) {
        const char *msg = nullptr;
        int at = 0;
        pcre *re = pcre_compile(regex.c_str(), 0, &msg, &at, pcre_maketables());
        if (!re) {
            err_ = "FileGrep::prepare::pcre_compile error";
            return -1;
        }
        pcre_extra *study = pcre_study(re, PCRE_STUDY_JIT_COMPILE, &msg);
        if (!study) { // no JIT / no study data counts as failure in the reference (Q12)
            pcre_free(re);
            err_ = "FileGrep::prepare::pcre_study error";
            return -1;
        }
        want_minlen = 1;
        pcre_fullinfo(re, study, PCRE_INFO_MINLENGTH, &want_minlen);
        cross_check = true;
        pcre_free_study(study);
        pcre_free(re);
    }
#endif
    if (db_) gscan_free(db_);
    db_ = nullptr;
    char reason[160] = {0};
    int got = 1;
    const int rc = gscan_compile(regex.data(), regex.size(), literal_ ? GSCAN_LITERAL : 0u, &db_, &got, reason, sizeof reason);
    if (rc == GSCAN_UNSUPPORTED) {
        err_ = std::string("FileGrep::prepare: pattern is outside the GPU engine's subset (") + reason + ")";
        return -1;
    }
    if (rc != GSCAN_OK) {
        err_ = "FileGrep::prepare::pcre_compile error";
        return -1;
    }
    if (cross_check && got != want_minlen) {
        err_ = "FileGrep::prepare: engine minlen " + std::to_string(got) + " disagrees with PCRE's " + std::to_string(want_minlen);
        return -1;
    }
    minlen_ = got;
    if (minlen_ < 0) return 0; // can match "": every file is skipped (Q2), nothing to open

    if (ctx_) gscan_close(ctx_);
    ctx_ = nullptr;
    const int orc = gscan_open(device_, chunk_size_, &ctx_);
    if (orc != GSCAN_OK) {
        err_ = "FileGrep::prepare::gscan_open: no usable HIP device " + std::to_string(device_) + " (rc " + std::to_string(orc) + ")";
        return -1;
    }
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

// Replaces grab.cc:131-239.  Geometry is the reference's: windows of chunk_size bytes that
// advance by chunk_size - 4 KiB, files shorter than minlen skipped unopened, per-chunk
// output flushed atomically and in file order, -s ends the file after the first chunk that
// printed.  Up to GSCAN_SLOTS chunks are in flight: while chunk k is on the GPU the host
// reads chunk k+1, and while chunk k is printed chunk k+1 is being copied and scanned.
int FileGrep::find(const char *path, const struct stat *st, int /*typeflag*/)
{
    const off_t size = st->st_size;
    if ((size_t)minlen_ > (size_t)size) return 0;

    int oflags = O_RDONLY | O_NOCTTY;
#ifdef __linux__
    if (st->st_uid == uid_ || uid_ == 0) oflags |= O_NOATIME; // do not dirty the inode (grab.cc:139-143)
#endif
    const int fd = open(path, oflags);
    if (fd < 0) {
        err_ = std::string("FileGrep::find::open: ") + strerror(errno);
        return -1;
    }
    if (size > 4 * 0x1000 && !single_) posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);

    struct InFlight {
        off_t off;
        size_t len;
    };
    std::deque<InFlight> flight;
    const unsigned rflags = report_flags();
    std::string text;
    bool printed_and_single = false;
    int status = 0;

    auto retire_oldest = [&](bool print) {
        const uint32_t *starts = nullptr;
        size_t n = 0;
        const void *bytes = nullptr;
        const InFlight job = flight.front();
        flight.pop_front();
        if (gscan_wait(ctx_, nullptr, &starts, &n, &bytes) != GSCAN_OK) {
            err_ = std::string("FileGrep::find::gscan_wait: ") + gscan_strerror(ctx_);
            status = -1;
            return;
        }
        if (!print) return;
        grab_report_chunk(db_, minlen_, rflags, path, (const char *)bytes, job.len, (long long)job.off, starts, n, text);
        if (!text.empty()) {
            emit(text);
            if (single_) printed_and_single = true;
        }
    };

    const off_t stride = (off_t)chunk_size_ - kOverlap;
    for (off_t off = 0; off < size && status == 0 && !printed_and_single; off += stride) {
        const size_t len = (size_t)std::min<off_t>(size - off, (off_t)chunk_size_);
        void *pinned = nullptr;
        if (gscan_acquire(ctx_, len, &pinned) != GSCAN_OK) {
            err_ = std::string("FileGrep::find::gscan_acquire: ") + gscan_strerror(ctx_);
            status = -1;
            break;
        }
        if (read_chunk(fd, pinned, len, off) < 0) {
            status = -1;
            break;
        }
        if (gscan_submit(ctx_, db_, pinned, len, (uint64_t)off) != GSCAN_OK) {
            err_ = std::string("FileGrep::find::gscan_submit: ") + gscan_strerror(ctx_);
            status = -1;
            break;
        }
        flight.push_back({off, len});
        if (flight.size() == GSCAN_SLOTS) retire_oldest(true);
    }
    while (!flight.empty()) retire_oldest(status == 0 && !printed_and_single);

    close(fd);
    return status;
}

int FileGrep::find(const std::string &path)
{
    struct stat st;
    if (stat(path.c_str(), &st) < 0) {
        err_ = std::string("FileGrep::find::stat: ") + strerror(errno);
        return -1;
    }
    if (S_ISREG(st.st_mode)) return find(path.c_str(), &st, FTW_F);
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
    return nftw(path.c_str(), on_entry, 1024, FTW_PHYS); // grab.cc:278
}
