// grab_cli.cc -- the `grab` command line over the MI355X scan engine.
//
// Command-line surface of the reference (/root/reference/src/main.cc:103-266): the same
// getopt letters "Rrn:IOlsL", the same chunk-size rules (-L halves it down to 32 MiB,
// -n > 1 quarters it), -n needs -r, usage goes to stdout with status 1, errors print
// why() and give status 255, success is status 0 whether or not anything matched.
// Additions: -S (pattern is a literal string; README:26 documents it for the reference's
// other branch), -H and -2 accepted and ignored (there is one engine).
//
// -n N is organised differently from the reference on purpose: there the tree walk runs
// to completion on one thread before N pthreads stripe the file list statically
// (main.cc:175-216); here the walk is a producer that feeds a bounded queue while N
// workers are already scanning, each worker owning a FileGrep bound to HIP device
// (i mod #devices) -- files are independent units, so an 8-GPU node is used with no
// collective at all.  Output order across files is unspecified, exactly as in the
// reference's threaded mode (its own check sorts: README:206-216).
#include <ftw.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "filegrep.h"

namespace {

// GRAB_TIMING=1: wall-clock marks of the run on stderr (startup cost is a large part of a sub-second scan)
const auto g_t0 = std::chrono::steady_clock::now();
void mark(const char *what)
{
    static const bool on = getenv("GRAB_TIMING") != nullptr;
    if (on) fprintf(stderr, "[grab timing] +%.3f s %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count(), what);
}

struct Options {
    std::map<std::string, size_t> cfg; // handed to FileGrep::config
    bool recursive = false;
    int workers = 0;
    std::string regex;
    std::vector<std::string> paths;
};

[[noreturn]] void usage(const char *argv0)
{
    std::cout << "Usage: " << argv0 << " [-rR] [-I] [-O] [-L] [-l] [-s] [-n <cores>] <regex> <path>\n";
    exit(1);
}

Options parse(int argc, char **argv)
{
    Options o;
    size_t chunk = size_t(1) << 30;
    for (int c; (c = getopt(argc, argv, "Rrn:IOlsLHS2")) != -1;) {
        switch (c) {
        case 'r': case 'R': o.recursive = true; break;
        case 's': o.cfg["single"] = 1; break;
        case 'O': o.cfg["offsets"] = 1; break;
        case 'l': o.cfg["noline"] = 1; break;
        case 'L':
            o.cfg["low_mem"] = 1;
            chunk = std::max(chunk >> 1, size_t(1) << 25);
            break;
        case 'I':
            if (isatty(1)) o.cfg["color"] = 1;
            break;
        case 'n': o.workers = atoi(optarg); break;
        case 'S': o.cfg["literal"] = 1; break;
        case 'H': case '2': break;
        default: usage(argv[0]);
        }
    }
    if (argc < optind + 2) usage(argv[0]);
    o.regex = argv[optind++];
    while (optind < argc) o.paths.push_back(argv[optind++]);
    if (o.workers > 1) chunk >>= 2;
    o.cfg["chunk_size"] = chunk;
    if (const char *b = getenv("GRAB_BATCH")) o.cfg["batch"] = size_t(atoll(b)); // largest file that is batched (0 = off)
    return o;
}

// ---- walker -> workers queue ----
struct Job {
    std::string path;
    struct stat st;
};

class JobQueue {
public:
    void push(Job &&j)
    {
        std::unique_lock<std::mutex> lk(m_);
        room_.wait(lk, [&] { return q_.size() < kMax; });
        q_.push_back(std::move(j));
        ready_.notify_one();
    }
    bool pop(Job &out)
    {
        std::unique_lock<std::mutex> lk(m_);
        ready_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        room_.notify_one();
        return true;
    }
    void close()
    {
        std::lock_guard<std::mutex> lk(m_);
        closed_ = true;
        ready_.notify_all();
    }

private:
    static constexpr size_t kMax = 1 << 16;
    std::mutex m_;
    std::condition_variable ready_, room_;
    std::deque<Job> q_;
    bool closed_ = false;
};

JobQueue *g_queue = nullptr;

int enqueue_entry(const char *path, const struct stat *st, int type, struct FTW *)
{
    if (type == FTW_F && S_ISREG(st->st_mode)) g_queue->push(Job{path, *st});
    return 0;
}

int run_workers(const Options &o)
{
    if (!o.recursive) {
        std::cerr << "Multicore support only for recursive grabs.\n";
        return -1;
    }
    const int ndev = std::max(1, gscan_device_count());
    std::vector<std::unique_ptr<FileGrep>> greps;
    for (int i = 0; i < o.workers; i++) {
        auto cfg = o.cfg;
        cfg["device"] = size_t(i % ndev);
        greps.emplace_back(new FileGrep);
        greps.back()->config(cfg);
        greps.back()->recurse();
        // The reference ignores prepare()'s result here (main.cc:198) and then silently matches
        // nothing; a scan that cannot run is reported instead.
        if (greps.back()->prepare(o.regex) < 0) {
            std::cerr << greps.back()->why() << std::endl;
            return -1;
        }
    }
    mark("contexts open, pattern compiled");
    JobQueue queue;
    g_queue = &queue;
    std::vector<std::thread> pool;
    for (int i = 0; i < o.workers; i++) {
        pool.emplace_back([&queue, g = greps[i].get()] {
            for (Job j; queue.pop(j);) g->find(j.path.c_str(), &j.st, FTW_F); // per-file errors ignored (main.cc:97)
            g->flush(); // what is still in flight or waiting in a half-filled batch
        });
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(i, &one); // worker i on CPU i, as main.cc:200-215; more workers than CPUs is fatal there too
        if (int r = pthread_setaffinity_np(pool.back().native_handle(), sizeof one, &one)) {
            std::cerr << "pthread_setaffinity_np:" << strerror(r) << " (more threads than cores?)" << std::endl;
            std::cout.flush();
            _exit(255);
        }
    }
    nftw(o.paths[0].c_str(), enqueue_entry, 1024, FTW_PHYS);
    queue.close();
    mark("walk done");
    for (auto &t : pool) t.join();
    mark("workers joined");
    greps.clear();
    mark("contexts closed");
    return 0;
}

int run_serial(const Options &o)
{
    FileGrep grep;
    auto cfg = o.cfg;
    if (const char *dev = getenv("GRAB_DEVICE")) cfg["device"] = size_t(atoi(dev));
    grep.config(cfg);
    if (grep.prepare(o.regex) < 0) {
        std::cerr << grep.why() << std::endl;
        return -1;
    }
    mark("context open, pattern compiled");
    if (o.recursive) {
        if (grep.find_recursive(o.paths[0]) < 0) {
            std::cerr << grep.why() << std::endl;
            return -1;
        }
        return 0;
    }
    if (o.paths.size() > 1) grep.show_path(true); // several paths: prefix every record (main.cc:249-250)
    for (const std::string &p : o.paths)
        if (grep.find(p) < 0) {
            std::cerr << grep.why() << std::endl;
            return -1;
        }
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    mark("main");
    const Options o = parse(argc, argv);
    const int rc = o.workers > 1 ? run_workers(o) : run_serial(o);
    mark("scan done, contexts closed");
    std::cout.flush();
    // GRAB_DIAG=1: say when the host matcher abandoned attempts at its resource limit (each ends its chunk silently, as a
    // pcre_exec error does in the reference: src/grab.cc:179); the differential tests skip such inputs
    if (getenv("GRAB_DIAG") && gscan_resource_errors())
        fprintf(stderr, "grab: %llu match attempts abandoned at the matcher's resource limit\n", (unsigned long long)gscan_resource_errors());
    return rc; // -1 -> exit status 255, like the reference's `return -1` from main
}
