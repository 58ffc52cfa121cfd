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
// (main.cc:175-216); here a multi-threaded walk (walk.h) feeds a bounded queue while N
// workers are already scanning, each worker owning a FileGrep bound to HIP device
// (i mod #devices) and running on that device's NUMA node -- files are independent units,
// so an 8-GPU node is used with no collective at all.  Output order across files is
// unspecified, exactly as in the reference's threaded mode (its own check sorts:
// README:206-216).  Without -n, the windows of ONE multi-window file are dealt out over
// all devices (FileGrep "devices"; GRAB_DEVICES overrides the count), printed in order.
#include <ftw.h>
#include <pthread.h>
#include <sched.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <deque>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fstream>

#include "filegrep.h"
#include "placement.h"
#include "walk.h"

namespace {

// GRAB_TIMING=1: wall-clock marks of the run on stderr (startup cost is a large part of a sub-second scan)
const auto g_t0 = std::chrono::steady_clock::now();
void mark(const char *what)
{
    static const bool on = getenv("GRAB_TIMING") != nullptr;
    if (on) fprintf(stderr, "[grab timing] +%.3f s %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count(), what);
}

void mark_memory(const char *when)
{
    if (!getenv("GRAB_TIMING")) return;
    std::ifstream st("/proc/self/status");
    std::string line, all;
    while (std::getline(st, line))
        if (!line.compare(0, 5, "VmRSS") || !line.compare(0, 5, "VmHWM") || !line.compare(0, 7, "RssAnon") || !line.compare(0, 8, "RssShmem") || !line.compare(0, 7, "RssFile")) {
            for (char &ch : line)
                if (ch == '\t') ch = ' ';
            all += line + "; ";
        }
    fprintf(stderr, "[grab timing] memory %s: %s\n", when, all.c_str());
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

// ---- walkers -> workers queue ----
struct Job {
    std::string path;
    struct stat st;
};

class JobQueue {
public:
    void push(Job &&j)
    {
        std::unique_lock<std::mutex> lk(m_);
        room_.wait(lk, [&] { return q_.size() < kMax || abandoned_; });
        if (abandoned_) return; // no worker is left to take it
        q_.push_back(std::move(j));
        ready_.notify_one();
    }
    bool try_pop(Job &out) // false: nothing there right now
    {
        std::lock_guard<std::mutex> lk(m_);
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        room_.notify_one();
        return true;
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
    void close() // the walk is over
    {
        std::lock_guard<std::mutex> lk(m_);
        closed_ = true;
        ready_.notify_all();
    }
    void abandon() // the workers cannot run: let the walkers finish into nothing
    {
        std::lock_guard<std::mutex> lk(m_);
        abandoned_ = true;
        q_.clear();
        room_.notify_all();
    }

private:
    static constexpr size_t kMax = 1 << 16;
    std::mutex m_;
    std::condition_variable ready_, room_;
    std::deque<Job> q_;
    bool closed_ = false, abandoned_ = false;
};

// the CPUs local to every device (sysfs local_cpulist of its PCI function; empty where that is unknown)
std::vector<std::vector<int>> device_cpu_lists(int ndev)
{
    std::vector<std::vector<int>> out((size_t)ndev);
    for (int d = 0; d < ndev; d++) {
        char list[1024];
        int cpus[1024];
        if (gscan_device_cpulist(d, list, sizeof list) <= 0) continue;
        const long n = gscan_parse_cpulist(list, cpus, 1024);
        for (long k = 0; k < n && k < 1024; k++) out[(size_t)d].push_back(cpus[k]);
    }
    return out;
}

// The HIP runtime initialises every device it can see, whether or not anything is ever sent there (on an 8-GPU node that is
// most of hipInit's time and of the exit's).  What the input can use is known before the first HIP call: explicit files of
// one window each need ONE device; `-n N` with N below the device count needs the first N.  Narrow HIP_VISIBLE_DEVICES to
// that -- within whatever list the caller has already set -- unless the caller chose devices itself (GRAB_DEVICE,
// GRAB_DEVICES, GRAB_ALL_DEVICES=1).
void narrow_visible_devices(size_t want)
{
    if (want == 0 || getenv("GRAB_DEVICE") || getenv("GRAB_DEVICES") || getenv("GRAB_ALL_DEVICES") || getenv("GSCAN_VIRTUAL_DEVICES")) return;
    const char *have = getenv("HIP_VISIBLE_DEVICES");
    if (!have || !*have) have = getenv("CUDA_VISIBLE_DEVICES"); // (the runtime honours it when the HIP one is unset: narrow within the caller's choice)
    std::string list;
    if (have && *have) { // the first `want` entries of the caller's list
        size_t n = 0;
        for (const char *q = have; *q && n < want; q++) {
            if (*q == ',' && ++n == want) break;
            list += *q;
        }
    } else {
        for (size_t i = 0; i < want; i++) list += (i ? "," : "") + std::to_string(i);
    }
    setenv("HIP_VISIBLE_DEVICES", list.c_str(), 1);
    static const bool on = getenv("GRAB_TIMING") != nullptr;
    if (on) fprintf(stderr, "[grab timing] visible devices narrowed to %s\n", list.c_str());
}

int run_workers(const Options &o)
{
    if (!o.recursive) {
        std::cerr << "Multicore support only for recursive grabs.\n";
        return -1;
    }
    // What the reference's prepare() would say, before anything is started.  It ignores the result in this mode
    // (main.cc:198): with a pattern PCRE rejects every pcre_exec fails, nothing is printed and the status is 0 -- same
    // here.  A pattern that is fine for PCRE and outside this engine is not the reference's case: reported, status 255.
    {
        std::string why;
        const int v = FileGrep::validate(o.regex, o.cfg.count("literal") != 0, why);
        if (v == -1) return 0;
        if (v != 0) {
            std::cerr << why << std::endl;
            return -1;
        }
    }
    // More threads than CPUs is fatal in the reference: it pins thread i to CPU i and pthread_setaffinity_np fails for a CPU
    // that is offline or outside the process's cpuset (main.cc:211-215).  It does NOT fail for a CPU that an inherited mask
    // merely leaves out (`taskset -c 8-15 grab -n 4 ...` runs): so the same call is made, on a scratch thread, and only its
    // verdict counts.  The workers themselves are bound to their device's NUMA node below (GRAB_PIN=cpu: to CPU i).
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    sched_getaffinity(0, sizeof allowed, &allowed);
    {
        int bad = 0;
        std::thread probe([&] {
            for (int i = 0; i < o.workers && !bad; i++) {
                cpu_set_t one;
                CPU_ZERO(&one);
                if (i >= CPU_SETSIZE) {
                    bad = EINVAL;
                    break;
                }
                CPU_SET(i, &one);
                bad = pthread_setaffinity_np(pthread_self(), sizeof one, &one);
            }
        });
        probe.join();
        if (bad) {
            std::cerr << "pthread_setaffinity_np:" << strerror(bad) << " (more threads than cores?)" << std::endl;
            return -1;
        }
    }

    JobQueue queue;
    // the walk starts at once, on its own threads; the workers open their devices meanwhile
    int walkers = 4;
    if (const char *w = getenv("GRAB_WALKERS")) walkers = std::max(1, atoi(w));
    std::thread walk([&] {
        grab_walk(o.paths[0], walkers, [&](std::string &&path, const struct stat &st) { queue.push(Job{std::move(path), st}); });
        queue.close();
        mark("walk done");
    });

    const int ndev = std::max(1, gscan_device_count());
    mark("runtime up");
    mark_memory("runtime up");
    std::mutex err_lock;
    std::string first_error;
    std::vector<std::thread> pool;
    const std::vector<WorkerPlace> places = grab_place_workers(o.workers, ndev, device_cpu_lists(ndev), allowed, getenv("GRAB_PIN"));
    for (int i = 0; i < o.workers; i++) {
        const int device = places[(size_t)i].device;
        const cpu_set_t cpus = places[(size_t)i].cpus;
        pool.emplace_back([&, device, cpus, i] {
            (void)pthread_setaffinity_np(pthread_self(), sizeof cpus, &cpus);
            // one per thread, like the reference (main.cc:195-199); the context is opened by the thread that uses it.  It is
            // NOT closed: once everything is retired and printed the process leaves through _exit (main), and freeing 8 x 3
            // windows of HBM, their pinned blocks and streams one by one first costs ~50 ms of a 1-2 s run for nothing.
            // (GRAB_CLOSE=1 closes them, for leak checkers.)
            static const bool close_ctx = getenv("GRAB_CLOSE") != nullptr;
            FileGrep *gp = new FileGrep;
            FileGrep &g = *gp;
            {
            auto cfg = o.cfg;
            cfg["device"] = size_t(device);
            cfg["silent_errors"] = 1; // per-file errors are ignored in this mode (main.cc:97), also the ones that surface after find() returned
            g.config(cfg);
            g.recurse();
            if (g.prepare(o.regex) < 0) { // the pattern is fine (checked above): the device could not be opened
                std::lock_guard<std::mutex> lk(err_lock);
                if (first_error.empty()) first_error = g.why();
                queue.abandon();
                return;
            }
            if (i == 0) {
                mark("worker 0: context open");
                mark_memory("worker 0 open");
            }
            for (Job j;;) { // per-file errors ignored (main.cc:97)
                if (!queue.try_pop(j)) {
                    g.idle(); // the walkers are behind: a half-filled batch of small files goes out now rather than when it is full
                    if (!queue.pop(j)) break;
                }
                g.find(j.path.c_str(), &j.st, FTW_F);
            }
            g.flush(); // what is still in flight or waiting in a half-filled batch
            if (g.failed()) { // a device error (per-file errors are ignored in this mode, a device that stopped scanning is not)
                std::lock_guard<std::mutex> lk(err_lock);
                if (first_error.empty()) first_error = g.why();
            }
            g.report_timing();
            if (i == 0) mark("worker 0: everything retired");
            }
            if (close_ctx) {
                delete gp;
                if (i == 0) mark("worker 0: context closed");
            }
        });
    }
    for (auto &t : pool) t.join();
    mark("workers joined");
    walk.join();
    if (!first_error.empty()) {
        std::cerr << first_error << std::endl;
        return -1;
    }
    return 0;
}

int run_serial(const Options &o)
{
    // (not closed before _exit either, see run_workers; GRAB_CLOSE=1 does)
    FileGrep *gp = new FileGrep;
    struct Closer {
        FileGrep *g;
        ~Closer()
        {
            if (getenv("GRAB_CLOSE")) delete g;
            else g->report_timing();
        }
    } closer{gp};
    FileGrep &grep = *gp;
    auto cfg = o.cfg;
    if (const char *dev = getenv("GRAB_DEVICE")) cfg["device"] = size_t(atoi(dev));
    // a file of several windows is spread over the node's GPUs (contexts beyond the first open when such a file turns up)
    if (const char *n = getenv("GRAB_DEVICES")) cfg["devices"] = size_t(std::max(1, atoi(n)));
    else cfg["devices"] = size_t(std::max(1, gscan_device_count()));
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

// GRAB_DETACH=1: the parent of the scanning child passes a terminating signal on and then takes it itself
volatile pid_t g_child = 0;
void forward_signal(int sig)
{
    if (g_child > 0) kill(g_child, sig);
    signal(sig, SIG_DFL);
    raise(sig);
}

} // namespace

int main(int argc, char **argv)
{
    mark("main");
    const Options o = parse(argc, argv);
    // Taking the process's GPU state apart costs the kernel 0.1 - 0.2 s at exit (profiles/r02_o_exit_cost_and_open_order.txt),
    // after the last byte of output.  GRAB_DETACH=1 (opt-in; round 2 had it on by default) runs the scan in a child: when it
    // has printed everything it closes its output, hands its exit status over a pipe and leaves; the parent returns that
    // status at once and the child's teardown goes on behind the caller's back.  The fork comes before the first HIP call:
    // the runtime is not up yet and there is one thread.  The pair still behaves like ONE process towards the outside: a
    // signal that reaches the parent (SIGINT, SIGTERM, SIGHUP, SIGQUIT) is passed on to the child before the parent takes
    // it itself, and the child asks the kernel for SIGKILL should the parent disappear in any other way -- no orphan keeps
    // scanning, writing to the caller's stdout or holding the GPUs.
    int status_fd = -1;
    if (getenv("GRAB_DETACH") && atoi(getenv("GRAB_DETACH")) != 0) {
        int pfd[2];
        if (pipe(pfd) == 0) {
            const pid_t parent = getpid();
            const pid_t child = fork();
            if (child > 0) {
                close(pfd[1]);
                g_child = child;
                for (int sig : {SIGINT, SIGTERM, SIGHUP, SIGQUIT}) {
                    struct sigaction sa;
                    memset(&sa, 0, sizeof sa);
                    sa.sa_handler = forward_signal;
                    sigaction(sig, &sa, nullptr);
                }
                int rc = 0;
                ssize_t got;
                do got = read(pfd[0], &rc, sizeof rc);
                while (got < 0 && errno == EINTR);
                if (got == (ssize_t)sizeof rc) _exit(rc & 255);
                int st = 0; // the child died without a word: its wait status says how
                while (waitpid(child, &st, 0) < 0 && errno == EINTR) {}
                if (WIFSIGNALED(st)) {
                    signal(WTERMSIG(st), SIG_DFL);
                    raise(WTERMSIG(st));
                }
                _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 255);
            }
            if (child == 0) {
                close(pfd[0]);
                status_fd = pfd[1];
                prctl(PR_SET_PDEATHSIG, SIGKILL);
                if (getppid() != parent) _exit(255); // (the parent went away between fork and prctl)
            } else { // no fork: carry on in this process
                close(pfd[0]);
                close(pfd[1]);
            }
        }
    }
    // Devices the input cannot use stay out of sight (narrow_visible_devices) -- decided here, before any helper thread exists
    // (setenv next to running threads is not safe): `-n N`: worker i drives device i mod #devices, fewer workers than devices
    // leave the rest idle; explicit paths: as many windows as the largest file has can be in flight at once.
    if (o.workers > 1) {
        if (o.recursive) narrow_visible_devices((size_t)o.workers);
    } else if (!o.recursive) {
        const size_t chunk = o.cfg.count("chunk_size") ? o.cfg.at("chunk_size") : (size_t(1) << 30), stride = chunk - 4096;
        size_t most = 1;
        for (const std::string &p : o.paths) {
            struct stat st;
            if (stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode)) most = std::max(most, ((size_t)st.st_size + stride - 1) / stride);
        }
        if (most <= 64) narrow_visible_devices(most);
    }
    // staging memory is mapped and touched by helper threads while the HIP runtime starts (no HIP call in there; behind the fork above: threads do not survive one): as many
    // blocks as the input can keep busy -- a tree: the reader pool's worth; explicit files: what their bytes fill, and the
    // helpers READ the files' first pieces into them while they are at it (gscan_prefault_files): by the time the runtime
    // answers its first call, a 256 MiB file sits in staging memory and only has to be registered and DMA'd
    {
        if (!o.recursive && o.workers <= 1) {
            unsigned long long total = 0;
            std::vector<const char *> names;
            for (const std::string &p : o.paths) {
                struct stat st;
                // (files up to the batching limit are queued by name and read by the reader pool as a batch: not these)
                const unsigned long long batch = o.cfg.count("batch") ? o.cfg.at("batch") : (2ull << 20);
                if (stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode) && (unsigned long long)st.st_size > batch) total += (unsigned long long)st.st_size, names.push_back(p.c_str());
            }
            const size_t blocks = (size_t)std::min<unsigned long long>(36, total / gscan_block_size() + 2);
            if (getenv("GRAB_NO_READ_AHEAD")) gscan_prefault(std::min<size_t>(18, blocks)); // (A/B runs)
            else gscan_prefault_files(blocks, names.data(), names.size());
        } else {
            gscan_prefault(18);
        }
    }
    const int rc = o.workers > 1 ? run_workers(o) : run_serial(o);
    mark("scan done");
    std::cout.flush();
    mark_memory("at exit"); // what the kernel will have to take apart when the process leaves
    mark("leaving");
    // GRAB_DIAG=1: say when the host matcher abandoned attempts at its resource limit (each ends its chunk silently, as a
    // pcre_exec error does in the reference: src/grab.cc:179); the differential tests skip such inputs
    if (getenv("GRAB_DIAG") && gscan_resource_errors())
        fprintf(stderr, "grab: %llu match attempts abandoned at the matcher's resource limit\n", (unsigned long long)gscan_resource_errors());
    // Everything is printed, every context is closed, every thread joined: leave without the HIP runtime's exit handlers (tens
    // of milliseconds of a run that takes one or two seconds).  -1 -> exit status 255, like the reference's `return -1` from main
    fflush(stderr);
    if (status_fd >= 0) { // (see the top of main)
        close(1);
        close(2);
        const int code = rc & 255;
        (void)!write(status_fd, &code, sizeof code);
    }
    if (const char *ms = getenv("GRAB_EXIT_SLEEP_MS")) usleep((useconds_t)atoi(ms) * 1000); // (diagnostic: what the exit costs against the process's age)
    if (getenv("GRAB_NORMAL_EXIT")) return rc & 255; // (profilers write their results from exit handlers)
    _exit(rc & 255);
}
