// engine.hip -- gscan context: streams, pinned/HBM buffers, double-buffered chunk
// pipeline, device-resident batch scans, and the C ABI of include/gscan.h.
//
// Data flow of the host-chunk path (what FileGrep::find drives, replacing the
// mmap -> pcre_exec loop of /root/reference/src/grab.cc:154-215):
//
//   reader threads: pread(2) into pinned blocks ──the device's copy streams: hipMemcpyAsync──► slot.d_text (HBM)
//        scan kernel on the first copy stream (behind its pieces there, after an event of every further copy stream)
//        ──► d_recs/d_desc/d_counter ──same stream: async D2H of counter + descriptors + first records──► event `done`
//   gscan_wait: sync `done`, fetch the rest if needed, stitch runs in tile order.
//
// With GSCAN_SLOTS = 3 the read + copy of chunk k+2 overlaps the scan of chunk k+1 and the report of chunk k.
//
// How the bytes get to the copy stream (measured on the MI355X box, profiles/r01_g_host_probe.txt:
// hipHostMalloc 0.22 s/GiB, page cache -> pinned 8-10 GB/s per thread and linear in threads,
// H2D 57 GB/s, register + unregister + munmap of a mapping ~55 ms/GiB per thread and slower with
// more threads):
//   gscan_submit_fd   a file range: the device's reader threads pread(2) it piecewise into a small pool
//                     of pinned blocks and DMA each piece as soon as it is read, spread over the
//                     context's copy streams; the thread that finishes the last piece launches the
//                     scan, the submitting thread does not wait; nothing is pinned per chunk
//   gscan_submit_segs many small files packed by the caller into the slot's pinned block: one
//                     H2D, one launch over a segment table
//   gscan_submit      a caller buffer: registered and DMA'd in place (>= 1 MiB) or staged
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "scan_args.h"
#include "../../include/gscan_test.h"


namespace gscan {
void nt_copy(void *dst, const void *src, size_t n); // hostcopy.cc
}

using gscan::Database;
using gscan::DevProgram;
using gscan::ScanArgs;

#include "db.h"

namespace {

constexpr size_t kPad = 4096;          // slack behind every text buffer
constexpr size_t kSpecPer = 256;        // records of EACH shard region fetched speculatively with the header
constexpr size_t kSpecRecs = kSpecPer * gscan::kShards;
constexpr size_t kCS = gscan::kCtrStride;                  // the shard counters sit one per 128-byte line (scan_args.h)
constexpr size_t kCounterWords = gscan::kShards * kCS + 4; // per-shard counts + overflow flag + records struck out by the second pass + bytes of line text gathered (k_lines) + records in the ordered copy (k_order_prefix)
constexpr size_t kGatherPinnedMax = 512u << 20;             // a window's gathered lines are fetched only up to this size (beyond: the host copies from the file)
constexpr size_t kCopyPiece = 32u << 20; // memcpy/H2D pipelining granule for foreign host buffers
constexpr size_t kMaxChunk = (1ull << 30) + 4096;

// ---- ingest configuration (environment, read once) ----
//   GSCAN_BLOCK_MIB     pinned pool block == read piece == one hipMemcpyAsync == batch buffer        (default 8: as fast as 16 in
//                       the steady state, half the pinned memory; profiles/r04_d_*)
//   GSCAN_READERS       reader threads per device; 0 or unset = auto: 8, fewer when the device's NUMA node has few CPUs per
//                       device (gscan_auto_readers: 8 GPUs x 8 readers must not outnumber the CPUs they are bound to)
//   GSCAN_COPY_STREAMS  copy streams per DEVICE, shared by its contexts; the scans ride on the first (default 2: a second
//                       stream's copy runs in the first one's gaps, 46.7 -> 49.2 GB/s; a third adds nothing, profiles/r04_f_*).
//                       The second one is made when the device has been handed GSCAN_SECOND_STREAM_MIB (1024) MiB, by a
//                       helper thread: a stream costs 7-12 ms to create and its share of the exit, and a run that is over in
//                       a tenth of a second -- one 256 MiB file: 0.101 against 0.123 s, profiles/r05_b_cfg1_* -- never needs it
//   GSCAN_NUMA          0: readers inherit the process's CPU mask; else the CPUs local to the device (default 1)
//   GSCAN_NT_COPY       1: readers pread into a cache-sized bounce buffer and stream it into the block with non-temporal
//                       stores (hostcopy.cc: no write-allocate, one DRAM crossing less per byte).  Off by default: measured
//                       slower than plain pread at one device AND under eight pools' load (profiles/r04_k_*, r05_a_n8_*)
// Test hooks (tests/test_gpu_pool.py): GSCAN_POOL_CAP caps the reader pool's blocks, GSCAN_FAIL_ALLOC_AFTER=n makes every
// allocation of a staging block after the n-th fail.  GSCAN_DIAG (measurements only -- the results are NOT scan results, said
// on stderr): 1 = the readers fill their blocks and give them straight back, no DMA, no scan (the host's page cache -> pinned
// ceiling); 2 = the readers do not read, the blocks go out as they are (the DMA side on its own); 3 = 2 without
// the scans and read-backs (the copies alone).
// What the sweeps of rounds 2 - 4 on the MI355X boxes say (profiles/r02_a_e2e_ingest_sweep.jsonl, r02_b_dma_probe.txt,
// r04_e_* .. r04_k_*): the link itself moves 57 GB/s in any piece size >= 16 MiB, also next to 16 busy pread threads; inside the
// pipeline the 64 GiB corpus goes through at 49-52 GB/s whatever the block size, the flavour of the pinned memory, the way the
// readers copy or how often an event is recorded behind a DMA; 8 readers are the best (more of them wait for blocks longer
// than they save reading).  The alternates those sweeps went through (per-context streams, a slab allocation, write-combined
// blocks, blocks made ahead, an event per k-th DMA, a mapped copy) are gone from the code; their numbers are in profiles/.
// GSCAN_TRACE=1: a time line of the pipeline on stderr -- "[gscan trace] +seconds-since-load thread what" -- for the runs
// that are over in a fraction of a second (where does a 256 MiB file's 0.1 s go?)
const auto g_trace_t0 = std::chrono::steady_clock::now();
const bool g_trace = getenv("GSCAN_TRACE") != nullptr;
const bool g_timing = getenv("GSCAN_TIMING") != nullptr; // where the readers' and gscan_wait's time goes, printed when a context closes
void trace(const char *fmt, ...)
{
    if (!g_trace) return;
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fprintf(stderr, "[gscan trace] +%.4f s t%05ld %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - g_trace_t0).count(),
            (long)(uintptr_t)pthread_self() % 100000, buf);
}

// Staging memory made ready BEFORE the HIP runtime is up (gscan_prefault): the command line knows it is going to scan the
// moment it starts, hipInit takes 50 ms, and a pinned block from hipHostMalloc costs 1.5 - 2 ms while the pipe fills (the
// runtime serves one at a time: a 256 MiB file spends 25 of its 30 ms of transfer waiting for blocks).  So a helper thread
// maps anonymous memory (huge pages where the kernel gives them) and touches it while the runtime starts; a block is then
// taken from there and only REGISTERED with the runtime (pages present, nothing to zero).
// Round 5: the helpers can also FILL blocks with the first pieces of the files the command line names (gscan_prefault_files):
// what a reader would pread after the runtime is up is read while it starts; the piece's reader then adopts the block as it
// is -- registers it and queues the DMA.  A 256 MiB file used to crawl through the pipe's ramp in 29 ms, after the runtime's
// 50 ms; now its bytes are in staging memory before the runtime answers its first call.
struct Prefault {
    char *base = nullptr;
    size_t stride = 0, count = 0;
    size_t ahead = 0;               // blocks [0, ahead) are read-ahead pieces, [ahead, count) plain touched blocks
    std::atomic<size_t> next{0};    // plain blocks handed out so far
    std::atomic<int> *state = nullptr; // per block: 0 not ready yet, 1 touched, 2 filled with its piece, 3 the read failed
                                       // (never freed: the helper threads are detached and may outlive static destruction)
    struct Piece {
        std::string path;
        dev_t dev = 0;
        ino_t ino = 0;
        off_t size = 0;               // what the file looked like when the helper read the piece: a file rewritten in place in
        struct timespec mtim{}, ctim{}; // between (same inode, same size) must not be scanned from stale staging bytes
        off_t off = 0;
        size_t n = 0;
        std::atomic<bool> claimed{false};
    };
    Piece *piece = nullptr;         // [ahead]; dev / ino are written by the helper before state goes to 2
    std::atomic<size_t> unclaimed{0};
};
Prefault g_prefault;

struct IngestCfg {
    size_t block;
    int readers;
    // A HIP stream is an HSA queue, and on this part a queue comes with ~177 MB of wave-save area that the runtime allocates
    // AND touches: 7-12 ms to create, ~4 ms of the process's exit, per stream (profiles/r02_r_resident_memory.txt, r04_c_*).
    // So the contexts of a device share the device's copy streams, and the scans and read-backs ride on the first of them
    // (a 64 MiB window scans in 15 us).
    int copy_streams;
    size_t second_after;  // bytes handed to a device before its second copy stream is made
    bool scan_stream;     // GSCAN_SCAN_STREAM=1 (measurements): the scans and read-backs on a stream of their own
    int ahead_threads;    // GSCAN_AHEAD_THREADS (measurements): helper threads of gscan_prefault_files
    bool numa;
    int nt_copy;          // GSCAN_NT_COPY
    long pool_cap;        // GSCAN_POOL_CAP (test hook): 0 = two blocks per reader
    long fail_alloc_after; // GSCAN_FAIL_ALLOC_AFTER (test hook): -1 = never
    int diag;             // GSCAN_DIAG: 0, 1 (no DMA, no scan), 2 (no read), 3 (no read, no scan)
    int virtual_devices;  // GSCAN_VIRTUAL_DEVICES (below)
    int virtual_fail_open; // GSCAN_VIRTUAL_FAIL_OPEN: gscan_open fails for this index (-1: none)
    bool prefault;        // GSCAN_PREFAULT=0 switches gscan_prefault off
};
const IngestCfg &ingest_cfg()
{
    static const IngestCfg c = [] {
        IngestCfg v;
        auto env = [](const char *n, long def, long lo, long hi) {
            const char *e = getenv(n);
            long x = e && *e ? atol(e) : def;
            return std::max(lo, std::min(hi, x));
        };
        v.block = (size_t)env("GSCAN_BLOCK_MIB", 8, 1, 64) << 20;
        v.readers = (int)env("GSCAN_READERS", 0, 0, 64); // 0: auto, per device (Ingest's constructor)
        const long hw = (long)std::thread::hardware_concurrency();
        if (hw > 0 && v.readers > hw) v.readers = (int)hw;
        v.copy_streams = (int)env("GSCAN_COPY_STREAMS", 2, 1, 2);
        v.second_after = (size_t)env("GSCAN_SECOND_STREAM_MIB", 1024, 0, 1 << 20) << 20;
        v.scan_stream = env("GSCAN_SCAN_STREAM", 0, 0, 1) != 0;
        v.ahead_threads = (int)env("GSCAN_AHEAD_THREADS", 8, 1, 32);
        v.numa = env("GSCAN_NUMA", 1, 0, 1) != 0;
        v.nt_copy = (int)env("GSCAN_NT_COPY", 0, 0, 1);
        v.pool_cap = env("GSCAN_POOL_CAP", 0, 0, 4096);
        v.fail_alloc_after = env("GSCAN_FAIL_ALLOC_AFTER", -1, -1, 1 << 20);
        v.diag = (int)env("GSCAN_DIAG", 0, 0, 3);
        v.virtual_devices = (int)env("GSCAN_VIRTUAL_DEVICES", 0, 0, 64);
        v.virtual_fail_open = (int)env("GSCAN_VIRTUAL_FAIL_OPEN", -1, -1, 64);
        v.prefault = env("GSCAN_PREFAULT", 1, 0, 1) != 0;
        // (GSCAN_DIAG makes the readers skip the read or the DMA: a measurement of the ingest pipe whose "results" are whatever
        // the staging blocks held.  It takes a second switch to get it, so that no stray variable turns a scan into that.)
        if (v.diag && !getenv("GSCAN_TEST_HOOKS")) {
            fprintf(stderr, "gscan: GSCAN_DIAG=%d ignored (it needs GSCAN_TEST_HOOKS=1 as well: a measurement of the ingest pipe, not a scan)\n", v.diag);
            v.diag = 0;
        }
        if (v.diag) fprintf(stderr, "gscan: GSCAN_DIAG=%d -- a measurement of the ingest pipe, NOT a scan: whatever is reported is meaningless\n", v.diag);
        return v;
    }();
    return c;
}
inline size_t block_bytes() { return ingest_cfg().block; }

// A pinned block of the process-wide pool.  ev is recorded after the last DMA out of the block.
struct PinBlock {
    void *p = nullptr;
    hipEvent_t ev = nullptr;
    bool registered = false; // p comes from the prefaulted arena and was hipHostRegister'ed (not hipHostMalloc'ed)
};

// One file range on its way to HBM (gscan_submit_fd): its pieces are read by the reader threads; whoever finishes the
// last piece runs `finish` (records the copy events and launches the scan), so the submitting thread never waits.
struct ReadGroup {
    std::mutex m;
    std::condition_variable cv;
    size_t pending = 0;
    int err = 0; // errno of the first failed read, -1 for a short file, -2 for a HIP failure
    bool finished = true;
    void (*finish)(ReadGroup *) = nullptr;
    void *ctx = nullptr, *slot = nullptr; // for `finish`
    int rc = 0;                           // what `finish` came to (a GSCAN_* code) ...
    std::string msg;                      // ... and its text
};
// One small file of a batch (gscan_submit_files): where it comes from and where its bytes go inside the batch's text.
struct FileItem {
    std::string path; // opened by the reader with `oflags` and closed after the read; empty: `fd` is the caller's
    int fd = -1;
    int oflags = 0;
    uint32_t len = 0;
    uint64_t dst_off = 0; // 16-byte aligned offset of the segment in the batch
    int err = 0;          // errno of open / pread, -1: the file is shorter than len
};
struct ReadTask {
    int fd;
    off_t off;
    size_t n;
    uint8_t *dst;       // device
    hipStream_t stream; // one of the submitting context's copy streams
    ReadGroup *grp;
    int ahead = -1;     // >= 0: the piece was read ahead into this block of the prefaulted arena (gscan_prefault_files)
    // a piece made of whole small files (gscan_submit_files): items[0..nitems) land at their dst_off - items[0].dst_off
    // inside the block, n covers them all, ONE DMA carries the lot
    FileItem *items = nullptr;
    uint32_t nitems = 0;
};

// "0-31,64-95" -> CPU numbers (the format of sysfs cpulist files)
size_t parse_cpulist(const char *list, std::vector<int> &out)
{
    out.clear();
    for (const char *q = list; q && *q;) {
        while (*q == ',' || *q == ' ' || *q == '\n' || *q == '\t') q++;
        if (*q < '0' || *q > '9') break;
        char *e = nullptr;
        long a = strtol(q, &e, 10), b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        if (b < a || b - a > 65536) break;
        for (long c = a; c <= b; c++) out.push_back((int)c);
        q = e;
    }
    return out.size();
}

int pci_cpulist(const char *root, const char *busid, char *buf, size_t cap)
{
    if (!root || !busid || !buf || cap < 2) return GSCAN_EINVAL;
    std::string id(busid);
    for (char &ch : id) ch = (char)tolower((unsigned char)ch);
    const std::string path = std::string(root) + "/" + id + "/local_cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return GSCAN_EIO;
    size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    while (n && (buf[n - 1] == '\n' || buf[n - 1] == ' ')) n--;
    buf[n] = 0;
    return (int)n;
}

// GSCAN_VIRTUAL_DEVICES=N (tests): the engine presents N device indices on a box that has fewer -- index v runs
// on HIP device v mod (the real count) but is a device of its own in every other respect: its own reader pool, pinned
// blocks, shared streams and placement (bus id 0000:XX:00.0 with XX = 0x0c + 0x10 v, looked up under $GSCAN_SYSFS_PCI like a
// real one) -- so that `grab -n` over an 8-GPU node's worth of device indices, and one file's windows dealt over 8 contexts,
// run for real on the one-GPU boxes there are.  GSCAN_VIRTUAL_FAIL_OPEN=v makes gscan_open fail for index v (a device that
// is busy or out of memory).
int virtual_devices() { return ingest_cfg().virtual_devices; }
int real_device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
int device_count()
{
    const int real = real_device_count();
    return real > 0 && virtual_devices() ? virtual_devices() : real;
}
int hip_device_of(int device)
{
    if (!virtual_devices()) return device;
    const int real = real_device_count();
    return real > 0 ? device % real : device;
}

int device_cpulist(int device, char *buf, size_t cap)
{
    char id[64] = {0};
    if (virtual_devices()) {
        snprintf(id, sizeof id, "0000:%02X:00.0", 0x0c + 0x10 * (device & 15));
    } else if (hipDeviceGetPCIBusId(id, (int)sizeof id, device) != hipSuccess) {
        (void)hipGetLastError();
        return GSCAN_EHIP;
    }
    const char *root = getenv("GSCAN_SYSFS_PCI");
    return pci_cpulist(root && *root ? root : "/sys/bus/pci/devices", id, buf, cap);
}

// Process-wide, one per device while any context on that device is open: pinned blocks + the reader threads that fill
// them.  Two block pools so that readers can never be starved by blocks parked in contexts' slots:
//   reader blocks (at most 2 per reader thread) cycle  free -> read -> DMA in flight -> free;
//   slot blocks back gscan_acquire and are cached here between contexts.
// Reference-counted by the contexts: the last gscan_close on a device joins the threads and frees the blocks.
class Ingest {
public:
    static Ingest *acquire(int device)
    {
        std::lock_guard<std::mutex> lk(gm());
        for (Ingest *i : all())
            if (i->device_ == device) {
                i->refs_++;
                return i;
            }
        all().push_back(new Ingest(device, (int)all().size() + 1));
        // the host's page cache feeds every pool of the process: with one more device being driven, the pools that are there
        // already may have to make do with fewer active readers (gscan_auto_readers: 24 in all)
        for (Ingest *i : all()) i->relimit((int)all().size());
        return all().back();
    }
    static void release(Ingest *g)
    {
        {
            std::lock_guard<std::mutex> lk(gm());
            if (--g->refs_ > 0) return;
            all().erase(std::find(all().begin(), all().end(), g));
            for (Ingest *i : all()) i->relimit((int)all().size());
        }
        delete g;
    }

    PinBlock *take_slot_block()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            if (!slot_free_.empty()) {
                PinBlock *b = slot_free_.back();
                slot_free_.pop_back();
                return b;
            }
        }
        return alloc_block();
    }
    void give_slot_block(PinBlock *b)
    {
        std::lock_guard<std::mutex> lk(m_);
        slot_free_.push_back(b);
    }

    void read(const ReadTask *t, size_t n)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (!started_) {
            started_ = true;
            for (int i = 0; i < readers_; i++) threads_.emplace_back([this, i] { reader_main(i); });
        }
        tasks_.insert(tasks_.end(), t, t + n);
        // (one task, one reader woken -- unless some readers sit out (relimit): notify_one might pick one of those, which goes back
        // to sleep without the task, and nobody else would hear of it)
        if (n == 1 && limit_.load(std::memory_order_relaxed) >= readers_) cv_tasks_.notify_one();
        else cv_tasks_.notify_all();
    }
    int readers() const { return readers_; }
    // The second copy stream, if the device has one by now.  It is made once the device has been handed `second_after` bytes,
    // by a helper thread (nobody waits for it: the pieces go on over the first stream until it is there).
    hipStream_t second_stream(size_t more_bytes)
    {
        const size_t had = handed_.fetch_add(more_bytes, std::memory_order_relaxed);
        if (ingest_cfg().copy_streams < 2) return nullptr;
        hipStream_t st = second_.load(std::memory_order_acquire);
        if (st || had + more_bytes < ingest_cfg().second_after) return st;
        bool expected = false;
        if (second_started_.compare_exchange_strong(expected, true)) {
            try {
                second_maker_ = std::thread([this] {
                    hipStream_t s = shared_stream(1);
                    second_.store(s, std::memory_order_release);
                    trace("ingest: second copy stream made");
                });
            } catch (...) { // (no thread to be had: one stream it is)
            }
        }
        return nullptr;
    }
    // the device's streams: created on first use (one at a time per device; devices do not wait for each other), destroyed
    // with the pool
    hipStream_t shared_stream(int k)
    {
        std::lock_guard<std::mutex> lk(streams_m_);
        if ((int)shared_.size() <= k) shared_.resize((size_t)k + 1, nullptr);
        if (!shared_[(size_t)k]) {
            (void)hipSetDevice(hip_device_of(device_));
            if (hipStreamCreateWithFlags(&shared_[(size_t)k], hipStreamNonBlocking) != hipSuccess) return shared_[(size_t)k] = nullptr;
        }
        return shared_[(size_t)k];
    }
    void report_if_timing()
    {
        if (g_timing && n_pieces_) report();
    }

private:
    static std::mutex &gm()
    {
        static std::mutex m;
        return m;
    }
    static std::vector<Ingest *> &all()
    {
        static std::vector<Ingest *> v;
        return v;
    }

    // driven: devices this process drives by now, this one included -- NOT the devices that are visible: a serial `grab -r`, or a
    // Python caller with one context, on an eight-GPU node drives one and gets the one-device reader count
    Ingest(int device, int driven) : device_(device)
    {
        readers_ = ingest_cfg().readers;
        // Where the readers run: on the CPUs of the device's NUMA node (the pinned blocks are first touched by them, and the
        // page cache -> pinned copy is the host's share of every byte), as far as the process is allowed there.  Without
        // that information: the process's own mask.
        // (the PROCESS's mask, i.e. the main thread's: the opener may be a worker that `grab -n` has bound to a single CPU)
        have_mask_ = sched_getaffinity(getpid(), sizeof mask_, &mask_) == 0;
        if (ingest_cfg().numa) {
            char list[1024];
            std::vector<int> cpus;
            cpu_set_t allowed, local;
            if (device_cpulist(device, list, sizeof list) > 0 && parse_cpulist(list, cpus) && sched_getaffinity(getpid(), sizeof allowed, &allowed) == 0) {
                CPU_ZERO(&local);
                int n = 0;
                for (int c : cpus)
                    if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) {
                        CPU_SET(c, &local);
                        n++;
                    }
                if (n > 0) {
                    mask_ = local;
                    have_mask_ = true;
                    numa_cpus_ = n;
                }
            }
        }
        if (readers_ <= 0) { // auto: as many as the CPUs this device can count on allow
            int sharing = 1, ndev = 0;
            char mine[1024], other[1024];
            if (numa_cpus_ > 0 && device_cpulist(device, mine, sizeof mine) > 0 && (ndev = device_count()) > 0) {
                sharing = 0;
                for (int d = 0; d < ndev; d++)
                    if (d == device || (device_cpulist(d, other, sizeof other) > 0 && !strcmp(mine, other))) sharing++;
            }
            int cpus = numa_cpus_;
            if (cpus <= 0) cpus = have_mask_ ? CPU_COUNT(&mask_) : (int)std::thread::hardware_concurrency();
            auto_cpus_ = cpus;
            auto_sharing_ = std::max(1, sharing);
            readers_ = gscan_auto_readers(cpus, auto_sharing_, std::max(1, driven));
        }
        limit_.store(readers_, std::memory_order_relaxed);
        cap_ = ingest_cfg().pool_cap > 0 ? (size_t)ingest_cfg().pool_cap : (size_t)readers_ * 2;
        // (The first host -> device transfer of a process costs the runtime 8 - 25 ms of set-up inside the call that asks for
        // it.  A helper thread making it -- 64 bytes -- while the opener creates the device's stream was tried: SLOWER, one
        // 256 MiB file 0.127 against 0.114 s, 16 GiB 0.631 against 0.609 s -- the two calls meet inside the runtime;
        // profiles/r05_d_cfg1_*.)
    }
    ~Ingest()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            cv_tasks_.notify_all();
        }
        for (std::thread &t : threads_) t.join();
        if (second_maker_.joinable()) second_maker_.join();
        (void)hipSetDevice(hip_device_of(device_));
        for (auto &l : lanes_)
            for (PinBlock *b : l->fifo) free_block(b, false); // (every context is closed: gscan_close has waited for the device)
        for (PinBlock *b : free_) free_block(b, false);
        for (PinBlock *b : slot_free_) free_block(b, false);
        for (hipStream_t st : shared_)
            if (st) (void)hipStreamDestroy(st);
    }
    // (under gm()) the process drives `driven` devices now: reader i of this pool takes tasks while i < limit_
    void relimit(int driven)
    {
        if (auto_cpus_ <= 0) return; // (GSCAN_READERS: the caller's count stands)
        const int want = std::min(readers_, gscan_auto_readers(auto_cpus_, auto_sharing_, std::max(1, driven)));
        if (limit_.exchange(want, std::memory_order_relaxed) != want) {
            std::lock_guard<std::mutex> lk(m_);
            cv_tasks_.notify_all();
        }
    }
    void free_block(PinBlock *b, bool wait)
    {
        if (wait) (void)hipEventSynchronize(b->ev);
        if (b->ev) (void)hipEventDestroy(b->ev);
        if (b->p && b->registered) (void)hipHostUnregister(b->p);
        else if (b->p) (void)hipHostFree(b->p);
        delete b;
    }

    PinBlock *alloc_block()
    {
        PinBlock *b = new (std::nothrow) PinBlock();
        if (!b) return nullptr;
        (void)hipSetDevice(hip_device_of(device_));
        if (ingest_cfg().fail_alloc_after >= 0 && n_made_.fetch_add(1) >= ingest_cfg().fail_alloc_after) { // (test hook: the runtime has no more pinned memory to give)
            delete b;
            return nullptr;
        }
        // a block that was mapped and touched while the runtime started (gscan_prefault): only registered here
        if (g_prefault.base && g_prefault.stride >= block_bytes() + kPad) {
            const size_t k = g_prefault.ahead + g_prefault.next.fetch_add(1, std::memory_order_relaxed);
            if (k < g_prefault.count) {
                while (!g_prefault.state[k].load(std::memory_order_acquire)) std::this_thread::yield();
                void *p = g_prefault.base + k * g_prefault.stride;
                if (hipHostRegister(p, block_bytes() + kPad, hipHostRegisterDefault) == hipSuccess) {
                    if (hipEventCreateWithFlags(&b->ev, hipEventDisableTiming) == hipSuccess) {
                        b->p = p;
                        b->registered = true;
                        return b;
                    }
                    (void)hipHostUnregister(p);
                }
                (void)hipGetLastError(); // (registration refused: the runtime's own allocator below)
            }
        }
        if (hipHostMalloc(&b->p, block_bytes() + kPad, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&b->ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (b->p) hipHostFree(b->p);
            delete b;
            return nullptr;
        }
        return b;
    }

    // A piece that was read ahead while the runtime started (gscan_prefault_files): its block joins the pool as it is, filled.
    // nullptr: the helper's read failed, or the runtime refuses the registration -- the piece is read the ordinary way.
    PinBlock *adopt_ahead_block(int k)
    {
        while (!g_prefault.state[k].load(std::memory_order_acquire)) std::this_thread::yield();
        if (g_prefault.state[k].load(std::memory_order_acquire) != 2) return nullptr;
        PinBlock *b = new (std::nothrow) PinBlock();
        if (!b) return nullptr;
        void *p = g_prefault.base + (size_t)k * g_prefault.stride;
        if (hipHostRegister(p, block_bytes() + kPad, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            delete b;
            return nullptr;
        }
        if (hipEventCreateWithFlags(&b->ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostUnregister(p);
            delete b;
            return nullptr;
        }
        b->p = p;
        b->registered = true;
        std::lock_guard<std::mutex> lk(m_);
        n_alloc_++; // (beyond cap_ if need be: the memory is there already; the block serves the pool from now on)
        return b;
    }

    // DMA bookkeeping, one Lane per copy stream.  The blocks whose DMA has been queued on the stream sit in `fifo` in stream
    // order (`order` makes "enqueue + event + push" one step), each with its event recorded right behind its DMA: a stream is
    // in order, so an event that has completed frees every block in front of it.  (An event behind every 4th / 8th DMA only:
    // 46.1 / 46.7 / 47.2 GB/s for K = 1 / 4 / 8 -- the event packets are not what keeps the pipe below the link's rate,
    // profiles/r04_g_*; taken out again.)
    struct Lane {
        hipStream_t st = nullptr;
        std::mutex order;
        std::deque<PinBlock *> fifo;
    };
    Lane *lane_for(hipStream_t st) // (under m_)
    {
        for (auto &l : lanes_)
            if (l->st == st) return l.get();
        lanes_.emplace_back(new Lane());
        lanes_.back()->st = st;
        return lanes_.back().get();
    }
    // free every block up to the last one of each lane whose event has completed (under m_)
    size_t reap()
    {
        size_t freed = 0;
        for (auto &l : lanes_) {
            size_t upto = 0;
            for (size_t i = 0; i < l->fifo.size(); i++) {
                if (hipEventQuery(l->fifo[i]->ev) != hipSuccess) {
                    (void)hipGetLastError();
                    break;
                }
                upto = i + 1;
            }
            for (size_t i = 0; i < upto; i++) {
                free_.push_back(l->fifo.front());
                l->fifo.pop_front();
            }
            freed += upto;
        }
        return freed;
    }

    PinBlock *take_reader_block()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (!free_.empty()) {
                PinBlock *b = free_.back();
                free_.pop_back();
                return b;
            }
            // a block whose DMA is over is as good as a free one -- and a new block costs 2-7 ms of hipHostMalloc (0.22 s per GiB,
            // one at a time inside the runtime) where an 8 MiB DMA takes 0.15 ms: the pool only grows while every block is busy
            if (reap()) continue;
            if (n_alloc_ < cap_) {
                n_alloc_++;
                lk.unlock();
                trace("reader: allocating a pinned block");
                PinBlock *b = alloc_block();
                trace("reader: pinned block allocated");
                if (b) return b;
                lk.lock();
                n_alloc_--;
                cv_blocks_.notify_all();               // (readers asleep below count on allocations still under way)
                if (n_alloc_ == 0) return nullptr;     // no pinned memory at all (and no other allocation under way)
                cap_ = n_alloc_;                       // the runtime has no more to give: the pool stays as big as it is (no retry per piece)
                continue;
            }
            // Every block is on its way out: wait for the oldest event of the lane with the most blocks in flight.  The waiter
            // TAKES that block out of the lane first: while it sleeps on the event no other thread can query it, free the
            // block or -- after a round through the free list -- record the event again (two threads on one hipEvent_t at a
            // time is not something the runtime promises to survive).
            Lane *pick = nullptr;
            for (auto &l : lanes_)
                if (!l->fifo.empty() && (!pick || l->fifo.size() > pick->fifo.size())) pick = l.get();
            if (!pick) {
                if (n_alloc_ == 0) return nullptr; // not one staging block to be had, and nobody is trying any more
                n_waits_cv_++;
                cv_blocks_.wait(lk); // (other waiters hold every block in flight, or readers hold them all: they will bring some back)
                continue;
            }
            PinBlock *take = pick->fifo.front();
            pick->fifo.pop_front();
            n_waits_ev_++;
            lk.unlock();
            (void)hipEventSynchronize(take->ev);
            return take;
        }
    }
    // the block goes back: straight to the free list, or -- its DMA queued on `st` by the caller, who holds that lane's
    // `order` -- to the tail of the lane
    void give_reader_block(PinBlock *b, Lane *lane)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (lane) lane->fifo.push_back(b);
        else free_.push_back(b);
        cv_blocks_.notify_one();
    }

    void reader_main(int index)
    {
        if (have_mask_) (void)pthread_setaffinity_np(pthread_self(), sizeof mask_, &mask_);
        (void)hipSetDevice(hip_device_of(device_));
        const int diag = ingest_cfg().diag;
        for (;;) {
            ReadTask t;
            double t0 = g_timing ? now() : 0;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_tasks_.wait(lk, [&] { return (!tasks_.empty() && index < limit_.load(std::memory_order_relaxed)) || stop_; });
                if (stop_ && (tasks_.empty() || index >= limit_.load(std::memory_order_relaxed))) return; // every context on the device is closed, nothing can be queued any more
                t = tasks_.front();
                tasks_.pop_front();
            }
            double t1 = g_timing ? now() : 0;
            int err = 0;
            trace("reader: task of %zu bytes taken", t.n);
            PinBlock *b = t.ahead >= 0 ? adopt_ahead_block(t.ahead) : nullptr;
            const bool filled = b != nullptr; // read ahead while the runtime started: nothing left to read
            if (!b) b = take_reader_block();
            trace("reader: block in hand");
            double t2 = g_timing ? now() : 0, t3 = t2;
            if (!b) {
                err = -2;
            } else {
                size_t got = filled ? t.n : 0;
                if (t.items) {
                    // whole small files, each opened, read and closed here: the worker that queued them has long gone on to the
                    // next batch.  A file that cannot be opened or has shrunk leaves its segment zero-filled and says so in
                    // its own err (gscan_last_file_errors): the batch goes on.
                    const uint64_t base = t.items[0].dst_off;
                    for (uint32_t k = 0; k < t.nitems; k++) {
                        FileItem &it = t.items[k];
                        char *at = (char *)b->p + (it.dst_off - base);
                        const int fd = it.path.empty() ? it.fd : open(it.path.c_str(), it.oflags);
                        size_t have = 0;
                        if (fd < 0) it.err = errno ? errno : EIO;
                        while (fd >= 0 && have < it.len && !it.err) {
                            const ssize_t r = pread(fd, at + have, it.len - have, (off_t)have);
                            if (r > 0) have += (size_t)r;
                            else if (r == 0) it.err = -1;
                            else if (errno != EINTR) it.err = errno;
                        }
                        if (have < it.len) memset(at + have, 0, it.len - have);
                        if (fd >= 0 && !it.path.empty()) close(fd);
                    }
                    got = t.n;
                }
                if (diag >= 2) got = t.n; // (GSCAN_DIAG=2, 3: nothing is read, the block goes out as it is)
                if (nt_copy_ && !t.items) {
                    // pread into a buffer that stays in this core's L2, stream it out to the block past the caches: the block's
                    // lines are never read into a cache for ownership (one DRAM crossing less per byte)
                    constexpr size_t kBounce = 256u << 10;
                    static thread_local char *bounce = nullptr;
                    if (!bounce && posix_memalign((void **)&bounce, 4096, kBounce) != 0) bounce = nullptr;
                    while (bounce && got < t.n && !err) {
                        const size_t want = std::min(kBounce, t.n - got);
                        size_t have = 0;
                        while (have < want && !err) {
                            const ssize_t r = pread(t.fd, bounce + have, want - have, t.off + (off_t)(got + have));
                            if (r > 0) have += (size_t)r;
                            else if (r == 0) err = -1;
                            else if (errno != EINTR) err = errno;
                        }
                        if (!err) gscan::nt_copy((char *)b->p + got, bounce, want);
                        got += want;
                    }
                }
                while (got < t.n && !err) {
                    const ssize_t r = pread(t.fd, (char *)b->p + got, t.n - got, t.off + (off_t)got);
                    if (r > 0) got += (size_t)r;
                    else if (r == 0) err = -1;
                    else if (errno != EINTR) err = errno;
                }
                t3 = g_timing ? now() : 0;
                trace("reader: %zu bytes read", t.n);
                if (!err && diag != 1) {
                    Lane *lane;
                    {
                        std::lock_guard<std::mutex> lk(m_);
                        lane = lane_for(t.stream);
                    }
                    std::lock_guard<std::mutex> ord(lane->order); // (enqueue + event + push: the lane's fifo is the stream's order)
                    if (hipMemcpyAsync(t.dst, b->p, t.n, hipMemcpyHostToDevice, t.stream) != hipSuccess || hipEventRecord(b->ev, t.stream) != hipSuccess) {
                        (void)hipGetLastError();
                        (void)hipStreamSynchronize(t.stream); // (whatever did get queued out of the block is over before it is reused)
                        err = -2;
                        give_reader_block(b, nullptr);
                    } else {
                        give_reader_block(b, lane);
                        if (g_timing && !first_dma_said_.exchange(true)) // (the ramp of a device: DESIGN.md 6's F(N) is made of these)
                            fprintf(stderr, "[gscan timing] device %d: first piece queued for DMA at +%.4f s\n", device_,
                                    std::chrono::duration<double>(std::chrono::steady_clock::now() - g_trace_t0).count());
                    }
                } else {
                    give_reader_block(b, nullptr); // (a failed read, or GSCAN_DIAG=1: the block is filled and given straight back)
                }
            }
            if (g_timing) {
                const double t4 = now();
                ns_idle_ += (uint64_t)((t1 - t0) * 1e9);
                ns_block_ += (uint64_t)((t2 - t1) * 1e9);
                ns_read_ += (uint64_t)((t3 - t2) * 1e9);
                ns_hip_ += (uint64_t)((t4 - t3) * 1e9);
                n_pieces_++;
                n_bytes_ += t.n;
            }
            bool last;
            {
                std::lock_guard<std::mutex> lk(t.grp->m);
                if (err && !t.grp->err) t.grp->err = err;
                last = --t.grp->pending == 0;
            }
            trace("reader: piece queued for DMA%s", last ? " (last of its range: launching)" : "");
            if (last) t.grp->finish(t.grp); // every piece is on its copy stream: launch the scan behind them
        }
    }

    // GSCAN_TIMING=1: where the reader threads spend their time, printed when a context closes
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    std::atomic<uint64_t> ns_idle_{0}, ns_block_{0}, ns_read_{0}, ns_hip_{0}, n_pieces_{0}, n_bytes_{0};
    void report()
    {
        fprintf(stderr, "[gscan timing] device %d readers %d (%d local CPUs) block %zu MiB pool %zu of %zu: pieces %llu bytes %llu | per reader: idle %.3f s  wait-for-block %.3f s  pread %.3f s  hip calls %.3f s | waits on an event %llu, on other readers %llu\n",
                device_, readers_, numa_cpus_, block_bytes() >> 20, n_alloc_, cap_, (unsigned long long)n_pieces_.load(), (unsigned long long)n_bytes_.load(), ns_idle_ / 1e9 / readers_,
                ns_block_ / 1e9 / readers_, ns_read_ / 1e9 / readers_, ns_hip_ / 1e9 / readers_, (unsigned long long)n_waits_ev_, (unsigned long long)n_waits_cv_);
    }
    int device_;
    int refs_ = 1;
    int readers_ = 8;
    std::atomic<int> limit_{8}; // readers that take tasks: readers_, fewer once the process drives more devices (relimit)
    int auto_cpus_ = 0, auto_sharing_ = 1; // what the automatic count was worked out from (0: the count was given)
    int numa_cpus_ = 0; // CPUs of the device's NUMA node the readers are bound to (0: not bound by NUMA)
    cpu_set_t mask_;
    bool have_mask_ = false;
    bool nt_copy_ = false;
    size_t cap_ = 16, n_alloc_ = 0;
    std::atomic<bool> first_dma_said_{false};
    std::atomic<size_t> handed_{0};             // bytes handed to this device so far (second_stream)
    std::atomic<hipStream_t> second_{nullptr};
    std::atomic<bool> second_started_{false};
    std::thread second_maker_;
    std::atomic<long> n_made_{0};           // staging blocks asked of the runtime so far (GSCAN_FAIL_ALLOC_AFTER)
    uint64_t n_waits_ev_ = 0, n_waits_cv_ = 0; // (under m_) how often the pool's slow paths ran: slept on a block's event / on the other readers
    bool started_ = false, stop_ = false;
    std::mutex m_, streams_m_;
    std::condition_variable cv_blocks_, cv_tasks_;
    std::vector<PinBlock *> free_, slot_free_;
    std::vector<std::unique_ptr<Lane>> lanes_;
    std::deque<ReadTask> tasks_;
    std::vector<std::thread> threads_;
    std::vector<hipStream_t> shared_;

public:
    // how often the pool's slow paths have run on this device (gscan_pool_stats: the tests that force them check that they did)
    void stats(uint64_t out[4])
    {
        std::lock_guard<std::mutex> lk(m_);
        out[0] = n_alloc_;
        out[1] = cap_;
        out[2] = n_waits_ev_;
        out[3] = n_waits_cv_;
    }
};

enum SlotState { FREE = 0, ACQUIRED, INFLIGHT };

struct Slot {
    SlotState state = FREE;
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    uint8_t *d_text = nullptr;
    size_t d_text_cap = 0;
    uint32_t *d_recs = nullptr;
    size_t rec_cap = 0;
    unsigned long long *d_desc = nullptr;
    size_t tiles_cap = 0;
    uint32_t *d_counter = nullptr;        // kCounterWords
    uint32_t *h_counter = nullptr;        // pinned, kCounterWords
    unsigned long long *h_desc = nullptr; // pinned
    size_t h_desc_cap = 0;
    uint32_t *h_spec = nullptr; // pinned, kShards rows of kSpecPer
    std::vector<uint32_t> sorted;
    // the result in text order, made on the device (k_order_*): [0, rec_cap) records, from rec_cap on their extras.  The host
    // fetches one linear range of each and hands it out as it is (out_starts / out_ext point into pinned memory then)
    uint32_t *d_sorted = nullptr;
    size_t sorted_cap = 0; // words
    uint32_t *d_dpos = nullptr;
    size_t dpos_cap = 0;
    bool ordered = false;
    const uint32_t *out_starts = nullptr, *out_ext = nullptr;
    // Speculative readback of an ordered result, issued with the launch: the first spec_n records + their extras and the
    // first gspec_n bytes of gathered line text, sized by what this context's last chunks produced (+25 %).  A chunk whose
    // result fits needs no transfer after its scan at all -- fetching it then costs a blocking round trip of ~2-3 ms behind
    // the 64 MiB window copies that are always in flight.
    uint32_t *h_big = nullptr; // pinned: [big_recs] records, then [big_recs * big_ew] extras
    size_t big_recs = 0, spec_n = 0;
    uint32_t big_ew = 0; // extras per record h_big has room for
    uint8_t *h_gspec = nullptr; // pinned
    size_t gspec_cap = 0, gspec_n = 0;
    const uint8_t *out_gather = nullptr;
    // per-record extras, parallel to the records: line extents (k_lines, 3 words each) or match ends (k_ends, 1 word)
    uint32_t *d_ext = nullptr;
    size_t ext_cap = 0;          // records
    uint32_t *h_ext_spec = nullptr; // pinned, kShards rows of kSpecPer * 3
    std::vector<uint32_t> sorted_ext;
    bool has_ext = false;
    bool resolved = false;      // the chunk went through k_resolve: the records are match starts, the extras their ends, the dropped ones are counted as struck
    uint32_t ext_words = 0;     // 4: {m1, lb, le, goff} per record ("line_extents"), 1: the match end ("match_ends"), 0: none
    // the text of the printed lines, gathered by k_lines ("line_extents"): device buffer, pinned copy of the used part
    uint8_t *d_gather = nullptr;
    size_t gather_cap = 0;
    size_t gather_bytes = 0;
    bool gather_ok = false;
    const void *ext = nullptr;  // caller's buffer this chunk was copied from (gscan_wait hands it back as *content)
    void *ext_reg = nullptr;    // ... registered with the runtime for direct DMA until the scan is done
    PinBlock *blk = nullptr; // pool block serving as this slot's pinned buffer (acquires of <= block_bytes() bytes)
    bool no_content = false;    // gscan_submit_fd: the bytes never sat in a host buffer of ours
    // a batch of small files read by the device's readers (gscan_submit_files): segment i is file i
    std::vector<FileItem> files;
    std::vector<int> file_err; // per segment, filled by gscan_wait_segs (gscan_last_file_errors)
    // multi-segment chunks (gscan_submit_segs, gscan_submit_files)
    std::vector<gscan_seg> segs;
    std::vector<uint32_t> tile_first; // first tile of segment i; [nseg] = n_tiles
    std::vector<size_t> seg_first;    // result: first record of segment i in `sorted`; [nseg] = total
    gscan::TileDesc *d_tiles = nullptr, *h_tiles = nullptr;
    size_t seg_tiles_cap = 0;
    hipEvent_t done = nullptr;
    hipEvent_t copied = nullptr, copied2 = nullptr; // behind a chunk's copies on the first copy stream (only when the scans have a stream of their own) / behind its pieces on the second
    hipStream_t second = nullptr;                  // the second copy stream, if this chunk's pieces use it
    std::unique_ptr<ReadGroup> grp;                       // gscan_submit_fd: the range's pieces (finished == true when idle)
    uint64_t tag = 0;
    size_t len = 0;
    uint32_t n_tiles = 0;
    uint32_t nw = 1; // waves per workgroup of the launch: descriptors per tile
    const gscan_db *db = nullptr;
    uint64_t seq = 0;
};

struct EvPair {
    hipEvent_t a, b;
};

} // namespace

struct gscan_ctx {
    int device = 0;  // the index gscan_open was given (what the caller counts in)
    int hip_dev = 0; // the HIP device behind it (the same, unless GSCAN_VIRTUAL_DEVICES maps several indices onto one)
    int cus = 256;
    size_t max_chunk = 0;
    // the device's streams (they belong to its Ingest, not to this context): GSCAN_COPY_STREAMS copy streams, the scans and
    // read-backs on the first of them
    hipStream_t copy = nullptr, compute = nullptr;
    Ingest *ingest = nullptr;
    Slot slot[GSCAN_SLOTS];
    uint64_t next_seq = 1;
    std::string err;
    // compiled program on the device
    void *h_arena = nullptr, *d_arena = nullptr; // one pinned and one device allocation behind all the small buffers below
    DevProgram *d_prog = nullptr;
    DevProgram *h_prog = nullptr; // pinned staging
    uint64_t prog_id = 0;
    uint32_t rec_permille = 0; // of the database uploaded last: DevProgram::est_permille if every hit of its windows becomes a record (resolve without K3's own VM in front), else 0
    hipEvent_t prog_ev = nullptr;      // behind the last upload of the program ...
    hipStream_t prog_stream = nullptr; // ... on this stream
    bool prog_pending = false;
    // options
    // defaults from the sweeps under profiles/: 12 KiB per wave (110 VGPRs -> 4 waves/SIMD) with
    // nontemporal loads, one workgroup per tile
    int k3_depth = 0; // option "k3_depth"
    int variant = 38; // 12 KiB per wave, nontemporal loads, K2's lane-table form for windows of <= 17 bytes
    int blocks_per_cu = 0;
    size_t register_min = 1u << 20; // caller buffers of at least this many bytes are registered and DMA'd in place
    bool line_extents = false;      // run k_lines after the scan when the pattern allows it ("line_extents" option)
    bool match_ends = false;        // run k_ends after the scan when the pattern allows it ("match_ends" option): -O -l without the text
    const Slot *last_waited = nullptr;
    // Pinned staging of what gscan_wait fetches AFTER the scan has finished and its sizes are known -- dense record lists (some
    // shard holds more than kSpecPer records: the used part of every shard region, one strided DMA each for records and
    // extras; a pageable destination made the runtime stage 64 short rows one by one, ~1.5 ms per copy) and the gathered
    // line text.  One of each per CONTEXT, not per slot, and grown in big steps: a pinned allocation costs milliseconds and
    // the runtime serialises them -- 24 slots growing their own buffers were a third of a 16 GiB run's wait time.
    uint32_t *h_dense = nullptr;
    size_t dense_cap = 0; // words
    uint8_t *h_gather = nullptr; // valid until the next gscan_wait* on this context
    size_t h_gather_cap = 0;
    std::atomic<size_t> hint_total{0}, hint_gather{0}; // what recent chunks produced (records, gathered bytes): sizes the speculative readback (written by the owner in gscan_wait, read by whichever thread launches: relaxed)
    // GSCAN_TIMING: where gscan_wait's time goes (seconds, this context)
    double tw_reads = 0, tw_scan = 0, tw_dense = 0, tw_gather = 0, tw_merge = 0, tw_alloc = 0;
    size_t tw_n = 0, tw_dense_bytes = 0, tw_gather_bytes = 0;
    // device-resident path
    size_t dev_cap_req = 0;
    uint32_t *dv_recs = nullptr;
    size_t dv_rec_cap = 0;
    uint32_t *dv_ends = nullptr; // the resolve pass's extras, parallel to dv_recs (databases with `resolve` only)
    size_t dv_ends_cap = 0;
    unsigned long long *dv_desc = nullptr;
    gscan::TileDesc *dv_tiles = nullptr;
    size_t dv_tiles_cap = 0;
    uint32_t *dv_counter = nullptr;
    std::vector<gscan_seg> dv_last_segs;
    std::vector<uint32_t> dv_tile_first_h;
    uint32_t dv_last_tile_bytes = 0;
    uint32_t dv_last_nw = 1;
    hipStream_t dv_stream = nullptr;
    // kernel timing
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
};

namespace {

// A reader thread that launches a scan on behalf of a context (fd_finish) must not write the context's error text while
// the owning thread may be reading it: it points this at the read group's own string for the duration.
thread_local std::string *t_err_sink = nullptr;

int fail(gscan_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (t_err_sink) *t_err_sink = buf;
    else if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return fail((c), GSCAN_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

void slot_drain_reads(Slot &s);

// (before a slot is sized for a chunk of this database: slot_reserve_device)
void note_db(gscan_ctx *c, const gscan_db *db) { c->rec_permille = db->db.prog.resolve && !db->db.prog.vm_filter ? db->db.prog.est_permille : 0u; }

int ensure_prog(gscan_ctx *c, const gscan_db *db, hipStream_t st)
{
    if (c->prog_id == db->db.id) {
        // uploaded on another stream and perhaps still on its way: this stream's kernels wait for it on the device
        if (c->prog_pending && st != c->prog_stream) HIPCHK(c, hipStreamWaitEvent(st, c->prog_ev, 0));
        return 0;
    }
    // A file range submitted with ANOTHER database may still be on its way in: its scan is launched later, by the reader
    // that finishes its last piece, and reads c->d_prog then.  Let every such range arrive and launch first -- the device
    // program is one per context, not one per slot.
    for (Slot &s : c->slot)
        if (s.state == INFLIGHT) slot_drain_reads(s);
    if (c->prog_id != 0) {
        // a pattern switch (rare: one per FileGrep::prepare): the staging copy is overwritten and so is the device copy that
        // earlier scans may still be reading -- everything that uses either has to be over
        if (c->prog_pending) HIPCHK(c, hipEventSynchronize(c->prog_ev));
        HIPCHK(c, hipStreamSynchronize(c->compute));
        if (c->dv_stream && c->dv_stream != c->compute) HIPCHK(c, hipStreamSynchronize(c->dv_stream));
    }
    memcpy(c->h_prog, &db->db.prog, sizeof(DevProgram));
    // NOT waited for: the scans that use it follow on the same stream (others wait for prog_ev on the device).  The first
    // transfer of a process costs the runtime 12 - 25 ms of set-up (profiles/r04_b_*): it now runs while the caller sizes its
    // slot and the readers read the first pieces of the file, instead of in front of them.
    HIPCHK(c, hipMemcpyAsync(c->d_prog, c->h_prog, sizeof(DevProgram), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->prog_ev, st));
    c->prog_stream = st;
    c->prog_pending = true;
    c->prog_id = db->db.id;
    return 0;
}

uint32_t grid_for(const gscan_ctx *c, const Database &db, uint32_t n_tiles)
{
    // kernels that stage a big LDS table once per workgroup always run as a persistent grid
    const uint32_t fixed = gscan::scan_persistent_blocks(db.tier, c->variant, db.prog);
    const uint32_t bpc = fixed ? fixed : (uint32_t)std::max(c->blocks_per_cu, 0);
    if (bpc == 0) return n_tiles;
    uint64_t g = (uint64_t)c->cus * (uint64_t)bpc;
    return (uint32_t)std::min<uint64_t>(g, n_tiles);
}

int slot_reserve_pinned(gscan_ctx *c, Slot &s, size_t len)
{
    if (len > s.pinned_cap) {
        if (s.pinned) hipHostFree(s.pinned);
        s.pinned = nullptr;
        s.pinned_cap = 0;
        size_t cap = std::max<size_t>((len + kPad + 0xfffff) & ~(size_t)0xfffff, 1u << 20);
        HIPCHK(c, hipHostMalloc(&s.pinned, cap, hipHostMallocDefault));
        s.pinned_cap = cap - kPad;
    }
    return 0;
}

int slot_reserve_device(gscan_ctx *c, Slot &s, size_t len)
{
    if (len > s.d_text_cap) {
        if (s.d_text) hipFree(s.d_text);
        s.d_text = nullptr;
        s.d_text_cap = 0;
        size_t cap = std::max<size_t>((len + kPad + 0xfffff) & ~(size_t)0xfffff, 1u << 20);
        // (the line pass's gather buffer -- as big again: the printed lines never overlap, their text fits the window -- is NOT
        // made here: round 3 took both out of one allocation, and a file that prints nothing paid for 514 MiB where it needed
        // 257, 15 ms of BASELINE configs[0]'s 100.  slot_launch makes it once this context's windows have had records.)
        trace("slot: text buffer, %zu MiB ...", cap >> 20);
        HIPCHK(c, hipMalloc((void **)&s.d_text, cap));
        trace("slot: ... allocated");
        s.d_text_cap = cap - kPad;
    }
    // descriptors: one per wave sub-tile, at the smallest sub-tile any variant uses (+ the padding of a last, partial tile)
    size_t tiles = len / gscan::kMinSubTileBytes + 2 * gscan::kMaxWavesPerTile;
    if (tiles > s.tiles_cap) {
        if (s.d_desc) hipFree(s.d_desc);
        if (s.h_desc) hipHostFree(s.h_desc);
        s.d_desc = nullptr;
        s.h_desc = nullptr;
        s.tiles_cap = 0;
        size_t cap = tiles + tiles / 4;
        HIPCHK(c, hipMalloc((void **)&s.d_desc, cap * 8));
        trace("slot: descriptors on the device");
        HIPCHK(c, hipHostMalloc((void **)&s.h_desc, cap * 8, hipHostMallocDefault));
        trace("slot: descriptors pinned (%zu KiB)", cap * 8 >> 10);
        s.tiles_cap = cap;
    }
    // records: one per 64 bytes of text to begin with (a shard that overflows makes the host regrow and rescan) -- more from the
    // start for a database whose every window hit is a record (the resolve pass's: no group-start compression) when its windows
    // are expected to hit often: twice the compiler's estimate, so that the first window of every slot is not scanned twice
    size_t per = len / 64;
    if (c->rec_permille > 16) per = std::max(per, (size_t)((double)len * std::min(250u, 2 * c->rec_permille) / 1000.0)); // (at most a record per four bytes up front: denser still, and the regrow path sizes it)
    size_t want = std::max<size_t>((per + gscan::kShards - 1) / gscan::kShards * gscan::kShards, kSpecRecs);
    if (want > s.rec_cap) {
        if (s.d_recs) hipFree(s.d_recs);
        s.d_recs = nullptr;
        s.rec_cap = 0;
        HIPCHK(c, hipMalloc((void **)&s.d_recs, want * 4));
        s.rec_cap = want;
    }
    return 0;
}

int slot_reserve(gscan_ctx *c, Slot &s, size_t len)
{
    int rc = slot_reserve_pinned(c, s, len);
    return rc ? rc : slot_reserve_device(c, s, len);
}

void slot_unregister(Slot &s)
{
    if (s.ext_reg) {
        (void)hipHostUnregister(s.ext_reg);
        s.ext_reg = nullptr;
    }
}

// Tile table of a multi-segment chunk: built in pinned memory, copied on the copy stream ahead of
// the `copied` event.  Single-segment chunks need none (the kernel tiles segment 0 in order).
int slot_build_tiles(gscan_ctx *c, Slot &s, uint32_t tile_bytes)
{
    const size_t nseg = s.segs.size();
    s.tile_first.assign(nseg + 1, 0);
    uint64_t nt = 0;
    for (size_t i = 0; i < nseg; i++) {
        s.tile_first[i] = (uint32_t)nt;
        nt += (s.segs[i].len + tile_bytes - 1) / tile_bytes;
    }
    s.tile_first[nseg] = (uint32_t)nt;
    if (nt + 1 > s.seg_tiles_cap) {
        if (s.d_tiles) hipFree(s.d_tiles);
        if (s.h_tiles) hipHostFree(s.h_tiles);
        s.d_tiles = nullptr;
        s.h_tiles = nullptr;
        s.seg_tiles_cap = 0;
        const size_t cap = (size_t)nt + nt / 2 + 64;
        HIPCHK(c, hipMalloc((void **)&s.d_tiles, cap * sizeof(gscan::TileDesc)));
        HIPCHK(c, hipHostMalloc((void **)&s.h_tiles, cap * sizeof(gscan::TileDesc), hipHostMallocDefault));
        s.seg_tiles_cap = cap;
    }
    for (size_t i = 0; i < nseg; i++)
        for (uint32_t t = s.tile_first[i]; t < s.tile_first[i + 1]; t++)
            s.h_tiles[t] = {s.segs[i].offset, s.segs[i].len, (t - s.tile_first[i]) * tile_bytes};
    if (nt) HIPCHK(c, hipMemcpyAsync(s.d_tiles, s.h_tiles, (size_t)nt * sizeof(gscan::TileDesc), hipMemcpyHostToDevice, c->copy));
    // descriptors: one per wave of every tile; make room (a batch of tiny files has far more tiles than len / tile size)
    const size_t nd = ((size_t)nt + 2) * gscan::kMaxWavesPerTile;
    if (nd > s.tiles_cap) {
        if (s.d_desc) hipFree(s.d_desc);
        if (s.h_desc) hipHostFree(s.h_desc);
        s.d_desc = nullptr;
        s.h_desc = nullptr;
        s.tiles_cap = 0;
        const size_t cap = nd + nd / 2 + 64;
        HIPCHK(c, hipMalloc((void **)&s.d_desc, cap * 8));
        HIPCHK(c, hipHostMalloc((void **)&s.h_desc, cap * 8, hipHostMallocDefault));
        s.tiles_cap = cap;
    }
    return 0;
}

// A free slot for a new chunk.  The slot the last gscan_wait* handed out still backs what the caller was given (starts, line
// extents, gathered text live in its pinned buffers): it is the last choice.
Slot *free_slot_for_submit(gscan_ctx *c)
{
    Slot *s = nullptr;
    for (Slot &x : c->slot)
        if (x.state == FREE && (!s || s == c->last_waited)) s = &x;
    return s;
}

int slot_launch(gscan_ctx *c, Slot &s)
{
    const Database &db = s.db->db;
    uint32_t tile_bytes = 0, nw = 1;
    gscan::scan_geometry(db.tier, c->variant, db.prog, &tile_bytes, &nw);
    const bool multi = !s.segs.empty();
    s.n_tiles = multi ? s.tile_first.back() : (uint32_t)((s.len + tile_bytes - 1) / tile_bytes);
    s.nw = nw;
    if ((size_t)s.n_tiles * nw > s.tiles_cap) return fail(c, GSCAN_EHIP, "descriptor buffer too small (%u tiles x %u waves)", s.n_tiles, nw);
    HIPCHK(c, hipMemsetAsync(s.d_counter, 0, kCounterWords * 4, c->compute));
    ScanArgs a;
    memset(&a, 0, sizeof a);
    a.base = s.d_text;
    a.tiles = multi ? s.d_tiles : nullptr; // nullptr: one segment, tiled in order
    a.seg0_off = 0;
    a.seg0_len = (uint32_t)s.len;
    a.n_tiles = s.n_tiles;
    a.cap_shard = (uint32_t)(s.rec_cap / gscan::kShards);
    a.recs = s.d_recs;
    a.desc = s.d_desc;
    a.counter = s.d_counter;
    a.prog = c->d_prog;
    gscan::fill_program(a, db.prog);
    if (c->k3_depth) a.k3_depth = (uint32_t)c->k3_depth;
    if (s.n_tiles) HIPCHK(c, gscan::launch_scan(db.tier, c->variant, a, grid_for(c, db, s.n_tiles), c->compute));
    if (s.n_tiles && gscan::scan_needs_settle(db.tier, db.prog)) HIPCHK(c, gscan::launch_settle(a, nw, c->compute));
    // (a database the device resolves always comes with its matches' ends: the list means nothing else)
    s.resolved = s.n_tiles && db.prog.resolve;
    s.ext_words = !s.n_tiles ? 0u : s.resolved ? 1u : (c->line_extents && db.prog.lines_ok) ? 4u : (c->match_ends && db.prog.ends_ok) ? 1u : 0u;
    s.has_ext = s.ext_words != 0;
    if (s.has_ext) {
        const size_t eb = 4 * (size_t)s.ext_words; // bytes per record
        if (s.ext_cap < s.rec_cap * s.ext_words) {
            if (s.d_ext) hipFree(s.d_ext);
            s.d_ext = nullptr;
            s.ext_cap = 0;
            HIPCHK(c, hipMalloc((void **)&s.d_ext, s.rec_cap * eb));
            s.ext_cap = s.rec_cap * s.ext_words;
        }
        if (!s.h_ext_spec) HIPCHK(c, hipHostMalloc((void **)&s.h_ext_spec, kSpecRecs * 16, hipHostMallocDefault));
        if (s.ext_words == 4) {
            // the printed lines never overlap (the loop restarts at the end of the line it printed): their text fits the window
            // ... once this context's windows have had records to print (hint_total: what its last chunks produced): until then
            // the pass runs without a gather buffer -- every printed line is marked "take the text from the chunk" -- which is what a
            // window without records costs nothing and the first windows of a dense output a few page faults
            const size_t want = std::min<size_t>(std::max<size_t>(s.len, 1u << 20), 0xfffffff0u);
            if (s.gather_cap < want && c->hint_total.load(std::memory_order_relaxed) > 0) {
                if (s.d_gather) hipFree(s.d_gather);
                s.d_gather = nullptr;
                s.gather_cap = 0;
                HIPCHK(c, hipMalloc((void **)&s.d_gather, want + want / 4));
                s.gather_cap = std::min<size_t>(want + want / 4, 0xfffffff0u);
            }
            const bool have_gather = s.gather_cap >= want;
            HIPCHK(c, gscan::launch_lines(a, nw, tile_bytes / nw, s.d_ext, have_gather ? s.d_gather : nullptr, have_gather ? (uint32_t)s.gather_cap : 0u, c->compute));
        } else if (s.resolved) {
            HIPCHK(c, gscan::launch_resolve(a, nw, s.d_ext, c->compute));
        } else {
            HIPCHK(c, gscan::launch_ends(a, nw, tile_bytes / nw, s.d_ext, c->compute));
        }
    }
    // The result once more in text order (not where a second pass may still strike records out: k3_settle's patterns keep the
    // host's merge over the shard regions).
    s.ordered = s.n_tiles && !gscan::scan_needs_settle(db.tier, db.prog);
    if (s.ordered) {
        const size_t n_desc = (size_t)s.n_tiles * nw, need = s.rec_cap * (1 + (size_t)s.ext_words);
        if (s.dpos_cap < n_desc) {
            if (s.d_dpos) hipFree(s.d_dpos);
            s.d_dpos = nullptr;
            s.dpos_cap = 0;
            HIPCHK(c, hipMalloc((void **)&s.d_dpos, (n_desc + n_desc / 4 + 64) * 4));
            s.dpos_cap = n_desc + n_desc / 4 + 64;
        }
        if (s.sorted_cap < need) {
            if (s.d_sorted) hipFree(s.d_sorted);
            s.d_sorted = nullptr;
            s.sorted_cap = 0;
            HIPCHK(c, hipMalloc((void **)&s.d_sorted, need * 4));
            s.sorted_cap = need;
        }
        HIPCHK(c, gscan::launch_order(a, nw, s.d_ext, s.ext_words, s.d_dpos, s.d_sorted, s.d_sorted + s.rec_cap, c->compute));
    }
    HIPCHK(c, hipMemcpyAsync(s.h_counter, s.d_counter, kCounterWords * 4, hipMemcpyDeviceToHost, c->compute));
    if (s.n_tiles)
        HIPCHK(c, hipMemcpyAsync(s.h_desc, s.d_desc, (size_t)s.n_tiles * nw * 8, hipMemcpyDeviceToHost, c->compute));
    s.spec_n = s.gspec_n = 0;
    if (s.ordered) { // the head of the ordered result: kSpecRecs records (enough for any sparse one), or what the last chunks suggest
        const size_t hint_t = c->hint_total.load(std::memory_order_relaxed), hint_g = c->hint_gather.load(std::memory_order_relaxed);
        const size_t want = std::min<size_t>(s.rec_cap, hint_t + hint_t / 4);
        if (want > kSpecRecs) {
            if (want > s.big_recs || s.ext_words > s.big_ew) { // (sized for the extras this context asks for: 1 word for match ends, 4 for line extents, none otherwise -- not for the most there could be)
                const size_t had = s.big_recs;
                if (s.h_big) hipHostFree(s.h_big);
                s.h_big = nullptr;
                s.big_recs = 0;
                const size_t cap = std::max(want + want / 4, had);
                HIPCHK(c, hipHostMalloc((void **)&s.h_big, cap * 4 * (1 + (size_t)s.ext_words), hipHostMallocDefault));
                s.big_recs = cap;
                s.big_ew = s.ext_words;
            }
            s.spec_n = want;
            HIPCHK(c, hipMemcpyAsync(s.h_big, s.d_sorted, want * 4, hipMemcpyDeviceToHost, c->compute));
            if (s.has_ext) HIPCHK(c, hipMemcpyAsync(s.h_big + s.big_recs, s.d_sorted + s.rec_cap, want * 4 * s.ext_words, hipMemcpyDeviceToHost, c->compute));
        } else {
            s.spec_n = std::min<size_t>(kSpecRecs, s.rec_cap);
            HIPCHK(c, hipMemcpyAsync(s.h_spec, s.d_sorted, s.spec_n * 4, hipMemcpyDeviceToHost, c->compute));
            if (s.has_ext) HIPCHK(c, hipMemcpyAsync(s.h_ext_spec, s.d_sorted + s.rec_cap, s.spec_n * 4 * s.ext_words, hipMemcpyDeviceToHost, c->compute));
        }
        if (s.ext_words == 4 && hint_g && s.gather_cap) { // ... and of the gathered line text
            const size_t gw = std::min<size_t>(std::min<size_t>(s.gather_cap, kGatherPinnedMax), hint_g + hint_g / 4);
            if (gw > s.gspec_cap) {
                if (s.h_gspec) hipHostFree(s.h_gspec);
                s.h_gspec = nullptr;
                s.gspec_cap = 0;
                HIPCHK(c, hipHostMalloc((void **)&s.h_gspec, gw + gw / 4, hipHostMallocDefault));
                s.gspec_cap = gw + gw / 4;
            }
            s.gspec_n = gw;
            HIPCHK(c, hipMemcpyAsync(s.h_gspec, s.d_gather, gw, hipMemcpyDeviceToHost, c->compute));
        }
    } else {
        // the head of every shard region in one strided copy: enough for any sparse result
        HIPCHK(c, hipMemcpy2DAsync(s.h_spec, kSpecPer * 4, s.d_recs, (size_t)a.cap_shard * 4, kSpecPer * 4, gscan::kShards,
                                   hipMemcpyDeviceToHost, c->compute));
        if (s.has_ext)
            HIPCHK(c, hipMemcpy2DAsync(s.h_ext_spec, kSpecPer * 4 * s.ext_words, s.d_ext, (size_t)a.cap_shard * 4 * s.ext_words, kSpecPer * 4 * s.ext_words,
                                       gscan::kShards, hipMemcpyDeviceToHost, c->compute));
    }
    HIPCHK(c, hipEventRecord(s.done, c->compute));
    return 0;
}

void free_slot(gscan_ctx *c, Slot &s)
{
    slot_unregister(s);
    if (s.blk && c->ingest) c->ingest->give_slot_block(s.blk); // back to the process-wide cache
    if (s.d_tiles) hipFree(s.d_tiles);
    if (s.h_tiles) hipHostFree(s.h_tiles);
    if (s.d_ext) hipFree(s.d_ext);
    if (s.h_ext_spec) hipHostFree(s.h_ext_spec);
    if (s.d_sorted) hipFree(s.d_sorted);
    if (s.d_dpos) hipFree(s.d_dpos);
    if (s.h_big) hipHostFree(s.h_big);
    if (s.h_gspec) hipHostFree(s.h_gspec);
    if (s.d_gather) hipFree(s.d_gather);
    if (s.pinned) hipHostFree(s.pinned);
    if (s.d_text) hipFree(s.d_text);
    if (s.d_recs) hipFree(s.d_recs);
    if (s.d_desc) hipFree(s.d_desc);
    if (s.h_desc) hipHostFree(s.h_desc);
    if (s.copied) hipEventDestroy(s.copied);
    if (s.copied2) hipEventDestroy(s.copied2);
    if (s.done) hipEventDestroy(s.done);
    s = Slot();
}

// gscan_submit_fd, second half: run by the reader thread that finished the range's last piece.  Every piece is on one of
// the context's copy streams by now; the scan goes behind them.
void fd_finish(ReadGroup *g)
{
    gscan_ctx *c = (gscan_ctx *)g->ctx;
    Slot &s = *(Slot *)g->slot;
    t_err_sink = &g->msg;
    int rc = [&]() -> int {
        if (g->err) {
            HIPCHK(c, hipStreamSynchronize(c->copy)); // pieces that did make it must not land after the slot is reused
            if (s.second) HIPCHK(c, hipStreamSynchronize(s.second));
            if (g->err == -1) return fail(c, GSCAN_EIO, "file shrank while reading");
            if (g->err == -2) return fail(c, GSCAN_EHIP, "staging block or DMA failed");
            return fail(c, GSCAN_EIO, "%s", strerror(g->err));
        }
        if (ingest_cfg().diag == 1 || ingest_cfg().diag == 3) { // nothing went to the device (or nothing is to be made of it): an empty result behind whatever the stream still holds
            memset(s.h_counter, 0, kCounterWords * 4);
            s.n_tiles = 0;
            s.nw = 1;
            s.ordered = s.has_ext = false;
            s.ext_words = 0;
            s.spec_n = s.gspec_n = 0;
            HIPCHK(c, hipEventRecord(s.done, c->compute));
            return 0;
        }
        // (the pieces on the first copy stream are in front of the scan as it is when the scans ride on that stream)
        if (c->compute != c->copy) {
            HIPCHK(c, hipEventRecord(s.copied, c->copy));
            HIPCHK(c, hipStreamWaitEvent(c->compute, s.copied, 0));
        }
        if (s.second) {
            HIPCHK(c, hipEventRecord(s.copied2, s.second));
            HIPCHK(c, hipStreamWaitEvent(c->compute, s.copied2, 0));
        }
        return slot_launch(c, s);
    }();
    t_err_sink = nullptr;
    std::lock_guard<std::mutex> lk(g->m);
    g->rc = rc;
    g->finished = true;
    g->cv.notify_all();
}

// the owner waits for a slot's file range to be read and its scan to be launched
void slot_drain_reads(Slot &s)
{
    if (!s.grp) return;
    std::unique_lock<std::mutex> lk(s.grp->m);
    s.grp->cv.wait(lk, [&] { return s.grp->finished; });
}

} // namespace

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

int gscan_compile(const char *pat, size_t len, unsigned flags, gscan_db **out, int *minlen, char *err,
                  size_t errcap)
{
    if (!pat || !out) return GSCAN_EINVAL;
    gscan_db *d = new (std::nothrow) gscan_db();
    if (!d) return GSCAN_ENOMEM;
    std::string why;
    int rc = gscan::compile_pattern(pat, len, flags, d->db, why);
    if (rc != 0) {
        if (err && errcap) snprintf(err, errcap, "%s", why.c_str());
        delete d;
        *out = nullptr;
        return rc < 0 ? GSCAN_EINVAL : GSCAN_UNSUPPORTED;
    }
    if (minlen) *minlen = d->db.minlen;
    *out = d;
    return GSCAN_OK;
}

void gscan_free(gscan_db *db) { delete db; }

int gscan_db_info(const gscan_db *db, gscan_info *info)
{
    if (!db || !info) return GSCAN_EINVAL;
    const Database &d = db->db;
    info->tier = d.tier;
    info->minlen = d.minlen;
    info->n_classes = (int)d.classes.size();
    info->has_tail = !d.alts.empty() && d.alts[0].has_tail;
    info->tail_extra = d.alts.empty() ? 0 : d.alts[0].tail_extra;
    info->anchor_off = (int)d.prog.anchor_off;
    info->anchor_len = (int)d.prog.anchor_len;
    info->is_literal = (int)d.prog.is_literal;
    info->n_alts = (int)d.alts.size();
    info->has_context = (d.dev_pre ? 1 : 0) | (d.dev_post ? 2 : 0);
    info->lines_ok = (int)d.prog.lines_ok;
    info->gapped = 0;
    for (const gscan::AltSeq &a : d.alts) info->gapped += a.gapped ? 1 : 0;
    info->ends_ok = (int)d.prog.ends_ok;
    info->textfree = d.solitary && !d.alts.empty() && !d.alts[0].has_tail ? 1 : 0;
    info->exact = d.exact ? 1 : 0;
    info->vm = d.prog.vm_filter ? 1 : 0;
    info->resolve = d.resolve ? 1 : 0;
    info->reach = (int)d.reach;
    info->n_windows = (int)d.dev_windows.size();
    return GSCAN_OK;
}

int gscan_db_alt_class(const gscan_db *db, int alt, int pos, uint8_t table[256], int *len)
{
    if (!db || !table) return GSCAN_EINVAL;
    const Database &d = db->db;
    if (alt < 0 || (size_t)alt >= d.alts.size()) return GSCAN_EINVAL;
    const gscan::AltSeq &a = d.alts[(size_t)alt];
    const gscan::ByteSet *s = nullptr;
    if (pos == -1) {
        if (!a.has_tail) return GSCAN_EINVAL;
        s = &a.tail;
    } else {
        if (pos < 0 || (size_t)pos >= a.window.size()) return GSCAN_EINVAL;
        s = &d.classes[a.window[(size_t)pos]];
    }
    for (int b = 0; b < 256; b++) table[b] = s->test((unsigned)b);
    if (len) *len = (int)a.window.size();
    return GSCAN_OK;
}

int gscan_db_class(const gscan_db *db, int pos, uint8_t table[256]) { return gscan_db_alt_class(db, 0, pos, table, nullptr); }

int gscan_device_count(void) { return device_count(); }

int gscan_open(int hip_device, size_t max_chunk, gscan_ctx **out)
{
    if (!out) return GSCAN_EINVAL;
    *out = nullptr;
    if (max_chunk == 0 || max_chunk > kMaxChunk) return GSCAN_ETOOBIG;
    const bool tm = g_timing;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char *what) {
        if (tm) {
            const double t1 = now();
            fprintf(stderr, "[gscan timing] gscan_open: %-28s %.4f s\n", what, t1 - t0);
            t0 = t1;
        }
    };
    const int n = device_count();
    if (n <= 0) return GSCAN_EHIP; // no device: there is no CPU path
    lap("hipGetDeviceCount (runtime init)");
    if (hip_device < 0 || hip_device >= n) return GSCAN_EINVAL;
    if (virtual_devices() && ingest_cfg().virtual_fail_open == hip_device) return GSCAN_EHIP;
    gscan_ctx *c = new (std::nothrow) gscan_ctx();
    if (!c) return GSCAN_ENOMEM;
    c->device = hip_device;
    c->hip_dev = hip_device_of(hip_device);
    c->max_chunk = max_chunk;
    auto bail = [&](int rc) {
        gscan_close(c);
        return rc;
    };
    if (hipSetDevice(c->hip_dev) != hipSuccess) return bail(GSCAN_EHIP);
    lap("hipSetDevice");
    c->ingest = Ingest::acquire(hip_device);
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->hip_dev) == hipSuccess && cus > 0) c->cus = cus;
    lap("device attribute");
    // The device's streams (Ingest::shared_stream): the first context of a device creates them -- ~6 ms each, one at a time
    // per DEVICE -- the others find them there.  Contexts on different devices do not wait for each other (until round 5 one
    // process-wide lock put every open in line: on an eight-GPU node the eighth device's first byte waited for fifteen
    // stream creations that were none of its business).
    if (!(c->copy = c->ingest->shared_stream(0))) return bail(GSCAN_EHIP);
    c->compute = c->copy; // the scans and read-backs ride on the first copy stream
    if (ingest_cfg().scan_stream && !(c->compute = c->ingest->shared_stream(2))) return bail(GSCAN_EHIP);
    lap("streams");
    // one pinned allocation for every small host-side buffer of the context (each hipHostMalloc costs about a millisecond)
    const size_t kHead = (kCounterWords * 4 + 63) & ~size_t(63); // the slot's counter words
    const size_t per_slot = kHead + kSpecRecs * 4;
    const size_t host_bytes = ((sizeof(DevProgram) + 63) & ~size_t(63)) + GSCAN_SLOTS * per_slot;
    if (hipHostMalloc((void **)&c->h_arena, host_bytes, hipHostMallocDefault) != hipSuccess) return bail(GSCAN_EHIP);
    lap("pinned arena");
    const size_t kDevHead = (kCounterWords * 4 + 255) & ~size_t(255);
    const size_t dev_bytes = ((sizeof(DevProgram) + 255) & ~size_t(255)) + (GSCAN_SLOTS + 1) * kDevHead;
    if (hipMalloc((void **)&c->d_arena, dev_bytes) != hipSuccess) return bail(GSCAN_EHIP);
    lap("device arena");
    {
        char *h = (char *)c->h_arena, *d = (char *)c->d_arena;
        c->h_prog = (DevProgram *)h;
        h += (sizeof(DevProgram) + 63) & ~size_t(63);
        c->d_prog = (DevProgram *)d;
        d += (sizeof(DevProgram) + 255) & ~size_t(255);
        for (Slot &s : c->slot) {
            s.h_counter = (uint32_t *)h;
            s.h_spec = (uint32_t *)(h + kHead);
            h += per_slot;
            s.d_counter = (uint32_t *)d;
            d += kDevHead;
        }
        c->dv_counter = (uint32_t *)d;
    }
    if (hipEventCreateWithFlags(&c->prog_ev, hipEventDisableTiming) != hipSuccess) return bail(GSCAN_EHIP);
    for (Slot &s : c->slot) {
        if (hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return bail(GSCAN_EHIP);
        if (hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess) return bail(GSCAN_EHIP);
        if (hipEventCreateWithFlags(&s.copied2, hipEventDisableTiming) != hipSuccess) return bail(GSCAN_EHIP);
    }
    lap("events");
    *out = c;
    return GSCAN_OK;
}

void gscan_close(gscan_ctx *c)
{
    if (!c) return;
    for (Slot &s : c->slot) slot_drain_reads(s); // a file range still being read: its last piece launches into this context
    if (c->ingest) c->ingest->report_if_timing();
    if (g_timing && c->tw_n)
        fprintf(stderr, "[gscan timing] context on device %d: %zu waits | reads still arriving %.3f s  scan + fixed readback %.3f s  dense records %.3f s (%.1f MB, of which pinned (re)allocation %.3f s)  gathered lines %.3f s (%.1f MB)  merge %.3f s\n",
                c->device, c->tw_n, c->tw_reads, c->tw_scan, c->tw_dense, c->tw_dense_bytes / 1e6, c->tw_alloc, c->tw_gather, c->tw_gather_bytes / 1e6, c->tw_merge);
    hipSetDevice(c->hip_dev);
    hipDeviceSynchronize();
    for (Slot &s : c->slot) free_slot(c, s);
    if (c->prog_ev) hipEventDestroy(c->prog_ev);
    if (c->d_arena) hipFree(c->d_arena);
    if (c->h_arena) hipHostFree(c->h_arena);
    if (c->h_dense) hipHostFree(c->h_dense);
    if (c->h_gather) hipHostFree(c->h_gather);
    if (c->dv_recs) hipFree(c->dv_recs);
    if (c->dv_ends) hipFree(c->dv_ends);
    if (c->dv_desc) hipFree(c->dv_desc);
    if (c->dv_tiles) hipFree(c->dv_tiles);
    for (auto &e : c->ev_pool) {
        hipEventDestroy(e.a);
        hipEventDestroy(e.b);
    }
    if (c->ingest) Ingest::release(c->ingest); // the last context of the device: reader threads joined, pinned pool freed
    delete c;
}

const char *gscan_strerror(const gscan_ctx *c) { return c ? c->err.c_str() : "no context"; }

namespace {
// the slot's pinned buffer for `len` bytes: a block of the process-wide pool when it fits, else the slot's own allocation
static int slot_pinned_for(gscan_ctx *c, Slot &s, size_t len, void **out)
{
    if (len <= block_bytes()) {
        if (!s.blk) s.blk = c->ingest->take_slot_block();
        if (!s.blk) return fail(c, GSCAN_ENOMEM, "no pinned memory for a staging block");
        *out = s.blk->p;
        return 0;
    }
    int rc = slot_reserve_pinned(c, s, len);
    if (rc) return rc;
    *out = s.pinned;
    return 0;
}
static bool slot_owns(const Slot &s, const void *p) { return p && (p == s.pinned || (s.blk && p == s.blk->p)); }
} // namespace

int gscan_acquire(gscan_ctx *c, size_t len, void **pinned)
{
    if (!c || !pinned) return GSCAN_EINVAL;
    if (len > c->max_chunk) return fail(c, GSCAN_ETOOBIG, "chunk of %zu bytes exceeds max_chunk %zu", len, c->max_chunk);
    HIPCHK(c, hipSetDevice(c->hip_dev));
    Slot *s = nullptr;
    for (Slot &x : c->slot)
        if (x.state == ACQUIRED) s = &x; // re-acquire: same slot
    if (!s) s = free_slot_for_submit(c);
    if (!s) return fail(c, GSCAN_EBUSY, "all %d slots in flight", GSCAN_SLOTS);
    int rc = slot_pinned_for(c, *s, len, pinned);
    if (rc) return rc;
    s->state = ACQUIRED;
    return GSCAN_OK;
}

size_t gscan_block_size(void) { return block_bytes(); }

namespace {
// the arena is bounded (at most 384 MiB whatever is asked), a helper thread that cannot be started leaves its
// blocks to the runtime's own allocator instead of taking the process down, and nothing crosses the extern "C" boundary.
struct AheadSrc {
    std::string path;
    off_t off;
    size_t n;
};
static int prefault_start(size_t plain, const std::vector<AheadSrc> &ahead_src)
{
    if (g_prefault.base || plain + ahead_src.size() == 0) return GSCAN_OK; // once per process
    if (!ingest_cfg().prefault) return GSCAN_OK;
    const size_t huge = size_t(2) << 20;
    const size_t stride = (block_bytes() + kPad + huge - 1) / huge * huge;
    const size_t most = std::max<size_t>(1, (size_t(384) << 20) / stride);
    const size_t ahead = std::min(ahead_src.size(), most);
    plain = std::min(std::min<size_t>(plain, 64), most - ahead);
    const size_t blocks = ahead + plain;
    if (blocks == 0) return GSCAN_OK;
    void *m = mmap(nullptr, stride * blocks + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return GSCAN_ENOMEM;
    char *base = (char *)(((uintptr_t)m + huge - 1) & ~(uintptr_t)(huge - 1));
    (void)madvise(base, stride * blocks, MADV_HUGEPAGE);
    std::atomic<int> *state = new (std::nothrow) std::atomic<int>[blocks];
    Prefault::Piece *piece = ahead ? new (std::nothrow) Prefault::Piece[ahead] : nullptr;
    if (!state || (ahead && !piece)) {
        delete[] state;
        delete[] piece;
        munmap(m, stride * blocks + huge);
        return GSCAN_ENOMEM;
    }
    for (size_t k = 0; k < blocks; k++) state[k].store(0, std::memory_order_relaxed);
    for (size_t k = 0; k < ahead; k++) {
        piece[k].path = ahead_src[k].path;
        piece[k].off = ahead_src[k].off;
        piece[k].n = ahead_src[k].n;
    }
    g_prefault.stride = stride;
    g_prefault.count = blocks;
    g_prefault.ahead = ahead;
    g_prefault.state = state;
    g_prefault.piece = piece;
    g_prefault.unclaimed.store(ahead, std::memory_order_relaxed);
    g_prefault.base = base;
    const size_t nth = std::min<size_t>(ahead ? (size_t)ingest_cfg().ahead_threads : 4, blocks);
    const size_t used = block_bytes() + kPad;
    for (size_t t = 0; t < nth; t++) {
        auto work = [t, nth, blocks, ahead, base, stride, used, state, piece] {
            int fd = -1;
            const std::string *open_path = nullptr;
            for (size_t k = t; k < blocks; k += nth) { // (block k is asked for k-th: the early ones first)
                char *at = base + k * stride;
                if (k >= ahead) {
                    for (size_t o = 0; o < used; o += 4096) at[o] = 0;
                    state[k].store(1, std::memory_order_release);
                    continue;
                }
                Prefault::Piece &pc = piece[k];
                if (!open_path || *open_path != pc.path) {
                    if (fd >= 0) close(fd);
                    fd = open(pc.path.c_str(), O_RDONLY | O_NOCTTY | O_CLOEXEC);
                    open_path = &pc.path;
                }
                struct stat st;
                bool ok = fd >= 0 && fstat(fd, &st) == 0;
                size_t got = 0;
                while (ok && got < pc.n) {
                    const ssize_t r = pread(fd, at + got, pc.n - got, pc.off + (off_t)got);
                    if (r > 0) got += (size_t)r;
                    else if (r == 0 || errno != EINTR) ok = false;
                }
                if (ok) {
                    pc.dev = st.st_dev;
                    pc.ino = st.st_ino;
                    pc.size = st.st_size;
                    pc.mtim = st.st_mtim;
                    pc.ctim = st.st_ctim;
                    for (size_t o = (pc.n + 4095) & ~size_t(4095); o < used; o += 4096) at[o] = 0; // (the slack behind the piece is staging memory too)
                }
                state[k].store(ok ? 2 : 3, std::memory_order_release);
            }
            if (fd >= 0) close(fd);
        };
        try {
            std::thread(work).detach();
        } catch (...) { // no thread to be had: this one's blocks are never ready -- marked failed, the runtime's allocator serves instead
            for (size_t k = t; k < blocks; k += nth) state[k].store(3, std::memory_order_release);
        }
    }
    return GSCAN_OK;
}
} // namespace

int gscan_prefault(size_t blocks)
{
    try {
        return prefault_start(blocks, {});
    } catch (...) {
        return GSCAN_ENOMEM;
    }
}

int gscan_prefault_files(size_t blocks, const char *const *paths, size_t npaths)
{
    std::vector<AheadSrc> src;
    const size_t blk = block_bytes();
    try {
        for (size_t i = 0; paths && i < npaths && src.size() < blocks; i++) {
            struct stat st;
            if (!paths[i] || stat(paths[i], &st) != 0 || !S_ISREG(st.st_mode)) continue;
            for (off_t o = 0; o < st.st_size && src.size() < blocks; o += (off_t)blk)
                src.push_back(AheadSrc{paths[i], o, (size_t)std::min<off_t>((off_t)blk, st.st_size - o)});
        }
        return prefault_start(blocks > src.size() ? blocks - src.size() : 0, src);
    } catch (...) {
        return GSCAN_ENOMEM;
    }
}

// Reader threads for one device whose NUMA node offers `local_cpus` CPUs to this process and is shared by `devices_sharing`
// devices, on a node that drives `devices_total`: 8 where there are CPUs to spare (the measured optimum on a one-GPU box: more
// of them only wait for blocks), half of the device's share of the node otherwise -- the other half is the workers' (report
// walk, batch reads) -- and never more than the node's page cache can feed: the host's copy ceiling (page cache -> pinned,
// no DMA) was measured with eight pools at 83 / 116 / 99 / 64 GB/s for 8 / 16 / 32 / 64 readers in all (profiles/r05_a_n8_*,
// a two-socket EPYC 9575F): past ~24 readers they take bandwidth from each other.  Never below 2.
int gscan_auto_readers(int local_cpus, int devices_sharing, int devices_total)
{
    const int by_host = std::max(2, std::min(8, 24 / std::max(1, devices_total)));
    if (local_cpus <= 0) return by_host;
    const int share = local_cpus / std::max(1, devices_sharing);
    return std::max(2, std::min(by_host, share / 2));
}

void gscan_ingest_info(size_t *block_bytes_out, int *readers, int *copy_streams)
{
    if (block_bytes_out) *block_bytes_out = ingest_cfg().block;
    if (readers) *readers = ingest_cfg().readers; // (0: auto -- gscan_auto_readers per device)
    if (copy_streams) *copy_streams = ingest_cfg().copy_streams;
}

int gscan_pool_stats(const gscan_ctx *c, uint64_t out[4])
{
    if (!c || !c->ingest || !out) return GSCAN_EINVAL;
    c->ingest->stats(out);
    return GSCAN_OK;
}

long gscan_parse_cpulist(const char *list, int *cpus, size_t cap)
{
    if (!list) return GSCAN_EINVAL;
    std::vector<int> v;
    parse_cpulist(list, v);
    for (size_t i = 0; i < v.size() && i < cap && cpus; i++) cpus[i] = v[i];
    return (long)v.size();
}

int gscan_pci_cpulist(const char *sysfs_pci_root, const char *busid, char *buf, size_t cap) { return pci_cpulist(sysfs_pci_root, busid, buf, cap); }

int gscan_device_cpulist(int hip_device, char *buf, size_t cap) { return device_cpulist(hip_device, buf, cap); }

int gscan_submit(gscan_ctx *c, const gscan_db *db, const void *host_bytes, size_t len, uint64_t tag)
{
    if (!c || !db || (!host_bytes && len)) return GSCAN_EINVAL;
    if (db->db.tier == GSCAN_TIER_NULL || db->db.tier == GSCAN_TIER_ANCHORED) return fail(c, GSCAN_EINVAL, "nothing to scan for this pattern (it matches \"\", or only at the restart position / chunk end)");
    if (len > c->max_chunk) return fail(c, GSCAN_ETOOBIG, "chunk of %zu bytes exceeds max_chunk %zu", len, c->max_chunk);
    HIPCHK(c, hipSetDevice(c->hip_dev));
    Slot *s = nullptr;
    for (Slot &x : c->slot)
        if (x.state == ACQUIRED) s = &x;
    const bool own = s && slot_owns(*s, host_bytes);
    if (!s) {
        s = free_slot_for_submit(c);
        if (!s) return fail(c, GSCAN_EBUSY, "all %d slots in flight", GSCAN_SLOTS);
    } else if (own && len > (host_bytes == s->pinned ? s->pinned_cap : block_bytes())) {
        return fail(c, GSCAN_EINVAL, "submitted %zu bytes into a smaller acquired buffer", len);
    }
    note_db(c, db);
    int rc = ensure_prog(c, db, c->compute);
    if (rc) return rc;
    rc = slot_reserve_device(c, *s, len);
    if (rc) return rc;
    s->ext = nullptr;
    s->no_content = false;
    s->segs.clear();
    s->files.clear();
    if (own) {
        s->ext = host_bytes;
        if (len) HIPCHK(c, hipMemcpyAsync(s->d_text, host_bytes, len, hipMemcpyHostToDevice, c->copy));
    } else {
        // The caller's own buffer (FileGrep hands over its mmap of the file chunk).  Big ones are registered
        // with the runtime and DMA'd in place -- no bounce copy: 57 GB/s from the page cache on MI355X,
        // against 8-10 GB/s per thread for memcpy into a pinned buffer (profiles/r01_g_host_probe.txt).
        // Small ones, or memory the runtime refuses to register, are staged through the slot's pinned buffer.
        s->ext = host_bytes;
        bool direct = false;
        if (len >= c->register_min) {
            if (hipHostRegister(const_cast<void *>(host_bytes), len, hipHostRegisterDefault) == hipSuccess) {
                s->ext_reg = const_cast<void *>(host_bytes);
                direct = true;
            } else {
                (void)hipGetLastError();
            }
        }
        if (direct) {
            HIPCHK(c, hipMemcpyAsync(s->d_text, host_bytes, len, hipMemcpyHostToDevice, c->copy));
        } else {
            void *stage = nullptr;
            rc = slot_pinned_for(c, *s, len, &stage);
            if (rc) return rc;
            for (size_t o = 0; o < len; o += kCopyPiece) { // memcpy of piece i+1 overlaps the DMA of piece i
                size_t n = std::min(kCopyPiece, len - o);
                memcpy((char *)stage + o, (const char *)host_bytes + o, n);
                HIPCHK(c, hipMemcpyAsync(s->d_text + o, (char *)stage + o, n, hipMemcpyHostToDevice, c->copy));
            }
        }
    }
    if (c->compute != c->copy) {
        HIPCHK(c, hipEventRecord(s->copied, c->copy));
        HIPCHK(c, hipStreamWaitEvent(c->compute, s->copied, 0));
    }
    s->second = nullptr;
    s->db = db;
    s->len = len;
    s->tag = tag;
    s->seq = c->next_seq++;
    rc = slot_launch(c, *s);
    if (rc) return rc;
    s->state = INFLIGHT;
    return GSCAN_OK;
}

int gscan_submit_segs(gscan_ctx *c, const gscan_db *db, const void *pinned, const gscan_seg *segs, size_t nseg,
                      uint64_t tag)
{
    if (!c || !db || !pinned || (!segs && nseg)) return GSCAN_EINVAL;
    if (db->db.tier == GSCAN_TIER_NULL || db->db.tier == GSCAN_TIER_ANCHORED) return fail(c, GSCAN_EINVAL, "nothing to scan for this pattern (it matches \"\", or only at the restart position / chunk end)");
    HIPCHK(c, hipSetDevice(c->hip_dev));
    Slot *s = nullptr;
    for (Slot &x : c->slot)
        if (x.state == ACQUIRED && slot_owns(x, pinned)) s = &x;
    if (!s) return fail(c, GSCAN_EINVAL, "gscan_submit_segs wants the buffer gscan_acquire handed out");
    const size_t cap = pinned == s->pinned ? s->pinned_cap : block_bytes();
    size_t used = 0;
    for (size_t i = 0; i < nseg; i++) {
        if (segs[i].offset & 15) return fail(c, GSCAN_EINVAL, "segment %zu is not 16-byte aligned", i);
        if (segs[i].offset + segs[i].len > cap) return fail(c, GSCAN_EINVAL, "segment %zu lies outside the acquired buffer", i);
        used = std::max<size_t>(used, segs[i].offset + segs[i].len);
    }
    note_db(c, db);
    int rc = ensure_prog(c, db, c->compute);
    if (rc) return rc;
    rc = slot_reserve_device(c, *s, used);
    if (rc) return rc;
    s->ext = pinned;
    s->no_content = false;
    s->files.clear();
    s->segs.assign(segs, segs + nseg);
    if (nseg == 0) s->segs.push_back({0, 0, 0}); // keeps the chunk on the multi-segment path with one empty segment
    {
        uint32_t tb = 0, nw = 1;
        gscan::scan_geometry(db->db.tier, c->variant, db->db.prog, &tb, &nw);
        rc = slot_build_tiles(c, *s, tb);
    }
    if (rc) return rc;
    if (used) HIPCHK(c, hipMemcpyAsync(s->d_text, pinned, used, hipMemcpyHostToDevice, c->copy));
    if (c->compute != c->copy) {
        HIPCHK(c, hipEventRecord(s->copied, c->copy));
        HIPCHK(c, hipStreamWaitEvent(c->compute, s->copied, 0));
    }
    s->second = nullptr;
    s->db = db;
    s->len = used;
    s->tag = tag;
    s->seq = c->next_seq++;
    rc = slot_launch(c, *s);
    if (rc) return rc;
    s->state = INFLIGHT;
    return GSCAN_OK;
}

int gscan_submit_fd(gscan_ctx *c, const gscan_db *db, int fd, long long file_off, size_t len, uint64_t tag)
{
    if (!c || !db || fd < 0 || file_off < 0) return GSCAN_EINVAL;
    if (db->db.tier == GSCAN_TIER_NULL || db->db.tier == GSCAN_TIER_ANCHORED) return fail(c, GSCAN_EINVAL, "nothing to scan for this pattern (it matches \"\", or only at the restart position / chunk end)");
    if (len > c->max_chunk) return fail(c, GSCAN_ETOOBIG, "chunk of %zu bytes exceeds max_chunk %zu", len, c->max_chunk);
    HIPCHK(c, hipSetDevice(c->hip_dev));
    Slot *s = free_slot_for_submit(c);
    if (!s) return fail(c, GSCAN_EBUSY, "no free slot (all in flight, or one is acquired)");
    note_db(c, db);
    trace("submit_fd: %zu bytes", len);
    int rc = 0;
    if (c->prog_id == 0 && len > s->d_text_cap) {
        // The first range of a context: the program's upload -- the process's first host-to-device transfer, 8 - 25 ms of
        // set-up inside the runtime (profiles/r04_b_*) -- and the slot's device buffers (hipMalloc: 15 ms per half GiB) are both
        // on the way to the first DMA, and neither needs the other: the buffers are made by a helper thread meanwhile.
        // BASELINE configs[0] is one such range and little else.
        std::string helper_err;
        int helper_rc = 0;
        std::thread helper([&] {
            t_err_sink = &helper_err;
            if (hipSetDevice(c->hip_dev) != hipSuccess) helper_rc = fail(c, GSCAN_EHIP, "hipSetDevice failed in the slot's helper thread");
            else helper_rc = slot_reserve_device(c, *s, len);
            t_err_sink = nullptr;
        });
        rc = ensure_prog(c, db, c->compute);
        trace("submit_fd: program on its way");
        helper.join();
        if (!rc && helper_rc) {
            c->err = helper_err;
            rc = helper_rc;
        }
        if (rc) return rc;
    } else {
        rc = ensure_prog(c, db, c->compute);
        if (rc) return rc;
        trace("submit_fd: program on the device");
        rc = slot_reserve_device(c, *s, len);
        if (rc) return rc;
    }
    trace("submit_fd: slot sized");
    // fan the range out to the device's reader threads: every piece is DMA'd on one of this context's copy streams the
    // moment it is read, and whoever finishes the last piece launches the scan (fd_finish).  This thread goes on.
    if (!s->grp) s->grp.reset(new (std::nothrow) ReadGroup());
    if (!s->grp) return fail(c, GSCAN_ENOMEM, "out of memory");
    ReadGroup &g = *s->grp;
    const size_t blk = block_bytes();
    g.pending = (len + blk - 1) / blk;
    g.err = 0;
    g.rc = 0;
    g.msg.clear();
    g.finish = fd_finish;
    g.ctx = c;
    g.slot = s;
    s->ext = nullptr;
    s->no_content = true;
    s->segs.clear();
    s->files.clear();
    s->db = db;
    s->len = len;
    s->tag = tag;
    s->seq = c->next_seq++;
    s->state = INFLIGHT;
    if (g.pending == 0) { // an empty range: nothing to read, launch right away
        g.finished = false;
        fd_finish(&g);
        return GSCAN_OK;
    }
    g.finished = false;
    std::vector<ReadTask> tasks;
    tasks.reserve(g.pending);
    size_t k = 0;
    // pieces that were read ahead while the runtime started (gscan_prefault_files): same file, same offset, same length
    struct stat fst;
    const bool look_ahead = g_prefault.unclaimed.load(std::memory_order_relaxed) > 0 && fstat(fd, &fst) == 0;
    s->second = c->ingest->second_stream(len); // (nullptr until the device has been handed enough to be worth a second stream)
    for (size_t o = 0; o < len; o += blk, k++) {
        tasks.push_back(ReadTask{fd, (off_t)(file_off + (long long)o), std::min(blk, len - o), s->d_text + o, (k & 1) && s->second ? s->second : c->copy, &g});
        if (look_ahead) {
            ReadTask &t = tasks.back();
            for (size_t a = 0; a < g_prefault.ahead; a++) {
                Prefault::Piece &pc = g_prefault.piece[a];
                if (pc.off != t.off || pc.n != t.n || pc.claimed.load(std::memory_order_relaxed)) continue;
                // (dev / ino are the helper's to write: wait for it -- it has had the whole start of the runtime; but not for ever:
                // a helper stuck on slow storage leaves the piece to the ordinary read)
                for (int spins = 0; !g_prefault.state[a].load(std::memory_order_acquire) && spins < 20000; spins++) std::this_thread::yield();
                if (g_prefault.state[a].load(std::memory_order_acquire) != 2 || pc.dev != fst.st_dev || pc.ino != fst.st_ino) continue;
                // the same file AS IT WAS READ: size and both time stamps unchanged since the helper's fstat
                if (pc.size != fst.st_size || pc.mtim.tv_sec != fst.st_mtim.tv_sec || pc.mtim.tv_nsec != fst.st_mtim.tv_nsec ||
                    pc.ctim.tv_sec != fst.st_ctim.tv_sec || pc.ctim.tv_nsec != fst.st_ctim.tv_nsec)
                    continue;
                if (pc.claimed.exchange(true)) continue;
                g_prefault.unclaimed.fetch_sub(1, std::memory_order_relaxed);
                t.ahead = (int)a;
                break;
            }
        }
    }
    c->ingest->read(tasks.data(), tasks.size());
    return GSCAN_OK;
}

int gscan_submit_files(gscan_ctx *c, const gscan_db *db, const gscan_file *files, size_t n, uint64_t tag)
{
    if (!c || !db || (!files && n)) return GSCAN_EINVAL;
    if (db->db.tier == GSCAN_TIER_NULL || db->db.tier == GSCAN_TIER_ANCHORED) return fail(c, GSCAN_EINVAL, "nothing to scan for this pattern (it matches \"\", or only at the restart position / chunk end)");
    const size_t blk = block_bytes();
    size_t used = 0;
    for (size_t i = 0; i < n; i++) {
        if (!files[i].path && files[i].fd < 0) return fail(c, GSCAN_EINVAL, "file %zu has neither a path nor a descriptor", i);
        if (files[i].len > blk) return fail(c, GSCAN_ETOOBIG, "file %zu is larger than a staging block (%zu bytes): hand it over with gscan_submit_fd", i, blk);
        used = ((used + 15) & ~size_t(15)) + files[i].len;
    }
    if (used > c->max_chunk) return fail(c, GSCAN_ETOOBIG, "batch of %zu bytes exceeds max_chunk %zu", used, c->max_chunk);
    HIPCHK(c, hipSetDevice(c->hip_dev));
    Slot *s = free_slot_for_submit(c);
    if (!s) return fail(c, GSCAN_EBUSY, "no free slot (all in flight, or one is acquired)");
    note_db(c, db);
    int rc = ensure_prog(c, db, c->compute);
    if (rc) return rc;
    rc = slot_reserve_device(c, *s, used);
    if (rc) return rc;
    s->files.resize(n);
    s->segs.resize(n);
    size_t at = 0;
    for (size_t i = 0; i < n; i++) {
        at = (at + 15) & ~size_t(15);
        FileItem &it = s->files[i];
        if (files[i].path) it.path = files[i].path;
        else it.path.clear();
        it.fd = files[i].fd;
        it.oflags = files[i].oflags;
        it.len = files[i].len;
        it.dst_off = at;
        it.err = 0;
        s->segs[i] = {(uint64_t)at, files[i].len, 0};
        at += files[i].len;
    }
    if (n == 0) s->segs.push_back({0, 0, 0}); // keeps the chunk on the multi-segment path with one empty segment
    {
        uint32_t tb = 0, nw = 1;
        gscan::scan_geometry(db->db.tier, c->variant, db->db.prog, &tb, &nw);
        rc = slot_build_tiles(c, *s, tb); // (the tile table goes out on the copy stream, ahead of the pieces)
    }
    if (rc) return rc;
    if (!s->grp) s->grp.reset(new (std::nothrow) ReadGroup());
    if (!s->grp) return fail(c, GSCAN_ENOMEM, "out of memory");
    ReadGroup &g = *s->grp;
    // pieces: runs of consecutive files that fit one staging block -- one reader fills the block, one DMA carries it
    s->second = c->ingest->second_stream(used);
    std::vector<ReadTask> tasks;
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        const uint64_t base = s->files[i].dst_off;
        while (j < n && s->files[j].dst_off + s->files[j].len - base <= blk) j++;
        ReadTask t{-1, 0, (size_t)(s->files[j - 1].dst_off + s->files[j - 1].len - base), s->d_text + base, (tasks.size() & 1) && s->second ? s->second : c->copy, &g};
        t.items = &s->files[i];
        t.nitems = (uint32_t)(j - i);
        tasks.push_back(t);
        i = j;
    }
    g.pending = tasks.size();
    g.err = 0;
    g.rc = 0;
    g.msg.clear();
    g.finish = fd_finish;
    g.ctx = c;
    g.slot = s;
    s->ext = nullptr;
    s->no_content = true;
    s->db = db;
    s->len = used;
    s->tag = tag;
    s->seq = c->next_seq++;
    s->state = INFLIGHT;
    g.finished = false;
    if (tasks.empty()) { // nothing to read (no files, or empty ones only... an empty file still is a task): launch right away
        fd_finish(&g);
        return GSCAN_OK;
    }
    c->ingest->read(tasks.data(), tasks.size());
    return GSCAN_OK;
}

int gscan_wait_segs(gscan_ctx *c, uint64_t *tag, const uint32_t **starts, const size_t **seg_first, size_t *nseg, const void **content)
{
    if (!c || !starts || !seg_first || !nseg) return GSCAN_EINVAL;
    Slot *s = nullptr;
    for (Slot &x : c->slot)
        if (x.state == INFLIGHT && (!s || x.seq < s->seq)) s = &x;
    if (!s) return fail(c, GSCAN_EEMPTY, "nothing in flight");
    // Whatever goes wrong from here on, the chunk's result is lost but its slot is free again: one device error does not
    // leave the context believing a slot is in flight for ever.
    struct Release {
        Slot *s;
        bool ok = false;
        ~Release()
        {
            if (!ok) {
                slot_unregister(*s);
                s->state = FREE;
            }
        }
    } release{s};
    const bool tw_on = g_timing;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tw0 = tw_on ? tnow() : 0;
    trace("wait: begin");
    slot_drain_reads(*s);
    trace("wait: range read and launched");
    if (tw_on) c->tw_reads += tnow() - tw0, tw0 = tnow(), c->tw_n++;
    if (s->grp && s->grp->rc) {
        c->err = s->grp->msg;
        const int rc = s->grp->rc;
        s->grp->rc = 0;
        return rc;
    }
    HIPCHK(c, hipSetDevice(c->hip_dev));
    HIPCHK(c, hipEventSynchronize(s->done));
    trace("wait: scan done");
    if (tw_on) c->tw_scan += tnow() - tw0, tw0 = tnow();
    slot_unregister(*s); // the DMA out of the caller's buffer is over (the text stays in HBM for a possible rescan)
    const size_t K = gscan::kShards;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (s->h_counter[K * kCS] == 0) break; // no shard overflowed
        if (attempt == 1) return fail(c, GSCAN_EHIP, "record buffer overflow persisted after regrow");
        // the text is still in HBM: size every shard for the fullest one (+25%) and rescan
        uint32_t worst = 0;
        for (size_t k = 0; k < K; k++) worst = std::max(worst, s->h_counter[k * kCS]);
        size_t want = ((size_t)worst + (size_t)worst / 4 + kSpecPer) * K;
        HIPCHK(c, hipStreamSynchronize(c->compute));
        if (s->d_recs) hipFree(s->d_recs);
        s->d_recs = nullptr;
        s->rec_cap = 0;
        HIPCHK(c, hipMalloc((void **)&s->d_recs, want * 4));
        s->rec_cap = want;
        int rc = slot_launch(c, *s);
        if (rc) return rc;
        HIPCHK(c, hipEventSynchronize(s->done));
    }
    const size_t cap_shard = s->rec_cap / K;
    const size_t struck = s->h_counter[K * kCS + 1]; // records the second pass struck out (kStruck in the buffer)
    size_t total = 0;
    bool spec_ok = true;
    for (size_t k = 0; k < K; k++) {
        total += s->h_counter[k * kCS];
        if (s->h_counter[k * kCS] > kSpecPer) spec_ok = false;
    }
    size_t fullest = 0;
    for (size_t k = 0; k < K; k++) fullest = std::max<size_t>(fullest, s->h_counter[k * kCS]);
    const size_t ew = s->ext_words;
    const uint32_t *dense = nullptr, *dense_ext = nullptr; // ordered: [total] records + [total * ew] extras; else [K][fullest] + [K][fullest * ew]
    auto reserve_dense = [&](size_t need) -> int {
        if (need > c->dense_cap) {
            const double ta = tw_on ? tnow() : 0;
            if (c->h_dense) hipHostFree(c->h_dense);
            c->h_dense = nullptr;
            c->dense_cap = 0;
            const size_t cap = std::max<size_t>(2 * need, 2u << 20);
            HIPCHK(c, hipHostMalloc((void **)&c->h_dense, cap * 4, hipHostMallocDefault));
            c->dense_cap = cap;
            if (tw_on) c->tw_alloc += tnow() - ta;
        }
        return 0;
    };
    // (a resolved chunk: the records k_resolve dropped are gone from the descriptors' runs and from the ordered copy -- what is
    // left is the list)
    if (s->resolved) {
        if (struck > total) return fail(c, GSCAN_EHIP, "the resolve pass dropped %zu of %zu records", struck, total);
        total -= struck;
    }
    const bool ordered = s->ordered && (struck == 0 || s->resolved);
    if (ordered && s->h_counter[K * kCS + 3] != total) return fail(c, GSCAN_EHIP, "ordered copy holds %u records, the shard counters %zu", s->h_counter[K * kCS + 3], total);
    if (ordered) {
        {
            const size_t h = c->hint_total.load(std::memory_order_relaxed);
            c->hint_total.store(std::max(total, h - h / 8), std::memory_order_relaxed); // (follows a growing result at once, a shrinking one slowly)
        }
        if (total <= s->spec_n) { // it came in with the counters: a sparse result, or one the last chunks foretold
            const bool big = s->spec_n > kSpecRecs;
            dense = big ? s->h_big : s->h_spec;
            dense_ext = big ? s->h_big + s->big_recs : s->h_ext_spec;
        } else { // ONE linear range each for the records and their extras
            const size_t need = total * (1 + (s->has_ext ? ew : 0));
            if (int rc = reserve_dense(need)) return rc;
            HIPCHK(c, hipMemcpyAsync(c->h_dense, s->d_sorted, total * 4, hipMemcpyDeviceToHost, c->compute));
            if (s->has_ext) HIPCHK(c, hipMemcpyAsync(c->h_dense + total, s->d_sorted + s->rec_cap, total * 4 * ew, hipMemcpyDeviceToHost, c->compute));
            HIPCHK(c, hipEventRecord(s->done, c->compute));
            HIPCHK(c, hipEventSynchronize(s->done));
            dense = c->h_dense;
            dense_ext = c->h_dense + total;
            if (tw_on) c->tw_dense += tnow() - tw0, tw0 = tnow(), c->tw_dense_bytes += need * 4;
        }
    } else if (!spec_ok) { // dense result: the used part of every shard region in ONE strided copy (descriptors deal the shards round robin: they fill evenly)
        const size_t need = K * fullest * (1 + (s->has_ext ? ew : 0));
        if (int rc = reserve_dense(need)) return rc;
        HIPCHK(c, hipMemcpy2DAsync(c->h_dense, fullest * 4, s->d_recs, cap_shard * 4, fullest * 4, K, hipMemcpyDeviceToHost, c->compute));
        dense = c->h_dense;
        if (s->has_ext) {
            HIPCHK(c, hipMemcpy2DAsync(c->h_dense + K * fullest, fullest * 4 * ew, s->d_ext, cap_shard * 4 * ew, fullest * 4 * ew, K, hipMemcpyDeviceToHost, c->compute));
            dense_ext = c->h_dense + K * fullest;
        }
        HIPCHK(c, hipEventRecord(s->done, c->compute));
        HIPCHK(c, hipEventSynchronize(s->done));
        if (tw_on) c->tw_dense += tnow() - tw0, tw0 = tnow(), c->tw_dense_bytes += need * 4;
    }
    s->gather_ok = false;
    s->gather_bytes = 0;
    s->out_gather = nullptr;
    if (s->ext_words == 4) { // the printed lines' text: the used part of the gather buffer, into pinned memory
        const size_t used = std::min<size_t>(s->h_counter[K * kCS + 2], s->gather_cap);
        {
            const size_t h = c->hint_gather.load(std::memory_order_relaxed);
            c->hint_gather.store(std::max(used, h - h / 8), std::memory_order_relaxed);
        }
        if (s->ordered && used <= s->gspec_n) { // it came in with the counters
            s->gather_ok = true;
            s->gather_bytes = used;
            s->out_gather = s->h_gspec;
        } else if (used <= kGatherPinnedMax) {
            if (used > c->h_gather_cap) {
                if (c->h_gather) hipHostFree(c->h_gather);
                c->h_gather = nullptr;
                c->h_gather_cap = 0;
                const size_t cap = std::max<size_t>(2 * used, 8u << 20);
                HIPCHK(c, hipHostMalloc((void **)&c->h_gather, cap, hipHostMallocDefault));
                c->h_gather_cap = cap;
            }
            if (used) {
                HIPCHK(c, hipMemcpyAsync(c->h_gather, s->d_gather, used, hipMemcpyDeviceToHost, c->compute));
                HIPCHK(c, hipEventRecord(s->done, c->compute));
                HIPCHK(c, hipEventSynchronize(s->done));
            }
            s->gather_ok = true;
            s->gather_bytes = used;
            s->out_gather = c->h_gather;
            if (tw_on) c->tw_gather += tnow() - tw0, tw0 = tnow(), c->tw_gather_bytes += used;
        }
    }
    const bool multi = !s->segs.empty();
    const size_t ns = multi ? s->segs.size() : 1;
    s->seg_first.assign(ns + 1, 0);
    size_t seg = 0;
    const size_t n_desc = (size_t)s->n_tiles * s->nw;
    if (ordered) {
        // the list is there as it is; the descriptors' counts say where the segments begin in it
        size_t at = 0;
        if (multi)
            for (size_t di = 0; di < n_desc; di++) {
                const uint32_t t = (uint32_t)(di / s->nw);
                while (seg + 1 <= ns && t >= s->tile_first[seg + 1]) s->seg_first[++seg] = at;
                at += (uint32_t)s->h_desc[di];
            }
        while (seg < ns) s->seg_first[++seg] = total; // trailing segments without tiles / records (one segment: [0, total))
        s->out_starts = dense;
        s->out_ext = s->has_ext ? dense_ext : nullptr;
    } else {
        s->sorted_ext.clear();
        if (s->has_ext) s->sorted_ext.reserve(total * ew);
        s->sorted.clear();
        s->sorted.reserve(total);
        for (size_t di = 0; di < n_desc; di++) { // descriptors are in (segment, text) order: concatenating their runs sorts the list
            const uint32_t t = (uint32_t)(di / s->nw);
            if (multi)
                while (seg + 1 <= ns && t >= s->tile_first[seg + 1]) s->seg_first[++seg] = s->sorted.size();
            unsigned long long d = s->h_desc[di];
            uint32_t cnt = (uint32_t)d;
            size_t base = (size_t)(d >> 32);
            if (!cnt) continue;
            const uint32_t *src = spec_ok ? s->h_spec + (base / cap_shard) * kSpecPer + base % cap_shard : dense + (base / cap_shard) * fullest + base % cap_shard;
            const uint32_t *ex = !s->has_ext ? nullptr
                                 : spec_ok   ? s->h_ext_spec + ((base / cap_shard) * kSpecPer + base % cap_shard) * ew
                                             : dense_ext + ((base / cap_shard) * fullest + base % cap_shard) * ew;
            if (struck == 0) {
                s->sorted.insert(s->sorted.end(), src, src + cnt);
                if (ex) s->sorted_ext.insert(s->sorted_ext.end(), ex, ex + (size_t)cnt * ew);
            } else { // (the second K3 pass struck records out: their extras go with them)
                for (uint32_t i = 0; i < cnt; i++)
                    if (src[i] != gscan::kStruck) {
                        s->sorted.push_back(src[i]);
                        if (ex) s->sorted_ext.insert(s->sorted_ext.end(), ex + (size_t)i * ew, ex + (size_t)(i + 1) * ew);
                    }
            }
        }
        if (s->sorted.size() + struck != total) return fail(c, GSCAN_EHIP, "descriptor total %zu + struck %zu != counter %zu", s->sorted.size(), struck, total);
        while (seg < ns) s->seg_first[++seg] = s->sorted.size(); // trailing segments without tiles / records
        s->out_starts = s->sorted.data();
        s->out_ext = s->has_ext ? s->sorted_ext.data() : nullptr;
    }
    s->file_err.clear();
    for (const FileItem &it : s->files) s->file_err.push_back(it.err);
    if (tag) *tag = s->tag;
    *starts = s->out_starts;
    *seg_first = s->seg_first.data();
    *nseg = ns;
    if (content) *content = s->no_content ? nullptr : s->ext;
    c->last_waited = s;
    s->state = FREE;
    release.ok = true;
    if (tw_on) c->tw_merge += tnow() - tw0;
    return GSCAN_OK;
}

const uint32_t *gscan_last_ext(const gscan_ctx *c)
{
    if (!c || !c->last_waited || c->last_waited->ext_words != 4) return nullptr;
    return c->last_waited->out_ext;
}

const uint8_t *gscan_last_gather(const gscan_ctx *c, size_t *bytes)
{
    if (bytes) *bytes = 0;
    if (!c || !c->last_waited || c->last_waited->ext_words != 4 || !c->last_waited->gather_ok) return nullptr;
    if (bytes) *bytes = c->last_waited->gather_bytes;
    return c->last_waited->out_gather ? c->last_waited->out_gather : (const uint8_t *)"";
}

const int *gscan_last_file_errors(const gscan_ctx *c, size_t *n)
{
    if (n) *n = 0;
    if (!c || !c->last_waited || c->last_waited->file_err.empty()) return nullptr;
    if (n) *n = c->last_waited->file_err.size();
    return c->last_waited->file_err.data();
}

const uint32_t *gscan_last_ends(const gscan_ctx *c)
{
    if (!c || !c->last_waited || c->last_waited->ext_words != 1) return nullptr;
    return c->last_waited->out_ext;
}

int gscan_wait(gscan_ctx *c, uint64_t *tag, const uint32_t **starts, size_t *n, const void **content)
{
    if (!n) return GSCAN_EINVAL;
    const size_t *first = nullptr;
    size_t ns = 0;
    int rc = gscan_wait_segs(c, tag, starts, &first, &ns, content);
    if (rc == GSCAN_OK) *n = first[ns];
    return rc;
}

int gscan_set_capacity(gscan_ctx *c, size_t n_records)
{
    if (!c) return GSCAN_EINVAL;
    c->dev_cap_req = n_records;
    return GSCAN_OK;
}

int gscan_set_option(gscan_ctx *c, const char *name, long value)
{
    if (!c || !name) return GSCAN_EINVAL;
    if (!strcmp(name, "variant")) {
        if (value != 13 && value != 14 && value != 38 && (value < 0 || value > 7 || (value & 3) == 3)) return GSCAN_EINVAL; // KiB per wave {16,8,12} | nontemporal<<2; 13: 768-thread workgroups for the table kernels; 38 = 6 + K2's lane-table form
        c->variant = (int)value;
        return GSCAN_OK;
    }
    if (!strcmp(name, "register_min")) { // 0 = register everything; a huge value = always stage through pinned memory
        if (value < 0) return GSCAN_EINVAL;
        c->register_min = (size_t)value;
        return GSCAN_OK;
    }
    if (!strcmp(name, "line_extents")) {
        c->line_extents = value != 0;
        return GSCAN_OK;
    }
    if (!strcmp(name, "match_ends")) {
        c->match_ends = value != 0;
        return GSCAN_OK;
    }
    if (!strcmp(name, "k3_depth")) { // K3's filter positions: 0 = the compiler's choice, else 3 or 4 (A/B runs, tests: both depths list the same records)
        if (value != 0 && value != 3 && value != 4) return GSCAN_EINVAL;
        c->k3_depth = (int)value;
        return GSCAN_OK;
    }
    if (!strcmp(name, "blocks_per_cu")) {
        if (value < 0 || value > 64) return GSCAN_EINVAL;
        c->blocks_per_cu = (int)value;
        return GSCAN_OK;
    }
    return GSCAN_EINVAL;
}

int gscan_scan_device(gscan_ctx *c, const gscan_db *db, const void *dev_base, const gscan_seg *segs, size_t nseg,
                      void *stream, gscan_dev_result *res)
{
    if (!c || !db || !res || (!segs && nseg)) return GSCAN_EINVAL;
    if (db->db.tier == GSCAN_TIER_NULL || db->db.tier == GSCAN_TIER_ANCHORED) return fail(c, GSCAN_EINVAL, "nothing to scan for this pattern (it matches \"\", or only at the restart position / chunk end)");
    HIPCHK(c, hipSetDevice(c->hip_dev));
    hipStream_t st = stream ? (hipStream_t)stream : c->compute;
    c->dv_stream = st;
    int rc = ensure_prog(c, db, st);
    if (rc) return rc;
    uint32_t tile_bytes = 0, nw = 1;
    gscan::scan_geometry(db->db.tier, c->variant, db->db.prog, &tile_bytes, &nw);
    c->dv_last_nw = nw;
    const size_t K = gscan::kShards;

    // tile table: rebuilt only when the segment table or the tile size changed
    bool same = c->dv_last_tile_bytes == tile_bytes && c->dv_last_segs.size() == nseg &&
                (nseg == 0 || !memcmp(c->dv_last_segs.data(), segs, nseg * sizeof(gscan_seg)));
    if (!same) {
        std::vector<uint32_t> &tf = c->dv_tile_first_h;
        tf.assign(nseg + 1, 0);
        uint64_t nt = 0;
        for (size_t i = 0; i < nseg; i++) {
            if (segs[i].len > kMaxChunk) return fail(c, GSCAN_ETOOBIG, "segment %zu longer than a chunk", i);
            if (segs[i].offset & 15) return fail(c, GSCAN_EINVAL, "segment %zu is not 16-byte aligned", i);
            tf[i] = (uint32_t)nt;
            nt += (segs[i].len + tile_bytes - 1) / tile_bytes;
        }
        if (nt >= 0xffffffffull) return fail(c, GSCAN_ETOOBIG, "too many tiles");
        tf[nseg] = (uint32_t)nt;
        std::vector<gscan::TileDesc> td((size_t)nt);
        for (size_t i = 0; i < nseg; i++)
            for (uint32_t t = tf[i]; t < tf[i + 1]; t++) td[t] = {segs[i].offset, segs[i].len, (t - tf[i]) * tile_bytes};
        HIPCHK(c, hipStreamSynchronize(st)); // earlier scans may still read the old table
        if (nt + 1 > c->dv_tiles_cap) {
            if (c->dv_desc) hipFree(c->dv_desc);
            if (c->dv_tiles) hipFree(c->dv_tiles);
            c->dv_desc = nullptr;
            c->dv_tiles = nullptr;
            c->dv_tiles_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->dv_desc, (size_t)(nt + 1) * gscan::kMaxWavesPerTile * 8));
            HIPCHK(c, hipMalloc((void **)&c->dv_tiles, (size_t)(nt + 1) * sizeof(gscan::TileDesc)));
            c->dv_tiles_cap = (size_t)nt + 1;
        }
        if (nt) HIPCHK(c, hipMemcpy(c->dv_tiles, td.data(), (size_t)nt * sizeof(gscan::TileDesc), hipMemcpyHostToDevice));
        c->dv_last_segs.assign(segs, segs + nseg);
        c->dv_last_tile_bytes = tile_bytes;
    }
    const uint32_t n_tiles = c->dv_tile_first_h.empty() ? 0 : c->dv_tile_first_h.back();

    // record buffer: the caller's capacity if set, else arena bytes / 16; kShards equal regions
    uint64_t total_bytes = 0;
    for (size_t i = 0; i < nseg; i++) total_bytes += segs[i].len;
    size_t want = c->dev_cap_req ? c->dev_cap_req : std::max<size_t>((size_t)(total_bytes / 16), 1u << 16);
    want = std::min<size_t>((want + K - 1) / K * K, 0xfffffff8u);
    if (want != c->dv_rec_cap && (c->dev_cap_req || want > c->dv_rec_cap)) {
        HIPCHK(c, hipStreamSynchronize(st));
        if (c->dv_recs) hipFree(c->dv_recs);
        c->dv_recs = nullptr;
        c->dv_rec_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->dv_recs, want * 4));
        c->dv_rec_cap = want;
    }
    if (db->db.prog.resolve && c->dv_ends_cap < c->dv_rec_cap) {
        HIPCHK(c, hipStreamSynchronize(st));
        if (c->dv_ends) hipFree(c->dv_ends);
        c->dv_ends = nullptr;
        c->dv_ends_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->dv_ends, c->dv_rec_cap * 4));
        c->dv_ends_cap = c->dv_rec_cap;
    }

    HIPCHK(c, hipMemsetAsync(c->dv_counter, 0, kCounterWords * 4, st));
    ScanArgs a;
    memset(&a, 0, sizeof a);
    a.base = (const uint8_t *)dev_base;
    a.tiles = c->dv_tiles;
    a.n_tiles = n_tiles;
    a.cap_shard = (uint32_t)(c->dv_rec_cap / K);
    a.recs = c->dv_recs;
    a.desc = c->dv_desc;
    a.counter = c->dv_counter;
    a.prog = c->d_prog;
    gscan::fill_program(a, db->db.prog);
    if (c->k3_depth) a.k3_depth = (uint32_t)c->k3_depth;
    if (c->ev_used == c->ev_pool.size() && c->ev_pool.size() < 4096) {
        EvPair e;
        HIPCHK(c, hipEventCreate(&e.a));
        HIPCHK(c, hipEventCreate(&e.b));
        c->ev_pool.push_back(e);
    }
    bool timed = c->ev_used < c->ev_pool.size();
    if (timed) HIPCHK(c, hipEventRecord(c->ev_pool[c->ev_used].a, st));
    if (n_tiles) HIPCHK(c, gscan::launch_scan(db->db.tier, c->variant, a, grid_for(c, db->db, n_tiles), st));
    if (n_tiles && gscan::scan_needs_settle(db->db.tier, db->db.prog)) HIPCHK(c, gscan::launch_settle(a, nw, st)); // inside the timed region
    if (n_tiles && db->db.prog.resolve) HIPCHK(c, gscan::launch_resolve(a, nw, c->dv_ends, st));                   // likewise
    if (timed) {
        HIPCHK(c, hipEventRecord(c->ev_pool[c->ev_used].b, st));
        c->ev_used++;
    }
    res->recs = c->dv_recs;
    res->desc = (const uint64_t *)c->dv_desc;
    res->n_tiles = (uint64_t)n_tiles * nw; // descriptors: one per wave sub-tile
    res->tile_bytes = tile_bytes / nw;
    res->total = 0;
    res->overflow = 0;
    res->ends = db->db.prog.resolve ? c->dv_ends : nullptr;
    return GSCAN_OK;
}

int gscan_dev_sync(gscan_ctx *c, gscan_dev_result *res)
{
    if (!c || !res) return GSCAN_EINVAL;
    HIPCHK(c, hipSetDevice(c->hip_dev));
    hipStream_t st = c->dv_stream ? c->dv_stream : c->compute;
    uint32_t h[kCounterWords] = {0};
    HIPCHK(c, hipMemcpyAsync(h, c->dv_counter, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    res->total = 0;
    for (size_t k = 0; k < (size_t)gscan::kShards; k++) res->total += h[k * kCS];
    res->overflow = h[gscan::kShards * kCS] != 0;
    if (!res->overflow) res->total -= h[gscan::kShards * kCS + 1]; // records struck out by the second pass
    return GSCAN_OK;
}

long gscan_dev_fetch(gscan_ctx *c, const gscan_dev_result *res, size_t seg, uint32_t *out, size_t cap)
{
    if (!c || !res) return GSCAN_EINVAL;
    if (seg + 1 >= c->dv_tile_first_h.size()) return fail(c, GSCAN_EINVAL, "segment index out of range");
    HIPCHK(c, hipSetDevice(c->hip_dev));
    hipStream_t st = c->dv_stream ? c->dv_stream : c->compute;
    HIPCHK(c, hipStreamSynchronize(st));
    const size_t t0 = (size_t)c->dv_tile_first_h[seg] * c->dv_last_nw, t1 = (size_t)c->dv_tile_first_h[seg + 1] * c->dv_last_nw; // descriptors of the segment
    std::vector<unsigned long long> d(t1 - t0);
    if (t1 > t0) HIPCHK(c, hipMemcpy(d.data(), c->dv_desc + t0, (size_t)(t1 - t0) * 8, hipMemcpyDeviceToHost));
    const size_t cap_shard = c->dv_rec_cap / gscan::kShards;
    size_t n = 0;
    for (unsigned long long v : d) {
        uint32_t cnt = (uint32_t)v;
        size_t base = (size_t)(v >> 32);
        if (!cnt) continue;
        if (base % cap_shard + cnt > cap_shard) return fail(c, GSCAN_EHIP, "record buffer overflowed; raise gscan_set_capacity");
        if (out && n + cnt <= cap) {
            HIPCHK(c, hipMemcpy(out + n, c->dv_recs + base, (size_t)cnt * 4, hipMemcpyDeviceToHost));
            size_t kept = 0;
            for (uint32_t i = 0; i < cnt; i++)
                if (out[n + i] != gscan::kStruck) out[n + kept++] = out[n + i];
            n += kept;
        } else {
            n += cnt; // sizing call: an upper bound
        }
    }
    return (long)n;
}

int gscan_kernel_time(gscan_ctx *c, double *sum_ms, uint64_t *launches, int reset)
{
    if (!c) return GSCAN_EINVAL;
    HIPCHK(c, hipSetDevice(c->hip_dev));
    double sum = 0;
    for (size_t i = 0; i < c->ev_used; i++) {
        float ms = 0;
        HIPCHK(c, hipEventSynchronize(c->ev_pool[i].b));
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        sum += ms;
    }
    if (sum_ms) *sum_ms = sum;
    if (launches) *launches = c->ev_used;
    if (reset) c->ev_used = 0;
    return GSCAN_OK;
}

} // extern "C"
