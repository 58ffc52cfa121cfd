// pipe_probe.hip -- what takes the host -> HBM pipe from the 57 GB/s the link moves in dma_probe to the 47-52 GB/s the
// engine's DMA side moves with the reads stubbed out (GSCAN_DIAG=2, profiles/r05_a_*)?  The engine's way of driving the copy
// streams, rebuilt one ingredient at a time.  16 GiB per mode as 8 MiB (or --mib) copies out of a ring of 16 staging blocks:
//
//   0  one thread, two streams alternating, blocks from hipHostMalloc, one sync at the end          (dma_probe's 57 GB/s)
//   1  + an event recorded behind every copy
//   2  + a block is reused only after its event has been waited for (hipEventSynchronize)
//   3  = 2 with blocks from an anonymous mapping (MADV_HUGEPAGE, touched) that is hipHostRegister'ed  (gscan_prefault's)
//   4  = 2 driven by EIGHT threads, two blocks each, enqueue + record under a per-stream mutex        (the reader pool)
//   5  = 4 with registered blocks
//   6  = 4 + on stream 0, behind every 8th copy: a memset, a small kernel and three small D2H copies  (slot_launch)
//   7  = 6 with registered blocks                                                                     (the engine as it is)
//   8  = 4 with hipEventQuery polling of the other threads' blocks before every wait                  (Ingest::reap)
//   9  = 4 on FOUR streams
//
//   pipe_probe [--mib 8] [--gib 16] [--modes 0,1,2,...]
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) printf("%s -> %s\n", #x, hipGetErrorString(e_)); \
    } while (0)

__global__ void k_small(uint32_t *p, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1;
}

struct Block {
    char *p = nullptr;
    hipEvent_t ev = nullptr;
    bool busy = false;
};

int main(int argc, char **argv)
{
    size_t piece = (size_t)8 << 20, total = (size_t)16 << 30;
    std::vector<int> modes = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--mib")) piece = (size_t)atoi(argv[i + 1]) << 20;
        else if (!strcmp(argv[i], "--gib")) total = (size_t)atoi(argv[i + 1]) << 30;
        else if (!strcmp(argv[i], "--modes")) {
            modes.clear();
            for (char *q = strtok(argv[i + 1], ","); q; q = strtok(nullptr, ",")) modes.push_back(atoi(q));
        }
    }
    CK(hipSetDevice(0));
    const size_t dev_bytes = (size_t)1 << 30;
    char *dev = nullptr;
    CK(hipMalloc((void **)&dev, dev_bytes));
    uint32_t *d_small = nullptr, *h_small = nullptr;
    CK(hipMalloc((void **)&d_small, 1 << 20));
    CK(hipHostMalloc((void **)&h_small, 1 << 20, hipHostMallocDefault));
    hipStream_t st[4];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int kBlocks = 16;
    std::vector<Block> pinned(kBlocks), reg(kBlocks);
    for (Block &b : pinned) {
        CK(hipHostMalloc((void **)&b.p, piece + 4096, hipHostMallocDefault));
        memset(b.p, 1, piece);
        CK(hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
    }
    {
        const size_t huge = (size_t)2 << 20, stride = (piece + 4096 + huge - 1) / huge * huge;
        char *m = (char *)mmap(nullptr, stride * kBlocks + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        char *base = (char *)(((uintptr_t)m + huge - 1) & ~(uintptr_t)(huge - 1));
        madvise(base, stride * kBlocks, MADV_HUGEPAGE);
        for (int k = 0; k < kBlocks; k++) {
            reg[k].p = base + k * stride;
            for (size_t o = 0; o < piece + 4096; o += 4096) reg[k].p[o] = 1;
            CK(hipHostRegister(reg[k].p, piece + 4096, hipHostRegisterDefault));
            CK(hipEventCreateWithFlags(&reg[k].ev, hipEventDisableTiming));
        }
    }
    const size_t n_pieces = total / piece;
    auto dst_of = [&](size_t i) { return dev + (i * piece) % (dev_bytes - piece + 1) / piece * piece; };
    for (int mode : modes) {
        const bool registered = mode == 3 || mode == 5 || mode == 7;
        std::vector<Block> &blk = registered ? reg : pinned;
        for (Block &b : blk) b.busy = false;
        const int ns = mode == 9 ? 4 : 2;
        const bool threaded = mode >= 4;
        const bool extras = mode == 6 || mode == 7;
        const bool poll = mode == 8;
        double best = 1e9;
        for (int rep = 0; rep < 2; rep++) {
            const double t0 = now();
            if (!threaded) {
                for (size_t i = 0; i < n_pieces; i++) {
                    Block &b = blk[i % kBlocks];
                    if (mode >= 2 && b.busy) CK(hipEventSynchronize(b.ev));
                    CK(hipMemcpyAsync(dst_of(i), b.p, piece, hipMemcpyHostToDevice, st[i % ns]));
                    if (mode >= 1) CK(hipEventRecord(b.ev, st[i % ns]));
                    b.busy = true;
                }
            } else {
                std::atomic<size_t> next{0};
                std::mutex order[4];
                std::vector<std::thread> th;
                for (int t = 0; t < 8; t++)
                    th.emplace_back([&, t] {
                        CK(hipSetDevice(0));
                        int mine = 0;
                        for (;;) {
                            const size_t i = next.fetch_add(1);
                            if (i >= n_pieces) break;
                            Block &b = blk[2 * t + (mine++ & 1)];
                            if (poll)
                                for (int k = 0; k < kBlocks; k++)
                                    if (blk[k].busy && hipEventQuery(blk[k].ev) != hipSuccess) (void)hipGetLastError();
                            if (b.busy) CK(hipEventSynchronize(b.ev));
                            const int s = (int)(i % ns);
                            std::lock_guard<std::mutex> lk(order[s]);
                            CK(hipMemcpyAsync(dst_of(i), b.p, piece, hipMemcpyHostToDevice, st[s]));
                            CK(hipEventRecord(b.ev, st[s]));
                            b.busy = true;
                            if (extras && s == 0 && i % 8 == 6) { // what follows a 64 MiB window's last piece on the first stream
                                CK(hipMemsetAsync(d_small, 0, 8192, st[0]));
                                hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st[0], d_small, 16384u);
                                CK(hipMemcpyAsync(h_small, d_small, 8200, hipMemcpyDeviceToHost, st[0]));
                                CK(hipMemcpyAsync(h_small + 4096, d_small + 4096, 40960, hipMemcpyDeviceToHost, st[0]));
                                CK(hipMemcpyAsync(h_small + 65536, d_small + 65536, 65536, hipMemcpyDeviceToHost, st[0]));
                            }
                        }
                    });
                for (auto &x : th) x.join();
            }
            for (int s = 0; s < ns; s++) CK(hipStreamSynchronize(st[s]));
            best = std::min(best, now() - t0);
            for (Block &b : blk) b.busy = false;
        }
        printf("mode %d  %4zu MiB pieces, %d streams, %s, %s%s%s: %6.2f GB/s\n", mode, piece >> 20, ns, registered ? "registered" : "hipHostMalloc",
               threaded ? "8 threads" : "1 thread", extras ? " + scan-launch extras" : "", poll ? " + event polling" : "", (double)(n_pieces * piece) / best / 1e9);
        fflush(stdout);
    }
    return 0;
}
