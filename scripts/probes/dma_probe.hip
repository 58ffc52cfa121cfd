// dma_probe.hip -- what the host -> HBM link of THIS box gives, and what takes it away (VERDICT r1 task 4).
//
//   dma_probe [file-in-page-cache]        (run once as is and once under HSA_ENABLE_SDMA=0: blit kernels instead of the SDMA engines)
//
// 1. pure DMA: hipMemcpyAsync pinned -> device in pieces of 8 / 16 / 32 / 1024 MiB on 1 and 2 streams, for the pinned-memory
//    flavours hipHostMalloc offers (default, non-coherent, write-combined);
// 2. the same while T threads copy page cache -> pinned (pread), i.e. what the engine's readers do next to the DMA;
// 3. a kernel reading the pinned memory itself (zero-copy over PCIe) with the scan kernels' load pattern.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                              \
    do {                                                                   \
        hipError_t e_ = (x);                                               \
        if (e_ != hipSuccess) printf("%s -> %s\n", #x, hipGetErrorString(e_)); \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read(const u32x4 *src, size_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const u32x4 v = __builtin_nontemporal_load(src + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/dev/shm/dma_probe.bin";
    const size_t N = (size_t)1 << 30;
    printf("HSA_ENABLE_SDMA=%s\n", getenv("HSA_ENABLE_SDMA") ? getenv("HSA_ENABLE_SDMA") : "(unset)");
    CK(hipSetDevice(0));
    char *dev = nullptr;
    CK(hipMalloc((void **)&dev, N));
    hipStream_t st[2];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    { // 2 GiB file in the page cache for the contention part
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
        std::vector<char> blk(1 << 20, 'x');
        for (size_t i = 0; i < (2 * N) >> 20; i++) {
            blk[0] = (char)i;
            if (write(fd, blk.data(), blk.size()) < 0) return 1;
        }
        close(fd);
    }
    struct Flavour {
        const char *name;
        unsigned flags;
    } flavours[] = {{"default", hipHostMallocDefault}, {"non-coherent", hipHostMallocNonCoherent}, {"write-combined", hipHostMallocWriteCombined}};
    for (const Flavour &f : flavours) {
        char *pin = nullptr;
        double t0 = now();
        if (hipHostMalloc((void **)&pin, N, f.flags) != hipSuccess) {
            (void)hipGetLastError();
            printf("%-14s hipHostMalloc refused\n", f.name);
            continue;
        }
        const double alloc_s = now() - t0;
        memset(pin, 1, N);
        for (size_t piece : {(size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20, N})
            for (int ns : {1, 2}) {
                double best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    t0 = now();
                    size_t k = 0;
                    for (size_t o = 0; o < N; o += piece, k++) CK(hipMemcpyAsync(dev + o, pin + o, piece, hipMemcpyHostToDevice, st[k % ns]));
                    for (int s = 0; s < ns; s++) CK(hipStreamSynchronize(st[s]));
                    best = std::min(best, now() - t0);
                }
                printf("%-14s (alloc %.3f s/GiB) pieces %4zu MiB, %d stream(s): %6.2f GB/s\n", f.name, alloc_s, piece >> 20, ns, N / best / 1e9);
            }
        // with T reader threads copying page cache -> another pinned buffer of the same flavour
        char *pin2 = nullptr;
        if (hipHostMalloc((void **)&pin2, N, f.flags) == hipSuccess) {
            int fd = open(path, O_RDONLY);
            for (int threads : {8, 16}) {
                std::atomic<bool> stop{false};
                std::atomic<size_t> copied{0};
                std::vector<std::thread> th;
                for (int t = 0; t < threads; t++)
                    th.emplace_back([&, t] {
                        const size_t span = N / threads;
                        while (!stop) {
                            for (size_t o = 0; o < span && !stop; o += 16 << 20) {
                                const size_t n = std::min<size_t>(16 << 20, span - o);
                                if (pread(fd, pin2 + t * span + o, n, (off_t)(t * span + o)) <= 0) return;
                                copied += n;
                            }
                        }
                    });
                usleep(50000);
                const size_t c0 = copied;
                t0 = now();
                for (int rep = 0; rep < 4; rep++) {
                    size_t k = 0;
                    for (size_t o = 0; o < N; o += 16 << 20, k++) CK(hipMemcpyAsync(dev + o, pin + o, 16 << 20, hipMemcpyHostToDevice, st[k & 1]));
                    for (auto &s : st) CK(hipStreamSynchronize(s));
                }
                const double dt = now() - t0;
                const size_t c1 = copied;
                stop = true;
                for (auto &x : th) x.join();
                printf("%-14s DMA 16 MiB pieces with %2d pread threads: DMA %6.2f GB/s, pread %6.2f GB/s\n", f.name, threads, 4.0 * N / dt / 1e9, (c1 - c0) / dt / 1e9);
            }
            close(fd);
            CK(hipHostFree(pin2));
        }
        { // zero-copy: a kernel reads the pinned memory over PCIe
            uint32_t *sink = nullptr;
            CK(hipMalloc((void **)&sink, 64));
            char *dptr = nullptr;
            if (hipHostGetDevicePointer((void **)&dptr, pin, 0) == hipSuccess) {
                for (int grid : {256, 1024, 4096}) {
                    double best = 1e9;
                    for (int rep = 0; rep < 3; rep++) {
                        t0 = now();
                        hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, st[0], (const u32x4 *)dptr, N / 16, sink);
                        CK(hipStreamSynchronize(st[0]));
                        best = std::min(best, now() - t0);
                    }
                    printf("%-14s kernel reads pinned memory (zero-copy), grid %4d: %6.2f GB/s\n", f.name, grid, N / best / 1e9);
                }
            } else {
                (void)hipGetLastError();
            }
            CK(hipFree(sink));
        }
        CK(hipHostFree(pin));
    }
    unlink(path);
    return 0;
}
