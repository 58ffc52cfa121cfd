// host_probe.hip -- costs of the host->HBM leg on the GPU box: runtime init, pinned/device allocation,
// page cache -> pinned copy (pread vs mmap+memcpy, 1..N threads), H2D from pinned / pageable / registered memory.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); } } while (0)
int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/dev/shm/probe.bin";
    const size_t N = (size_t)1 << 30;
    double t0 = now();
    CK(hipInit(0));
    int n = 0; CK(hipGetDeviceCount(&n)); CK(hipSetDevice(0)); CK(hipFree(0));
    printf("hip init + first context: %.3f s (devices %d)\n", now() - t0, n);
    { // file of 1 GiB in page cache
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
        std::vector<char> blk(1 << 20, 'x');
        for (size_t i = 0; i < N >> 20; i++) { blk[0] = (char)i; if (write(fd, blk.data(), blk.size()) < 0) return 1; }
        close(fd);
    }
    for (size_t mb : {32, 256, 1024}) {
        void *p = nullptr; t0 = now(); CK(hipHostMalloc(&p, mb << 20, hipHostMallocDefault)); double a = now() - t0;
        t0 = now(); CK(hipHostFree(p)); double f = now() - t0;
        void *d = nullptr; t0 = now(); CK(hipMalloc(&d, mb << 20)); double da = now() - t0; t0 = now(); CK(hipFree(d));
        printf("hipHostMalloc %4zu MiB: %.4f s (free %.4f)   hipMalloc: %.4f s (free %.4f)\n", mb, a, f, da, now() - t0);
    }
    char *pin = nullptr; CK(hipHostMalloc((void **)&pin, N, hipHostMallocDefault));
    char *dev = nullptr; CK(hipMalloc((void **)&dev, N));
    int fd = open(path, O_RDONLY);
    for (int threads : {1, 2, 4, 8, 16}) {
        for (int mode = 0; mode < 2; mode++) {
            char *map = mode ? (char *)mmap(nullptr, N, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, fd, 0) : nullptr;
            t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < threads; t++)
                th.emplace_back([&, t] {
                    size_t lo = N / threads * t, hi = t == threads - 1 ? N : N / threads * (t + 1);
                    if (mode) memcpy(pin + lo, map + lo, hi - lo);
                    else for (size_t o = lo; o < hi;) { ssize_t r = pread(fd, pin + o, hi - o, (off_t)o); if (r <= 0) break; o += (size_t)r; }
                });
            for (auto &x : th) x.join();
            double dt = now() - t0;
            if (map) munmap(map, N);
            printf("page cache -> pinned, %2d thread(s), %s: %.2f GB/s\n", threads, mode ? "mmap+memcpy" : "pread      ", N / dt / 1e9);
        }
    }
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; rep++) { t0 = now(); CK(hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); printf("H2D pinned 1 GiB: %.2f GB/s\n", N / (now() - t0) / 1e9); }
    for (size_t piece : {(size_t)1 << 20, (size_t)8 << 20, (size_t)32 << 20}) {
        t0 = now();
        for (size_t o = 0; o < N; o += piece) CK(hipMemcpyAsync(dev + o, pin + o, piece, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        printf("H2D pinned in %zu MiB pieces: %.2f GB/s\n", piece >> 20, N / (now() - t0) / 1e9);
    }
    { // pageable: straight from the mapping
        char *map = (char *)mmap(nullptr, N, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, fd, 0);
        for (int rep = 0; rep < 2; rep++) { t0 = now(); CK(hipMemcpy(dev, map, N, hipMemcpyHostToDevice)); printf("H2D pageable (mmap of page cache) 1 GiB: %.2f GB/s\n", N / (now() - t0) / 1e9); }
        t0 = now(); hipError_t e = hipHostRegister(map, N, hipHostRegisterDefault); double rt = now() - t0;
        printf("hipHostRegister(file mapping, 1 GiB): %s in %.3f s\n", hipGetErrorString(e), rt);
        if (e == hipSuccess) {
            for (int rep = 0; rep < 2; rep++) { t0 = now(); CK(hipMemcpyAsync(dev, map, N, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); printf("H2D registered mapping 1 GiB: %.2f GB/s\n", N / (now() - t0) / 1e9); }
            t0 = now(); CK(hipHostUnregister(map)); printf("unregister %.3f s\n", now() - t0);
        } else (void)hipGetLastError();
        munmap(map, N);
    }
    { // anonymous memory registered
        char *anon = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        memset(anon, 1, N);
        t0 = now(); hipError_t e = hipHostRegister(anon, N, hipHostRegisterDefault);
        printf("hipHostRegister(anon 1 GiB): %s in %.3f s\n", hipGetErrorString(e), now() - t0);
        if (e == hipSuccess) CK(hipHostUnregister(anon));
        munmap(anon, N);
    }
    // kernel reading host pinned memory directly (zero-copy over PCIe)
    close(fd); unlink(path);
    return 0;
}
