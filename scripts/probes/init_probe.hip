// init_probe.hip -- where a one-shot process's start-up goes: each first call of the HIP runtime timed on its own.
//   hipcc --offload-arch=gfx950 -O2 init_probe.hip -o init_probe -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void touch(int *p) { p[threadIdx.x] = (int)threadIdx.x; }

int main(int argc, char **argv)
{
    const bool hsa_first = argc > 1 && argv[1][0] == 'h';
    double t = now();
    auto lap = [&](const char *what) {
        const double n = now();
        long rss[3] = {0, 0, 0}; // anon, file, shmem (kB)
        if (FILE *f = fopen("/proc/self/status", "r")) {
            char line[256];
            while (fgets(line, sizeof line, f)) {
                sscanf(line, "RssAnon: %ld", &rss[0]);
                sscanf(line, "RssFile: %ld", &rss[1]);
                sscanf(line, "RssShmem: %ld", &rss[2]);
            }
            fclose(f);
        }
        printf("%-34s %8.4f s   RssAnon %7ld kB  RssFile %7ld kB  RssShmem %7ld kB\n", what, n - t, rss[0], rss[1], rss[2]);
        t = now();
    };
    if (hsa_first) {
        hsa_init();
        lap("hsa_init");
    }
    int n = 0;
    hipGetDeviceCount(&n);
    lap("hipGetDeviceCount (hipInit)");
    hipSetDevice(0);
    lap("hipSetDevice");
    hipStream_t s[4];
    for (int i = 0; i < 4; i++) {
        hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        lap("hipStreamCreate");
    }
    void *d = nullptr;
    hipMalloc(&d, 256u << 20);
    lap("hipMalloc 256 MiB");
    void *h = nullptr;
    hipHostMalloc(&h, 16u << 20, 0);
    lap("hipHostMalloc 16 MiB");
    hipMemcpyAsync(d, h, 16u << 20, hipMemcpyHostToDevice, s[0]);
    hipStreamSynchronize(s[0]);
    lap("first H2D 16 MiB");
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s[1], (int *)d);
    hipStreamSynchronize(s[1]);
    lap("first kernel (module load)");
    hipFree(d);
    lap("hipFree");
    hipHostFree(h);
    lap("hipHostFree");
    for (int i = 0; i < 4; i++) hipStreamDestroy(s[i]);
    lap("4 x hipStreamDestroy");
    // what the process still OWNS when it leaves, for the parent to time the exit against (argv: h|x  vram_GiB  pinned_MiB  streams)
    const long vram_gib = argc > 2 ? atol(argv[2]) : 0, pinned_mib = argc > 3 ? atol(argv[3]) : 0, streams = argc > 4 ? atol(argv[4]) : 0;
    for (long i = 0; i < vram_gib * 4; i++) hipMalloc(&d, 256u << 20);
    for (long i = 0; i < pinned_mib / 16; i++) {
        hipHostMalloc(&h, 16u << 20, 0);
        memset(h, 1, 16u << 20);
    }
    for (long i = 0; i < streams; i++) hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking);
    lap("left allocated at exit");
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    printf("exit_at %.6f\n", ts.tv_sec + ts.tv_nsec * 1e-9);
    fflush(stdout);
    _exit(0);
}
