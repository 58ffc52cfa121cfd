// exit_age_probe.hip -- what makes a GPU process's exit cost 0.1 s once it is older than about half a second?
//   exit_age_probe <what> <sleep_ms>      what: 0 = hipInit only, 1 = + one stream, 2 = + a 64 MiB pinned block and one DMA,
//                                               3 = 2 + a kernel launch
// The process does <what>, sleeps, prints a time stamp and leaves through _exit; the parent (scripts below) measures how long
// after that stamp the child is gone.
//   hipcc --offload-arch=gfx950 -O2 exit_age_probe.hip -o exit_age_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>

__global__ void touch(int *p) { p[threadIdx.x] = (int)threadIdx.x; }

int main(int argc, char **argv)
{
    const int what = argc > 1 ? atoi(argv[1]) : 0, sleep_ms = argc > 2 ? atoi(argv[2]) : 0;
    int n = 0;
    hipGetDeviceCount(&n);
    hipSetDevice(0);
    hipStream_t s = nullptr;
    void *h = nullptr, *d = nullptr;
    if (what >= 1) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (what >= 2) {
        hipHostMalloc(&h, 64u << 20, 0);
        hipMalloc(&d, 64u << 20);
        hipMemcpyAsync(d, h, 64u << 20, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
    }
    if (what >= 3) {
        hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, (int *)d);
        hipStreamSynchronize(s);
    }
    usleep((useconds_t)sleep_ms * 1000);
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    printf("%.6f\n", ts.tv_sec + ts.tv_nsec * 1e-9);
    fflush(stdout);
    _exit(0);
}
