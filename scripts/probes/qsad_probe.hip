#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(uint64_t *o, uint64_t src, uint32_t ref) {
    o[0] = __builtin_amdgcn_qsad_pk_u16_u8(src, ref, 0ull);
    o[1] = __builtin_amdgcn_mqsad_pk_u16_u8(src, ref, 0ull);
}
template <int MODE>
__global__ void rate(uint32_t *o, uint32_t n) {
    uint64_t a = threadIdx.x * 0x0101010101ull + 12345, acc = 0; uint32_t r = 0x64636261u + blockIdx.x; uint32_t b = threadIdx.x, c = 7;
    for (uint32_t i = 0; i < n; i++) {
        if (MODE == 0) { acc = __builtin_amdgcn_qsad_pk_u16_u8(a, r, acc); a += acc; }
        if (MODE == 1) { b = __builtin_amdgcn_alignbyte(b, c, 1) ^ r; c += b; }
        if (MODE == 2) { acc = __builtin_amdgcn_mqsad_pk_u16_u8(a, r, acc); a += acc; }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc + (uint32_t)a + b + c;
}
int main() {
    uint64_t *d; hipMalloc(&d, 64); uint64_t h[2];
    const char *s = "abcdefgh"; uint64_t src; memcpy(&src, s, 8);
    for (const char *ref : {"abcd", "bcde", "cdef", "defg", "abc\0", "\0bcd", "bcd\0"}) {
        uint32_t r; memcpy(&r, ref, 4);
        hipLaunchKernelGGL(k, 1, 1, 0, 0, d, src, r); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("src abcdefgh ref %02x%02x%02x%02x: qsad [%u %u %u %u]  mqsad [%u %u %u %u]\n", (unsigned char)ref[0], (unsigned char)ref[1], (unsigned char)ref[2], (unsigned char)ref[3],
               (unsigned)(h[0] & 0xffff), (unsigned)(h[0] >> 16 & 0xffff), (unsigned)(h[0] >> 32 & 0xffff), (unsigned)(h[0] >> 48),
               (unsigned)(h[1] & 0xffff), (unsigned)(h[1] >> 16 & 0xffff), (unsigned)(h[1] >> 32 & 0xffff), (unsigned)(h[1] >> 48));
    }
    uint32_t *o; hipMalloc(&o, 1024 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9;
        for (int it = 0; it < 3; it++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate<0>, 1024 * 8, 256, 0, 0, o, 4096u);
            if (mode == 1) hipLaunchKernelGGL(rate<1>, 1024 * 8, 256, 0, 0, o, 4096u);
            if (mode == 2) hipLaunchKernelGGL(rate<2>, 1024 * 8, 256, 0, 0, o, 4096u);
            hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double iters = 1024.0 * 8 * 4 * 4096; // wave-iterations
        printf("mode %d (%s): %.3f ms  -> %.2f ns per wave-iteration (dependent chain)\n", mode, mode == 0 ? "qsad + add64" : mode == 1 ? "alignbyte+xor + add" : "mqsad + add64", best, best * 1e6 / iters * (256 * 4 * 4) / 1.0);
    }
    return 0;
}
