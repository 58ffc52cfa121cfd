// hostcopy.cc -- page cache -> pinned staging block without going through the CPU caches.
//
// The readers' share of every byte is one copy: file pages -> the pinned block the DMA engine reads next.  pread(2) does it
// with ordinary stores: the block's lines are first read for ownership, then sit dirty in the writing core's L2 / its CCD's L3
// until the DMA's reads pull them out again.  A mapping of the file range + a copy with non-temporal stores writes the block
// straight to DRAM (no read for ownership, nothing dirty left in the caches for the device's reads to chase); the sources are
// streamed through with ordinary loads.  Measured on the MI355X box: profiles/r02_*e2e*.
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace gscan {

__attribute__((target("avx2"))) static void nt_copy_avx2(void *dst, const void *src, size_t n)
{
    uint8_t *d = (uint8_t *)dst;
    const uint8_t *s = (const uint8_t *)src;
    // head: up to the destination's next 32-byte boundary
    const size_t head = ((uintptr_t)d & 31) ? 32 - ((uintptr_t)d & 31) : 0;
    if (head) {
        const size_t h = head < n ? head : n;
        memcpy(d, s, h);
        d += h, s += h, n -= h;
    }
    for (; n >= 128; d += 128, s += 128, n -= 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(s)), b = _mm256_loadu_si256((const __m256i *)(s + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(s + 64)), e = _mm256_loadu_si256((const __m256i *)(s + 96));
        _mm256_stream_si256((__m256i *)(d), a);
        _mm256_stream_si256((__m256i *)(d + 32), b);
        _mm256_stream_si256((__m256i *)(d + 64), c);
        _mm256_stream_si256((__m256i *)(d + 96), e);
    }
    if (n) memcpy(d, s, n);
    _mm_sfence(); // the streamed stores are visible before the DMA is queued
}

// dst <- src, n bytes, the destination written with non-temporal stores where the CPU has AVX2 (else a plain memcpy).
void nt_copy(void *dst, const void *src, size_t n)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 4096) nt_copy_avx2(dst, src, n);
    else memcpy(dst, src, n);
}

} // namespace gscan
