/*
 * linesum.c -- TEST INFRASTRUCTURE ONLY (checker, like the rest of oracle/).
 *
 * An order-independent digest of a stream of lines, for comparing the output of `grab -n N` (whose order across files is
 * unspecified, /root/reference/README.md:206-216 sorts before it compares) with the reference's at sizes where sorting
 * gigabytes of output is out of proportion: 172.9 M lines for BASELINE configs[2] at 64 GiB.
 *
 *     <producer> | linesum      prints: "<lines> <sum of the 64-bit hashes of all lines, hex> <xor of them, hex> <bytes>"
 *
 * Two outputs have the same multiset of lines iff (with overwhelming probability) all four numbers agree.  The hash is
 * a 64-bit multiply-xorshift over 8-byte words of the line (the newline excluded), seeded with its length.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static inline uint64_t mix(uint64_t h)
{
    h ^= h >> 32;
    h *= 0xd6e8feb86659fd93ull;
    h ^= h >> 29;
    h *= 0xd6e8feb86659fd93ull;
    h ^= h >> 32;
    return h;
}

static uint64_t line_hash(const unsigned char *p, size_t n)
{
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)n;
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = mix(h ^ w) + 0x632be59bd9b4e019ull;
        p += 8;
        n -= 8;
    }
    if (n) {
        uint64_t w = 0;
        memcpy(&w, p, n);
        h = mix(h ^ w) + 0x632be59bd9b4e019ull;
    }
    return mix(h);
}

int main(void)
{
    const size_t cap = (size_t)64 << 20;
    unsigned char *buf = malloc(cap);
    if (!buf) return 2;
    size_t have = 0;
    uint64_t lines = 0, sum = 0, x = 0, bytes = 0;
    for (;;) {
        ssize_t r = read(0, buf + have, cap - have);
        if (r < 0) return 2;
        if (r == 0) break;
        bytes += (uint64_t)r;
        have += (size_t)r;
        size_t at = 0;
        for (;;) {
            unsigned char *nl = memchr(buf + at, '\n', have - at);
            if (!nl) break;
            const uint64_t h = line_hash(buf + at, (size_t)(nl - (buf + at)));
            sum += h;
            x ^= h;
            lines++;
            at = (size_t)(nl - buf) + 1;
        }
        memmove(buf, buf + at, have - at);
        have -= at;
        if (have == cap) return 3; /* a line longer than 64 MiB: not this tool's business */
    }
    if (have) { /* a last line without a newline */
        const uint64_t h = line_hash(buf, have);
        sum += h;
        x ^= h;
        lines++;
    }
    printf("%llu %016llx %016llx %llu\n", (unsigned long long)lines, (unsigned long long)sum, (unsigned long long)x, (unsigned long long)bytes);
    return 0;
}
