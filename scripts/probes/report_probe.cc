// report_probe.cc -- what a printed line costs the host, per path of grab_report_chunk (filegrep.cc): the reference's loop over
// the window's text (host walk), the same with the line extents + gathered line text the device's k_lines pass supplies,
// -O -l with and without the device's match ends.  Pure host: candidates by brute force (gscan_match_at at every offset).
//   g++ -O2 -std=c++17 report_probe.cc -I../../grab_amd/csrc -I../../include -L../../grab_amd/lib -lgrabhost -lgscan -Wl,-rpath,$PWD/../../grab_amd/lib -o /tmp/report_probe
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "filegrep.h"

// (optional) the gathered text in memory from the HIP runtime's pinned allocator, as the engine hands it over
extern "C" int hipHostMalloc(void **, size_t, unsigned);
extern "C" int hipInit(unsigned);

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 64) << 20;
    const char *pat = argc > 2 ? argv[2] : "[A-Za-z_][A-Za-z0-9_]{15,}";
    static const char alphabet[] = "abcdefghijklmnopqrstuvwxyz     _0123456789ABCDEF(){};=.,\n";
    std::vector<char> text(n);
    uint64_t x = 0x67726162u;
    for (size_t i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        text[i] = alphabet[(x >> 11) % 57];
    }
    gscan_db *db = nullptr;
    int minlen = 0;
    char err[128];
    if (gscan_compile(pat, strlen(pat), 0, &db, &minlen, err, sizeof err) != GSCAN_OK) return 1;
    std::vector<uint32_t> starts, ends;
    bool prev = false;
    for (size_t p = 0; p + (size_t)minlen <= n; p++) {
        const bool c = gscan_match_at(db, text.data(), n, (uint32_t)p) != 0;
        if (c && !prev) starts.push_back((uint32_t)p), ends.push_back(gscan_match_end(db, text.data(), n, (uint32_t)p));
        prev = c;
    }
    printf("%zu MiB, %zu group starts\n", n >> 20, starts.size());
    const char *path = "some/dir/f000123.txt";
    auto run = [&](const char *label, unsigned flags, const uint32_t *ext, const uint32_t *en, const uint8_t *gather) {
        double best = 1e9;
        std::string out;
        for (int rep = 0; rep < 5; rep++) {
            out.clear();
            const double t0 = now();
            grab_report_chunk(db, minlen, flags | GRAB_PREFIX, path, text.data(), n, 0, starts.data(), starts.size(), out, ext, en, gather);
            best = std::min(best, now() - t0);
        }
        size_t lines = 0;
        for (char ch : out) lines += ch == '\n';
        printf("%-44s %7.2f ms  %8zu lines  %6.1f ns/line  %5.1f MB out\n", label, best * 1e3, lines, best / lines * 1e9, out.size() / 1e6);
        return out;
    };
    run("-O -l  host walk", GRAB_OFFSETS | GRAB_NOLINE, nullptr, nullptr, nullptr);
    run("-O -l  device match ends (text-free)", GRAB_OFFSETS | GRAB_NOLINE, nullptr, ends.data(), nullptr);
    const std::string want = run("-O     host walk", GRAB_OFFSETS, nullptr, nullptr, nullptr);
    run("(lines) host walk", 0, nullptr, nullptr, nullptr);
    // what k_lines would hand over: per group start {m1, lb, le, goff} + the gathered text, from the loop itself
    std::vector<uint32_t> ext(starts.size() * 4, 0);
    std::string gathered;
    {
        gscan_cursor cur;
        cur.ready = 0;
        size_t s = 0, li = 0;
        while (s + (size_t)minlen < n) {
            uint32_t b0, b1;
            if (gscan_next_match(db, text.data(), n, starts.data(), starts.size(), &cur, (uint32_t)s, &b0, &b1) != 1) break;
            size_t lo = b0 - std::min<size_t>(b0 - s, 511);
            const void *nl = memrchr(text.data() + lo, '\n', b0 - lo);
            const size_t lb = nl ? (size_t)((const char *)nl - text.data()) + 1 : lo;
            const size_t hi = std::min(n, (size_t)b1 + 511);
            const void *nr = memchr(text.data() + b1, '\n', hi - b1);
            const size_t le = nr ? (size_t)((const char *)nr - text.data()) : hi;
            while (li < starts.size() && starts[li] < b0) li++;
            if (li < starts.size() && starts[li] == b0) {
                ext[4 * li] = b1, ext[4 * li + 1] = (uint32_t)lb, ext[4 * li + 2] = (uint32_t)le, ext[4 * li + 3] = (uint32_t)gathered.size();
                gathered.append(text.data() + lb, le - lb);
            }
            s = le;
        }
    }
    const std::string got = run("-O     device line pass + gathered text", GRAB_OFFSETS, ext.data(), nullptr, (const uint8_t *)gathered.data());
    run("(lines) device line pass + gathered text", 0, ext.data(), nullptr, (const uint8_t *)gathered.data());
    if (hipInit(0) == 0) { // the same from pinned (hipHostMalloc) memory: what the device's DMA writes into
        void *pin = nullptr;
        if (hipHostMalloc(&pin, gathered.size() + 64, 0) == 0 && pin) {
            double t0 = now();
            memcpy(pin, gathered.data(), gathered.size());
            const double tw = now() - t0;
            std::string back(gathered.size(), 0);
            t0 = now();
            memcpy(&back[0], pin, gathered.size());
            printf("pinned buffer: %.1f MB written at %.2f GB/s, read back at %.2f GB/s\n", gathered.size() / 1e6, gathered.size() / tw / 1e9, gathered.size() / (now() - t0) / 1e9);
            run("-O     device line pass + gathered text (pinned)", GRAB_OFFSETS, ext.data(), nullptr, (const uint8_t *)pin);
        }
    }
    for (size_t i = 0; i < starts.size(); i++) ext[4 * i + 3] = 0xffffffffu;
    run("-O     device line pass, text from the window", GRAB_OFFSETS, ext.data(), nullptr, nullptr);
    printf("outputs %s\n", got == want ? "identical" : "DIFFER");
    return 0;
}
