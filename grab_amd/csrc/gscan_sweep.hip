// gscan_sweep.hip -- native A/B harness for the scan kernels (no Python, no torch).
//
//   gscan_sweep [--gib G] [--seg-mib M] [--pattern P]... [--iters N] [--variants 0,1,2,4,5,6,38] [--bpc 0,4,8,16] [--k3-depth 0,3,4]
//   (--pattern may be given several times: the patterns run one after the other on the same arena; variant -1 = the engine's default)
//
// Fills a G GiB arena in HBM with synthetic text (57-symbol alphabet, SURVEY.md 8d
// distribution, xorshift stream), splits it into M MiB segments, and for every
// (variant, blocks_per_cu) pair runs N timed launches of gscan_scan_device, interleaved
// round-robin over the pairs so clock/thermal drift hits all of them alike.  Prints one
// line per pair: mean/min kernel time (HIP events on the launch stream) and GB/s.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gscan.h"

extern "C" int gscan_kernel_time(gscan_ctx *, double *, uint64_t *, int);

// ---- read-ceiling probe: the scan kernels' exact load pattern (256-thread workgroup, each wave
// ITER x 1 KiB via buffer_load_dwordx4 issued up front) with a trivial reduction instead of the
// scan.  What this reaches is the practical ceiling for "read every byte once" on this box.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// NT: false = default cache policy, true = nontemporal; AUX >= 0 overrides with an explicit cache-policy operand
// (gfx940+: bit 0 sc0, bit 1 nt, bit 4 sc1)
template <int ITER, bool NT, int AUX = -1>
__global__ __launch_bounds__(256) void k0_read_probe(const uint8_t *base, uint32_t n_tiles, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr uint32_t kTile = 4 * ITER * 1024;
    uint32_t acc = 0;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t addr = (uint64_t)base + (uint64_t)t * kTile;
        const uint32_t alo = __builtin_amdgcn_readfirstlane((uint32_t)addr), ahi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)ahi << 32) | alo), 0, (int)kTile, 0x00020000);
        u32x4 buf[ITER];
        const int v0 = (int)(wave * ITER * 1024 + lane * 16);
#pragma unroll
        for (int k = 0; k < ITER; k++) buf[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, v0 + k * 1024, 0, AUX >= 0 ? AUX : (NT ? 2 : 0)));
#pragma unroll
        for (int k = 0; k < ITER; k++) acc ^= buf[k].x ^ buf[k].y ^ buf[k].z ^ buf[k].w;
    }
    if (acc == 0x9e3779b9u) sink[0] = acc; // never true on text; keeps the loads alive
}

template <int ITER, bool NT, int AUX = -1>
static float probe(const uint8_t *arena, size_t total, uint32_t *sink, int bpc, int iters)
{
    const uint32_t n_tiles = (uint32_t)(total / (4 * ITER * 1024));
    const uint32_t grid = bpc > 0 ? std::min<uint32_t>(n_tiles, 256u * bpc) : n_tiles;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int i = 0; i < iters + 1; i++) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL((k0_read_probe<ITER, NT, AUX>), dim3(grid), dim3(256), 0, 0, arena, n_tiles, sink);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (i > 0 && ms < best) best = ms;
    }
    return best;
}

static std::vector<long> parse_list(const char *s)
{
    std::vector<long> v;
    for (const char *p = s; *p;) {
        v.push_back(strtol(p, (char **)&p, 10));
        if (*p == ',') p++;
    }
    return v;
}

int main(int argc, char **argv)
{
    double gib = 8;
    int seg_mib = 64, iters = 10;
    std::vector<std::string> patterns;
    std::vector<long> variants = {0, 1, 2, 4, 5, 6}, bpcs = {0, 8}, depths = {0}; // depth: K3's filter positions (0 = the compiler's choice)
    int plant_every_mib = 1;
    bool ceiling = false;
    for (int i = 1; i < argc; i++) {
        auto is = [&](const char *f) { return !strcmp(argv[i], f) && i + 1 < argc; };
        if (is("--gib")) gib = atof(argv[++i]);
        else if (is("--seg-mib")) seg_mib = atoi(argv[++i]);
        else if (is("--pattern")) patterns.push_back(argv[++i]);
        else if (is("--iters")) iters = atoi(argv[++i]);
        else if (is("--variants")) variants = parse_list(argv[++i]);
        else if (is("--bpc")) bpcs = parse_list(argv[++i]);
        else if (is("--k3-depth")) depths = parse_list(argv[++i]);
        else if (is("--plant-mib")) plant_every_mib = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--ceiling")) ceiling = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (patterns.empty()) patterns.push_back("foobardoesnotexist");
    const size_t seg_bytes = (size_t)seg_mib << 20;
    const size_t nseg = (size_t)(gib * 1024 / seg_mib);
    const size_t total = nseg * seg_bytes;

    static const char alphabet[] = "abcdefghijklmnopqrstuvwxyz     _0123456789ABCDEF(){};=.,\n";
    std::vector<uint8_t> block(seg_bytes);
    uint64_t x = 0x67726162u;
    for (size_t i = 0; i < seg_bytes; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        block[i] = (uint8_t)alphabet[(x >> 11) % 57];
    }
    if (plant_every_mib > 0)
        for (size_t at = 300000; at + 64 < seg_bytes; at += (size_t)plant_every_mib << 20) memcpy(&block[at], "foobardoesnotexist", 18);

    uint8_t *arena = nullptr;
    if (hipMalloc((void **)&arena, total + 4096) != hipSuccess) { fprintf(stderr, "hipMalloc %zu failed\n", total); return 1; }
    (void)hipMemcpy(arena, block.data(), seg_bytes, hipMemcpyHostToDevice);
    for (size_t s = 1; s < nseg; s++) (void)hipMemcpy(arena + s * seg_bytes, arena, seg_bytes, hipMemcpyDeviceToDevice);
    (void)hipDeviceSynchronize();

    if (ceiling) {
        uint32_t *sink = nullptr;
        (void)hipMalloc((void **)&sink, 64);
        printf("# read-ceiling probe, %.2f GiB, best of %d\n", total / 1073741824.0, iters);
        for (int bpc : {0, 8, 16}) {
            printf("probe ITER16      bpc %2d : %8.1f GB/s\n", bpc, total / probe<16, false>(arena, total, sink, bpc, iters) / 1e6);
            printf("probe ITER16 nt   bpc %2d : %8.1f GB/s\n", bpc, total / probe<16, true>(arena, total, sink, bpc, iters) / 1e6);
            printf("probe ITER8       bpc %2d : %8.1f GB/s\n", bpc, total / probe<8, false>(arena, total, sink, bpc, iters) / 1e6);
            printf("probe ITER8  nt   bpc %2d : %8.1f GB/s\n", bpc, total / probe<8, true>(arena, total, sink, bpc, iters) / 1e6);
            printf("probe ITER32 nt   bpc %2d : %8.1f GB/s\n", bpc, total / probe<32, true>(arena, total, sink, bpc, iters) / 1e6);
        }
        // cache-policy operand of the loads (ITER 12, one workgroup per tile -- the scan kernels' shape)
        printf("probe ITER12 aux 0 (default)    : %8.1f GB/s\n", total / probe<12, false, 0>(arena, total, sink, 0, iters) / 1e6);
        printf("probe ITER12 aux 1 (sc0)        : %8.1f GB/s\n", total / probe<12, false, 1>(arena, total, sink, 0, iters) / 1e6);
        printf("probe ITER12 aux 2 (nt)         : %8.1f GB/s\n", total / probe<12, false, 2>(arena, total, sink, 0, iters) / 1e6);
        printf("probe ITER12 aux 3 (sc0 nt)     : %8.1f GB/s\n", total / probe<12, false, 3>(arena, total, sink, 0, iters) / 1e6);
        printf("probe ITER12 aux 16 (sc1)       : %8.1f GB/s\n", total / probe<12, false, 16>(arena, total, sink, 0, iters) / 1e6);
        printf("probe ITER12 aux 17 (sc0 sc1)   : %8.1f GB/s\n", total / probe<12, false, 17>(arena, total, sink, 0, iters) / 1e6);
        printf("probe ITER12 aux 18 (nt sc1)    : %8.1f GB/s\n", total / probe<12, false, 18>(arena, total, sink, 0, iters) / 1e6);
        printf("probe ITER12 aux 19 (sc0 nt sc1): %8.1f GB/s\n", total / probe<12, false, 19>(arena, total, sink, 0, iters) / 1e6);
        (void)hipFree(sink);
    }

    gscan_ctx *ctx = nullptr;
    if (gscan_open(0, 1u << 30, &ctx) != GSCAN_OK) { fprintf(stderr, "gscan_open failed\n"); return 1; }
    std::vector<gscan_seg> segs(nseg);
    for (size_t s = 0; s < nseg; s++) segs[s] = {s * seg_bytes, (uint32_t)seg_bytes, 0};
    gscan_set_capacity(ctx, 1u << 28);
    for (const std::string &pattern : patterns) {
        gscan_db *db = nullptr;
        char err[128];
        int minlen = 0;
        if (gscan_compile(pattern.data(), pattern.size(), 0, &db, &minlen, err, sizeof err) != GSCAN_OK) { fprintf(stderr, "compile: %s\n", err); return 1; }
        gscan_info info;
        gscan_db_info(db, &info);

        struct Cell { long variant, bpc, depth; double sum = 0, best = 1e30; int n = 0; uint64_t matches = 0; };
        std::vector<Cell> cells;
        for (long v : variants) for (long b : bpcs) for (long d : depths) { Cell c; c.variant = v; c.bpc = b; c.depth = d; cells.push_back(c); }
        printf("# arena %.2f GiB in %zu x %d MiB segments, pattern '%s' (tier %d, minlen %d), %d iters\n", total / 1073741824.0, nseg, seg_mib, pattern.c_str(), info.tier, minlen, iters);
        for (int it = -1; it < iters; it++) { // it == -1: warm-up round
            for (Cell &c : cells) {
                if (c.variant >= 0) gscan_set_option(ctx, "variant", c.variant);
                gscan_set_option(ctx, "blocks_per_cu", c.bpc);
                gscan_set_option(ctx, "k3_depth", c.depth);
                gscan_dev_result res;
                int rc = gscan_scan_device(ctx, db, arena, segs.data(), nseg, nullptr, &res);
                if (rc != GSCAN_OK) { fprintf(stderr, "scan failed: %s\n", gscan_strerror(ctx)); return 1; }
                gscan_dev_sync(ctx, &res);
                double ms = 0; uint64_t n = 0;
                gscan_kernel_time(ctx, &ms, &n, 1);
                if (it >= 0) { c.sum += ms; c.best = ms < c.best ? ms : c.best; c.n++; }
                c.matches = res.total;
                if (res.overflow) fprintf(stderr, "overflow (total %llu)\n", (unsigned long long)res.total);
            }
        }
        for (const Cell &c : cells) {
            const double bytes = (double)total + 4.0 * c.matches;
            if (depths.size() > 1 || depths[0] != 0) printf("k3_depth %ld ", c.depth);
            printf("variant %ld bpc %2ld : mean %8.3f ms  min %8.3f ms  -> %8.1f GB/s mean, %8.1f GB/s best   matches %llu\n", c.variant, c.bpc,
                   c.sum / c.n, c.best, bytes / (c.sum / c.n) / 1e6, bytes / c.best / 1e6, (unsigned long long)c.matches);
        }
        gscan_free(db);
    }
    gscan_close(ctx);
    (void)hipFree(arena);
    return 0;
}
