// engine.hip -- gscan context: streams, pinned/HBM buffers, double-buffered chunk
// pipeline, device-resident batch scans, and the C ABI of include/gscan.h.
//
// Data flow of the host-chunk path (what FileGrep::find drives, replacing the
// mmap -> pcre_exec loop of /root/reference/src/grab.cc:154-215):
//
//   read(2) into slot.pinned ──copy stream: hipMemcpyAsync──► slot.d_text (HBM)
//        event `copied` ──compute stream waits──► scan kernel ──► d_recs/d_desc/d_counter
//        ──compute stream: async D2H of counter + descriptors + first records──► event `done`
//   gscan_wait: sync `done`, fetch the rest if needed, stitch runs in tile order.
//
// With GSCAN_SLOTS = 2 the copy of chunk k+1 overlaps the scan and report of chunk k.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "scan_args.h"


using gscan::Database;
using gscan::DevProgram;
using gscan::ScanArgs;

struct gscan_db {
    Database db;
};

namespace {

constexpr size_t kPad = 4096;          // slack behind every text buffer
constexpr size_t kSpecPer = 2048;       // records of EACH shard region fetched speculatively with the header
constexpr size_t kSpecRecs = kSpecPer * gscan::kShards;
constexpr size_t kCounterWords = gscan::kShards + 1; // per-shard counts + overflow flag
constexpr size_t kCopyPiece = 32u << 20; // memcpy/H2D pipelining granule for foreign host buffers
constexpr size_t kMaxChunk = (1ull << 30) + 4096;

enum SlotState { FREE = 0, ACQUIRED, INFLIGHT };

struct Slot {
    SlotState state = FREE;
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    uint8_t *d_text = nullptr;
    size_t d_text_cap = 0;
    uint32_t *d_recs = nullptr;
    size_t rec_cap = 0;
    unsigned long long *d_desc = nullptr;
    size_t tiles_cap = 0;
    uint32_t *d_counter = nullptr;        // kCounterWords
    uint32_t *h_counter = nullptr;        // pinned, kCounterWords
    unsigned long long *h_desc = nullptr; // pinned
    size_t h_desc_cap = 0;
    uint32_t *h_spec = nullptr; // pinned, kShards rows of kSpecPer
    std::vector<uint32_t> raw, sorted;
    hipEvent_t copied = nullptr, done = nullptr;
    uint64_t tag = 0;
    size_t len = 0;
    uint32_t n_tiles = 0;
    const gscan_db *db = nullptr;
    uint64_t seq = 0;
};

struct EvPair {
    hipEvent_t a, b;
};

} // namespace

struct gscan_ctx {
    int device = 0;
    int cus = 256;
    size_t max_chunk = 0;
    hipStream_t copy = nullptr, compute = nullptr;
    Slot slot[GSCAN_SLOTS];
    uint64_t next_seq = 1;
    std::string err;
    // compiled program on the device
    DevProgram *d_prog = nullptr;
    DevProgram *h_prog = nullptr; // pinned staging
    uint64_t prog_id = 0;
    // options
    // defaults from the sweeps under profiles/: 12 KiB per wave (110 VGPRs -> 4 waves/SIMD) with
    // nontemporal loads, one workgroup per tile
    int variant = 6;
    int blocks_per_cu = 0;
    // device-resident path
    size_t dev_cap_req = 0;
    uint32_t *dv_recs = nullptr;
    size_t dv_rec_cap = 0;
    unsigned long long *dv_desc = nullptr;
    gscan::TileDesc *dv_tiles = nullptr;
    size_t dv_tiles_cap = 0;
    uint32_t *dv_counter = nullptr;
    std::vector<gscan_seg> dv_last_segs;
    std::vector<uint32_t> dv_tile_first_h;
    uint32_t dv_last_tile_bytes = 0;
    hipStream_t dv_stream = nullptr;
    // kernel timing
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
};

namespace {

int fail(gscan_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return fail((c), GSCAN_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

int ensure_prog(gscan_ctx *c, const gscan_db *db, hipStream_t st)
{
    if (c->prog_id == db->db.id) return 0;
    // The staging copy is overwritten: every earlier upload must have left it.  Uploads are
    // rare (one per pattern), so a stream sync here costs nothing measurable.
    HIPCHK(c, hipStreamSynchronize(c->compute));
    if (c->dv_stream && c->dv_stream != c->compute) HIPCHK(c, hipStreamSynchronize(c->dv_stream));
    memcpy(c->h_prog, &db->db.prog, sizeof(DevProgram));
    HIPCHK(c, hipMemcpyAsync(c->d_prog, c->h_prog, sizeof(DevProgram), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    c->prog_id = db->db.id;
    return 0;
}

uint32_t grid_for(const gscan_ctx *c, const Database &db, uint32_t n_tiles)
{
    // kernels that stage a big LDS table once per workgroup always run as a persistent grid
    const uint32_t fixed = gscan::scan_persistent_blocks(db.tier, db.prog.n_classes);
    const uint32_t bpc = fixed ? fixed : (uint32_t)std::max(c->blocks_per_cu, 0);
    if (bpc == 0) return n_tiles;
    uint64_t g = (uint64_t)c->cus * (uint64_t)bpc;
    return (uint32_t)std::min<uint64_t>(g, n_tiles);
}

int slot_reserve(gscan_ctx *c, Slot &s, size_t len)
{
    if (len > s.pinned_cap) {
        if (s.pinned) hipHostFree(s.pinned);
        s.pinned = nullptr;
        s.pinned_cap = 0;
        size_t cap = std::max<size_t>((len + kPad + 0xfffff) & ~(size_t)0xfffff, 1u << 20);
        HIPCHK(c, hipHostMalloc(&s.pinned, cap, hipHostMallocDefault));
        s.pinned_cap = cap - kPad;
    }
    if (len > s.d_text_cap) {
        if (s.d_text) hipFree(s.d_text);
        s.d_text = nullptr;
        s.d_text_cap = 0;
        size_t cap = std::max<size_t>((len + kPad + 0xfffff) & ~(size_t)0xfffff, 1u << 20);
        HIPCHK(c, hipMalloc((void **)&s.d_text, cap));
        s.d_text_cap = cap - kPad;
    }
    // tiles at the smallest tile size any variant uses
    size_t tiles = len / gscan::scan_min_tile_bytes() + 2;
    if (tiles > s.tiles_cap) {
        if (s.d_desc) hipFree(s.d_desc);
        if (s.h_desc) hipHostFree(s.h_desc);
        s.d_desc = nullptr;
        s.h_desc = nullptr;
        s.tiles_cap = 0;
        size_t cap = tiles + tiles / 4;
        HIPCHK(c, hipMalloc((void **)&s.d_desc, cap * 8));
        HIPCHK(c, hipHostMalloc((void **)&s.h_desc, cap * 8, hipHostMallocDefault));
        s.tiles_cap = cap;
    }
    size_t want = std::max<size_t>((len / 64 + gscan::kShards - 1) / gscan::kShards * gscan::kShards, kSpecRecs);
    if (want > s.rec_cap) {
        if (s.d_recs) hipFree(s.d_recs);
        s.d_recs = nullptr;
        s.rec_cap = 0;
        HIPCHK(c, hipMalloc((void **)&s.d_recs, want * 4));
        s.rec_cap = want;
    }
    return 0;
}

int slot_launch(gscan_ctx *c, Slot &s)
{
    const Database &db = s.db->db;
    const uint32_t tile_bytes = gscan::scan_tile_bytes(db.tier, c->variant, db.prog.n_classes);
    s.n_tiles = (uint32_t)((s.len + tile_bytes - 1) / tile_bytes);
    HIPCHK(c, hipMemsetAsync(s.d_counter, 0, kCounterWords * 4, c->compute));
    ScanArgs a;
    memset(&a, 0, sizeof a);
    a.base = s.d_text;
    a.tiles = nullptr; // one segment, tiled in order
    a.seg0_off = 0;
    a.seg0_len = (uint32_t)s.len;
    a.n_tiles = s.n_tiles;
    a.cap_shard = (uint32_t)(s.rec_cap / gscan::kShards);
    a.recs = s.d_recs;
    a.desc = s.d_desc;
    a.counter = s.d_counter;
    a.prog = c->d_prog;
    gscan::fill_program(a, db.prog);
    if (s.n_tiles) HIPCHK(c, gscan::launch_scan(db.tier, c->variant, a, grid_for(c, db, s.n_tiles), c->compute));
    HIPCHK(c, hipMemcpyAsync(s.h_counter, s.d_counter, kCounterWords * 4, hipMemcpyDeviceToHost, c->compute));
    if (s.n_tiles)
        HIPCHK(c, hipMemcpyAsync(s.h_desc, s.d_desc, (size_t)s.n_tiles * 8, hipMemcpyDeviceToHost, c->compute));
    // the head of every shard region in one strided copy: enough for any sparse result
    HIPCHK(c, hipMemcpy2DAsync(s.h_spec, kSpecPer * 4, s.d_recs, (size_t)a.cap_shard * 4, kSpecPer * 4, gscan::kShards,
                               hipMemcpyDeviceToHost, c->compute));
    HIPCHK(c, hipEventRecord(s.done, c->compute));
    return 0;
}

void free_slot(Slot &s)
{
    if (s.pinned) hipHostFree(s.pinned);
    if (s.d_text) hipFree(s.d_text);
    if (s.d_recs) hipFree(s.d_recs);
    if (s.d_desc) hipFree(s.d_desc);
    if (s.d_counter) hipFree(s.d_counter);
    if (s.h_counter) hipHostFree(s.h_counter);
    if (s.h_desc) hipHostFree(s.h_desc);
    if (s.h_spec) hipHostFree(s.h_spec);
    if (s.copied) hipEventDestroy(s.copied);
    if (s.done) hipEventDestroy(s.done);
    s = Slot();
}

} // namespace

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

int gscan_compile(const char *pat, size_t len, unsigned flags, gscan_db **out, int *minlen, char *err,
                  size_t errcap)
{
    if (!pat || !out) return GSCAN_EINVAL;
    gscan_db *d = new (std::nothrow) gscan_db();
    if (!d) return GSCAN_ENOMEM;
    std::string why;
    int rc = gscan::compile_pattern(pat, len, flags, d->db, why);
    if (rc != 0) {
        if (err && errcap) snprintf(err, errcap, "%s", why.c_str());
        delete d;
        *out = nullptr;
        return rc < 0 ? GSCAN_EINVAL : GSCAN_UNSUPPORTED;
    }
    if (minlen) *minlen = d->db.minlen;
    *out = d;
    return GSCAN_OK;
}

void gscan_free(gscan_db *db) { delete db; }

int gscan_db_info(const gscan_db *db, gscan_info *info)
{
    if (!db || !info) return GSCAN_EINVAL;
    const Database &d = db->db;
    info->tier = d.tier;
    info->minlen = d.minlen;
    info->n_classes = (int)d.classes.size();
    info->has_tail = !d.alts.empty() && d.alts[0].has_tail;
    info->tail_extra = d.alts.empty() ? 0 : d.alts[0].tail_extra;
    info->anchor_off = (int)d.prog.anchor_off;
    info->anchor_len = (int)d.prog.anchor_len;
    info->is_literal = (int)d.prog.is_literal;
    info->n_alts = (int)d.alts.size();
    return GSCAN_OK;
}

int gscan_db_alt_class(const gscan_db *db, int alt, int pos, uint8_t table[256], int *len)
{
    if (!db || !table) return GSCAN_EINVAL;
    const Database &d = db->db;
    if (alt < 0 || (size_t)alt >= d.alts.size()) return GSCAN_EINVAL;
    const gscan::AltSeq &a = d.alts[(size_t)alt];
    const gscan::ByteSet *s = nullptr;
    if (pos == -1) {
        if (!a.has_tail) return GSCAN_EINVAL;
        s = &a.tail;
    } else {
        if (pos < 0 || (size_t)pos >= a.window.size()) return GSCAN_EINVAL;
        s = &d.classes[a.window[(size_t)pos]];
    }
    for (int b = 0; b < 256; b++) table[b] = s->test((unsigned)b);
    if (len) *len = (int)a.window.size();
    return GSCAN_OK;
}

int gscan_db_class(const gscan_db *db, int pos, uint8_t table[256]) { return gscan_db_alt_class(db, 0, pos, table, nullptr); }

namespace {
// The alternative pcre_exec's match at p goes through: the first one, in priority order, whose
// window fits into the chunk and matches there (pattern.h).  nullptr: no match starts at p.
const gscan::AltSeq *alt_at(const Database &d, const uint8_t *content, size_t clen, size_t p)
{
    for (const gscan::AltSeq &a : d.alts) {
        const size_t m = a.window.size();
        if (p + m > clen) continue;
        const uint8_t *t = content + p;
        size_t i = 0;
        while (i < m && d.classes[a.window[i]].test(t[i])) i++;
        if (i == m) return &a;
    }
    return nullptr;
}
} // namespace

int gscan_match_at(const gscan_db *db, const void *content, size_t clen, uint32_t p)
{
    const Database &d = db->db;
    if (d.minlen <= 0) return 0;
    return alt_at(d, (const uint8_t *)content, clen, p) != nullptr;
}

uint32_t gscan_match_end(const gscan_db *db, const void *content, size_t clen, uint32_t start)
{
    const Database &d = db->db;
    const uint8_t *t = (const uint8_t *)content;
    const gscan::AltSeq *a = d.minlen > 0 ? alt_at(d, t, clen, start) : nullptr;
    if (!a) return start; // not a match start
    size_t e = (size_t)start + a->window.size();
    if (a->has_tail) {
        uint64_t extra = 0;
        while (e < clen && extra < (uint64_t)a->tail_extra && a->tail.test(t[e])) {
            e++;
            extra++;
        }
    }
    return (uint32_t)e;
}

int gscan_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gscan_open(int hip_device, size_t max_chunk, gscan_ctx **out)
{
    if (!out) return GSCAN_EINVAL;
    *out = nullptr;
    if (max_chunk == 0 || max_chunk > kMaxChunk) return GSCAN_ETOOBIG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GSCAN_EHIP; // no device: there is no CPU path
    if (hip_device < 0 || hip_device >= n) return GSCAN_EINVAL;
    gscan_ctx *c = new (std::nothrow) gscan_ctx();
    if (!c) return GSCAN_ENOMEM;
    c->device = hip_device;
    c->max_chunk = max_chunk;
    auto bail = [&](int rc) {
        gscan_close(c);
        return rc;
    };
    if (hipSetDevice(hip_device) != hipSuccess) return bail(GSCAN_EHIP);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, hip_device) == hipSuccess) c->cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking) != hipSuccess) return bail(GSCAN_EHIP);
    if (hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking) != hipSuccess) return bail(GSCAN_EHIP);
    if (hipMalloc((void **)&c->d_prog, sizeof(DevProgram)) != hipSuccess) return bail(GSCAN_EHIP);
    if (hipHostMalloc((void **)&c->h_prog, sizeof(DevProgram), hipHostMallocDefault) != hipSuccess) return bail(GSCAN_EHIP);
    for (Slot &s : c->slot) {
        if (hipMalloc((void **)&s.d_counter, 64) != hipSuccess) return bail(GSCAN_EHIP);
        if (hipHostMalloc((void **)&s.h_counter, 64, hipHostMallocDefault) != hipSuccess) return bail(GSCAN_EHIP);
        if (hipHostMalloc((void **)&s.h_spec, kSpecRecs * 4, hipHostMallocDefault) != hipSuccess) return bail(GSCAN_EHIP);
        if (hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess) return bail(GSCAN_EHIP);
        if (hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return bail(GSCAN_EHIP);
    }
    if (hipMalloc((void **)&c->dv_counter, 64) != hipSuccess) return bail(GSCAN_EHIP);
    *out = c;
    return GSCAN_OK;
}

void gscan_close(gscan_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (Slot &s : c->slot) free_slot(s);
    if (c->d_prog) hipFree(c->d_prog);
    if (c->h_prog) hipHostFree(c->h_prog);
    if (c->dv_recs) hipFree(c->dv_recs);
    if (c->dv_desc) hipFree(c->dv_desc);
    if (c->dv_tiles) hipFree(c->dv_tiles);
    if (c->dv_counter) hipFree(c->dv_counter);
    for (auto &e : c->ev_pool) {
        hipEventDestroy(e.a);
        hipEventDestroy(e.b);
    }
    if (c->copy) hipStreamDestroy(c->copy);
    if (c->compute) hipStreamDestroy(c->compute);
    delete c;
}

const char *gscan_strerror(const gscan_ctx *c) { return c ? c->err.c_str() : "no context"; }

int gscan_acquire(gscan_ctx *c, size_t len, void **pinned)
{
    if (!c || !pinned) return GSCAN_EINVAL;
    if (len > c->max_chunk) return fail(c, GSCAN_ETOOBIG, "chunk of %zu bytes exceeds max_chunk %zu", len, c->max_chunk);
    HIPCHK(c, hipSetDevice(c->device));
    for (Slot &s : c->slot)
        if (s.state == ACQUIRED) { // re-acquire: same slot
            int rc = slot_reserve(c, s, len);
            if (rc) return rc;
            *pinned = s.pinned;
            return GSCAN_OK;
        }
    for (Slot &s : c->slot)
        if (s.state == FREE) {
            int rc = slot_reserve(c, s, len);
            if (rc) return rc;
            s.state = ACQUIRED;
            *pinned = s.pinned;
            return GSCAN_OK;
        }
    return fail(c, GSCAN_EBUSY, "all %d slots in flight", GSCAN_SLOTS);
}

int gscan_submit(gscan_ctx *c, const gscan_db *db, const void *host_bytes, size_t len, uint64_t tag)
{
    if (!c || !db || (!host_bytes && len)) return GSCAN_EINVAL;
    if (db->db.tier == GSCAN_TIER_NULL) return fail(c, GSCAN_EINVAL, "a pattern that can match the empty string scans nothing");
    if (len > c->max_chunk) return fail(c, GSCAN_ETOOBIG, "chunk of %zu bytes exceeds max_chunk %zu", len, c->max_chunk);
    HIPCHK(c, hipSetDevice(c->device));
    Slot *s = nullptr;
    for (Slot &x : c->slot)
        if (x.state == ACQUIRED) s = &x;
    if (!s) {
        void *p;
        int rc = gscan_acquire(c, len, &p);
        if (rc) return rc;
        for (Slot &x : c->slot)
            if (x.state == ACQUIRED) s = &x;
    } else if (len > s->pinned_cap) {
        return fail(c, GSCAN_EINVAL, "submitted %zu bytes into a slot acquired for %zu", len, s->pinned_cap);
    }
    int rc = ensure_prog(c, db, c->compute);
    if (rc) return rc;
    if (host_bytes == s->pinned) {
        if (len) HIPCHK(c, hipMemcpyAsync(s->d_text, s->pinned, len, hipMemcpyHostToDevice, c->copy));
    } else {
        rc = slot_reserve(c, *s, len);
        if (rc) return rc;
        for (size_t o = 0; o < len; o += kCopyPiece) { // memcpy of piece i+1 overlaps the DMA of piece i
            size_t n = std::min(kCopyPiece, len - o);
            memcpy((char *)s->pinned + o, (const char *)host_bytes + o, n);
            HIPCHK(c, hipMemcpyAsync(s->d_text + o, (char *)s->pinned + o, n, hipMemcpyHostToDevice, c->copy));
        }
    }
    HIPCHK(c, hipEventRecord(s->copied, c->copy));
    HIPCHK(c, hipStreamWaitEvent(c->compute, s->copied, 0));
    s->db = db;
    s->len = len;
    s->tag = tag;
    s->seq = c->next_seq++;
    rc = slot_launch(c, *s);
    if (rc) return rc;
    s->state = INFLIGHT;
    return GSCAN_OK;
}

int gscan_wait(gscan_ctx *c, uint64_t *tag, const uint32_t **starts, size_t *n, const void **content)
{
    if (!c || !starts || !n) return GSCAN_EINVAL;
    Slot *s = nullptr;
    for (Slot &x : c->slot)
        if (x.state == INFLIGHT && (!s || x.seq < s->seq)) s = &x;
    if (!s) return fail(c, GSCAN_EEMPTY, "nothing in flight");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(s->done));
    const size_t K = gscan::kShards;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (s->h_counter[K] == 0) break; // no shard overflowed
        if (attempt == 1) return fail(c, GSCAN_EHIP, "record buffer overflow persisted after regrow");
        // the text is still in HBM: size every shard for the fullest one (+25%) and rescan
        uint32_t worst = 0;
        for (size_t k = 0; k < K; k++) worst = std::max(worst, s->h_counter[k]);
        size_t want = ((size_t)worst + (size_t)worst / 4 + kSpecPer) * K;
        HIPCHK(c, hipStreamSynchronize(c->compute));
        if (s->d_recs) hipFree(s->d_recs);
        s->d_recs = nullptr;
        s->rec_cap = 0;
        HIPCHK(c, hipMalloc((void **)&s->d_recs, want * 4));
        s->rec_cap = want;
        int rc = slot_launch(c, *s);
        if (rc) return rc;
        HIPCHK(c, hipEventSynchronize(s->done));
    }
    const size_t cap_shard = s->rec_cap / K;
    size_t total = 0;
    bool spec_ok = true;
    for (size_t k = 0; k < K; k++) {
        total += s->h_counter[k];
        if (s->h_counter[k] > kSpecPer) spec_ok = false;
    }
    if (!spec_ok) { // dense result: fetch each shard region's used part
        s->raw.resize(s->rec_cap);
        for (size_t k = 0; k < K; k++)
            if (s->h_counter[k])
                HIPCHK(c, hipMemcpy(s->raw.data() + k * cap_shard, s->d_recs + k * cap_shard, (size_t)s->h_counter[k] * 4,
                                    hipMemcpyDeviceToHost));
    }
    s->sorted.clear();
    s->sorted.reserve(total);
    for (uint32_t t = 0; t < s->n_tiles; t++) { // tiles are in text order: concatenating their runs sorts the list
        unsigned long long d = s->h_desc[t];
        uint32_t cnt = (uint32_t)d;
        size_t base = (size_t)(d >> 32);
        if (!cnt) continue;
        const uint32_t *src = spec_ok ? s->h_spec + (base / cap_shard) * kSpecPer + base % cap_shard : s->raw.data() + base;
        s->sorted.insert(s->sorted.end(), src, src + cnt);
    }
    if (s->sorted.size() != total) return fail(c, GSCAN_EHIP, "descriptor total %zu != counter %zu", s->sorted.size(), total);
    if (tag) *tag = s->tag;
    *starts = s->sorted.data();
    *n = s->sorted.size();
    if (content) *content = s->pinned;
    s->state = FREE;
    return GSCAN_OK;
}

int gscan_set_capacity(gscan_ctx *c, size_t n_records)
{
    if (!c) return GSCAN_EINVAL;
    c->dev_cap_req = n_records;
    return GSCAN_OK;
}

int gscan_set_option(gscan_ctx *c, const char *name, long value)
{
    if (!c || !name) return GSCAN_EINVAL;
    if (!strcmp(name, "variant")) {
        if (value < 0 || value > 7 || (value & 3) == 3) return GSCAN_EINVAL; // KiB per wave {16,8,12} | nontemporal<<2
        c->variant = (int)value;
        return GSCAN_OK;
    }
    if (!strcmp(name, "blocks_per_cu")) {
        if (value < 0 || value > 64) return GSCAN_EINVAL;
        c->blocks_per_cu = (int)value;
        return GSCAN_OK;
    }
    return GSCAN_EINVAL;
}

int gscan_scan_device(gscan_ctx *c, const gscan_db *db, const void *dev_base, const gscan_seg *segs, size_t nseg,
                      void *stream, gscan_dev_result *res)
{
    if (!c || !db || !res || (!segs && nseg)) return GSCAN_EINVAL;
    if (db->db.tier == GSCAN_TIER_NULL) return fail(c, GSCAN_EINVAL, "a pattern that can match the empty string scans nothing");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->compute;
    c->dv_stream = st;
    int rc = ensure_prog(c, db, st);
    if (rc) return rc;
    const uint32_t tile_bytes = gscan::scan_tile_bytes(db->db.tier, c->variant, db->db.prog.n_classes);
    const size_t K = gscan::kShards;

    // tile table: rebuilt only when the segment table or the tile size changed
    bool same = c->dv_last_tile_bytes == tile_bytes && c->dv_last_segs.size() == nseg &&
                (nseg == 0 || !memcmp(c->dv_last_segs.data(), segs, nseg * sizeof(gscan_seg)));
    if (!same) {
        std::vector<uint32_t> &tf = c->dv_tile_first_h;
        tf.assign(nseg + 1, 0);
        uint64_t nt = 0;
        for (size_t i = 0; i < nseg; i++) {
            if (segs[i].len > kMaxChunk) return fail(c, GSCAN_ETOOBIG, "segment %zu longer than a chunk", i);
            if (segs[i].offset & 15) return fail(c, GSCAN_EINVAL, "segment %zu is not 16-byte aligned", i);
            tf[i] = (uint32_t)nt;
            nt += (segs[i].len + tile_bytes - 1) / tile_bytes;
        }
        if (nt >= 0xffffffffull) return fail(c, GSCAN_ETOOBIG, "too many tiles");
        tf[nseg] = (uint32_t)nt;
        std::vector<gscan::TileDesc> td((size_t)nt);
        for (size_t i = 0; i < nseg; i++)
            for (uint32_t t = tf[i]; t < tf[i + 1]; t++) td[t] = {segs[i].offset, segs[i].len, (t - tf[i]) * tile_bytes};
        HIPCHK(c, hipStreamSynchronize(st)); // earlier scans may still read the old table
        if (nt + 1 > c->dv_tiles_cap) {
            if (c->dv_desc) hipFree(c->dv_desc);
            if (c->dv_tiles) hipFree(c->dv_tiles);
            c->dv_desc = nullptr;
            c->dv_tiles = nullptr;
            c->dv_tiles_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->dv_desc, (size_t)(nt + 1) * 8));
            HIPCHK(c, hipMalloc((void **)&c->dv_tiles, (size_t)(nt + 1) * sizeof(gscan::TileDesc)));
            c->dv_tiles_cap = (size_t)nt + 1;
        }
        if (nt) HIPCHK(c, hipMemcpy(c->dv_tiles, td.data(), (size_t)nt * sizeof(gscan::TileDesc), hipMemcpyHostToDevice));
        c->dv_last_segs.assign(segs, segs + nseg);
        c->dv_last_tile_bytes = tile_bytes;
    }
    const uint32_t n_tiles = c->dv_tile_first_h.empty() ? 0 : c->dv_tile_first_h.back();

    // record buffer: the caller's capacity if set, else arena bytes / 16; kShards equal regions
    uint64_t total_bytes = 0;
    for (size_t i = 0; i < nseg; i++) total_bytes += segs[i].len;
    size_t want = c->dev_cap_req ? c->dev_cap_req : std::max<size_t>((size_t)(total_bytes / 16), 1u << 16);
    want = std::min<size_t>((want + K - 1) / K * K, 0xfffffff8u);
    if (want != c->dv_rec_cap && (c->dev_cap_req || want > c->dv_rec_cap)) {
        HIPCHK(c, hipStreamSynchronize(st));
        if (c->dv_recs) hipFree(c->dv_recs);
        c->dv_recs = nullptr;
        c->dv_rec_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->dv_recs, want * 4));
        c->dv_rec_cap = want;
    }

    HIPCHK(c, hipMemsetAsync(c->dv_counter, 0, kCounterWords * 4, st));
    ScanArgs a;
    memset(&a, 0, sizeof a);
    a.base = (const uint8_t *)dev_base;
    a.tiles = c->dv_tiles;
    a.n_tiles = n_tiles;
    a.cap_shard = (uint32_t)(c->dv_rec_cap / K);
    a.recs = c->dv_recs;
    a.desc = c->dv_desc;
    a.counter = c->dv_counter;
    a.prog = c->d_prog;
    gscan::fill_program(a, db->db.prog);
    if (c->ev_used == c->ev_pool.size() && c->ev_pool.size() < 4096) {
        EvPair e;
        HIPCHK(c, hipEventCreate(&e.a));
        HIPCHK(c, hipEventCreate(&e.b));
        c->ev_pool.push_back(e);
    }
    bool timed = c->ev_used < c->ev_pool.size();
    if (timed) HIPCHK(c, hipEventRecord(c->ev_pool[c->ev_used].a, st));
    if (n_tiles) HIPCHK(c, gscan::launch_scan(db->db.tier, c->variant, a, grid_for(c, db->db, n_tiles), st));
    if (timed) {
        HIPCHK(c, hipEventRecord(c->ev_pool[c->ev_used].b, st));
        c->ev_used++;
    }
    res->recs = c->dv_recs;
    res->desc = (const uint64_t *)c->dv_desc;
    res->n_tiles = n_tiles;
    res->tile_bytes = tile_bytes;
    res->total = 0;
    res->overflow = 0;
    return GSCAN_OK;
}

int gscan_dev_sync(gscan_ctx *c, gscan_dev_result *res)
{
    if (!c || !res) return GSCAN_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->dv_stream ? c->dv_stream : c->compute;
    uint32_t h[kCounterWords] = {0};
    HIPCHK(c, hipMemcpyAsync(h, c->dv_counter, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    res->total = 0;
    for (size_t k = 0; k < (size_t)gscan::kShards; k++) res->total += h[k];
    res->overflow = h[gscan::kShards] != 0;
    return GSCAN_OK;
}

long gscan_dev_fetch(gscan_ctx *c, const gscan_dev_result *res, size_t seg, uint32_t *out, size_t cap)
{
    if (!c || !res) return GSCAN_EINVAL;
    if (seg + 1 >= c->dv_tile_first_h.size()) return fail(c, GSCAN_EINVAL, "segment index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->dv_stream ? c->dv_stream : c->compute;
    HIPCHK(c, hipStreamSynchronize(st));
    uint32_t t0 = c->dv_tile_first_h[seg], t1 = c->dv_tile_first_h[seg + 1];
    std::vector<unsigned long long> d(t1 - t0);
    if (t1 > t0) HIPCHK(c, hipMemcpy(d.data(), c->dv_desc + t0, (size_t)(t1 - t0) * 8, hipMemcpyDeviceToHost));
    const size_t cap_shard = c->dv_rec_cap / gscan::kShards;
    size_t n = 0;
    for (unsigned long long v : d) {
        uint32_t cnt = (uint32_t)v;
        size_t base = (size_t)(v >> 32);
        if (!cnt) continue;
        if (base % cap_shard + cnt > cap_shard) return fail(c, GSCAN_EHIP, "record buffer overflowed; raise gscan_set_capacity");
        if (out && n + cnt <= cap) HIPCHK(c, hipMemcpy(out + n, c->dv_recs + base, (size_t)cnt * 4, hipMemcpyDeviceToHost));
        n += cnt;
    }
    return (long)n;
}

int gscan_kernel_time(gscan_ctx *c, double *sum_ms, uint64_t *launches, int reset)
{
    if (!c) return GSCAN_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    double sum = 0;
    for (size_t i = 0; i < c->ev_used; i++) {
        float ms = 0;
        HIPCHK(c, hipEventSynchronize(c->ev_pool[i].b));
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        sum += ms;
    }
    if (sum_ms) *sum_ms = sum;
    if (launches) *launches = c->ev_used;
    if (reset) c->ev_used = 0;
    return GSCAN_OK;
}

} // extern "C"
