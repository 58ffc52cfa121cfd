// capi_host.cc -- extern "C" facade declared in include/grab_host.h.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "../../include/grab_host.h"
#include "filegrep.h"
#include "placement.h"
#include "walk.h"

struct grab_filegrep {
    FileGrep g;
};

extern "C" {

grab_filegrep *grab_filegrep_new(void) { return new (std::nothrow) grab_filegrep(); }
void grab_filegrep_free(grab_filegrep *g) { delete g; }
const char *grab_filegrep_why(grab_filegrep *g) { return g->g.why(); }
void grab_filegrep_recurse(grab_filegrep *g) { g->g.recurse(); }
void grab_filegrep_show_path(grab_filegrep *g, int on) { g->g.show_path(on != 0); }
void grab_filegrep_config(grab_filegrep *g, const char *key, size_t value)
{
    std::map<std::string, size_t> kv;
    kv[key] = value;
    g->g.config(kv);
}
int grab_filegrep_prepare(grab_filegrep *g, const char *regex, size_t len) { return g->g.prepare(std::string(regex, len)); }
int grab_filegrep_find(grab_filegrep *g, const char *path) { return g->g.find(std::string(path)); }
int grab_filegrep_find3(grab_filegrep *g, const char *path, const struct stat *st, int typeflag) { return g->g.find(path, st, typeflag); }
int grab_filegrep_flush(grab_filegrep *g) { return g->g.flush(); }
int grab_filegrep_find_recursive(grab_filegrep *g, const char *path) { return g->g.find_recursive(std::string(path)); }
int grab_filegrep_engine_option(grab_filegrep *g, const char *name, long value) { return g->g.engine_option(name, value); }

int grab_report_chunk_c(const gscan_db *db, unsigned flags, const char *path, const void *content, size_t clen,
                        long long off, const uint32_t *starts, size_t nstarts, char **out, size_t *outlen)
{
    return grab_report_chunk_ends_c(db, flags, path, content, clen, off, starts, nullptr, nstarts, out, outlen);
}

int grab_report_chunk_ends_c(const gscan_db *db, unsigned flags, const char *path, const void *content, size_t clen,
                             long long off, const uint32_t *starts, const uint32_t *ends, size_t nstarts, char **out, size_t *outlen)
{
    return grab_report_chunk_ext_c(db, flags, path, content, clen, off, starts, ends, nullptr, nullptr, nstarts, out, outlen);
}

int grab_report_chunk_ext_c(const gscan_db *db, unsigned flags, const char *path, const void *content, size_t clen, long long off,
                            const uint32_t *starts, const uint32_t *ends, const uint32_t *ext, const unsigned char *gather, size_t nstarts,
                            char **out, size_t *outlen)
{
    if (!db || !out || !outlen) return -1;
    gscan_info info;
    if (gscan_db_info(db, &info) != GSCAN_OK) return -1;
    std::string text;
    grab_report_chunk(db, info.minlen, flags, path ? path : "", (const char *)content, clen, off, starts, nstarts, text, ext, ends, gather);
    char *buf = (char *)malloc(text.size() + 1);
    if (!buf) return -1;
    memcpy(buf, text.data(), text.size());
    buf[text.size()] = 0;
    *out = buf;
    *outlen = text.size();
    return 0;
}

void grab_free(void *p) { free(p); }

long grab_walk_parallel(const char *root, int threads, grab_walk_fn fn, void *arg)
{
    if (!root || !fn) return -1;
    return (long)grab_walk(root, threads, [&](std::string &&path, const struct stat &st) { fn(path.c_str(), &st, arg); });
}

int grab_validate(const char *regex, size_t len, int literal, char *why, size_t whycap)
{
    std::string w;
    const int rc = FileGrep::validate(std::string(regex, len), literal != 0, w);
    if (why && whycap) snprintf(why, whycap, "%s", w.c_str());
    return rc;
}


// `grab -n workers` on a node with ndev devices (placement.h): worker i's device and its CPUs as a bitmap of bytes_each
// bytes (bit c of byte c / 8 = CPU c).  dev_cpulists[d] / allowed are sysfs cpulist strings ("0-63,128-191"; NULL or "":
// unknown / every CPU below 8 * bytes_each).
int grab_place_workers_c(int workers, int ndev, const char *const *dev_cpulists, const char *allowed, const char *pin, int *devices_out,
                         unsigned char *cpu_bits_out, size_t bytes_each)
{
    if (workers < 0 || ndev < 1 || !devices_out || !cpu_bits_out) return -1;
    auto parse = [](const char *list, std::vector<int> &out) {
        out.clear();
        if (!list) return;
        int cpus[4096];
        const long n = gscan_parse_cpulist(list, cpus, 4096);
        for (long k = 0; k < n && k < 4096; k++) out.push_back(cpus[k]);
    };
    std::vector<std::vector<int>> dev((size_t)ndev);
    for (int d = 0; d < ndev; d++) parse(dev_cpulists ? dev_cpulists[d] : nullptr, dev[(size_t)d]);
    cpu_set_t mask;
    CPU_ZERO(&mask);
    std::vector<int> al;
    parse(allowed, al);
    if (al.empty())
        for (size_t c = 0; c < bytes_each * 8 && c < (size_t)CPU_SETSIZE; c++) CPU_SET(c, &mask);
    for (int c : al)
        if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &mask);
    const std::vector<WorkerPlace> pl = grab_place_workers(workers, ndev, dev, mask, pin);
    memset(cpu_bits_out, 0, bytes_each * (size_t)workers);
    for (int i = 0; i < workers; i++) {
        devices_out[i] = pl[(size_t)i].device;
        for (size_t c = 0; c < bytes_each * 8 && c < (size_t)CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &pl[(size_t)i].cpus)) cpu_bits_out[(size_t)i * bytes_each + c / 8] |= (unsigned char)(1u << (c % 8));
    }
    return 0;
}
} // extern "C"
