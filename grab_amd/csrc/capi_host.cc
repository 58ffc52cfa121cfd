// capi_host.cc -- extern "C" facade declared in include/grab_host.h.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "../../include/grab_host.h"
#include "filegrep.h"
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
    if (!db || !out || !outlen) return -1;
    gscan_info info;
    if (gscan_db_info(db, &info) != GSCAN_OK) return -1;
    std::string text;
    grab_report_chunk(db, info.minlen, flags, path ? path : "", (const char *)content, clen, off, starts, nstarts, text, nullptr, ends);
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

} // extern "C"
