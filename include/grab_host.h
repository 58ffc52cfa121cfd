/*
 * grab_host.h -- C facade over the host-side FileGrep (grab_amd/csrc/filegrep.h), so that
 * non-C++ callers (the Python tests, other language bindings) can drive the same object
 * the `grab` binary uses.  One function per FileGrep method of
 * /root/reference/src/grab.h:55-85; same 0 / -1 + why() convention.
 */
#ifndef GRAB_HOST_H
#define GRAB_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "gscan.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct grab_filegrep grab_filegrep;

grab_filegrep *grab_filegrep_new(void);                               /* FileGrep()            grab.h:57 */
void grab_filegrep_free(grab_filegrep *g);                            /* ~FileGrep()           grab.h:59 */
const char *grab_filegrep_why(grab_filegrep *g);                      /* why()                 grab.h:61 */
void grab_filegrep_recurse(grab_filegrep *g);                         /* recurse()             grab.h:66 */
void grab_filegrep_show_path(grab_filegrep *g, int on);               /* show_path(bool)       grab.h:71 */
void grab_filegrep_config(grab_filegrep *g, const char *key, size_t value); /* config(map), one key at a time  grab.h:78 */
int grab_filegrep_prepare(grab_filegrep *g, const char *regex, size_t len); /* prepare(string)  grab.h:76 */
int grab_filegrep_find(grab_filegrep *g, const char *path);           /* find(string)          grab.h:80 */
int grab_filegrep_find_recursive(grab_filegrep *g, const char *path); /* find_recursive(string) grab.h:84 */
int grab_filegrep_engine_option(grab_filegrep *g, const char *name, long value);

/*
 * The per-chunk printing rule of grab.cc:171-213 as a pure host function: given the
 * chunk bytes and the engine's ascending candidate starts, produce exactly what the
 * reference prints for that chunk.  flags: 1 -O, 2 -l, 4 -s, 8 path prefix, 16 colour.
 * *out is malloc'd; release with grab_free.
 */
int grab_report_chunk_c(const gscan_db *db, unsigned flags, const char *path, const void *content,
                        size_t clen, long long off, const uint32_t *starts, size_t nstarts,
                        char **out, size_t *outlen);
void grab_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
