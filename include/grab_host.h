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
#include <sys/stat.h>

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
/* find(path, st, typeflag): THE HOT PATH -- what the nftw callback (grab.cc:265-268) and the worker threads (main.cc:85-99)
 * call per file.  Here the file's windows / its batch may still be in flight when it returns (the next file is read
 * while this one is on the GPU); grab_filegrep_flush scans and prints whatever is pending -- call it after the last
 * file.  grab_filegrep_find, _find_recursive and _free flush by themselves.                                  grab.h:82 */
int grab_filegrep_find3(grab_filegrep *g, const char *path, const struct stat *st, int typeflag);
int grab_filegrep_flush(grab_filegrep *g);
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
/* The same with the match ends the device measured (gscan_last_ends; NULL: none): under -O -l (flags 1 | 2) the walk
 * then never reads `content`, except for a match whose end the device left open (ends[i] == 0). */
int grab_report_chunk_ends_c(const gscan_db *db, unsigned flags, const char *path, const void *content,
                             size_t clen, long long off, const uint32_t *starts, const uint32_t *ends, size_t nstarts,
                             char **out, size_t *outlen);
/* ... and with the device's line pass for the line-printing modes (gscan_last_ext: 4 words per start; gscan_last_gather: the
 * printed lines' text; either may be NULL). */
int grab_report_chunk_ext_c(const gscan_db *db, unsigned flags, const char *path, const void *content, size_t clen,
                            long long off, const uint32_t *starts, const uint32_t *ends, const uint32_t *ext,
                            const unsigned char *gather, size_t nstarts, char **out, size_t *outlen);
void grab_free(void *p);

/*
 * Where the threads of `grab -n workers` run on a node with ndev devices: worker i drives device i mod ndev and is bound to
 * the CPUs local to that device, cut to the process's CPU mask (pin "cpu": the reference's rule, CPU i -- main.cc:200-215;
 * "none": the process's mask).  dev_cpulists[d] = sysfs local_cpulist of device d, allowed = the process's CPUs (cpulist
 * strings; NULL or "" = unknown / all).  Fills devices_out[workers] and one CPU bitmap of bytes_each bytes per worker.
 * Pure: the layout of an 8-GPU node can be checked anywhere (tests/test_host_cpu.py).
 */
int grab_place_workers_c(int workers, int ndev, const char *const *dev_cpulists, const char *allowed, const char *pin,
                         int *devices_out, unsigned char *cpu_bits_out, size_t bytes_each);

/*
 * The tree walk of `grab -n` (grab_amd/csrc/walk.h): `threads` walkers report every regular file nftw(root, fn, 1024,
 * FTW_PHYS) would report as FTW_F (src/main.cc:74-83,178), in no particular order; fn is called CONCURRENTLY from the
 * walker threads.  Returns the number of files reported, -1 on bad arguments.
 */
typedef void (*grab_walk_fn)(const char *path, const struct stat *st, void *arg);
long grab_walk_parallel(const char *root, int threads, grab_walk_fn fn, void *arg);

/* FileGrep::prepare's verdict on the pattern text without opening a device: 0 fine; -1 PCRE rejects it (why = the
 * reference's message, src/grab.cc:108,117); -2 valid PCRE outside the engine's subset. */
int grab_validate(const char *regex, size_t len, int literal, char *why, size_t whycap);

#ifdef __cplusplus
}
#endif
#endif
