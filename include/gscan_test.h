/*
 * gscan_test.h -- test and diagnostic hooks of libgscan.so.
 *
 * Not part of what a binding of the engine needs (include/gscan.h is): the pattern compiler's tables, the host matcher's
 * verdict at one offset, the device's VM run on the host, the staging pool's counters.  tests/ and bench.py's checker use
 * them to compare the product's parts with the oracle one by one.  Same conventions as gscan.h.
 */
#ifndef GSCAN_TEST_H
#define GSCAN_TEST_H

#include "gscan.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 256-entry membership table (1 byte each) of window position `pos`; pos == -1: the tail class (alternative 0) */
GSCAN_API int gscan_db_class(const gscan_db *db, int pos, uint8_t table[256]);
/* the same for alternative `alt`; *len (optional) receives that alternative's window length */
GSCAN_API int gscan_db_alt_class(const gscan_db *db, int alt, int pos, uint8_t table[256], int *len);
/* 1 if the pattern matches AT offset p of content[0..clen) with the subject starting at p, else 0 */
GSCAN_API int gscan_match_at(const gscan_db *db, const void *content, size_t clen, uint32_t p);
/* ovector[1] for a match starting at `start` of content[0..clen), subject starting there: src/grab.cc:178 semantics */
GSCAN_API uint32_t gscan_match_end(const gscan_db *db, const void *content, size_t clen, uint32_t start);
/* the offsets gscan_next_match tests itself because a window there would end with the chunk (patterns with
 * look-ahead context only: foo\b, foo$ ...); exported for tests.  Returns how many there are; fills at most cap. */
GSCAN_API size_t gscan_tail_positions(const gscan_db *db, size_t clen, uint32_t *out, size_t cap);
/* The device's VM (grab_amd/csrc/vm.h), run on the host -- for tests and diagnostics.
 *   gscan_vm_verdict  at offset p with the subject starting at subject_start: 0 no match starts at p, 1 a match starts at p,
 *                     2 the VM gave up (step / stack limit), -1 the pattern has no VM program.  Never 0 where
 *                     gscan_match_info finds a match.
 *   gscan_vm_filter   what the K3 kernel does with its filter hits when gscan_info.vm is set: of hits[0..n) (offsets of device
 *                     windows) the ones it keeps, in order, into kept (may be NULL); returns how many, -1 if vm is not set.
 *   gscan_vm_pair     the device's two-byte table (DevProgram::vm_pair): 1 a match may begin with the bytes b0 b1, 0 none can,
 *                     -1 the pattern has no table.  gscan_prefix_viable: the probe the table is built from -- may a match
 *                     begin at offset 0 of some subject that starts with these n bytes?  (0 only if the host matcher fails
 *                     without looking at or beyond byte n.) */
GSCAN_API int gscan_vm_verdict(const gscan_db *db, const void *content, size_t clen, uint32_t subject_start, uint32_t p);
/* the VM's full answer: as gscan_vm_verdict, and for 1 the match's end (ovector[1]) and whether its path closed a capturing group
 * -- what the device's resolve pass (gscan_info.resolve) writes next to every record */
GSCAN_API int gscan_vm_match(const gscan_db *db, const void *content, size_t clen, uint32_t subject_start, uint32_t p, uint32_t *end, int *captures);
/* k_resolve on the host (tests): of hits[0..n) -- offsets where a start window fits -- the ones at which the VM finds a match
 * or gives up, into starts, and what the device writes next to them into ends; returns how many, -1 if there is no program */
GSCAN_API long gscan_vm_resolve(const gscan_db *db, const void *content, size_t clen, const uint32_t *hits, size_t n, uint32_t *starts, uint32_t *ends);
GSCAN_API int gscan_vm_pair(const gscan_db *db, unsigned b0, unsigned b1);
GSCAN_API int gscan_prefix_viable(const gscan_db *db, const void *bytes, size_t n);
GSCAN_API long gscan_vm_filter(const gscan_db *db, const void *content, size_t clen, const uint32_t *hits, size_t n, uint32_t *kept);
/* What the kernels scan for alternative `alt`: the membership table of DEVICE window position `pos` (the window plus
 * its context positions), the device window length, and the shift from a device hit to the reported match start. */
GSCAN_API int gscan_db_dev_window(const gscan_db *db, int alt, int pos, uint8_t table[256], int *len, int *shift);
/* gscan_device_cpulist for an explicit sysfs root + bus id (no device needed) */
GSCAN_API int gscan_pci_cpulist(const char *sysfs_pci_root, const char *busid, char *buf, size_t cap);
/* the staging-block pool of the context's device, for tests and diagnostics: out[0] blocks allocated, out[1] the pool's cap,
 * out[2] how often a reader had to sleep on a block's DMA event because every block was in flight, out[3] how often it had
 * to sleep until another reader brought a block back.  (tests/test_gpu_pool.py forces these slow paths and checks they ran.) */
GSCAN_API int gscan_pool_stats(const gscan_ctx *ctx, uint64_t out[4]);
/* reader threads a device gets when GSCAN_READERS is unset (*readers == 0 above): 8, fewer when the device's share of its
 * NUMA node's CPUs (local_cpus / devices_sharing that node) is small or when the node drives so many devices (devices_total)
 * that 8 readers each would outrun what the host's page cache can feed (24 in all); exported for tests */
GSCAN_API int gscan_auto_readers(int local_cpus, int devices_sharing, int devices_total);

#ifdef __cplusplus
}
#endif
#endif /* GSCAN_TEST_H */
