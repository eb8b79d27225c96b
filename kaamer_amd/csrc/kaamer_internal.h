// kaamer_internal.h — declarations shared by the translation units of
// libkaamer_hip.so.  Not part of the ABI.
#pragma once
#include <stdint.h>

#include "../../include/kaamer_hip.h"
#include "kaamer_layout.h"

struct kaamer_image {
    kh_image_header hdr;
    kh_bucket *buckets;
    uint32_t *arena;
};

// records the thread-local error string and returns `code`
int kaamer_fail(int code, const char *fmt, ...);
int kaamer_image_alloc(kaamer_image *img, uint64_t n_buckets, uint64_t arena_words);
extern "C" void kaamer_stats_from_header(const kh_image_header *h, kaamer_image_stats *out);

// an image whose two arrays live on a device (builder_device.hip); the owner hipFree()s them
struct kaamer_device_image {
    kh_image_header hdr;
    kh_bucket *d_buckets;
    uint32_t *d_arena;
};
int kaamer_build_on_device(const uint8_t *seqs, const uint64_t *offsets, const uint32_t *ids, uint32_t n_proteins,
                           uint32_t shard, uint32_t n_shards, double load, int device, kaamer_device_image *out);
extern "C" void kaamer_proteins_raw(const kaamer_proteins *p, const uint8_t **seqs, const uint64_t **offsets, const uint32_t **ids, uint32_t *n);
