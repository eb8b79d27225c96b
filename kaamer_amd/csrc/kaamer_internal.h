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
