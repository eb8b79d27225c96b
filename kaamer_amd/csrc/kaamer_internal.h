// kaamer_internal.h — declarations shared by the translation units of
// libkaamer_hip.so.  Not part of the ABI.
#pragma once
#include <stdint.h>

#include <string>

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

// The readers take the bytes of a file: when they start with the gzip signature (what http.DetectContentType calls
// "application/x-gzip": search.go:255-263, inputEMBL.go:76-84) the text is inflated first, all members of the stream
// (Go's gzip.Reader is multistream).  A stream that breaks off or is damaged yields the text up to there, as the
// reference's scanner does (it stops at the read error, whatever was read is processed, the error is not looked at).
// Returns 0, or KAAMER_E_FORMAT when not even the first header is a gzip header (gzip.NewReader fails: no input).
bool kaamer_is_gzip(const char *text, uint64_t len);
int kaamer_gunzip(const char *text, uint64_t len, std::string *out);
