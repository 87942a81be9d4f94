/* ORACLE (test infrastructure only) -- SHA-256 (FIPS 180-4) + RFC 9380 expand_message_xmd / gnark fr.Hash */
#ifndef ORC_SHA256_H
#define ORC_SHA256_H
#include <stddef.h>
#include <stdint.h>
#include "field.h"
void orc_sha256(const uint8_t* msg, size_t len, uint8_t out[32]);
void orc_expand_xmd(const uint8_t* msg, size_t mlen, const uint8_t* dst, size_t dlen, uint8_t* out, size_t outlen);
/* count elements of Fr: 48 bytes each, big-endian, reduced */
void orc_hash_to_fr(const uint8_t* msg, size_t mlen, const char* dst, fe* out, int count);
#endif
