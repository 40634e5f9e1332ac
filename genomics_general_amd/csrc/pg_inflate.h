// what the host layer needs of the device inflater (pg_inflate_core.h / pg_inflate.hip): error bits and the member record
#pragma once
#include <stdint.h>

enum {
    PGI_ERR_BTYPE = 1,    // block type 3
    PGI_ERR_STORED = 2,   // stored block: LEN != ~NLEN
    PGI_ERR_CODE = 4,     // invalid / over-subscribed / incomplete code, invalid symbol
    PGI_ERR_DIST = 8,     // distance reaches in front of the member's first byte
    PGI_ERR_OUT = 16,     // more (or fewer) bytes than the member's ISIZE
    PGI_ERR_IN = 32,      // the stream runs past the member's compressed bytes
    PGI_ERR_CRC = 64      // CRC-32 of the inflated bytes differs from the member's trailer (k_crc32)
};

struct PgiMember {
    uint32_t in_off;      // byte offset of the deflate stream in the block's compressed bytes
    uint32_t in_len;      // its length (member size - header - 8)
    uint64_t out_off;     // where its text goes (byte offset in the output)
    uint32_t out_len;     // ISIZE of the trailer
    uint32_t crc;         // CRC-32 of the trailer
};
