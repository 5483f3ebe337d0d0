// Host-side helpers of the C ABI (no device code).
//   dpd_crc32c: CRC32C (Castagnoli) as used by the TF checkpoint container (dpdist_amd/tf_checkpoint.py, row f3) --
//   the pure-Python fallback manages ~1 MB/s, this table-driven loop several hundred.
#include <stddef.h>
#include <stdint.h>

#include "../../include/dpdist_capi.h"

namespace {
struct CrcTable {
    uint32_t t[8][256];
    CrcTable() {
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][n] = c;
        }
        for (uint32_t n = 0; n < 256; ++n)
            for (int s = 1; s < 8; ++s) t[s][n] = (t[s - 1][n] >> 8) ^ t[0][t[s - 1][n] & 0xFF];
    }
};
}  // namespace

extern "C" uint32_t dpd_crc32c(const void* data, size_t n, uint32_t crc) {
    static const CrcTable tab;
    const uint8_t* p = (const uint8_t*)data;
    crc = ~crc;
    while (n >= 8) {   // slicing-by-8
        const uint32_t lo = crc ^ ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
        crc = tab.t[7][lo & 0xFF] ^ tab.t[6][(lo >> 8) & 0xFF] ^ tab.t[5][(lo >> 16) & 0xFF] ^ tab.t[4][lo >> 24] ^
              tab.t[3][p[4]] ^ tab.t[2][p[5]] ^ tab.t[1][p[6]] ^ tab.t[0][p[7]];
        p += 8;
        n -= 8;
    }
    while (n--) crc = tab.t[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
