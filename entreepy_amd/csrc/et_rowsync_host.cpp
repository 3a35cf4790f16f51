// et_rowsync_host.cpp -- which code tables the row walk (et_rowsync.h) applies to.  No HIP here.
#include "et_rowsync.h"

namespace et {

bool row_code_of(const et_codebook *cb, RowCode *rc) {
    if (!cb || !rc || cb->min_length < 7 || cb->max_length > 8) return false;  // (7 / 7: 128 codewords of 7 bits, t = 128)
    bool seen7[128] = {}, seen8[256] = {};
    uint32_t n7 = 0, n8 = 0;
    for (int s = 0; s < 256; ++s) {
        const uint32_t len = cb->length[s];
        if (len == 0) continue;
        if (len == 7) {
            const uint32_t v = cb->data[s] & 0x7fu;
            if (seen7[v]) return false;
            seen7[v] = true;
            ++n7;
        } else if (len == 8) {
            const uint32_t v = cb->data[s] & 0xffu;
            if (seen8[v]) return false;
            seen8[v] = true;
            ++n8;
        } else {
            return false;
        }
    }
    if (2 * n7 + n8 != 256) return false;  // complete: every bit pattern is a codeword's beginning
    for (uint32_t v = 0; v < n7; ++v)
        if (!seen7[v]) return false;  // the 7-bit codes are the values 0 .. n7 - 1 (what encode.zig's two-queue merge hands out)
    for (uint32_t v = 0; v < 256; ++v)
        if (seen8[v] && (v >> 1) < n7) return false;  // prefix-free
    rc->t = n7;
    return true;
}

}  // namespace et
