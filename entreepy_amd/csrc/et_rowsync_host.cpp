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

// A complete code of L- and (L + 1)-bit codewords, L <= 7: does a walk that begins at a wrong bit meet the true one soon enough for
// the tree walk's run-in (et_treewalk.hip: 128 bits, then the seams settle what is left)?  Two walks a few bits apart move against
// each other whenever one of them reads a short codeword and the other a long one -- with p = the share of L-bit patterns that ARE
// codewords that happens with q = 2 p (1 - p) per codeword -- and meet after a random walk over the R = L + 1 - p phases: about
// (R^2 / 6) / q codewords of R bits.  Measured on uniform draws over k symbols (profiles/r04_flat_alphabets.jsonl: every k from 3
// to 124 that was tried): the tree walk settles streams up to an estimate of ~170 bits when short codewords are the rare ones and
// up to ~320 when long ones are; past that most blocks give up, the exit maps run after all and the attempt has cost a sweep.
// (L = 7, 1 GiB: 150..205 symbols settle and decode at 630-810 GB/s where the row walk gives 445-580; 129..139 and 210..255 do not
// or settle more slowly than the row walk takes them: tools/probe/ab_row_vs_tw.sh.)
bool quick_to_synchronise(const et_codebook *cb) {
    if (!cb || cb->n_coded <= 2 || cb->max_length != cb->min_length + 1 || cb->min_length > 7) return false;
    const uint32_t L = cb->min_length;
    uint32_t n_short = 0;
    for (int s = 0; s < 256; ++s) n_short += cb->length[s] == L ? 1u : 0u;
    const double p = static_cast<double>(n_short) / static_cast<double>(1u << L);
    if (p <= 0.0 || p >= 1.0) return false;
    const double q = 2.0 * p * (1.0 - p), R = L + 1 - p;
    const double bits = R * R / 6.0 / q * R;
    return bits < (p >= 0.5 ? 330.0 : 160.0);
}

}  // namespace et
