/*
 * et_cpu_fast.c -- a FAST CPU variant of the same path, for bench.py's CPU baselines only.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE (see et_oracle.h): the product never calls this.
 * SURVEY.md 8d asks for two CPU figures beside the GPU's: the faithful restatement on one
 * core (et_oracle.c: bit-serial like encode.zig:303-315 and decode.zig:143-203) and "the
 * fast chunk-parallel CPU variant on all host cores".  This file is the second one: the
 * per-chunk pieces, written the way one would write them for a CPU today (64-bit
 * accumulator pack, 12-bit lookup decode); bench.py runs one chunk per host thread
 * (ctypes releases the GIL) and does the cheap sequential steps in between.  Output is
 * bit-identical to et_oracle_encode / the input again, which tests/test_oracle.py checks.
 *
 * Chunk-parallel decode has the GPU path's problem -- the format has no index -- and
 * solves it the same way: every chunk is walked from its byte boundary first (counting,
 * and marking which bit positions of its first ET_FAST_MARKS bits are codeword boundaries),
 * then the true start of chunk t (= the true end of chunk t-1) is walked forward until it
 * lands on a marked boundary, from where the first walk's count and exit hold.
 */
#include <stdlib.h>
#include <string.h>

#include "et_oracle.h"

#define ET_FAST_LUT_BITS 12

struct et_fast_tables {
    uint16_t lut[1u << ET_FAST_LUT_BITS]; /* (len << 8) | sym for codes <= 12 bits, else 0 */
    /* two codewords per lookup where both fit the 12 bits:
     * sym1 | sym2 << 8 | len1 << 16 | (len1 + len2) << 20 | n << 24  (n = 0: take the slow way) */
    uint32_t lut2[1u << ET_FAST_LUT_BITS];
    uint32_t n_long;                       /* codes longer than 12 bits, left-aligned in 32 bits */
    uint32_t long_code[256];
    uint8_t long_len[256], long_sym[256];
    uint32_t max_len;
};

size_t et_fast_tables_size(void) { return sizeof(struct et_fast_tables); }

/* Returns 0, or ET_ORACLE_FORMAT for a table this variant does not take (a code longer
 * than 32 bits: encode.zig:141's u32 wraps there, quirk Q3 -- left to the restatement). */
int et_fast_build_tables(const et_oracle_dict *dict, struct et_fast_tables *t)
{
    memset(t, 0, sizeof *t);
    for (int s = 0; s < 256; s++) {
        const uint32_t len = dict->length[s];
        if (len == 0) continue;
        if (len > 32) return ET_ORACLE_FORMAT;
        if (len > t->max_len) t->max_len = len;
        const uint32_t code = len == 32 ? dict->data[s] : (dict->data[s] & ((1u << len) - 1u));
        if (len <= ET_FAST_LUT_BITS) {
            const uint32_t lo = code << (ET_FAST_LUT_BITS - len), n = 1u << (ET_FAST_LUT_BITS - len);
            for (uint32_t k = 0; k < n; k++) t->lut[lo + k] = (uint16_t)((len << 8) | (uint32_t)s);
        } else {
            t->long_code[t->n_long] = code << (32 - len);
            t->long_len[t->n_long] = (uint8_t)len;
            t->long_sym[t->n_long] = (uint8_t)s;
            t->n_long++;
        }
    }
    for (uint32_t w = 0; w < (1u << ET_FAST_LUT_BITS); w++) {
        const uint16_t e1 = t->lut[w];
        if (!e1) continue;
        const uint32_t len1 = e1 >> 8;
        uint32_t v = (uint32_t)(e1 & 0xffu) | (len1 << 16) | (len1 << 20) | (1u << 24);
        const uint16_t e2 = t->lut[(w << len1) & ((1u << ET_FAST_LUT_BITS) - 1u)];
        if (e2 && len1 + (e2 >> 8) <= ET_FAST_LUT_BITS)
            v = (uint32_t)(e1 & 0xffu) | ((uint32_t)(e2 & 0xffu) << 8) | (len1 << 16) | ((len1 + (e2 >> 8)) << 20) | (2u << 24);
        t->lut2[w] = v;
    }
    return 0;
}

/* Pack `text` at bit position start_bit of `out` (MSB first, as std.io.bitWriter(.big)).
 * Bytes wholly inside the chunk are stored; the first and last byte may be shared with
 * the neighbouring chunks and are OR-ed in atomically (the caller zeroes them first).
 * Returns the end bit. */
uint64_t et_fast_pack(const et_oracle_dict *dict, const uint8_t *text, size_t n, uint8_t *out, uint64_t start_bit)
{
    uint64_t pos = start_bit >> 3;           /* next byte to emit */
    uint64_t acc = 0;                        /* pending bits, right-aligned */
    unsigned nb = (unsigned)(start_bit & 7); /* pending bit count; the leading start_bit & 7 are zeros that get OR-ed */
    int first = 1;
    uint64_t bits = start_bit;
    for (size_t i = 0; i < n; i++) {
        const unsigned len = dict->length[text[i]];
        if (len == 0) continue;
        const uint64_t code = len == 32 ? dict->data[text[i]] : (dict->data[text[i]] & ((1u << len) - 1u));
        acc = (acc << len) | code; /* nb < 32 here, len <= 32 */
        nb += len;
        bits += len;
        if (nb >= 32) {
            const uint32_t w = __builtin_bswap32((uint32_t)(acc >> (nb - 32)));
            nb -= 32;
            if (first) { /* the chunk's first byte may hold the previous chunk's last bits */
                __atomic_fetch_or(&out[pos], (uint8_t)w, __ATOMIC_RELAXED);
                memcpy(out + pos + 1, (const uint8_t *)&w + 1, 3);
                first = 0;
            } else {
                memcpy(out + pos, &w, 4);
            }
            pos += 4;
        }
    }
    while (nb >= 8) {
        const uint8_t b = (uint8_t)(acc >> (nb - 8));
        nb -= 8;
        if (first) {
            __atomic_fetch_or(&out[pos], b, __ATOMIC_RELAXED);
            first = 0;
        } else {
            out[pos] = b;
        }
        pos++;
    }
    if (nb) __atomic_fetch_or(&out[pos], (uint8_t)((acc << (8 - nb)) & 0xffu), __ATOMIC_RELAXED); /* shared with the next chunk */
    return bits;
}

static inline uint32_t peek32(const uint8_t *body, uint64_t total_bytes, uint64_t p)
{
    const uint64_t byte = p >> 3;
    uint64_t w = 0;
    if (byte + 8 <= total_bytes) {
        memcpy(&w, body + byte, 8);
        w = __builtin_bswap64(w);
    } else {
        for (unsigned k = 0; k < 8; k++) w = (w << 8) | (byte + k < total_bytes ? body[byte + k] : 0u);
    }
    return (uint32_t)((w << (p & 7)) >> 32);
}

/* One codeword at bit p: returns its length (0: no code matches) and the symbol. */
static inline unsigned one_code(const struct et_fast_tables *t, uint32_t window, uint8_t *sym)
{
    const uint16_t e = t->lut[window >> (32 - ET_FAST_LUT_BITS)];
    if (e) {
        *sym = (uint8_t)e;
        return e >> 8;
    }
    for (uint32_t k = 0; k < t->n_long; k++) {
        const unsigned len = t->long_len[k];
        if (((window ^ t->long_code[k]) >> (32 - len)) == 0) {
            *sym = t->long_sym[k];
            return len;
        }
    }
    return 0;
}

#define ET_FAST_NO_MARK 0xffffffffu

/* Walk codewords from bit start_bit while the position is < end_bit (and a whole code is
 * left before total_bits) and fewer than max_syms symbols were seen.  Symbols go to `out`
 * when it is not NULL.  marks (optional, n_marks entries, pre-filled with NO_MARK):
 * marks[p - mark_base] = number of symbols in front of codeword boundary p.
 * Returns the symbol count; *exit_bit = where the walk stopped. */
uint64_t et_fast_walk(const struct et_fast_tables *t, const uint8_t *body, uint64_t total_bytes, uint64_t start_bit,
                      uint64_t end_bit, uint64_t max_syms, uint8_t *out, uint64_t *exit_bit, uint32_t *marks,
                      uint64_t mark_base, uint32_t n_marks)
{
    const uint64_t total_bits = total_bytes * 8;
    uint64_t p = start_bit, count = 0;
    const uint64_t marks_end = marks ? mark_base + n_marks : 0;
    /* two codewords per lookup while nothing has to be looked at in between: outside the
     * marked window, 8 readable bytes ahead, two symbols still wanted, and the first
     * codeword ends in front of end_bit (else the second one belongs to the next chunk) */
    const uint64_t fast_end = total_bytes >= 8 ? (total_bytes - 8) * 8 : 0;
    for (;;) {
        while (p >= marks_end && p < fast_end && count + 2 <= max_syms) {
            uint64_t w;
            memcpy(&w, body + (p >> 3), 8);
            const uint32_t window = (uint32_t)((__builtin_bswap64(w) << (p & 7)) >> 32);
            const uint32_t e = t->lut2[window >> (32 - ET_FAST_LUT_BITS)];
            if ((e >> 24) != 2 || p + ((e >> 16) & 15u) >= end_bit) break;
            if (out) {
                out[count] = (uint8_t)e;
                out[count + 1] = (uint8_t)(e >> 8);
            }
            count += 2;
            p += (e >> 20) & 15u;
        }
        if (!(p < end_bit && count < max_syms)) break;
        if (marks && p >= mark_base && p < marks_end) marks[p - mark_base] = (uint32_t)count;
        uint8_t sym = 0;
        const unsigned len = one_code(t, peek32(body, total_bytes, p), &sym);
        if (len == 0 || p + len > total_bits) break;
        if (out) out[count] = sym;
        count++;
        p += len;
    }
    *exit_bit = p;
    return count;
}

/* From the true position p (>= mark_base), walk until a boundary the first walk marked.
 * Returns 1 and (*extra = symbols walked here, *at = index into marks of the meeting point),
 * or 0 when the two walks do not meet inside the marked window. */
int et_fast_merge(const struct et_fast_tables *t, const uint8_t *body, uint64_t total_bytes, uint64_t p,
                  const uint32_t *marks, uint64_t mark_base, uint32_t n_marks, uint64_t *extra, uint32_t *at)
{
    const uint64_t total_bits = total_bytes * 8;
    uint64_t n = 0;
    while (p >= mark_base && p - mark_base < n_marks) {
        if (marks[p - mark_base] != ET_FAST_NO_MARK) {
            *extra = n;
            *at = (uint32_t)(p - mark_base);
            return 1;
        }
        uint8_t sym;
        const unsigned len = one_code(t, peek32(body, total_bytes, p), &sym);
        if (len == 0 || p + len > total_bits) return 0;
        p += len;
        n++;
    }
    return 0;
}
