// et_tables.cpp -- host-side builders of the decode lookup tables (formats: et_tables.h).
#include "et_tables.h"

#include <cstring>
#include <vector>

namespace et {

// First-level table indexed by the next lut_bits bits: the symbol whose code is a
// prefix of the index and, when a second whole code also fits in the remaining bits,
// that one too (layout: et_kernels.h LUT_*).  Codes longer than lut_bits go to a short
// list searched linearly (they are the rare symbols by construction).

void build_decode_tables(const et_codebook *cb, uint32_t lut_bits_max, uint32_t max_syms, uint32_t *lut, uint32_t *longc, uint16_t *sub,
                         HostDecodeTables *out) {
    const uint32_t k = cb->max_length < lut_bits_max ? (cb->max_length ? cb->max_length : 1) : lut_bits_max;
    const uint32_t n = 1u << k;
    const uint32_t sub_bits = cb->max_length > k ? (cb->max_length - k < DEC_SUB_BITS_MAX ? cb->max_length - k : DEC_SUB_BITS_MAX) : 0;
    std::vector<uint16_t> single(n, 0);  // (len << 8) | sym of the code that prefixes the index
    std::vector<int> sub_of(n, -1);      // second-level table of a first-level index
    uint32_t nl = 0, n_sub = 0;
    for (int s = 0; s < 256; ++s) {
        const uint32_t len = cb->length[s];
        if (!len) continue;
        const uint32_t code = cb->data[s];
        const uint32_t meta = (len << 8) | static_cast<uint32_t>(s);
        if (len <= k) {
            const uint32_t first = code << (k - len), span = 1u << (k - len);
            for (uint32_t i = 0; i < span; ++i) single[first + i] = static_cast<uint16_t>(meta);
        } else {
            longc[2 * nl] = code << (32 - len);
            longc[2 * nl + 1] = meta;
            ++nl;
            const uint32_t prefix = code >> (len - k), rest_bits = len - k;
            if (sub_of[prefix] < 0 && n_sub < DEC_SUB_TABLES_MAX) {
                sub_of[prefix] = static_cast<int>(n_sub);
                std::memset(sub + (static_cast<size_t>(n_sub) << sub_bits), 0, sizeof(uint16_t) << sub_bits);
                ++n_sub;
            }
            if (sub_of[prefix] >= 0 && rest_bits <= sub_bits) {
                const uint32_t rest = code & ((1u << rest_bits) - 1u);
                uint16_t *t = sub + (static_cast<size_t>(sub_of[prefix]) << sub_bits);
                const uint32_t first = rest << (sub_bits - rest_bits), span = 1u << (sub_bits - rest_bits);
                for (uint32_t i = 0; i < span; ++i) t[first + i] = static_cast<uint16_t>(meta);
            }
        }
    }
    for (uint32_t v = 0; v < n; ++v) {
        // greedily take whole codes out of the k-bit index: up to three symbols
        uint32_t entry = 0, used = 0, cnt = 0;
        while (cnt < max_syms) {
            const uint32_t rest = (v << used) & (n - 1);  // the remaining k - used bits, left-aligned in k
            const uint32_t e = single[rest], len = e >> 8;
            if (!len || used + len > k) break;
            entry |= (e & 0xffu) << (8 * cnt);
            used += len;
            ++cnt;
        }
        if (cnt) entry |= (used << LUT_LEN_SHIFT) | (cnt << LUT_N_SHIFT);
        else if (sub_of[v] >= 0) entry = static_cast<uint32_t>(sub_of[v]) | (1u << LUT_SUB_SHIFT);
        lut[v] = entry;
    }
    out->lut_bits = k;
    out->n_long = nl;
    out->sub_bits = sub_bits;
    out->n_sub = n_sub;
}

// Step table of k_dec_sync_reg (et_kernels.h STEP_*): index = the next k bits, entry =
// what a lookup adds to the walk state; second-level tables for the codes longer than k
// follow it.  Returns k.
uint32_t build_step_table(const et_codebook *cb, uint32_t bits_max, uint32_t *steps, uint32_t *sub_bits_out, uint32_t *n_sub_out) {
    const uint32_t k = cb->max_length < bits_max ? (cb->max_length ? cb->max_length : 1) : bits_max;
    const uint32_t n = 1u << k;
    const uint32_t sub_bits = cb->max_length > k ? (cb->max_length - k < DEC_SUB_BITS_MAX ? cb->max_length - k : DEC_SUB_BITS_MAX) : 0;
    uint32_t *ssub = steps + n;
    std::vector<uint8_t> first(n, 0);  // length of the code that prefixes the index
    std::vector<uint8_t> table_of(n, 0);  // 1 + second-level table of an index that is the prefix of longer codes
    uint32_t n_sub = 0;
    for (int s = 0; s < 256; ++s) {
        const uint32_t len = cb->length[s];
        if (!len) continue;
        if (len <= k) {
            std::memset(first.data() + (cb->data[s] << (k - len)), static_cast<int>(len), static_cast<size_t>(1) << (k - len));
            continue;
        }
        const uint32_t prefix = cb->data[s] >> (len - k), rest_bits = len - k;
        if (!table_of[prefix] && n_sub < 15 && ((n_sub + 1) << sub_bits) <= DEC_STEP_SUB_WORDS) {
            std::memset(ssub + (static_cast<size_t>(n_sub) << sub_bits), 0, sizeof(uint32_t) << sub_bits);
            table_of[prefix] = static_cast<uint8_t>(++n_sub);
        }
        if (table_of[prefix] && rest_bits <= sub_bits) {
            uint32_t *t = ssub + (static_cast<size_t>(table_of[prefix] - 1) << sub_bits);
            const uint32_t lo = (cb->data[s] & ((1u << rest_bits) - 1u)) << (sub_bits - rest_bits);
            for (uint32_t i = 0; i < (1u << (sub_bits - rest_bits)); ++i) t[lo + i] = (1u << 16) - len;
        }
    }
    for (uint32_t v = 0; v < n; ++v) {
        uint32_t used = 0, cnt = 0;
        for (;;) {
            const uint32_t len = first[(v << used) & (n - 1)];
            if (!len || used + len > k) break;
            used += len;
            ++cnt;
        }
        steps[v] = cnt ? (static_cast<uint32_t>(first[v]) << 28) + (cnt << 16) - used : STEP_ESCAPE + (static_cast<uint32_t>(table_of[v]) << 28);
    }
    *sub_bits_out = sub_bits;
    *n_sub_out = n_sub;
    return k;
}

// Step table of k_dec_write_reg (et_kernels.h WSTEP_*): as build_step_table, with symbols.
uint32_t build_write_step_table(const et_codebook *cb, uint32_t bits_max, uint32_t *steps, uint32_t *sub_bits_out, uint32_t *n_sub_out) {
    const uint32_t k = cb->max_length < bits_max ? (cb->max_length ? cb->max_length : 1) : bits_max;
    const uint32_t n = 1u << k;
    const uint32_t sub_bits = cb->max_length > k ? (cb->max_length - k < DEC_SUB_BITS_MAX ? cb->max_length - k : DEC_SUB_BITS_MAX) : 0;
    uint32_t *wsub = steps + n;
    std::vector<uint16_t> first(n, 0);  // (len << 8) | sym of the code that prefixes the index
    std::vector<uint8_t> table_of(n, 0);
    uint32_t n_sub = 0;
    for (int s = 0; s < 256; ++s) {
        const uint32_t len = cb->length[s];
        if (!len) continue;
        if (len <= k) {
            const uint32_t lo = cb->data[s] << (k - len);
            for (uint32_t i = 0; i < (1u << (k - len)); ++i) first[lo + i] = static_cast<uint16_t>((len << 8) | static_cast<uint32_t>(s));
            continue;
        }
        const uint32_t prefix = cb->data[s] >> (len - k), rest_bits = len - k;
        if (!table_of[prefix] && n_sub < 254 && ((n_sub + 1) << sub_bits) <= DEC_STEP_SUB_WORDS) {
            std::memset(wsub + (static_cast<size_t>(n_sub) << sub_bits), 0, sizeof(uint32_t) << sub_bits);
            table_of[prefix] = static_cast<uint8_t>(++n_sub);
        }
        if (table_of[prefix] && rest_bits <= sub_bits) {
            uint32_t *t = wsub + (static_cast<size_t>(table_of[prefix] - 1) << sub_bits);
            const uint32_t lo = (cb->data[s] & ((1u << rest_bits) - 1u)) << (sub_bits - rest_bits);
            for (uint32_t i = 0; i < (1u << (sub_bits - rest_bits)); ++i) t[lo + i] = (static_cast<uint32_t>(s) << 16) | ((1u << 10) - len);
        }
    }
    for (uint32_t v = 0; v < n; ++v) {
        uint32_t used = 0, cnt = 0, syms = 0;
        while (cnt < 2) {
            const uint32_t f = first[(v << used) & (n - 1)], len = f >> 8;
            if (!len || used + len > k) break;
            syms |= (f & 0xffu) << (16 + 8 * cnt);
            used += len;
            ++cnt;
        }
        steps[v] = cnt ? syms | (((cnt << 10) - used) & 0xffffu) : (static_cast<uint32_t>(table_of[v]) << 24) | WSTEP_ESCAPE;
    }
    *sub_bits_out = sub_bits;
    *n_sub_out = n_sub;
    return k;
}

// The decisions of the three builders above without the fills (same loops, same order).
void plan_tables(const et_codebook *cb, uint32_t lut_bits_max, uint32_t max_syms, uint32_t step_bits_max, uint32_t wstep_bits_max, TablePlan *p) {
    std::memcpy(p->data, cb->data, sizeof p->data);
    std::memcpy(p->length, cb->length, sizeof p->length);
    std::memset(p->lut_sub, 0, 256);
    std::memset(p->step_sub, 0, 256);
    std::memset(p->wstep_sub, 0, 256);
    std::memset(p->long_idx, 0, 256);
    const uint32_t longest = cb->max_length ? cb->max_length : 1;
    auto width = [&](uint32_t most) { return cb->max_length < most ? longest : most; };
    auto sub_width = [&](uint32_t k) { return cb->max_length > k ? (cb->max_length - k < DEC_SUB_BITS_MAX ? cb->max_length - k : DEC_SUB_BITS_MAX) : 0u; };
    p->lut_bits = width(lut_bits_max);
    p->step_bits = width(step_bits_max);
    p->wstep_bits = width(wstep_bits_max);
    p->sub_bits = sub_width(p->lut_bits);
    p->step_sub_bits = sub_width(p->step_bits);
    p->wstep_sub_bits = sub_width(p->wstep_bits);
    p->max_syms = max_syms;
    p->pad_ = 0;
    uint8_t of_lut[1u << DEC_LUT_BITS_MAX], of_step[1u << DEC_STEP_BITS_MAX], of_w[1u << DEC_LUT_BITS_MAX];  // (stack: this runs once per decode call)
    std::memset(of_lut, 0, static_cast<size_t>(1) << p->lut_bits);
    std::memset(of_step, 0, static_cast<size_t>(1) << p->step_bits);
    std::memset(of_w, 0, static_cast<size_t>(1) << p->wstep_bits);
    uint32_t nl = 0, n_sub = 0, n_step = 0, n_w = 0;
    for (int s = 0; s < 256; ++s) {
        const uint32_t len = cb->length[s];
        if (!len) continue;
        if (len > p->lut_bits) {
            p->long_idx[s] = static_cast<uint8_t>(nl++);
            const uint32_t prefix = cb->data[s] >> (len - p->lut_bits);
            if (!of_lut[prefix] && n_sub < DEC_SUB_TABLES_MAX) of_lut[prefix] = static_cast<uint8_t>(++n_sub);
            p->lut_sub[s] = of_lut[prefix];
        }
        if (len > p->step_bits) {
            const uint32_t prefix = cb->data[s] >> (len - p->step_bits);
            if (!of_step[prefix] && n_step < 15 && ((n_step + 1) << p->step_sub_bits) <= DEC_STEP_SUB_WORDS) of_step[prefix] = static_cast<uint8_t>(++n_step);
            p->step_sub[s] = of_step[prefix];
        }
        if (len > p->wstep_bits) {
            const uint32_t prefix = cb->data[s] >> (len - p->wstep_bits);
            if (!of_w[prefix] && n_w < 254 && ((n_w + 1) << p->wstep_sub_bits) <= DEC_STEP_SUB_WORDS) of_w[prefix] = static_cast<uint8_t>(++n_w);
            p->wstep_sub[s] = of_w[prefix];
        }
    }
    p->n_long = nl;
    p->n_sub = n_sub;
    p->n_step_sub = n_step;
    p->n_wstep_sub = n_w;
}

}  // namespace et
