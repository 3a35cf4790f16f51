// et_treewalk_host.cpp -- host part of the tree walk (et_treewalk.h): the code table as a tree, and a
// reference fill of the walk's table.  Plain C++ (also built by tests/test_sanitizers.py with g++).
#include "et_treewalk.h"

#include <cstring>

namespace et {

int tw_build_tree(const et_codebook *cb, TwTree *tree) {
    if (cb->n_coded < 2) return ET_ERR_UNSUPPORTED;
    constexpr int16_t NONE = -1;
    tree->n_int = 1;
    tree->pad_ = 0;
    for (uint32_t i = 0; i < 2 * TW_MAX_NODES; ++i) tree->child[i] = NONE;
    for (int s = 0; s < 256; ++s) {
        const uint32_t len = cb->length[s];
        if (!len) continue;
        if (len > 32) return ET_ERR_UNSUPPORTED;
        uint32_t node = 0;
        for (uint32_t i = len; i-- > 0;) {  // first stream bit = bit len-1 of the code (encode.zig:293,311)
            const uint32_t bit = (cb->data[s] >> i) & 1u;
            int16_t &c = tree->child[2 * node + bit];
            if (i == 0) {
                if (c != NONE) return ET_ERR_UNSUPPORTED;  // not prefix-free
                c = static_cast<int16_t>(TW_LEAF0 - s);
            } else {
                if (c <= TW_LEAF0) return ET_ERR_UNSUPPORTED;  // a shorter code is a prefix of this one
                if (c == NONE) {
                    if (tree->n_int >= TW_MAX_NODES) return ET_ERR_UNSUPPORTED;
                    c = static_cast<int16_t>(tree->n_int++);
                }
                node = static_cast<uint32_t>(c);
            }
        }
    }
    for (uint32_t i = 0; i < 2 * tree->n_int; ++i)
        if (tree->child[i] == NONE) return ET_ERR_UNSUPPORTED;  // not a full tree (an encoder's always is)
    return ET_OK;
}

void tw_fill_table(const TwTree *tree, uint16_t *table) {
    const uint32_t rows = tw_rows(tree->n_int);
    for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t f = 0; f < 256; ++f) {
            uint32_t node = r < tree->n_int ? r : 0, n = 0, first = 0;
            const uint32_t skip = r < tree->n_int ? 0 : r - tree->n_int + 1;
            for (uint32_t i = skip; i < 8; ++i) {
                const int16_t c = tree->child[2 * node + ((f >> (7 - i)) & 1u)];
                if (c >= 0) {
                    node = static_cast<uint32_t>(c);
                } else {
                    if (n++ == 0) first = i;
                    node = 0;
                }
            }
            table[(r << 8) + f] = static_cast<uint16_t>(node | (n << TW_N_SHIFT) | (first << TW_OFF_SHIFT));
        }
}

}  // namespace et
