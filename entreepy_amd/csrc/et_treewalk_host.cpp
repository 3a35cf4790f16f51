// et_treewalk_host.cpp -- host part of the tree walk (et_treewalk.h): the code table as a tree, and a
// reference fill of the walk's table.  Plain C++ (also built by tests/test_sanitizers.py with g++).
#include "et_treewalk.h"

#include <cstring>

namespace et {

int tw_build_tree(const et_codebook *cb, TwTree *tree, bool complete) {
    if (cb->n_coded < (complete ? 1u : 2u)) return ET_ERR_UNSUPPORTED;
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
        if (tree->child[i] == NONE) {
            if (!complete) return ET_ERR_UNSUPPORTED;  // not a full tree (an encoder's always is)
            tree->child[i] = TW_LEAF0;  // a code that no symbol has: it decodes as byte 0
        }
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

// true: the tables fit with sub-tables of at most `cap` index bits
static bool chain_try(const TwTree *tree, const uint8_t *height, uint32_t cap, ChainPlan *plan) {
    for (uint32_t i = 0; i < TW_MAX_NODES; ++i) plan->table_of[i] = -1;
    plan->n_tables = 1;
    plan->n_entries = 1u << CH_ROOT_BITS;
    plan->sub_bits = cap;
    plan->pad_ = 0;
    plan->tab[0] = ChainTable{0, static_cast<uint8_t>(CH_ROOT_BITS), 0, 0};
    plan->table_of[0] = 0;
    for (uint32_t t = 0; t < plan->n_tables; ++t) {
        // internal nodes exactly tab[t].bits levels below tab[t].node
        struct Item {
            uint16_t node, depth;
        } stack[2 * 32 + 2];
        uint32_t sp = 0;
        stack[sp++] = Item{plan->tab[t].node, 0};
        while (sp) {
            const Item it = stack[--sp];
            if (it.depth == plan->tab[t].bits) {
                if (plan->table_of[it.node] >= 0) continue;
                if (plan->n_tables >= CH_MAX_TABLES) return false;
                const uint32_t bits = height[it.node] < cap ? height[it.node] : cap;
                if (plan->n_entries + (1u << bits) > CH_MAX_ENTRIES) return false;
                plan->table_of[it.node] = static_cast<int16_t>(plan->n_tables);
                plan->tab[plan->n_tables++] = ChainTable{it.node, static_cast<uint8_t>(bits), 0, plan->n_entries};
                plan->n_entries += 1u << bits;
                continue;
            }
            for (uint32_t bit = 0; bit < 2; ++bit) {
                const int16_t c = tree->child[2 * it.node + bit];
                if (c >= 0) stack[sp++] = Item{static_cast<uint16_t>(c), static_cast<uint16_t>(it.depth + 1)};
            }
        }
    }
    return true;
}

void tw_chain_plan(const TwTree *tree, ChainPlan *plan) {
    // a child's number is larger than its parent's (tw_build_tree numbers nodes as it first walks through them)
    uint8_t height[TW_MAX_NODES] = {};
    for (uint32_t node = tree->n_int; node-- > 0;) {
        uint32_t h = 0;
        for (uint32_t bit = 0; bit < 2; ++bit) {
            const int16_t c = tree->child[2 * node + bit];
            const uint32_t hc = c >= 0 ? height[c] : 0u;
            if (hc > h) h = hc;
        }
        height[node] = static_cast<uint8_t>(h + 1);
    }
    for (uint32_t cap = CH_SUB_BITS_MAX; cap >= 1; --cap)
        if (chain_try(tree, height, cap, plan)) return;
    // (cap 1: at most one two-entry table per internal node -- always fits)
}

void tw_chain_fill(const TwTree *tree, const ChainPlan *plan, uint64_t *table) {
    for (uint32_t t = 0; t < plan->n_tables; ++t)
        for (uint32_t i = 0; i < (1u << plan->tab[t].bits); ++i) table[plan->tab[t].first + i] = tw_chain_entry(tree, plan, t, i);
}

}  // namespace et
