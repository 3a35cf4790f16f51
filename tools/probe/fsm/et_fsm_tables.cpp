// et_fsm_tables.cpp -- host side of the fixed-rate decode tables (formats: et_fsm.h): the code
// tree, the choice of step widths, and reference fills the device-built tables are tested against.
#include "et_fsm.h"

#include <cstring>

namespace et {

int fsm_build_tree(const et_codebook *cb, FsmTree *tree) {
    tree->n_int = 1;
    tree->child[0] = tree->child[1] = FSM_NONE;
    for (int s = 0; s < 256; ++s) {
        const uint32_t len = cb->length[s];
        if (!len) continue;
        if (len > 32) return ET_ERR_UNSUPPORTED;
        uint32_t node = 0;
        for (uint32_t i = len; i-- > 0;) {  // first stream bit = bit len-1 of the code (encode.zig:293,311)
            const uint32_t bit = (cb->data[s] >> i) & 1u;
            int32_t &c = tree->child[2 * node + bit];
            if (i == 0) {
                c = FSM_LEAF0 - s;
            } else {
                if (c < 0) {  // (a leaf here would mean the set is not prefix-free: et_parse_header rejects that)
                    if (tree->n_int >= FSM_MAX_NODES) return ET_ERR_UNSUPPORTED;
                    c = static_cast<int32_t>(tree->n_int);
                    tree->child[2 * tree->n_int] = tree->child[2 * tree->n_int + 1] = FSM_NONE;
                    ++tree->n_int;
                }
                node = static_cast<uint32_t>(c);
            }
        }
    }
    return ET_OK;
}

namespace {

// One table entry: walk the k bits of `field` (first bit = bit k-1) from `row`.
struct Step {
    uint32_t next, n;
    uint8_t sym[32];
};

Step walk_field(const FsmTree *tree, uint32_t k, uint32_t row, uint32_t field) {
    Step st;
    st.n = 0;
    uint32_t node = row < tree->n_int ? row : 0;
    const uint32_t skip = row < tree->n_int ? 0 : row - tree->n_int + 1;  // entry rows S_1 .. S_{k-1}
    for (uint32_t i = skip; i < k; ++i) {
        const int32_t c = tree->child[2 * node + ((field >> (k - 1 - i)) & 1u)];
        if (c >= 0) {
            node = static_cast<uint32_t>(c);
        } else {
            if (c <= FSM_LEAF0) st.sym[st.n++] = static_cast<uint8_t>(FSM_LEAF0 - c);
            node = 0;
        }
    }
    st.next = node;
    return st;
}

}  // namespace

uint32_t fsm_max_symbols_per_entry(const FsmTree *tree, uint32_t k) {
    uint32_t most = 0;
    const uint32_t rows = fsm_rows(tree->n_int, k);
    for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t f = 0; f < (1u << k); ++f) {
            const uint32_t n = walk_field(tree, k, r, f).n;
            if (n > most) most = n;
        }
    return most;
}

void fsm_fill_sync(const FsmTree *tree, uint32_t k, uint16_t *table) {
    const uint32_t rows = fsm_rows(tree->n_int, k);
    for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t f = 0; f < (1u << k); ++f) {
            const Step st = walk_field(tree, k, r, f);
            table[(static_cast<size_t>(r) << k) + f] = static_cast<uint16_t>(st.next | (st.n << FSM_SYNC_N_SHIFT));
        }
}

void fsm_fill_write(const FsmTree *tree, uint32_t k, uint32_t *table) {
    const uint32_t rows = fsm_rows(tree->n_int, k);
    for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t f = 0; f < (1u << k); ++f) {
            const Step st = walk_field(tree, k, r, f);  // (n <= 2: fsm_plan)
            uint32_t e = (st.next << fsm_write_row_shift(k)) | (8u * st.n);
            if (st.n > 0) e |= static_cast<uint32_t>(st.sym[0]) << 16;
            if (st.n > 1) e |= static_cast<uint32_t>(st.sym[1]) << 24;
            table[(static_cast<size_t>(r) << (fsm_write_row_shift(k) - 2)) + f] = e;
        }
}

int fsm_plan(const et_codebook *cb, const FsmTree *tree, uint32_t lds_budget_sync, uint32_t lds_budget_write, FsmPlan *plan) {
    std::memset(plan, 0, sizeof *plan);
    plan->n_int = tree->n_int;
    const uint32_t sync_widths[3] = {8, 4, 2}, write_widths[3] = {6, 4, 2};
    for (uint32_t k : sync_widths) {
        const uint64_t rows = fsm_rows(tree->n_int, k);
        if (rows <= FSM_MAX_ROWS && (rows << k) * sizeof(uint16_t) <= lds_budget_sync) {
            plan->k_sync = k;
            plan->rows_sync = static_cast<uint32_t>(rows);
            break;
        }
    }
    // Two symbols per write entry: completions inside k bits are at least min_length apart and the
    // first needs one bit, so 1 + 2 * min_length > k rules a third one out; a corrupted dictionary
    // (missing branches count as boundaries too, but carry no symbol) is measured entry by entry.
    for (uint32_t k : write_widths) {
        const uint64_t rows = fsm_rows(tree->n_int, k);
        if (tree->n_int > fsm_write_max_targets(k) || (rows << fsm_write_row_shift(k)) > lds_budget_write) continue;
        const bool two = 1 + 2 * cb->min_length > k || fsm_max_symbols_per_entry(tree, k) <= 2;
        if (two) {
            plan->k_write = k;
            plan->rows_write = static_cast<uint32_t>(rows);
            break;
        }
    }
    if (!plan->k_sync || !plan->k_write) return ET_ERR_UNSUPPORTED;
    const uint32_t min_len = cb->min_length ? cb->min_length : 1;
    plan->max_per_lane = 256 / min_len + 2;
    return ET_OK;
}

}  // namespace et
