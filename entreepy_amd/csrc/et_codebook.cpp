// et_codebook.cpp -- host half of the path: code construction, header/dictionary
// writer and parser.  North star: "code-length assignment on the host (src/queue.zig
// tree build unchanged)"; this file reproduces the reference's results exactly but is
// organised around flat arrays instead of the reference's node pool + ring queues.
//
// Reference behaviour reproduced (all /root/reference/src):
//   encode.zig:54-79    leaf order = (count asc, byte asc); book_index u8 saturation
//   encode.zig:102-138  two-queue merge, tie -> leaf queue, left = first pick
//   encode.zig:141-214  code = path bits in a u32 (high bits fall off), length u8
//   encode.zig:259-299  header + bit-packed dictionary
//   decode.zig:34-141   header + dictionary parse, body offset
#include "entreepy_hip.h"

#include <algorithm>
#include <cstring>

namespace {

struct MergeNode {
    uint64_t weight;
    int16_t left, right;  // node ids; -1 for leaves
    int16_t symbol;       // leaf symbol, -1 for internal nodes
};

// MSB-first bit sink over a caller buffer (std.io.bitWriter(.big) semantics,
// encode.zig:256-257), 64-bit staging instead of one call per bit.
class BitSink {
public:
    BitSink(uint8_t *buf, size_t cap) : buf_(buf), cap_(cap) {}
    void put(uint64_t value, unsigned nbits) {  // low nbits of value, MSB first; nbits <= 32
        if (nbits == 0) return;
        value &= (nbits >= 64) ? ~0ull : ((1ull << nbits) - 1);
        acc_ = (acc_ << nbits) | value;
        fill_ += nbits;
        while (fill_ >= 8) {
            fill_ -= 8;
            emit(static_cast<uint8_t>(acc_ >> fill_));
        }
    }
    void pad_to_byte() {  // flushBits(): zero fill (encode.zig:298)
        if (fill_) {
            emit(static_cast<uint8_t>(acc_ << (8 - fill_)));
            fill_ = 0;
        }
        acc_ = 0;
    }
    size_t bytes() const { return pos_; }
    bool overflowed() const { return overflow_; }

private:
    void emit(uint8_t b) {
        if (pos_ < cap_) buf_[pos_] = b; else overflow_ = true;
        ++pos_;
    }
    uint8_t *buf_;
    size_t cap_;
    size_t pos_ = 0;
    uint64_t acc_ = 0;
    unsigned fill_ = 0;
    bool overflow_ = false;
};

// Bits the reference emits for one code, in order: for j = len..1 the bit
// (data >> ((j-1) & 31)) & 1 (encode.zig:291-295).  For len <= 32 that is the low
// `len` bits of data; beyond 32 the u5 truncation makes the low 32 bits repeat.
void put_code(BitSink &sink, uint32_t data, unsigned len) {
    if (len == 0) return;
    unsigned head = ((len - 1) & 31u) + 1;  // bits (len-1)&31 .. 0 come first
    sink.put(data, head);
    for (unsigned rest = len - head; rest > 0; rest -= 32) sink.put(data, 32);
}

class BitSource {
public:
    BitSource(const uint8_t *buf, size_t len) : buf_(buf), nbits_(static_cast<uint64_t>(len) * 8) {}
    bool get(unsigned nbits, uint64_t &value) {  // nbits <= 64
        if (pos_ + nbits > nbits_) return false;
        uint64_t v = 0;
        for (unsigned k = 0; k < nbits; ++k, ++pos_) v = (v << 1) | ((buf_[pos_ >> 3] >> (7 - (pos_ & 7))) & 1u);
        value = v;
        return true;
    }
    uint64_t pos() const { return pos_; }

private:
    const uint8_t *buf_;
    uint64_t nbits_;
    uint64_t pos_ = 0;
};

}  // namespace

// (here, with the rest of the plain host code: every host of the C ABI's status codes links this file)
extern "C" size_t et_encode_bound(size_t n) { return (n + 7200 + 15) & ~static_cast<size_t>(15); }  // encode.zig:253-254

extern "C" const char *et_strerror(int status) {
    switch (status) {
        case ET_OK: return "ok";
        case ET_ERR_EMPTY: return "empty input (error.QueueEmpty)";
        case ET_ERR_NOMEM: return "out of memory";
        case ET_ERR_CAP: return "output buffer too small";
        case ET_ERR_FORMAT: return "malformed .et stream";
        case ET_ERR_HIP: return "HIP runtime error";
        case ET_ERR_ARG: return "invalid argument";
        case ET_ERR_UNSUPPORTED: return "unsupported stream (code length > 32)";
        case ET_ERR_IO: return "file read/write error";
        case ET_ERR_RCCL: return "RCCL / exchange failure";
        default: return "unknown status";
    }
}

extern "C" int et_build_codebook(const uint64_t hist[256], et_codebook *cb) {
    if (!hist || !cb) return ET_ERR_ARG;
    std::memset(cb, 0, sizeof(*cb));

    // Leaves in (count, byte) ascending order (encode.zig:54-74).
    int order[256];
    int n_present = 0;
    for (int s = 0; s < 256; ++s)
        if (hist[s] >= 1) order[n_present++] = s;
    std::stable_sort(order, order + n_present, [&](int a, int b) { return hist[a] < hist[b]; });
    // book_index is a u8 that stops at 255 (encode.zig:57,70,79): with all 256 byte
    // values present only the first 255 of the order become leaves; the last one
    // (most frequent, highest byte on ties) keeps length 0.
    const int n_leaves = std::min(n_present, 255);
    if (n_leaves == 0) return ET_ERR_EMPTY;  // both queues empty -> error.QueueEmpty (encode.zig:137-138)

    MergeNode nodes[511];
    for (int i = 0; i < n_leaves; ++i) nodes[i] = MergeNode{hist[order[i]], -1, -1, static_cast<int16_t>(order[i])};

    // Two-queue merge (encode.zig:102-135).  Leaves are consumed through `lf`,
    // internal nodes are created in non-decreasing weight order and consumed through
    // `in`, so both "queues" are just cursors into `nodes`.
    int lf = 0, in = n_leaves, n_nodes = n_leaves;
    auto take = [&]() -> int {
        const bool leaf_avail = lf < n_leaves, int_avail = in < n_nodes;
        if (!int_avail) return lf++;
        if (!leaf_avail) return in++;
        return (nodes[lf].weight <= nodes[in].weight) ? lf++ : in++;  // tie -> leaf (encode.zig:113)
    };
    while ((n_leaves - lf) + (n_nodes - in) > 1) {
        const int a = take();
        const int b = take();
        nodes[n_nodes] = MergeNode{nodes[a].weight + nodes[b].weight, static_cast<int16_t>(a), static_cast<int16_t>(b), -1};
        ++n_nodes;
    }
    const int root = (lf < n_leaves) ? lf : in;  // encode.zig:137-138

    // Codes top-down.  A parent is always created after its children, so walking
    // node ids downwards visits every parent before its children.  left = <<1|0,
    // right = <<1|1 in a u32, length in a u8 (encode.zig:181-197).
    uint32_t path[511];
    uint8_t depth[511];
    path[root] = 0;
    depth[root] = 0;
    for (int id = root; id >= n_leaves; --id) {
        const MergeNode &nd = nodes[id];
        path[nd.left] = path[id] << 1;
        path[nd.right] = (path[id] << 1) | 1u;
        depth[nd.left] = depth[nd.right] = static_cast<uint8_t>(depth[id] + 1);
    }
    uint32_t min_len = 0, max_len = 0, coded = 0;
    for (int i = 0; i < n_leaves; ++i) {
        const int s = nodes[i].symbol;
        cb->data[s] = path[i];
        cb->length[s] = depth[i];
        if (depth[i]) {
            ++coded;
            min_len = (min_len == 0) ? depth[i] : std::min<uint32_t>(min_len, depth[i]);
            max_len = std::max<uint32_t>(max_len, depth[i]);
        }
    }
    cb->n_coded = coded;
    cb->min_length = min_len;
    cb->max_length = max_len;

    // -d dump order: stack DFS that pushes right then left (encode.zig:171-213),
    // i.e. pre-order with the left subtree first.
    int stack[512];
    int top = 0, out = 0;
    stack[top++] = root;
    while (top) {
        const MergeNode &nd = nodes[stack[--top]];
        if (nd.symbol >= 0) {
            cb->dfs_order[out++] = static_cast<uint8_t>(nd.symbol);
        } else {
            stack[top++] = nd.right;
            stack[top++] = nd.left;
        }
    }
    return ET_OK;
}

extern "C" int et_write_header(const et_codebook *cb, uint64_t text_len, uint8_t *out, size_t cap, size_t *header_len) {
    if (!cb || !out || !header_len) return ET_ERR_ARG;
    BitSink sink(out, cap);
    sink.put(0xe7c0de, 24);  // encode.zig:262
    sink.put(0x01, 8);       // encode.zig:266
    uint32_t d = 0;          // encode.zig:270-275
    for (int s = 0; s < 256; ++s) d += cb->length[s] > 0;
    if (d > 0) --d;
    sink.put(d, 8);
    sink.put(text_len & 0xffffffffull, 32);  // encode.zig:279: 32-bit field, wraps at 4 GiB
    for (int s = 0; s < 256; ++s) {          // encode.zig:285-297
        if (!cb->length[s]) continue;
        sink.put(static_cast<uint64_t>(s), 8);
        sink.put(cb->length[s], 8);
        put_code(sink, cb->data[s], cb->length[s]);
    }
    sink.pad_to_byte();  // encode.zig:298
    if (sink.overflowed()) return ET_ERR_CAP;
    *header_len = sink.bytes();
    return ET_OK;
}

extern "C" int et_plan_shards(const uint64_t *hists, uint32_t world, et_codebook *cb, uint8_t *header, size_t header_cap,
                              size_t *header_len, uint64_t *start_bits) {
    if (!hists || !world || !cb || !header || !header_len || !start_bits) return ET_ERR_ARG;
    uint64_t total[256] = {0}, n = 0;
    for (uint32_t r = 0; r < world; ++r)
        for (int s = 0; s < 256; ++s) total[s] += hists[static_cast<size_t>(r) * 256 + s];
    for (int s = 0; s < 256; ++s) n += total[s];
    int rc = et_build_codebook(total, cb);
    if (rc != ET_OK) return rc;
    rc = et_write_header(cb, n, header, header_cap, header_len);
    if (rc != ET_OK) return rc;
    start_bits[0] = 8 * static_cast<uint64_t>(*header_len);
    for (uint32_t r = 0; r < world; ++r) {
        uint64_t bits = 0;
        et_codebook_bits(cb, hists + static_cast<size_t>(r) * 256, &bits);
        start_bits[r + 1] = start_bits[r] + bits;
    }
    return ET_OK;
}

extern "C" int et_codebook_bits(const et_codebook *cb, const uint64_t hist[256], uint64_t *bits) {
    if (!cb || !hist || !bits) return ET_ERR_ARG;
    uint64_t total = 0;
    for (int s = 0; s < 256; ++s) total += hist[s] * cb->length[s];
    *bits = total;
    return ET_OK;
}

// encode.zig:221-247, the -d self-check, loop for loop -- including its k = 0 round, which compares the bit
// ABOVE each code (bit `length`, index truncated to u5).
extern "C" int et_prefix_collisions(const et_codebook *cb, uint8_t *pairs, size_t cap_pairs, size_t *n_pairs) {
    if (!cb || !n_pairs || (cap_pairs && !pairs)) return ET_ERR_ARG;
    size_t n = 0;
    for (unsigned i = 0; i < 256; ++i)
        for (unsigned j = 0; j < 256; ++j) {
            if (cb->length[i] == 0 || cb->length[j] == 0 || i == j) continue;
            bool is_prefix = true;
            const unsigned shorter = cb->length[i] < cb->length[j] ? cb->length[i] : cb->length[j];
            for (unsigned k = 0; k <= shorter; ++k) {
                const unsigned b1 = (cb->data[i] >> ((cb->length[i] - k) & 31u)) & 1u;
                const unsigned b2 = (cb->data[j] >> ((cb->length[j] - k) & 31u)) & 1u;
                if (b1 != b2) {
                    is_prefix = false;
                    break;
                }
            }
            if (is_prefix) {
                if (n < cap_pairs) {
                    pairs[2 * n] = static_cast<uint8_t>(i);
                    pairs[2 * n + 1] = static_cast<uint8_t>(j);
                }
                ++n;
            }
        }
    *n_pairs = n;
    return ET_OK;
}

extern "C" int et_check_magic(const uint8_t first4[4], const char **why) {
    const char *reason = nullptr;
    if (!first4) return ET_ERR_ARG;
    if (first4[0] != 0xe7 || first4[1] != 0xc0 || first4[2] != 0xde) reason = "not an .et file (magic e7 c0 de missing)";
    else if (first4[3] != 0x01) reason = "unknown .et format version";
    if (why) *why = reason;
    return reason ? ET_ERR_FORMAT : ET_OK;
}

extern "C" int et_decoded_size(const uint8_t *compressed, size_t len, size_t *n_symbols) {
    if (!compressed || !n_symbols) return ET_ERR_ARG;
    if (len < 5) return ET_ERR_FORMAT;
    *n_symbols = (static_cast<size_t>(compressed[1]) << 24) | (static_cast<size_t>(compressed[2]) << 16) |
                 (static_cast<size_t>(compressed[3]) << 8) | compressed[4];  // decode.zig:36-42
    return ET_OK;
}

extern "C" int et_parse_header(const uint8_t *compressed, size_t len, et_codebook *cb, uint64_t *n_symbols, size_t *body_offset) {
    if (!compressed || !cb || !n_symbols || !body_offset) return ET_ERR_ARG;
    if (len < 5) return ET_ERR_FORMAT;
    std::memset(cb, 0, sizeof(*cb));
    size_t n = 0;
    et_decoded_size(compressed, len, &n);
    *n_symbols = n;

    // decode.zig:34: D + 1 entries follow -- except that a stream with no coded
    // symbol at all also stores D = 0 (encode.zig:270-275) and carries no dictionary;
    // the reference tells the two apart only by running out of bytes (decode.zig:66).
    const unsigned entries = static_cast<unsigned>(compressed[0]) + 1;
    BitSource src(compressed + 5, len - 5);
    uint32_t min_len = 0, max_len = 0, coded = 0;
    for (unsigned e = 0; e < entries; ++e) {
        uint64_t sym, clen, code;
        if (!src.get(8, sym)) {
            if (e == 0 && len == 5) break;  // header-only stream: single-symbol input
            return ET_ERR_FORMAT;
        }
        if (!src.get(8, clen)) return ET_ERR_FORMAT;
        if (clen == 0) return ET_ERR_FORMAT;
        if (clen > 32) return ET_ERR_UNSUPPORTED;  // reference entry is [32]u8 (decode.zig:49,124)
        if (!src.get(static_cast<unsigned>(clen), code)) return ET_ERR_FORMAT;
        if (cb->length[sym]) return ET_ERR_FORMAT;  // duplicate symbol
        cb->data[sym] = static_cast<uint32_t>(code);
        cb->length[sym] = static_cast<uint8_t>(clen);
        cb->dfs_order[coded++] = static_cast<uint8_t>(sym);
        min_len = (min_len == 0) ? static_cast<uint32_t>(clen) : std::min<uint32_t>(min_len, static_cast<uint32_t>(clen));
        max_len = std::max<uint32_t>(max_len, static_cast<uint32_t>(clen));
    }
    cb->n_coded = coded;
    cb->min_length = min_len;
    cb->max_length = max_len;
    *body_offset = 5 + static_cast<size_t>((src.pos() + 7) / 8);  // decode.zig:156

    // The decoder needs a prefix-free set: sort left-aligned codes and make sure no
    // code's interval [c, c + 2^(32-len)) reaches the next one.
    struct Iv { uint64_t lo, hi; };
    Iv iv[256];
    unsigned k = 0;
    for (int s = 0; s < 256; ++s)
        if (cb->length[s]) {
            const uint64_t lo = static_cast<uint64_t>(cb->data[s]) << (32 - cb->length[s]);
            iv[k++] = Iv{lo, lo + (1ull << (32 - cb->length[s]))};
        }
    std::sort(iv, iv + k, [](const Iv &a, const Iv &b) { return a.lo < b.lo; });
    for (unsigned i = 0; i + 1 < k; ++i)
        if (iv[i].hi > iv[i + 1].lo) return ET_ERR_FORMAT;
    return ET_OK;
}
