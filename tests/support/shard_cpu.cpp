// shard_cpu.cpp -- TEST INFRASTRUCTURE: a CPU stand-in for ONE rank's GPU, so that the product's group sequence
// (entreepy_amd/csrc/et_shard_seq.cpp, compiled into this library unchanged) runs in the CPU test suite over gloo.
// "Device" memory is host memory; every compute step is the oracle's (oracle/et_oracle.c).  Nothing of the product
// links or loads this file: libentreepy_hip.so's groups only ever sit on an et_ctx (csrc/et_shard_hip.cpp).
#include <unistd.h>

#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "et_oracle.h"
#include "et_shard_seq.h"

#include <atomic>

namespace {

std::atomic<int> g_fail_patches{0};  // tests: the next so many patch_word calls fail (et_cpu_fail_next_patches)

struct OracleBackend : et_shard::Backend {
    std::string err;
    uint64_t counts[256] = {};
    bool have_counts = false;
    // the range synchronised last
    std::vector<uint8_t> syms;
    struct {
        const et_codebook *cb_copy = nullptr;
        et_codebook cb;
        const uint8_t *range = nullptr;
        size_t range_bytes = 0, tail_bytes = 0;
        bool has_front = false, valid = false;
    } maps;

    const char *last_error() const override { return err.c_str(); }
    int bad(int rc, const char *what) {
        err = what;
        return rc;
    }

    int histogram_begin(const void *d_text, size_t n, void *d_row) override {
        et_oracle_histogram(static_cast<const uint8_t *>(d_text), n, counts);
        have_counts = true;
        if (d_row) std::memcpy(d_row, counts, sizeof counts);
        return ET_OK;
    }
    int histogram_host(uint64_t out[256]) override {
        if (!have_counts) return bad(ET_ERR_ARG, "no current histogram");
        std::memcpy(out, counts, sizeof counts);
        return ET_OK;
    }
    int histogram_known(const uint64_t c[256]) override { return std::memcmp(c, counts, sizeof counts) ? bad(ET_ERR_ARG, "not the counts of the last histogram") : ET_OK; }

    // a piece as et_encode_head_shard_device / et_encode_body_device leave it: every word it touches fully written
    int pack(const et_codebook *cb, const void *d_text, size_t n, void *d_out, size_t cap, const uint8_t *header, size_t header_len, uint64_t start_bit, uint64_t *end_bit) {
        et_oracle_dict d;
        std::memcpy(d.data, cb->data, sizeof d.data);
        std::memcpy(d.length, cb->length, sizeof d.length);
        const uint8_t *text = static_cast<const uint8_t *>(d_text);
        uint64_t bits = 0;
        for (size_t i = 0; i < n; ++i) bits += cb->length[text[i]];
        uint8_t *out = static_cast<uint8_t *>(d_out);
        if (bits == 0) {
            const size_t need = header_len ? ((header_len + 3) / 4) * 4 : 4;
            if (need > cap) return bad(ET_ERR_CAP, "body does not fit d_out");
            std::memset(out, 0, need);
            if (header_len) std::memcpy(out, header, header_len);
            *end_bit = start_bit;
            return ET_OK;
        }
        const uint64_t end = start_bit + bits, need = ((end + 31) / 32) * 4;
        if (need > cap) return bad(ET_ERR_CAP, "body does not fit d_out");
        std::memset(out, 0, need);
        if (header_len) std::memcpy(out, header, header_len);
        const int64_t got = et_oracle_pack_body(&d, text, n, out, need, start_bit);
        if (got < 0 || static_cast<uint64_t>(got) != end) return bad(ET_ERR_HIP, "oracle pack failed");
        *end_bit = end;
        return ET_OK;
    }
    int encode_head(const et_codebook *cb, const void *d_text, size_t n, void *d_out, size_t cap, const uint8_t *header, size_t header_len, uint64_t *end_bit) override {
        return pack(cb, d_text, n, d_out, cap, header, header_len, 8 * header_len, end_bit);
    }
    int encode_body(const et_codebook *cb, const void *d_text, size_t n, void *d_out, size_t cap, uint64_t start_bit, uint64_t *end_bit) override {
        return pack(cb, d_text, n, d_out, cap, nullptr, 0, start_bit, end_bit);
    }
    int read_first_last(const void *d_out, uint64_t n_words, uint32_t fl[2]) override {
        std::memcpy(&fl[0], d_out, 4);
        std::memcpy(&fl[1], static_cast<const uint8_t *>(d_out) + (n_words - 1) * 4, 4);
        return ET_OK;
    }
    int patch_word(void *d_out, uint64_t word, uint32_t value) override {
        if (g_fail_patches.load() > 0 && g_fail_patches.fetch_sub(1) > 0) return bad(ET_ERR_HIP, "patch_word: injected failure");
        std::memcpy(static_cast<uint8_t *>(d_out) + word * 4, &value, 4);
        return ET_OK;
    }
    int drain() override { return ET_OK; }
    int to_fd(const void *d_src, size_t len, int fd, uint64_t off) override {
        const uint8_t *p = static_cast<const uint8_t *>(d_src);
        while (len) {
            const ssize_t w = ::pwrite(fd, p, len, static_cast<off_t>(off));
            if (w <= 0) return bad(ET_ERR_IO, "pwrite");
            p += w;
            off += static_cast<uint64_t>(w);
            len -= static_cast<size_t>(w);
        }
        return ET_OK;
    }
    int copy(void *d_dst, const void *d_src, size_t len) override {
        std::memcpy(d_dst, d_src, len);
        return ET_OK;
    }
    int read_head(const void *d_src, size_t len, uint8_t *host) override {
        std::memcpy(host, d_src, len);
        return ET_OK;
    }

    // ---- cold decode of a block range: a bit-serial walk (what et_decode_range_sync computes on the GPU) ----------
    struct Walker {
        std::unordered_map<uint64_t, int> table;  // len << 32 | code -> symbol
        uint32_t max_len = 0;
        const uint8_t *range;
        int64_t lo_bit, hi_bit;  // readable bits, relative to the range's first bit
        Walker(const et_codebook *cb, const uint8_t *r, size_t range_bytes, size_t tail_bytes, bool has_front) : range(r) {
            for (int s = 0; s < 256; ++s)
                if (cb->length[s]) {
                    table[(static_cast<uint64_t>(cb->length[s]) << 32) | cb->data[s]] = s;
                    if (cb->length[s] > max_len) max_len = cb->length[s];
                }
            lo_bit = has_front ? -128 : 0;
            hi_bit = static_cast<int64_t>(range_bytes + tail_bytes) * 8;
        }
        int bit(int64_t p) const { return (range[p >> 3] >> (7 - (p & 7))) & 1; }  // (p >> 3 floors: bytes in front of the range have negative indices)
        // -> length of the codeword at p (0: cut by the end of what is readable); *sym = -1: no symbol's code, one bit passed over
        uint32_t step(int64_t p, int *sym) const {
            uint64_t val = 0;
            for (uint32_t ln = 1; ln <= max_len; ++ln) {
                if (p + ln > hi_bit) return 0;
                val = (val << 1) | static_cast<uint64_t>(bit(p + ln - 1));
                auto it = table.find((static_cast<uint64_t>(ln) << 32) | val);
                if (it != table.end()) {
                    *sym = it->second;
                    return ln;
                }
            }
            *sym = -1;
            return 1;
        }
    };
    int walk_range(const et_codebook *cb, const uint8_t *range, size_t range_bytes, size_t tail_bytes, bool has_front, int32_t in_start_bit, et_range_info *info, bool keep) {
        if (cb->max_length > 32) return bad(ET_ERR_UNSUPPORTED, "code length > 32");
        const Walker w(cb, range, range_bytes, tail_bytes, has_front);
        int64_t p;
        int sym = 0;
        if (in_start_bit >= 0) {
            p = in_start_bit;
        } else {  // run in over the 128 bits in front of the range
            if (!has_front) return bad(ET_ERR_ARG, "an unknown start needs the 16 bytes in front of the range");
            p = -128;
            while (p < 0) {
                const uint32_t ln = w.step(p, &sym);
                if (!ln) break;
                p += ln;
            }
        }
        const int64_t range_bits = static_cast<int64_t>(range_bytes) * 8;
        info->start_bit = static_cast<uint32_t>(p);
        uint64_t n = 0;
        if (keep) syms.clear();
        while (p < range_bits) {
            const uint32_t ln = w.step(p, &sym);
            if (!ln) {
                p = range_bits;
                break;
            }
            if (sym >= 0) {
                ++n;
                if (keep) syms.push_back(static_cast<uint8_t>(sym));
            }
            p += ln;
        }
        info->exit_bit = static_cast<uint32_t>(p - range_bits);
        info->n_symbols = n;
        info->sweeps = 1;
        info->reserved = 0;
        return ET_OK;
    }
    int range_sync(const et_codebook *cb, const void *d_range, size_t range_bytes, size_t tail_bytes, int has_front, int32_t in_start_bit, et_range_info *info) override {
        return walk_range(cb, static_cast<const uint8_t *>(d_range), range_bytes, tail_bytes, has_front != 0, in_start_bit, info, true);
    }
    int range_maps(const et_codebook *cb, const void *d_range, size_t range_bytes, size_t tail_bytes, int32_t in_start_bit, uint8_t map[32], uint32_t *n_starts) override {
        maps.cb = *cb;
        maps.range = static_cast<const uint8_t *>(d_range);
        maps.range_bytes = range_bytes;
        maps.tail_bytes = tail_bytes;
        maps.valid = true;
        *n_starts = cb->max_length;
        for (uint32_t p = 0; p < 32; ++p) {
            map[p] = static_cast<uint8_t>(p);
            if (in_start_bit < 0 && p >= cb->max_length) continue;
            et_range_info i;
            const int rc = walk_range(cb, maps.range, range_bytes, tail_bytes, false, in_start_bit >= 0 ? in_start_bit : static_cast<int32_t>(p), &i, false);
            if (rc != ET_OK) return rc;
            map[p] = static_cast<uint8_t>(i.exit_bit);
        }
        return ET_OK;
    }
    int range_resolve(uint32_t in_start_bit, et_range_info *info) override {
        if (!maps.valid) return bad(ET_ERR_ARG, "et_decode_range_resolve needs et_decode_range_maps first");
        return walk_range(&maps.cb, maps.range, maps.range_bytes, maps.tail_bytes, false, static_cast<int32_t>(in_start_bit), info, true);
    }
    int range_write(uint64_t max_symbols, void *d_out, size_t cap, size_t *out_len) override {
        const size_t n = syms.size() < max_symbols ? syms.size() : static_cast<size_t>(max_symbols);
        if (n > cap) return bad(ET_ERR_CAP, "output buffer too small");
        std::memcpy(d_out, syms.data(), n);
        *out_len = n;
        return ET_OK;
    }
};

}  // namespace

// Test hook: the next n patch_word calls of any backend in this process fail with ET_ERR_HIP (a failure BEHIND the merge's exchange).
extern "C" void et_cpu_fail_next_patches(int n) { g_fail_patches.store(n); }

// The product's et_group_create with a stand-in where the et_ctx would be (ctx is ignored).
extern "C" int et_group_create(et_ctx *, int rank, int world, et_allgather_fn allgather, void *user, et_group **out) {
    if (!out || (world > 1 && !allgather)) return ET_ERR_ARG;
    return et_shard::group_new(new OracleBackend(), new et_shard::CallbackExchange(allgather, user), rank, world, out);
}
