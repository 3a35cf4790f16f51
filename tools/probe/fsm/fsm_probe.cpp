// fsm_probe.cpp -- standalone check + timing of the fixed-rate decode walks (et_fsm.hip) against a
// bit-serial tree walk on the host.  Not part of the library; links the kernels' object directly.
//   fsm_probe <text-file for the byte distribution> [MiB of text = 256] [alphabet: 0 = the file's]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "et_fsm_kernels.h"
#include "entreepy_hip.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(2);                                                               \
        }                                                                          \
    } while (0)

static uint64_t rng_state = 0x5EED0004ull;
static inline uint64_t next64() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const size_t n = (argc > 2 ? atol(argv[2]) : 256) << 20;
    const int alphabet = argc > 3 ? atoi(argv[3]) : 0;
    // byte distribution -> 16-bit lookup
    std::vector<double> p(256, 0.0);
    if (alphabet > 0) {
        for (int i = 0; i < alphabet; ++i) p[(i + 1) & 255] = 1.0 / (1.0 + i * (alphabet > 100 ? 0.15 : 0.0));  // flat, or Zipf-ish for big alphabets
    } else {
        FILE *f = fopen(argv[1], "rb");
        if (!f) return 1;
        int c;
        while ((c = fgetc(f)) != EOF) p[c] += 1;
        fclose(f);
    }
    double sum = 0;
    for (double v : p) sum += v;
    std::vector<uint8_t> lut(65536);
    {
        double acc = 0;
        int pos = 0;
        for (int s = 0; s < 256; ++s) {
            acc += p[s] / sum * 65536.0;
            while (pos < 65536 && pos < acc) lut[pos++] = static_cast<uint8_t>(s);
        }
        while (pos < 65536) lut[pos++] = lut[pos - 1];
    }
    std::vector<uint8_t> text(n);
    for (size_t i = 0; i + 4 <= n; i += 4) {
        const uint64_t r = next64();
        text[i] = lut[r & 0xffff];
        text[i + 1] = lut[(r >> 16) & 0xffff];
        text[i + 2] = lut[(r >> 32) & 0xffff];
        text[i + 3] = lut[(r >> 48) & 0xffff];
    }

    et_ctx *ctx = nullptr;
    if (et_ctx_create(0, &ctx) != ET_OK) return 2;
    uint8_t *d_text, *d_et, *d_out;
    const size_t cap = et_encode_bound(n);
    CK(hipMalloc(&d_text, n + 64));
    CK(hipMalloc(&d_et, cap + 64));
    CK(hipMalloc(&d_out, n + 64));
    CK(hipMemcpy(d_text, text.data(), n, hipMemcpyHostToDevice));
    size_t et_len = 0;
    if (et_encode_device(ctx, d_text, n, d_et, cap, &et_len) != ET_OK) {
        fprintf(stderr, "encode: %s\n", et_last_error(ctx));
        return 2;
    }
    CK(hipDeviceSynchronize());
    std::vector<uint8_t> et(et_len);
    CK(hipMemcpy(et.data(), d_et, et_len, hipMemcpyDeviceToHost));
    et_codebook cb;
    uint64_t n_symbols = 0;
    size_t body_off = 0;
    if (et_parse_header(et.data() + 4, et_len - 4, &cb, &n_symbols, &body_off) != ET_OK) return 3;
    body_off += 4;
    printf("text %zu B -> .et %zu B, body at %zu, %u symbols coded, lengths %u..%u\n", n, et_len, body_off, cb.n_coded, cb.min_length, cb.max_length);

    static et::FsmTree tree;
    et::FsmPlan plan;
    if (et::fsm_build_tree(&cb, &tree) != ET_OK || et::fsm_plan(&cb, &tree, 136 * 1024, 72 * 1024, &plan) != ET_OK) return 4;
    printf("tree: %u internal nodes; sync K=%u (%u rows, %u KiB), write K=%u (%u rows, %u KiB), <= %u symbols per lane\n", plan.n_int, plan.k_sync,
           plan.rows_sync, (plan.rows_sync << plan.k_sync) * 2 / 1024, plan.k_write, plan.rows_write, et::fsm_write_table_words(plan.rows_write, plan.k_write) * 4 / 1024,
           plan.max_per_lane);
    std::vector<uint16_t> h_sync((static_cast<size_t>(plan.rows_sync) << plan.k_sync) + 8);
    std::vector<uint32_t> h_write(et::fsm_write_table_words(plan.rows_write, plan.k_write), 0u);
    et::fsm_fill_sync(&tree, plan.k_sync, h_sync.data());
    et::fsm_fill_write(&tree, plan.k_write, h_write.data());
    uint16_t *d_sync;
    uint32_t *d_write;
    CK(hipMalloc(&d_sync, h_sync.size() * 2 + 64));
    CK(hipMalloc(&d_write, h_write.size() * 4 + 64));
    CK(hipMemcpy(d_sync, h_sync.data(), h_sync.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_write, h_write.data(), h_write.size() * 4, hipMemcpyHostToDevice));

    // the stream as the kernels see it
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_et + body_off);
    const uint32_t *words = reinterpret_cast<const uint32_t *>(a & ~static_cast<uintptr_t>(3));
    const uint32_t first_bit = static_cast<uint32_t>(a & 3) * 8;
    const uint64_t n_bytes = (a & 3) + (et_len - body_off);
    const uint64_t n_subs = (n_bytes * 8 + 255) / 256;
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + 255) / 256);
    const uint8_t *hb = et.data() + body_off - (a & 3);  // host view from the aligned base

    // host: bit-serial walk
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> want(n_subs, 0), want_off(n_subs, 0);
    std::vector<uint32_t> want_exit(n_blocks, 0), want_count(n_blocks, 0);
    {
        uint32_t node = 0;
        uint64_t emitted = 0;
        std::vector<uint32_t> ends(n_subs, 0), begins(n_subs, 0), in_row(n_subs + 1, 0), first_begin(n_subs + 1, 0xffffffffu);
        uint64_t code_begin = first_bit;
        first_begin[first_bit / 256] = first_bit % 256;
        begins[first_bit / 256] = 0;
        for (uint64_t bit = first_bit; bit < n_bytes * 8; ++bit) {
            if ((bit & 255) == 0) in_row[bit >> 8] = node;
            const int32_t c = tree.child[2 * node + ((hb[bit >> 3] >> (7 - (bit & 7))) & 1)];
            if (c >= 0) {
                node = c;
            } else {
                if (c <= et::FSM_LEAF0) {
                    ends[bit >> 8] += 1;
                    begins[code_begin >> 8] += 1;
                    ++emitted;
                }
                node = 0;
                code_begin = bit + 1;
                if ((code_begin >> 8) < n_subs && first_begin[code_begin >> 8] == 0xffffffffu) first_begin[code_begin >> 8] = code_begin & 255;
            }
        }
        in_row[n_subs] = 0;
        for (uint64_t s = 0; s < n_subs; ++s) {
            const uint32_t ex = (s + 1 < n_subs) ? in_row[s + 1] : 0u;
            want[s] = in_row[s] | (ex << 11) | (ends[s] << 22);
            const uint32_t fb = first_begin[s] == 0xffffffffu ? 0u : first_begin[s];
            want_off[s] = fb | (begins[s] << 16);
            want_count[s >> 8] += ends[s];
        }
        for (uint32_t b = 0; b < n_blocks; ++b) {
            const uint64_t last = std::min<uint64_t>(n_subs, (static_cast<uint64_t>(b) + 1) * 256);
            want_exit[b] = last < n_subs ? in_row[last] : 0u;
        }
        printf("host walk: %llu symbols end inside the stream (declared %llu), %.1f s\n", (unsigned long long)emitted, (unsigned long long)n_symbols,
               std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }

    uint32_t *d_state, *d_exit, *d_count, *d_flags;
    unsigned long long *d_off;
    CK(hipMalloc(&d_state, n_subs * 4 + 64));
    CK(hipMalloc(&d_exit, n_blocks * 4 + 64));
    CK(hipMalloc(&d_count, n_blocks * 4 + 64));
    CK(hipMalloc(&d_flags, 64));
    CK(hipMalloc(&d_off, (n_blocks + 1) * 8 + 64));
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    et::FsmTables ft = {d_sync, d_write, plan.n_int, plan.k_sync, plan.k_write, plan.rows_sync, plan.rows_write, plan.max_per_lane, 0, 0, 0};
    const char *env;
    if ((env = getenv("FSM_SYNC_THREADS"))) ft.sync_threads = atoi(env);
    if ((env = getenv("FSM_WRITE_THREADS"))) ft.write_threads = atoi(env);
    if ((env = getenv("FSM_WRITE_WINDOW"))) ft.write_window = atoi(env);

    int bad = 0;
    // ---- D1 ----
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        CK(hipMemsetAsync(d_flags, 0, 64, stream));
        CK(hipEventRecord(e0, stream));
        et::launch_fsm_sync(stream, words, n_bytes, first_bit, n_subs, ft, d_state, d_exit, d_count, d_flags, 6, et::DEC_HAVE_START);
        CK(hipEventRecord(e1, stream));
        CK(hipStreamSynchronize(stream));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("k_fsm_sync<%u> x%u threads: %.3f ms  (%.0f GB/s of stream, %.3f of 8 TB/s)\n", plan.k_sync, ft.sync_threads, best, n_bytes / best / 1e6,
           n_bytes / best / 1e6 / 8000.0);
    {
        std::vector<uint32_t> got(n_subs), gexit(n_blocks), gcount(n_blocks), flags(16);
        CK(hipMemcpy(got.data(), d_state, n_subs * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gexit.data(), d_exit, n_blocks * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gcount.data(), d_count, n_blocks * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(flags.data(), d_flags, 64, hipMemcpyDeviceToHost));
        // the first sweep settles lanes inside a group; a group's first lane keeps its run-in (repaired by the next sweep)
        size_t diff = 0, diff_first = 0;
        const uint64_t group_subs = 256;
        std::vector<uint8_t> bad_group((n_subs + group_subs - 1) / group_subs, 0);
        for (uint64_t s = 0; s < n_subs; s += group_subs)
            if ((got[s] & 0x7ff) != (want[s] & 0x7ff)) {
                bad_group[s / group_subs] = 1;
                ++diff_first;
            }
        for (uint64_t s = 0; s < n_subs; ++s)
            if (!bad_group[s / group_subs] && got[s] != want[s]) {
                if (diff < 5) printf("  sub %llu: got %08x want %08x\n", (unsigned long long)s, got[s], want[s]);
                ++diff;
            }
        size_t bdiff = 0;
        for (uint32_t b = 0; b < n_blocks; ++b)
            if (!bad_group[(static_cast<uint64_t>(b) * 256) / group_subs] && (gexit[b] != want_exit[b] || gcount[b] != want_count[b])) {
                if (bdiff < 5) printf("  block %u: exit %u/%u count %u/%u\n", b, gexit[b], want_exit[b], gcount[b], want_count[b]);
                ++bdiff;
            }
        printf("  groups whose run-in missed: %zu of %zu; other subsequences that differ: %zu; blocks: %zu; gave up: %u\n", diff_first, bad_group.size(), diff,
               bdiff, flags[1]);
        bad += diff != 0 || bdiff != 0;
    }

    // ---- D3 from the true state ----
    std::vector<unsigned long long> h_off(n_blocks + 1, 0);
    for (uint32_t b = 0; b < n_blocks; ++b) h_off[b + 1] = h_off[b] + want_count[b];
    CK(hipMemcpy(d_off, h_off.data(), (n_blocks + 1) * 8, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 1) {  // offsets of blocks in the "begins" convention
            std::vector<uint32_t> cnt(n_blocks, 0);
            for (uint64_t s = 0; s < n_subs; ++s) cnt[s >> 8] += want_off[s] >> 16;
            for (uint32_t b = 0; b < n_blocks; ++b) h_off[b + 1] = h_off[b] + cnt[b];
            CK(hipMemcpy(d_off, h_off.data(), (n_blocks + 1) * 8, hipMemcpyHostToDevice));
        }
        CK(hipMemcpy(d_state, mode ? want_off.data() : want.data(), n_subs * 4, hipMemcpyHostToDevice));
        best = 1e9f;
        for (int it = 0; it < 6; ++it) {
            CK(hipMemsetAsync(d_out, 0, n, stream));
            CK(hipMemsetAsync(d_flags, 0, 64, stream));
            CK(hipEventRecord(e0, stream));
            et::launch_fsm_write(stream, words, n_bytes, first_bit, n_subs, ft, d_state, d_off, n_symbols, d_out, mode == 1, true);
            CK(hipEventRecord(e1, stream));
            CK(hipStreamSynchronize(stream));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("k_fsm_write<%u> x%u threads, %s: %.3f ms  (%.0f GB/s of stream + text, %.3f of 8 TB/s)\n", plan.k_write, ft.write_threads,
               mode ? "offset mode" : "state mode", best, (n_bytes + n) / best / 1e6, (n_bytes + n) / best / 1e6 / 8000.0);
        std::vector<uint8_t> back(n);
        CK(hipMemcpy(back.data(), d_out, n, hipMemcpyDeviceToHost));
        size_t diff = 0, first = 0;
        for (size_t i = 0; i < n; ++i)
            if (back[i] != text[i]) {
                if (!diff) first = i;
                ++diff;
            }
        printf("  output bytes that differ: %zu (first at %zu)\n", diff, first);
        bad += diff != 0;
    }
    printf(bad ? "FAILED\n" : "ok\n");
    et_ctx_destroy(ctx);
    return bad ? 1 : 0;
}
