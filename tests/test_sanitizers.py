"""CPU: the oracle and the product's host-side C++ (code construction, header writer,
parser) under AddressSanitizer + UBSan.  (GPU ASan is not available on this pool.)"""
import os
import subprocess
import sys
import textwrap

import pytest

from tests.conftest import ROOT

DRIVER = textwrap.dedent(r"""
    #include "entreepy_hip.h"
    #include "et_oracle.h"
    #include <cstdio>
    #include <cstdlib>
    #include <cstring>
    #include <vector>
    // deterministic fuzz of the host half against the oracle, no GPU involved
    static uint64_t rng = 88172645463325252ull;
    static uint64_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
    int main() {
        for (int it = 0; it < 4000; ++it) {
            uint64_t hist[256] = {0};
            int k = next() % 257, mode = it % 4;
            for (int i = 0; i < k; ++i) {
                uint64_t v = mode == 0 ? 1 + next() % 3 : mode == 1 ? 1 + next() % 100000 : mode == 2 ? 1ull << (next() % 45) : 7;
                hist[next() % 256] = v;
            }
            et_codebook cb; et_oracle_dict od; int leaves = 0; uint8_t order[256];
            int a = et_build_codebook(hist, &cb), b = et_oracle_build_dict(hist, &od, order, &leaves);
            if ((a != 0) != (b != 0)) { std::printf("status mismatch\n"); return 1; }
            if (a) continue;
            if (std::memcmp(cb.data, od.data, sizeof od.data) || std::memcmp(cb.length, od.length, 256)) { std::printf("table mismatch\n"); return 1; }
            uint8_t h1[8192], h2[8192]; size_t n1 = 0; uint64_t n = next();
            if (et_write_header(&cb, n, h1, sizeof h1, &n1) != ET_OK) { std::printf("header failed\n"); return 1; }
            int64_t n2 = et_oracle_write_header(&od, n, h2, sizeof h2);
            if ((int64_t)n1 != n2 || std::memcmp(h1, h2, n1)) { std::printf("header mismatch\n"); return 1; }
            et_codebook back; uint64_t ns = 0; size_t off = 0;
            int rc = et_parse_header(h1 + 4, n1 - 4, &back, &ns, &off);
            if (cb.max_length <= 32 && cb.n_coded && (rc != ET_OK || off + 4 != n1)) { std::printf("parse failed %d\n", rc); return 1; }
            // truncated and bit-flipped headers must be rejected or parsed, never crash
            for (int t = 0; t < 8; ++t) {
                std::vector<uint8_t> m(h1 + 4, h1 + n1);
                if (!m.empty()) m[next() % m.size()] ^= (uint8_t)(1u << (next() % 8));
                m.resize(next() % (m.size() + 1));
                et_parse_header(m.data(), m.size(), &back, &ns, &off);
            }
        }
        std::printf("ok\n");
        return 0;
    }
""")


@pytest.mark.skipif(subprocess.run(["which", "g++"], capture_output=True).returncode != 0, reason="g++ missing")
def test_host_code_and_oracle_under_asan_ubsan(tmp_path):
    src = tmp_path / "driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           f"-I{ROOT}/include", f"-I{ROOT}/oracle", str(src), f"{ROOT}/entreepy_amd/csrc/et_codebook.cpp", "-x", "c", f"{ROOT}/oracle/et_oracle.c", "-o", str(exe)]
    subprocess.check_call(cmd)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


TABLES_DRIVER = textwrap.dedent(r"""
    #include "entreepy_hip.h"
    #include "et_tables.h"
    #include <cstdio>
    #include <cstdlib>
    #include <cstring>
    #include <vector>
    // The decode lookup tables (et_tables.cpp) against a brute-force decoder, for random code
    // tables from the product's own code construction: every first-level entry of every
    // format, and the second level of every code longer than the index.
    static uint64_t rng = 0x9E3779B97F4A7C15ull;
    static uint64_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
    static const et_codebook *CB;
    // the code that prefixes the `avail` bits at the top of w: returns symbol or -1, *len
    static int first_code(uint32_t w, uint32_t avail, uint32_t *len) {
        for (int s = 0; s < 256; ++s) {
            const uint32_t l = CB->length[s];
            if (!l || l > avail) continue;
            const uint32_t code = l == 32 ? CB->data[s] : (CB->data[s] & ((1u << l) - 1u));
            if ((w >> (32 - l)) == code) { *len = l; return s; }
        }
        return -1;
    }
    #define FAIL(...) do { std::printf(__VA_ARGS__); std::printf(" (it %d)\n", it); return 1; } while (0)
    int main() {
        std::vector<uint32_t> lut(1u << et::DEC_LUT_BITS_MAX), longc(512), steps((1u << et::DEC_STEP_BITS_MAX) + et::DEC_STEP_SUB_WORDS + 8);
        std::vector<uint16_t> sub((et::DEC_SUB_TABLES_MAX << et::DEC_SUB_BITS_MAX) + 64);
        int checked = 0;
        for (int it = 0; it < 700; ++it) {
            uint64_t hist[256] = {0};
            const int k = 1 + next() % 256, mode = it % 4;
            for (int i = 0; i < k; ++i) hist[next() % 256] = mode == 0 ? 1 + next() % 3 : mode == 1 ? 1 + next() % 100000 : mode == 2 ? 1ull << (next() % 30) : 7;
            et_codebook cb;
            if (et_build_codebook(hist, &cb) != ET_OK || cb.n_coded == 0 || cb.max_length > 32) continue;
            CB = &cb;
            ++checked;
            // --- sync step table, several index widths
            const uint32_t widths[3] = {12, 13, 9};
            for (uint32_t want : widths) {
                uint32_t sb = 0, ns = 0;
                const uint32_t K = et::build_step_table(&cb, want, steps.data(), &sb, &ns);
                if (K > want || K > cb.max_length || (ns << sb) > et::DEC_STEP_SUB_WORDS) FAIL("step table shape");
                for (uint32_t v = 0; v < (1u << K); ++v) {
                    uint32_t used = 0, cnt = 0, first = 0, l = 0;
                    while (used < K && first_code(v << (32 - K) << used, K - used, &l) >= 0) { if (!cnt) first = l; used += l; ++cnt; }
                    const uint32_t e = steps[v];
                    if (cnt) { if (e != (first << 28) + (cnt << 16) - used) FAIL("step entry %u", v); }
                    else if ((e & 0x0fffffffu) != et::STEP_ESCAPE) FAIL("escape entry %u", v);
                }
                for (int s = 0; s < 256; ++s) {
                    const uint32_t l = cb.length[s];
                    if (l <= K) continue;
                    const uint32_t code = l == 32 ? cb.data[s] : (cb.data[s] & ((1u << l) - 1u));
                    const uint32_t e = steps[code >> (l - K)], t = e >> 28;
                    if ((e & 0x0fffffffu) != et::STEP_ESCAPE) FAIL("long code %d has no escape", s);
                    if (t && l - K <= sb) {
                        const uint32_t rest = (code & ((1u << (l - K)) - 1u)) << (sb - (l - K)) | (static_cast<uint32_t>(next()) & ((1u << (sb - (l - K))) - 1u));
                        if (t > ns || steps[(1u << K) + ((t - 1) << sb) + rest] != (1u << 16) - l) FAIL("second level of %d", s);
                    }
                }
            }
            // --- write step table
            {
                uint32_t sb = 0, ns = 0;
                const uint32_t K = et::build_write_step_table(&cb, 11, steps.data(), &sb, &ns);
                if ((ns << sb) > et::DEC_STEP_SUB_WORDS) FAIL("write table shape");
                for (uint32_t v = 0; v < (1u << K); ++v) {
                    uint32_t used = 0, cnt = 0, syms = 0, l = 0;
                    int s;
                    while (cnt < 2 && used < K && (s = first_code(v << (32 - K) << used, K - used, &l)) >= 0) { syms |= static_cast<uint32_t>(s) << (16 + 8 * cnt); used += l; ++cnt; }
                    const uint32_t e = steps[v];
                    if (cnt) { if (e != (syms | (((cnt << 10) - used) & 0xffffu))) FAIL("write entry %u", v); }
                    else if ((e & 0xffffu) != et::WSTEP_ESCAPE) FAIL("write escape %u", v);
                }
                for (int s = 0; s < 256; ++s) {
                    const uint32_t l = cb.length[s];
                    if (l <= K) continue;
                    const uint32_t code = l == 32 ? cb.data[s] : (cb.data[s] & ((1u << l) - 1u));
                    const uint32_t e = steps[code >> (l - K)], t = e >> 24;
                    if ((e & 0xffffu) != et::WSTEP_ESCAPE) FAIL("write: long code %d has no escape", s);
                    if (t && l - K <= sb) {
                        const uint32_t rest = (code & ((1u << (l - K)) - 1u)) << (sb - (l - K));
                        if (t > ns || steps[(1u << K) + ((t - 1) << sb) + rest] != ((static_cast<uint32_t>(s) << 16) | ((1u << 10) - l))) FAIL("write second level of %d", s);
                    }
                }
            }
            // --- older format
            {
                et::HostDecodeTables ht;
                et::build_decode_tables(&cb, 11, et::DEC_WRITE_SYMS, lut.data(), longc.data(), sub.data(), &ht);
                const uint32_t K = ht.lut_bits;
                for (uint32_t v = 0; v < (1u << K); ++v) {
                    uint32_t used = 0, cnt = 0, syms = 0, l = 0;
                    int s;
                    while (cnt < et::DEC_WRITE_SYMS && used < K && (s = first_code(v << (32 - K) << used, K - used, &l)) >= 0) { syms |= static_cast<uint32_t>(s) << (8 * cnt); used += l; ++cnt; }
                    const uint32_t e = lut[v];
                    if (cnt && e != (syms | (used << et::LUT_LEN_SHIFT) | (cnt << et::LUT_N_SHIFT))) FAIL("lut entry %u", v);
                    if (!cnt && ((e >> et::LUT_N_SHIFT) & 3u)) FAIL("lut escape %u", v);
                }
                uint32_t n_long = 0;
                for (int s = 0; s < 256; ++s) n_long += cb.length[s] > K;
                if (n_long != ht.n_long) FAIL("long list length");
                for (uint32_t i = 0; i < ht.n_long; ++i) {
                    const uint32_t meta = longc[2 * i + 1], l = meta >> 8, s = meta & 0xffu;
                    const uint32_t code = l == 32 ? cb.data[s] : (cb.data[s] & ((1u << l) - 1u));
                    if (cb.length[s] != l || longc[2 * i] != code << (32 - l)) FAIL("long list entry %u", i);
                }
            }
        }
        std::printf(checked > 300 ? "ok\n" : "too few code tables\n");
        return 0;
    }
""")


@pytest.mark.skipif(subprocess.run(["which", "g++"], capture_output=True).returncode != 0, reason="g++ missing")
def test_decode_table_builders_under_asan_ubsan(tmp_path):
    """et_tables.cpp (the four lookup-table formats of the decode kernels) against a
    brute-force decoder, under AddressSanitizer + UBSan."""
    src = tmp_path / "tables_driver.cpp"
    src.write_text(TABLES_DRIVER)
    exe = tmp_path / "tables_driver"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           f"-I{ROOT}/include", f"-I{ROOT}/entreepy_amd/csrc", str(src), f"{ROOT}/entreepy_amd/csrc/et_tables.cpp",
           f"{ROOT}/entreepy_amd/csrc/et_codebook.cpp", "-o", str(exe)]
    subprocess.check_call(cmd)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


FAST_DRIVER = textwrap.dedent(r"""
    #include "et_oracle.h"
    #include <stdio.h>
    #include <stdlib.h>
    #include <string.h>
    /* oracle/et_cpu_fast.c (bench.py's all-cores CPU baseline) against the restatement,
       chunks run one after the other: pack at bit offsets, walk + mark + merge + write. */
    static uint64_t rng = 0x2545F4914F6CDD1Dull;
    static uint64_t next(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
    #define MARKS 4096u
    int main(void) {
        for (int it = 0; it < 60; ++it) {
            const size_t n = 1 + next() % 60000;
            const int alphabet = 1 + next() % 256, skew = it % 3;
            uint8_t *text = malloc(n), *et = malloc(n + 7200), *mine = calloc(n + 7200 + 16, 1), *back = malloc(n + 16);
            for (size_t i = 0; i < n; ++i) {
                uint64_t r = next();
                text[i] = (uint8_t)(skew == 0 ? r % alphabet : skew == 1 ? (r % 7 ? r % 3 : r % alphabet) : __builtin_ctzll(r | (1ull << 40)) % alphabet);
            }
            const int64_t want = et_oracle_encode(text, n, et, n + 7200);
            if (want < 0) { printf("oracle encode failed\n"); return 1; }
            const int parts = 1 + it % 4;
            uint64_t occ[256], hist[4][256], starts[5];
            size_t cut[5];
            for (int k = 0; k <= parts; ++k) cut[k] = n * k / parts;
            memset(occ, 0, sizeof occ);
            for (int k = 0; k < parts; ++k) {
                et_oracle_histogram(text + cut[k], cut[k + 1] - cut[k], hist[k]);
                for (int s = 0; s < 256; ++s) occ[s] += hist[k][s];
            }
            et_oracle_dict d;
            if (et_oracle_build_dict(occ, &d, NULL, NULL)) { printf("dict failed\n"); return 1; }
            const int64_t hb = et_oracle_write_header(&d, n, mine, n + 7200);
            starts[0] = 8 * (uint64_t)hb;
            for (int k = 0; k < parts; ++k) {
                uint64_t b = 0;
                for (int s = 0; s < 256; ++s) b += hist[k][s] * d.length[s];
                starts[k + 1] = starts[k] + b;
            }
            for (int k = 0; k < parts; ++k)
                if (et_fast_pack(&d, text + cut[k], cut[k + 1] - cut[k], mine, starts[k]) != starts[k + 1]) { printf("pack end mismatch\n"); return 1; }
            if ((int64_t)((starts[parts] + 7) / 8) != want || memcmp(mine, et, (size_t)want)) { printf("encode mismatch (it %d)\n", it); return 1; }
            /* decode */
            et_oracle_dict pd; uint64_t body_start = 0; uint32_t body_len = 0;
            if (et_oracle_parse_dict(et + 4, (size_t)want - 4, &pd, &body_start, &body_len)) {
                int longest = 0;
                for (int s = 0; s < 256; ++s) if (d.length[s] > longest) longest = d.length[s];
                if (longest > 32 || n >= 256) { free(text); free(et); free(mine); free(back); continue; }  /* Q3 tables are not this variant's */
                printf("parse_dict failed\n"); return 1;
            }
            struct et_fast_tables *t = malloc(et_fast_tables_size());
            if (et_fast_build_tables(&pd, t)) { printf("tables failed\n"); return 1; }
            const uint8_t *body = et + 4 + body_start;
            const uint64_t body_bytes = (uint64_t)want - 4 - body_start;
            uint64_t bcut[5], cnt[4], st[4], ex = 0, exit_bit = 0;
            for (int k = 0; k <= parts; ++k) bcut[k] = 8 * (body_bytes * k / parts);
            uint32_t *marks = malloc(MARKS * sizeof(uint32_t));
            st[0] = 0;
            cnt[0] = et_fast_walk(t, body, body_bytes, 0, bcut[1], ~0ull, NULL, &exit_bit, NULL, 0, 0);
            for (int k = 1; k < parts; ++k) {
                memset(marks, 0xff, MARKS * sizeof(uint32_t));
                const uint64_t c = et_fast_walk(t, body, body_bytes, bcut[k], bcut[k + 1], ~0ull, NULL, &ex, marks, bcut[k], MARKS);
                uint64_t extra = 0; uint32_t at = 0;
                st[k] = exit_bit;
                if (exit_bit >= bcut[k + 1]) cnt[k] = 0;
                else if (et_fast_merge(t, body, body_bytes, exit_bit, marks, bcut[k], MARKS, &extra, &at)) { cnt[k] = extra + c - marks[at]; exit_bit = ex; }
                else cnt[k] = et_fast_walk(t, body, body_bytes, exit_bit, bcut[k + 1], ~0ull, NULL, &exit_bit, NULL, 0, 0);
            }
            uint64_t off = 0;
            for (int k = 0; k < parts; ++k) {
                uint64_t lim = cnt[k];
                if (off + lim > body_len) lim = body_len > off ? body_len - off : 0;
                if (lim && et_fast_walk(t, body, body_bytes, st[k], ~0ull, lim, back + off, &ex, NULL, 0, 0) != lim) { printf("short write\n"); return 1; }
                off += lim;
            }
            /* all 256 values present: the reference drops the most frequent symbol (Q1), the text does not come back */
            int present = 0;
            for (int s = 0; s < 256; ++s) present += occ[s] != 0;
            if (present < 256 && (off != n || memcmp(back, text, n))) { printf("decode mismatch (it %d, parts %d, n %zu, got %llu)\n", it, parts, n, (unsigned long long)off); return 1; }
            free(marks); free(t); free(text); free(et); free(mine); free(back);
        }
        printf("ok\n");
        return 0;
    }
""")


@pytest.mark.skipif(subprocess.run(["which", "gcc"], capture_output=True).returncode != 0, reason="gcc missing")
def test_fast_cpu_variant_under_asan_ubsan(tmp_path):
    src = tmp_path / "fast_driver.c"
    src.write_text(FAST_DRIVER)
    exe = tmp_path / "fast_driver"
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", f"-I{ROOT}/oracle",
           str(src), f"{ROOT}/oracle/et_oracle.c", f"{ROOT}/oracle/et_cpu_fast.c", "-o", str(exe)]
    subprocess.check_call(cmd)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


TREEWALK_DRIVER = textwrap.dedent(r"""
    #include "entreepy_hip.h"
    #include "et_treewalk.h"
    #include <cstdio>
    #include <cstring>
    #include <vector>
    // The tree walk's host part (et_treewalk_host.cpp) against a bit-serial decoder: for code tables of the
    // product's own construction, walking random bytes through the table from any row must end in the node,
    // and count the codewords, that decoding the same bits one at a time does.
    static uint64_t rng = 0x2545F4914F6CDD1Dull;
    static uint64_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
    int main() {
        static et::TwTree tree;
        std::vector<uint16_t> table((et::TW_MAX_NODES + et::TW_ENTRY_ROWS) * 256);
        for (int it = 0; it < 300; ++it) {
            uint64_t hist[256] = {0};
            int k = 2 + next() % 255;
            for (int i = 0; i < k; ++i) hist[next() % 256] = 1 + next() % (1ull << (next() % 24));
            et_codebook cb;
            if (et_build_codebook(hist, &cb) != ET_OK || cb.n_coded < 2 || cb.max_length > 32) continue;
            if (et::tw_build_tree(&cb, &tree) != ET_OK) { std::puts("a Huffman table was turned away"); return 1; }
            if (tree.n_int != cb.n_coded - 1) { std::puts("node count"); return 1; }
            et::tw_fill_table(&tree, table.data());
            for (int w = 0; w < 200; ++w) {
                uint32_t row = next() % tree.n_int, node = row, count = 0, first = 99;
                uint8_t byte = (uint8_t)next();
                for (int i = 0; i < 8; ++i) {
                    int16_t c = tree.child[2 * node + ((byte >> (7 - i)) & 1)];
                    if (c >= 0) node = c; else { if (!count) first = i; ++count; node = 0; }
                }
                uint16_t e = table[(row << 8) + byte];
                if ((e & et::TW_ROW_MASK) != node || ((e >> et::TW_N_SHIFT) & 15u) != count || (count && (uint32_t)(e >> et::TW_OFF_SHIFT) != first)) {
                    std::puts("entry mismatch");
                    return 1;
                }
            }
            // the write walk's chained tables: plan within its bounds, and random bits decoded through the tables
            // (following each entry's own next-table / next-shift fields) give what the tree gives bit by bit
            static et::ChainPlan plan;
            static uint64_t chain[et::CH_MAX_ENTRIES];
            et::tw_chain_plan(&tree, &plan);
            if (plan.n_tables < 1 || plan.n_tables > et::CH_MAX_TABLES || plan.n_entries > et::CH_MAX_ENTRIES || plan.tab[0].bits != et::CH_ROOT_BITS) { std::puts("plan bounds"); return 1; }
            et::tw_chain_fill(&tree, &plan, chain);
            for (int w = 0; w < 60; ++w) {
                uint8_t bits[160];
                for (int i = 0; i < 160; ++i) bits[i] = (uint8_t)(next() & 1);
                // bit-serial reference: symbols and the positions where they end, for the first 96 bits
                std::vector<int> want_sym, want_end;
                { uint32_t node = 0; for (int i = 0; i < 128; ++i) { int16_t c = tree.child[2 * node + bits[i]]; if (c >= 0) node = c; else { want_sym.push_back(et::TW_LEAF0 - c); want_end.push_back(i + 1); node = 0; } } }
                uint32_t pos = 0, t_off = 0, shift = 32 - et::CH_ROOT_BITS; size_t got = 0;
                while (pos < 90) {
                    uint32_t window = 0;
                    for (int i = 0; i < 32; ++i) window = (window << 1) | bits[pos + i];
                    const uint64_t e = chain[t_off / 8 + (window >> shift)];
                    const uint32_t lo = (uint32_t)e, hi = (uint32_t)(e >> 32);
                    const int adv = (int16_t)(lo & 0xffffu);
                    const int n = (adv + 15) >> 10, used = (n << 10) - adv;
                    if (used < 1 || used > (int)et::CH_ROOT_BITS || n < 0 || n > 2) { std::puts("chain entry fields"); return 1; }
                    if (n >= 1) { if (got >= want_sym.size() || want_sym[got] != (int)((lo >> 16) & 0xff) || want_end[got] != (int)(pos + (lo >> 24))) { std::puts("chain first symbol"); return 1; } ++got; }
                    if (n == 2) { if (got >= want_sym.size() || want_sym[got] != (int)((hi >> 16) & 0xff) || want_end[got] != (int)(pos + used)) { std::puts("chain second symbol"); return 1; } ++got; }
                    pos += used;
                    t_off = hi & 0xffffu; shift = hi >> 24;
                    if (t_off / 8 >= plan.n_entries || shift < 32 - et::CH_ROOT_BITS || shift > 31) { std::puts("chain next table"); return 1; }
                }
            }
        }
        // dictionaries that are not full trees (or not prefix-free) are turned away, never walked
        et_codebook bad;
        std::memset(&bad, 0, sizeof bad);
        bad.length[1] = 1; bad.data[1] = 0; bad.length[2] = 2; bad.data[2] = 2; bad.n_coded = 2; bad.min_length = 1; bad.max_length = 2;
        if (et::tw_build_tree(&bad, &tree) == ET_OK) { std::puts("a gap was accepted"); return 1; }
        bad.length[3] = 2; bad.data[3] = 1; bad.n_coded = 3;  // "01" under the leaf "0"
        if (et::tw_build_tree(&bad, &tree) == ET_OK) { std::puts("a prefix clash was accepted"); return 1; }
        std::printf("ok\n");
        return 0;
    }
""")


@pytest.mark.skipif(subprocess.run(["which", "g++"], capture_output=True).returncode != 0, reason="g++ missing")
def test_treewalk_host_part_under_asan_ubsan(tmp_path):
    src = tmp_path / "tw.cpp"
    src.write_text(TREEWALK_DRIVER)
    exe = tmp_path / "tw"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           f"-I{ROOT}/include", f"-I{ROOT}/entreepy_amd/csrc", str(src), f"{ROOT}/entreepy_amd/csrc/et_treewalk_host.cpp",
           f"{ROOT}/entreepy_amd/csrc/et_codebook.cpp", "-o", str(exe)]
    subprocess.check_call(cmd)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


GROUP_DRIVER = textwrap.dedent(r"""
    #include "entreepy_hip.h"
    #include "et_oracle.h"
    #include <condition_variable>
    #include <cstdio>
    #include <cstring>
    #include <mutex>
    #include <thread>
    #include <vector>
    // The group sequence (et_shard_seq.cpp, the file the product is built from) over the CPU stand-in, rank THREADS of one
    // process: ragged cuts, dirty buffers, cold decode whole and windowed, a failure injected on one rank -- under ASan + UBSan.
    static uint64_t rng = 0x243F6A8885A308D3ull;
    static uint64_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
    struct Exchange {
        std::mutex m; std::condition_variable cv; int world = 1, arrived = 0, gen = 0; std::vector<uint8_t> slots;
        void meet(std::unique_lock<std::mutex> &lk) { const int g = gen; if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); } else cv.wait(lk, [&] { return gen != g; }); }
        static int gather(void *user, const void *send, void *recv, size_t bytes) {
            auto *x = static_cast<std::pair<Exchange *, int> *>(user); Exchange &e = *x->first;
            std::unique_lock<std::mutex> lk(e.m);
            if (e.slots.size() < bytes * e.world) e.slots.resize(bytes * e.world);
            std::memcpy(e.slots.data() + bytes * x->second, send, bytes);
            e.meet(lk); std::memcpy(recv, e.slots.data(), bytes * e.world); e.meet(lk);
            return 0;
        }
    };
    int main() {
        int bad = 0, relayed = 0, images = 0, decodes = 0;
        for (int trial = 0; trial < 120; ++trial) {
            const int world = 1 + next() % 4;
            const size_t n = trial % 9 == 0 ? 1 + next() % 40 : 1 + next() % 90000;
            std::vector<uint8_t> text(n);
            const int mode = trial % 3, k = 2 + next() % 200;
            for (auto &b : text) b = mode == 0 ? (uint8_t)(next() % k) : mode == 1 ? (uint8_t)(__builtin_ctzll(next() | (1ull << 40))) : (uint8_t)(32 + next() % 64 % (1 + next() % 64));
            std::vector<size_t> cuts(world + 1, 0); cuts[world] = n;
            for (int r = 1; r < world; ++r) cuts[r] = next() % (n + 1);
            std::sort(cuts.begin(), cuts.end());
            std::vector<uint8_t> want(n + 7200);
            const int64_t want_len = et_oracle_encode(text.data(), n, want.data(), want.size());
            if (want_len < 0) { std::printf("oracle encode failed\n"); return 1; }
            std::vector<uint8_t> image(((size_t)want_len + 3) / 4 * 4, 0xEE);
            Exchange ex; ex.world = world;
            std::vector<std::pair<Exchange *, int>> who(world);
            std::vector<et_group *> grp(world, nullptr);
            for (int r = 0; r < world; ++r) { who[r] = {&ex, r}; if (et_group_create(nullptr, r, world, Exchange::gather, &who[r], &grp[r]) != ET_OK) return 1; }
            std::vector<std::vector<uint8_t>> enc(world);
            for (int r = 0; r < world; ++r) enc[r].assign(et_encode_bound(cuts[r + 1] - cuts[r]) + 64, 0xA5);
            std::vector<int> rc(world, 0);
            auto run = [&](auto body) { std::vector<std::thread> th; for (int r = 0; r < world; ++r) th.emplace_back([&, r] { rc[r] = body(r); }); for (auto &t : th) t.join(); };
            const int victim = (int)(next() % world);
            if (trial % 4 == 1) {  // one rank without its output buffer: everybody must return that
                run([&](int r) { et_shard_info i; return et_encode_sharded(grp[r], text.data() + cuts[r], cuts[r + 1] - cuts[r], r == victim ? nullptr : enc[r].data(), enc[r].size(), &i); });
                for (int r = 0; r < world; ++r) if (rc[r] != ET_ERR_ARG) { ++bad; std::printf("trial %d: rank %d returned %d for rank %d's failure\n", trial, r, rc[r], victim); }
                ++relayed;
            }
            run([&](int r) {
                et_shard_info i;
                int c = et_encode_sharded(grp[r], text.data() + cuts[r], cuts[r + 1] - cuts[r], enc[r].data(), enc[r].size(), &i);
                if (c == ET_OK) c = et_shard_merge_seams(grp[r], enc[r].data());
                if (c == ET_OK) c = et_shard_place(grp[r], enc[r].data(), image.data(), image.size());
                return c;
            });
            bool all_ok = true, all_cap = true;
            for (int r = 0; r < world; ++r) { all_ok = all_ok && rc[r] == ET_OK; all_cap = all_cap && rc[r] == ET_ERR_CAP; }
            if (all_cap) { for (auto g : grp) et_group_destroy(g); continue; }  // (a shard of rare symbols that outgrows et_encode_bound: said by all)
            if (!all_ok || std::memcmp(image.data(), want.data(), (size_t)want_len)) { ++bad; std::printf("trial %d: image differs (world %d, n %zu)\n", trial, world, n); for (auto g : grp) et_group_destroy(g); continue; }
            ++images;
            // cold decode: whole stream on every rank, or each rank's own window
            const uint8_t *comp = want.data() + 4; const size_t len = (size_t)want_len - 4;
            std::vector<uint8_t> truth(n + 64); const int64_t truth_len = et_oracle_decode(comp, len, truth.data(), truth.size());
            et_codebook cb; uint64_t ns = 0; size_t off = 0;
            if (truth_len < 0 || et_parse_header(comp, len, &cb, &ns, &off) != ET_OK || cb.max_length > 32) { for (auto g : grp) et_group_destroy(g); continue; }
            std::vector<std::vector<uint8_t>> out(world, std::vector<uint8_t>(n + 64));
            std::vector<size_t> wrote(world, 0); std::vector<uint64_t> first(world, 0);
            const bool windowed = next() & 1;
            // (the stand-in wants 4-byte aligned streams, like the GPU: comp = want + 4 of a vector's data is)
            std::vector<std::vector<uint32_t>> win(world);
            run([&](int r) {
                if (!windowed) return et_decode_sharded(grp[r], comp, len, out[r].data(), out[r].size(), &wrote[r], &first[r]);
                uint64_t wo = 0, wl = 0, mine = 0;
                const size_t head_len = len < 8192 ? len : 8192;
                int c = et_decode_shard_window(comp, head_len, len, r, world, &wo, &wl);
                if (c != ET_OK) return c;
                win[r].assign((wl + 3) / 4 + 1, 0);
                std::memcpy(win[r].data(), comp + wo, wl);
                c = et_decode_sharded_begin(grp[r], comp, head_len, len, wl ? win[r].data() : nullptr, wo, wl, ~0ull, &mine, &first[r]);
                if (c == ET_OK) c = et_decode_sharded_write(grp[r], out[r].data(), out[r].size(), &wrote[r]);
                return c;
            });
            std::vector<uint8_t> got;
            bool ok = true; uint64_t pos = 0;
            for (int r = 0; r < world; ++r) { ok = ok && rc[r] == ET_OK && (wrote[r] == 0 || first[r] == pos); pos += wrote[r]; got.insert(got.end(), out[r].begin(), out[r].begin() + wrote[r]); }
            if (!ok || (int64_t)got.size() != truth_len || std::memcmp(got.data(), truth.data(), got.size())) { ++bad; std::printf("trial %d: decode differs (world %d, n %zu, windowed %d)\n", trial, world, n, (int)windowed); }
            else ++decodes;
            for (auto g : grp) et_group_destroy(g);
        }
        std::printf("%s images %d decodes %d relayed %d\n", bad ? "FAILED" : "ok", images, decodes, relayed);
        return bad ? 1 : 0;
    }
""")


@pytest.mark.skipif(subprocess.run(["which", "g++"], capture_output=True).returncode != 0, reason="g++ missing")
def test_group_sequence_under_asan_ubsan(tmp_path):
    """csrc/et_shard_seq.cpp -- the group sequence of the product, as it is -- over the CPU stand-in (tests/support/shard_cpu.cpp),
    rank threads of one process, under AddressSanitizer + UBSan: images equal the oracle's, cold decodes (whole and windowed) return
    the text, an injected failure comes back from every rank."""
    src = tmp_path / "driver.cpp"
    src.write_text("#include <algorithm>\n" + GROUP_DRIVER)
    exe = tmp_path / "driver"
    obj = tmp_path / "oracle.o"
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-c", f"{ROOT}/oracle/et_oracle.c", "-o", str(obj)])
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           f"-I{ROOT}/include", f"-I{ROOT}/oracle", f"-I{ROOT}/entreepy_amd/csrc", str(src), f"{ROOT}/entreepy_amd/csrc/et_shard_seq.cpp",
           f"{ROOT}/tests/support/shard_cpu.cpp", f"{ROOT}/entreepy_amd/csrc/et_codebook.cpp", str(obj), "-o", str(exe)]
    subprocess.check_call(cmd)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().startswith("ok"), out.stdout[-3000:] + out.stderr[-3000:]
