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
