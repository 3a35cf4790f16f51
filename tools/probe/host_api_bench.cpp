// PCIe-inclusive rate of the host-pointer and file entry points, measured from C++ (no
// Python allocations in the way).  NEVER the headline `value` of bench.py.
//   hipcc -O2 -Iinclude -o build/host_api_bench tools/probe/host_api_bench.cpp -Lentreepy_amd -lentreepy_hip -Wl,-rpath,$PWD/entreepy_amd
//   build/host_api_bench <text file> [tmp dir]
// "cold" = input in a freshly malloc'ed + filled buffer and output into a fresh buffer (what a
// one-shot caller has), "warm" = the same buffers again.  raw = one hipMemcpy per direction
// straight from/to the caller's pageable memory (what the library did before et_io).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "entreepy_hip.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string tmp = argc > 2 ? argv[2] : "/tmp";
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const size_t n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> src(n);
    if (fread(src.data(), 1, n, f) != n) return 2;
    fclose(f);
    et_ctx *ctx = nullptr;
    if (et_ctx_create(0, &ctx) != ET_OK) return 3;
    et_ctx_reserve(ctx, n);
    const size_t bound = et_encode_bound(n);
    size_t m = 0;
    {   // warm the library (workspaces, staging buffers)
        std::vector<uint8_t> o(bound), b(n + 64);
        size_t k = 0;
        if (et_encode(ctx, src.data(), n, o.data(), bound, &m) != ET_OK) return 4;
        if (et_decode(ctx, o.data() + 4, m - 4, b.data(), b.size(), &k) != ET_OK || k != n || memcmp(b.data(), src.data(), n)) return 5;
    }
    std::vector<uint8_t> et(bound);
    et_encode(ctx, src.data(), n, et.data(), bound, &m);
    for (int rep = 0; rep < 2; ++rep) {
        // cold buffers
        uint8_t *in = static_cast<uint8_t *>(malloc(n)), *out = static_cast<uint8_t *>(malloc(bound));
        memcpy(in, src.data(), n);
        double t = now();
        size_t k = 0;
        et_encode(ctx, in, n, out, bound, &k);
        const double e_cold = now() - t;
        t = now();
        et_encode(ctx, in, n, out, bound, &k);
        const double e_warm = now() - t;
        uint8_t *cin = static_cast<uint8_t *>(malloc(m)), *cout = static_cast<uint8_t *>(malloc(n + 64));
        memcpy(cin, et.data(), m);
        t = now();
        et_decode(ctx, cin + 4, m - 4, cout, n + 64, &k);
        const double d_cold = now() - t;
        t = now();
        et_decode(ctx, cin + 4, m - 4, cout, n + 64, &k);
        const double d_warm = now() - t;
        if (k != n || memcmp(cout, src.data(), n)) return 6;
        printf("rep %d host pointers: et_encode cold %.2f GB/s (%.1f ms) warm %.2f GB/s | et_decode cold %.2f GB/s (%.1f ms) warm %.2f GB/s  [GB/s of text, n = %zu, .et = %zu]\n", rep,
               n / e_cold / 1e9, e_cold * 1e3, n / e_warm / 1e9, n / d_cold / 1e9, d_cold * 1e3, n / d_warm / 1e9, n, m);
        // the old way: one hipMemcpy per direction on the caller's cold pageable memory
        void *d_in, *d_out;
        hipMalloc(&d_in, n + 16);
        hipMalloc(&d_out, bound + 16);
        uint8_t *in2 = static_cast<uint8_t *>(malloc(n)), *out2 = static_cast<uint8_t *>(malloc(bound));
        memcpy(in2, src.data(), n);
        t = now();
        hipMemcpy(d_in, in2, n, hipMemcpyHostToDevice);
        et_encode_device(ctx, d_in, n, d_out, bound, &k);
        hipMemcpy(out2, d_out, k, hipMemcpyDeviceToHost);
        const double raw = now() - t;
        printf("rep %d raw hipMemcpy + et_encode_device + hipMemcpy, cold buffers: %.2f GB/s (%.1f ms)\n", rep, n / raw / 1e9, raw * 1e3);
        hipFree(d_in); hipFree(d_out);
        free(in); free(out); free(cin); free(cout); free(in2); free(out2);
    }
    // files (page cache warm: the file was just written / read)
    const std::string p_in = tmp + "/et_bench_in.txt", p_et = tmp + "/et_bench.et", p_back = tmp + "/et_bench_back.txt";
    int fd = open(p_in.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (write(fd, src.data(), n) != static_cast<ssize_t>(n)) return 7;
    close(fd);
    for (int rep = 0; rep < 2; ++rep) {
        int fi = open(p_in.c_str(), O_RDONLY), fo = open(p_et.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        size_t a = 0, b = 0;
        double t = now();
        int rc = et_encode_fd(ctx, fi, fo, &a, &b);
        const double e = now() - t;
        close(fi); close(fo);
        if (rc != ET_OK) return 8;
        fi = open(p_et.c_str(), O_RDONLY); fo = open(p_back.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        t = now();
        rc = et_decode_fd(ctx, fi, 4, fo, &a, &b);
        const double d = now() - t;
        close(fi); close(fo);
        if (rc != ET_OK || b != n) return 9;
        printf("rep %d files in %s: et_encode_fd %.2f GB/s (%.1f ms) | et_decode_fd %.2f GB/s (%.1f ms)\n", rep, tmp.c_str(), n / e / 1e9, e * 1e3, n / d / 1e9, d * 1e3);
    }
    unlink(p_in.c_str()); unlink(p_et.c_str()); unlink(p_back.c_str());
    et_ctx_destroy(ctx);
    return 0;
}
