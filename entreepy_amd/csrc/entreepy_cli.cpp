// entreepy_cli.cpp -- the `entreepy` command: same surface as the reference's
// src/main.zig (options -h -p -t -d, commands c/d, -o), with encode()/decode()
// replaced by libentreepy_hip.so.  The reference's host is Zig; no Zig toolchain
// exists in this image, so the tested host driver is C++ (INTEGRATION.md has the Zig
// shim a maintainer would use instead).
#include "entreepy_hip.h"

#include <fcntl.h>
#include <unistd.h>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// main.zig:45-67, verbatim.
const char kHelp[] =
    "Entreepy - Text compression tool\n"
    "\n"
    "Usage: entreepy [options] [command] [file] [command options]\n"
    "\n"
    "Options:\n"
    "    -h, --help     show help\n"
    "    -p, --print    print decompressed text to stdout\n"
    "    -t, --test     test/dry run, does not write to file\n"
    "    -d, --debug    print huffman code dictionary and performance times to stdout\n"
    "\n"
    "Commands:\n"
    "    c    compress a file\n"
    "    d    decompress a file\n"
    "\n"
    "Command Options:\n"
    "    -o, --output    output file (default: [file].et or decoded_[file])\n"
    "\n"
    "Examples:\n"
    "    entreepy -d c text.txt -o text.txt.et\n"
    "    entreepy -ptd d text.txt.et -o decoded_text.txt\n";

enum class Mode { None, Compress, Decompress };

struct Options {
    bool print = false, debug = false, dry = false;
    int gpus = 1;  // --gpus N (not in the reference): shard the file over N GPUs of the node
    Mode mode = Mode::None;
    std::string in_path, out_path;
    bool have_in = false;
};

// utils.zig:3-13: f32 byte count, 1024 divisors, two decimals above 1 KiB.
std::string format_file_size(float byte_count) {
    char buf[64];
    if (byte_count < 1024.0f) std::snprintf(buf, sizeof buf, "%g B", static_cast<double>(byte_count));
    else if (byte_count < 1024.0f * 1024.0f) std::snprintf(buf, sizeof buf, "%.2f KB", static_cast<double>(byte_count / 1024.0f));
    else if (byte_count < 1024.0f * 1024.0f * 1024.0f) std::snprintf(buf, sizeof buf, "%.2f MB", static_cast<double>(byte_count / (1024.0f * 1024.0f)));
    else std::snprintf(buf, sizeof buf, "%.2f GB", static_cast<double>(byte_count / (1024.0f * 1024.0f * 1024.0f)));
    return buf;
}

// main.zig:73-146.  Returns 0 to continue, 1 to exit successfully (help), 2 on error.
int parse_cli(int argc, char **argv, Options &o) {
    enum { Normal, OutPath, InPath, Gpus } state = Normal;
    if (argc <= 1) {  // main.zig:148-152
        std::fputs(kHelp, stdout);
        return 1;
    }
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        switch (state) {
            case Normal:
                if (!arg.empty() && arg[0] == '-') {
                    bool stop = false;
                    for (size_t k = 1; k < arg.size() && !stop; ++k) {
                        switch (arg[k]) {
                            case 'h': std::fputs(kHelp, stdout); return 1;
                            case 'p': o.print = true; break;
                            case 'd': o.debug = true; break;
                            case 't': o.dry = true; break;
                            case 'o': state = OutPath; break;
                            case '-': {
                                const std::string name = arg.substr(2);
                                if (name == "help") { std::fputs(kHelp, stdout); return 1; }
                                else if (name == "print") o.print = true;
                                else if (name == "debug") o.debug = true;
                                else if (name == "test") o.dry = true;
                                else if (name == "output") state = OutPath;
                                else if (name == "gpus") state = Gpus;
                                else { std::fprintf(stderr, "error: invalid option: %s\n\n", arg.c_str()); return 2; }
                                stop = true;
                                break;
                            }
                            default: std::fprintf(stderr, "error: invalid option: %s\n\n", arg.c_str()); return 2;
                        }
                    }
                } else if (!arg.empty() && (arg[0] == 'c' || arg[0] == 'd')) {  // main.zig:123-130
                    o.mode = arg[0] == 'c' ? Mode::Compress : Mode::Decompress;
                    state = InPath;
                } else {
                    std::fprintf(stderr, "error: invalid command: %s\n\n", arg.c_str());
                    return 2;
                }
                break;
            case InPath: o.in_path = arg; o.have_in = true; state = Normal; break;
            case OutPath: o.out_path = arg; state = Normal; break;
            case Gpus:
                o.gpus = std::atoi(arg.c_str());
                if (o.gpus < 1 || o.gpus > 64) { std::fprintf(stderr, "error: invalid --gpus: %s\n\n", arg.c_str()); return 2; }
                state = Normal;
                break;
        }
    }
    if (o.mode == Mode::None) return 1;
    if (!o.have_in) { std::fprintf(stderr, "error: no input file\n"); return 2; }
    // main.zig:154-170 (the intended defaults; the reference reads an undefined slice here)
    if (o.out_path.empty()) {
        if (o.mode == Mode::Compress) {
            o.out_path = o.in_path + ".et";
        } else {
            const size_t slash = o.in_path.find_last_of('/');
            const std::string dir = slash == std::string::npos ? "" : o.in_path.substr(0, slash);
            std::string name = slash == std::string::npos ? o.in_path : o.in_path.substr(slash + 1);
            if (name.size() >= 3 && name.compare(name.size() - 3, 3, ".et") == 0) name.resize(name.size() - 3);
            o.out_path = (dir.empty() ? "" : dir + "/") + "decoded_" + name;
        }
    }
    return 0;
}

bool read_file(const std::string &path, std::vector<uint8_t> &data) {  // main.zig:34-40
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const std::streamsize n = f.tellg();
    f.seekg(0);
    data.resize(static_cast<size_t>(n));
    return n == 0 || static_cast<bool>(f.read(reinterpret_cast<char *>(data.data()), n));
}

// main.zig:199 leaves "validate the file" as a TODO; a file without the magic is refused here.
std::string g_magic_why;
bool magic_ok(int fd, et_ctx *) {
    uint8_t first4[4] = {0, 0, 0, 0};
    const char *why = nullptr;
    if (::pread(fd, first4, 4, 0) != 4) {
        g_magic_why = "file shorter than the 4-byte magic";
        return false;
    }
    if (et_check_magic(first4, &why) == ET_OK) return true;
    g_magic_why = why ? why : "bad magic";
    return false;
}

void dump_dictionary(const et_codebook &cb) {  // encode.zig:204-212
    const unsigned leaves = cb.n_coded ? cb.n_coded : 1;
    for (unsigned i = 0; i < leaves; ++i) {
        const unsigned s = cb.dfs_order[i];
        std::printf("%c %u - ", static_cast<int>(s), s);
        for (unsigned j = cb.length[s]; j > 0; --j) std::printf("%u", (cb.data[s] >> ((j - 1) & 31u)) & 1u);
        std::printf("\n");
    }
}

// encode.zig:221-247, the -d self-check (the loop itself: et_prefix_collisions); no newline, as there.
void check_prefix_collisions(const et_codebook &cb) {
    std::vector<uint8_t> pairs(2 * 256 * 255);
    size_t n = 0;
    if (et_prefix_collisions(&cb, pairs.data(), pairs.size() / 2, &n) != ET_OK) return;
    for (size_t k = 0; k < n; ++k)
        std::printf("Found colliding prefix codes for %u %c and %u %c", pairs[2 * k], static_cast<int>(pairs[2 * k]), pairs[2 * k + 1], static_cast<int>(pairs[2 * k + 1]));
}

// ---- --gpus N: one file over N GPUs ---------------------------------------------------------------
// One host thread per rank, each with its own et_ctx (rank r on device r modulo the devices present, so the
// path also runs on a one-GPU box); the ranks' exchange is an all-gather through this process's memory.
struct ThreadExchange {
    std::mutex m;
    std::condition_variable cv;
    int world = 1, arrived = 0, generation = 0;
    std::vector<uint8_t> slots;
    static int gather(void *user, const void *send, void *recv, size_t bytes) {
        auto *x = static_cast<std::pair<ThreadExchange *, int> *>(user);
        ThreadExchange &e = *x->first;
        std::unique_lock<std::mutex> lk(e.m);
        if (e.slots.size() < bytes * e.world) e.slots.resize(bytes * e.world);
        std::memcpy(e.slots.data() + bytes * x->second, send, bytes);
        const int gen = e.generation;
        if (++e.arrived == e.world) {  // last in: everybody's bytes are there
            e.arrived = 0;
            ++e.generation;
            e.cv.notify_all();
        } else {
            e.cv.wait(lk, [&] { return e.generation != gen; });
        }
        std::memcpy(recv, e.slots.data(), bytes * e.world);
        // nobody may overwrite a slot before all have copied: a second rendezvous
        const int gen2 = e.generation;
        if (++e.arrived == e.world) {
            e.arrived = 0;
            ++e.generation;
            e.cv.notify_all();
        } else {
            e.cv.wait(lk, [&] { return e.generation != gen2; });
        }
        return 0;
    }
};

struct RankResult {
    int rc = ET_OK;
    bool local = false;  // the failure is this rank's own (its text says what happened), not one it heard of in an exchange
    std::string err;
    size_t out_bytes = 0;  // bytes this rank contributed
    size_t in_bytes = 0;   // bytes of the input file it read
    et_codebook cb = {};
};

// Every rank makes the same group calls in the same order; the library carries each rank's status in the rows of
// every exchange, so a rank whose own preparation failed (no memory, a short read) makes its call with nothing to
// contribute and all ranks return together (include/entreepy_hip.h, "FAILURES").
int run_sharded(const Options &opt, int in_fd, int out_fd, size_t *in_size, size_t *written, et_codebook *cb_out, std::string *err) {
    struct stat_holder { off_t size; } st{::lseek(in_fd, 0, SEEK_END)};
    if (st.size < 0) { *err = "input is not a regular file"; return ET_ERR_IO; }
    const size_t file_size = static_cast<size_t>(st.size);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) { *err = "no HIP device"; return ET_ERR_HIP; }
    const int world = opt.gpus;
    ThreadExchange ex;
    ex.world = world;
    std::vector<RankResult> res(world);
    std::vector<std::thread> threads;
    const bool compress = opt.mode == Mode::Compress;
    const size_t skip = compress ? 0 : 4;
    if (!compress && file_size < 9) { *err = "file shorter than its header"; return ET_ERR_FORMAT; }
    *in_size = file_size - skip;
    // contexts and groups first, here: a rank that cannot even join is found before anybody waits for it
    std::vector<et_ctx *> ctxs(world, nullptr);
    std::vector<et_group *> grps(world, nullptr);
    std::vector<std::pair<ThreadExchange *, int>> who(world);
    int rc0 = ET_OK;
    for (int r = 0; r < world && rc0 == ET_OK; ++r) {
        who[r] = {&ex, r};
        if ((rc0 = et_ctx_create(r % n_dev, &ctxs[r])) != ET_OK) *err = "et_ctx_create";
        else if ((rc0 = et_group_create(ctxs[r], r, world, ThreadExchange::gather, &who[r], &grps[r])) != ET_OK) *err = "et_group_create";
    }
    // a cold stream's header and dictionary: read once, parsed by every rank (decode.zig:34-141)
    std::vector<uint8_t> head;
    if (rc0 == ET_OK && !compress) {
        head.resize(std::min<size_t>(file_size - 4, 8192));
        if (::pread(in_fd, head.data(), head.size(), 4) != static_cast<ssize_t>(head.size())) { rc0 = ET_ERR_IO; *err = "reading the header"; }  // main.zig:204: text_in[4..]
    }
    for (int r = 0; r < world && rc0 == ET_OK; ++r) {
        threads.emplace_back([&, r] {
            RankResult &out = res[r];
            et_ctx *ctx = ctxs[r];
            et_group *grp = grps[r];
            void *d_in = nullptr, *d_out = nullptr;
            auto bail = [&](int rc, const char *what, bool local) {
                if (out.rc != ET_OK && out.local) return;  // (this rank's own first cause stays: what the exchanges relay afterwards is its echo)
                out.rc = rc;
                out.local = local;
                out.err = std::string(what) + ": " + (*et_group_last_error(grp) ? et_group_last_error(grp) : et_last_error(ctx));
            };
            (void)hipSetDevice(r % n_dev);
            int rc = ET_OK;
            bool ok = true;
            // (tests: ET_CLI_TEST_FAIL_RANK=r makes rank r's own preparation fail as an unreadable input range would)
            static const int fail_rank = [] { const char *e = std::getenv("ET_CLI_TEST_FAIL_RANK"); return e ? std::atoi(e) : -1; }();
            if (r == fail_rank) {
                bail(ET_ERR_IO, "reading the input", true);
                ok = false;
            }
            if (compress) {
                const size_t lo = file_size * r / world, hi = file_size * (r + 1) / world, n = hi - lo;
                const size_t cap = et_encode_bound(n);
                if (ok && (hipMalloc(&d_in, n + 16) != hipSuccess || hipMalloc(&d_out, cap + 16) != hipSuccess)) { bail(ET_ERR_NOMEM, "hipMalloc", true); ok = false; }
                if (ok && (rc = et_fd_to_device(ctx, in_fd, lo, n, d_in)) != ET_OK) { bail(rc, "reading the input", true); ok = false; }
                out.in_bytes = n;
                et_shard_info info{};
                // (a rank without its buffers still makes the call: its status travels in its row of the exchange)
                rc = ok ? et_encode_sharded(grp, d_in, n, d_out, cap, &info) : et_encode_sharded(grp, nullptr, 1, nullptr, 0, &info);
                if (rc != ET_OK) {
                    if (ok) bail(rc, "et_encode_sharded", false);
                } else if ((rc = et_shard_merge_seams(grp, d_out)) != ET_OK) {
                    bail(rc, "et_shard_merge_seams", false);
                } else if (out_fd >= 0 && (rc = et_shard_write_fd(grp, d_out, out_fd)) != ET_OK) {
                    bail(rc, "et_shard_write_fd", true);
                }
                if (out.rc == ET_OK) {
                    const uint64_t lo_b = info.owned_word_lo * 4, hi_b = std::min<uint64_t>(info.owned_word_hi * 4, info.file_bytes);
                    out.out_bytes = hi_b > lo_b ? static_cast<size_t>(hi_b - lo_b) : 0;
                    et_group_codebook(grp, &out.cb);
                }
            } else {
                // this rank's window of the stream: its 8 KiB-block range with 16 bytes on either side -- not the file
                const size_t len = file_size - 4;
                uint64_t w_off = 0, w_len = 0;
                if (ok && (rc = et_decode_shard_window(head.data(), head.size(), len, r, world, &w_off, &w_len)) != ET_OK) { bail(rc, "header", true); ok = false; }
                if (ok && w_len && hipMalloc(&d_in, w_len + 32) != hipSuccess) { bail(ET_ERR_NOMEM, "hipMalloc", true); ok = false; }
                if (ok && w_len && (rc = et_fd_to_device(ctx, in_fd, 4 + w_off, w_len, d_in)) != ET_OK) { bail(rc, "reading the input", true); ok = false; }
                out.in_bytes = ok ? static_cast<size_t>(w_len) : 0;
                uint64_t n_mine = 0, first = 0;
                rc = et_decode_sharded_begin(grp, ok ? head.data() : nullptr, head.size(), len, d_in, w_off, static_cast<size_t>(w_len), ~0ull, &n_mine, &first);
                size_t wrote = 0;
                if (rc != ET_OK) {
                    if (ok) bail(rc, "et_decode_sharded", false);
                } else if (n_mine) {  // the output is sized by the rank's own share, known now
                    if (hipMalloc(&d_out, n_mine + 64) != hipSuccess) bail(ET_ERR_NOMEM, "hipMalloc", true);
                    else if ((rc = et_decode_sharded_write(grp, d_out, n_mine + 64, &wrote)) != ET_OK) bail(rc, "et_decode_sharded_write", true);
                    else if (out_fd >= 0 && wrote && (rc = et_device_to_fd(ctx, d_out, wrote, out_fd, first)) != ET_OK) bail(rc, "writing the output", true);
                }
                if (out.rc == ET_OK) out.out_bytes = wrote;
            }
            (void)hipStreamSynchronize(static_cast<hipStream_t>(et_ctx_stream(ctx)));
            if (d_in) (void)hipFree(d_in);
            if (d_out) (void)hipFree(d_out);
        });
    }
    for (auto &t : threads) t.join();
    for (et_group *g : grps)
        if (g) et_group_destroy(g);
    for (et_ctx *c : ctxs)
        if (c) et_ctx_destroy(c);
    if (rc0 != ET_OK) return rc0;
    *written = 0;
    const RankResult *bad = nullptr;
    for (const RankResult &r : res) {
        if (r.rc != ET_OK && (!bad || (r.local && !bad->local))) bad = &r;  // (the rank it happened to says it best)
        *written += r.out_bytes;
    }
    if (bad) {
        *err = bad->err;
        return bad->rc;
    }
    if (opt.debug && !compress)  // (-d: what each rank read of the input -- the dictionary once, and its own window)
        for (int r = 0; r < world; ++r) std::printf("rank %d read %zu of %zu bytes\n", r, res[r].in_bytes, file_size - 4);
    *cb_out = res[0].cb;
    return ET_OK;
}

}  // namespace

int main(int argc, char **argv) {
    Options opt;
    const int pr = parse_cli(argc, argv, opt);
    if (pr == 1) return 0;
    if (pr == 2) return 1;

    // The file goes through the library's chunked pinned-buffer pipeline (et_encode_fd /
    // et_decode_fd) instead of main.zig:34-40's read-all; only `d -p`, which also wants the
    // decoded bytes on stdout, reads it into memory the reference's way.
    const int in_fd = ::open(opt.in_path.c_str(), O_RDONLY);
    if (in_fd < 0) {
        std::fprintf(stderr, "error: FileNotFound: %s\n", opt.in_path.c_str());
        return 1;
    }
    int out_fd = -1;
    if (!opt.dry) {  // main.zig:191-197: created (truncated) before coding starts
        out_fd = ::open(opt.out_path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (out_fd < 0) {
            std::fprintf(stderr, "error: cannot create %s\n", opt.out_path.c_str());
            return 1;
        }
    }

    et_ctx *ctx = nullptr;
    int rc = et_ctx_create(0, &ctx);
    if (rc != ET_OK) {
        std::fprintf(stderr, "error: %s (no usable MI355X / HIP device; entreepy-hip has no CPU path)\n", et_strerror(rc));
        return 1;
    }

    const auto t0 = std::chrono::steady_clock::now();
    size_t written = 0, reported = 0, in_size_bytes = 0;
    std::string sharded_err;
    if (opt.gpus > 1 && !(opt.mode == Mode::Decompress && opt.print)) {
        if (opt.mode == Mode::Decompress && !magic_ok(in_fd, ctx)) {
            rc = ET_ERR_FORMAT;
        } else {
            et_codebook cb{};
            rc = run_sharded(opt, in_fd, out_fd, &in_size_bytes, &written, &cb, &sharded_err);
            if (rc == ET_OK && opt.mode == Mode::Compress) {
                if (opt.debug) {
                    dump_dictionary(cb);
                    check_prefix_collisions(cb);
                    std::printf("\nbits in output: %zu\n", written * 8);
                }
                reported = written;
            } else if (rc == ET_OK && !opt.dry) {
                reported = written;
            }
        }
    } else if (opt.mode == Mode::Compress) {
        rc = et_encode_fd(ctx, in_fd, out_fd, &in_size_bytes, &written);  // encode.zig:319: bytes land in the file
        if (rc == ET_OK) {
            if (opt.debug) {
                et_codebook cb;
                if (et_last_codebook(ctx, &cb) == ET_OK) {
                    dump_dictionary(cb);
                    check_prefix_collisions(cb);
                }
            }
            if (opt.debug) std::printf("\nbits in output: %zu\n", written * 8);  // encode.zig:320
            reported = written;  // encode.zig:331,336: counts the bytes even with -t
        }
    } else if (!magic_ok(in_fd, ctx)) {
        rc = ET_ERR_FORMAT;
    } else if (!opt.print) {
        rc = et_decode_fd(ctx, in_fd, 4, out_fd, &in_size_bytes, &written);  // main.zig:204: text_in[4..]
        if (rc == ET_OK && !opt.dry) reported = written;  // decode.zig:185-188
    } else {
        std::vector<uint8_t> in, out;
        if (!read_file(opt.in_path, in) || in.size() < 9) {
            rc = ET_ERR_FORMAT;
        } else {
            in_size_bytes = in.size() - 4;
            size_t n = 0;
            et_decoded_size(in.data() + 4, in.size() - 4, &n);
            out.resize(n + 64);
            rc = et_decode(ctx, in.data() + 4, in.size() - 4, out.data(), out.size(), &written);  // main.zig:204
            if (rc == ET_OK) {
                if (!opt.dry) {  // decode.zig:185-188
                    size_t done = 0;
                    while (done < written) {
                        const ssize_t w = ::write(out_fd, out.data() + done, written - done);
                        if (w <= 0) break;
                        done += static_cast<size_t>(w);
                    }
                    reported = written;
                }
                std::fwrite(out.data(), 1, written, stdout);  // decode.zig:189
            }
        }
    }
    if (rc != ET_OK) {
        std::fprintf(stderr, "error: %s: %s\n", et_strerror(rc),
                     !g_magic_why.empty() ? g_magic_why.c_str() : (!sharded_err.empty() ? sharded_err.c_str() : et_last_error(ctx)));
        et_ctx_destroy(ctx);
        ::close(in_fd);
        if (out_fd >= 0) ::close(out_fd);
        return 1;
    }
    // encode.zig:334 / decode.zig:217 (decode reports the compressed_text length, i.e. file - 4)
    const float in_size = static_cast<float>(in_size_bytes);
    std::fprintf(stderr, "%s => %s\n", format_file_size(in_size).c_str(), format_file_size(static_cast<float>(reported)).c_str());
    if (opt.debug) {  // encode.zig:26-28 / decode.zig:15-17 (deferred to function exit)
        const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        std::printf("time taken: %lld\xce\xbcs\n", static_cast<long long>(us));
    }
    et_ctx_destroy(ctx);
    ::close(in_fd);
    if (out_fd >= 0) ::close(out_fd);
    return 0;
}
