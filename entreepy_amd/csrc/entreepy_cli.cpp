// entreepy_cli.cpp -- the `entreepy` command: same surface as the reference's
// src/main.zig (options -h -p -t -d, commands c/d, -o), with encode()/decode()
// replaced by libentreepy_hip.so.  The reference's host is Zig; no Zig toolchain
// exists in this image, so the tested host driver is C++ (INTEGRATION.md has the Zig
// shim a maintainer would use instead).
#include "entreepy_hip.h"

#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
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
    enum { Normal, OutPath, InPath } state = Normal;
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
    if (opt.mode == Mode::Compress) {
        rc = et_encode_fd(ctx, in_fd, out_fd, &in_size_bytes, &written);  // encode.zig:319: bytes land in the file
        if (rc == ET_OK) {
            if (opt.debug) {
                et_codebook cb;
                if (et_last_codebook(ctx, &cb) == ET_OK) dump_dictionary(cb);
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
        std::fprintf(stderr, "error: %s: %s\n", et_strerror(rc), g_magic_why.empty() ? et_last_error(ctx) : g_magic_why.c_str());
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
