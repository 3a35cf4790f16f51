// et_sharded.cpp -- one stream over several GPUs behind the C ABI (include/entreepy_hip.h, "groups").
//
// The reference encodes one text with one code table into one image (encode.zig:43-47 histogram,
// :54-214 table, :303-319 body + a single writeAll).  A group of GPUs does the same for a text that
// is split into contiguous chunks, one per rank:
//   encode   K1 on the local chunk -> ONE exchange (all-gather of the 256 x u64 local histograms; their
//            sum is the histogram of encode.zig:43-47, each row gives a shard's bit count) -> the same
//            code table and header on every rank (et_plan_shards) -> K2 + K4 at the shard's bit offset.
//   concat   the bit-offset-adjusted concatenation of encode.zig:319's image: a 32-bit word two shards
//            share belongs to the first of them; et_shard_merge_seams hands that owner the bits of its
//            successors (one tiny exchange of first/last words), after which the pieces are disjoint
//            word ranges that go to a file (pwrite per shard) or to one GPU's image (RCCL send/recv
//            over xGMI, or a device copy for ranks that share an address space).
//   decode   a cold .et stream: ranges cut at multiples of 8 KiB, every rank synchronises its range,
//            one exchange of (start, exit, symbols) -- or of the 32-byte exit maps for codes that do
//            not self-synchronise --, repair where a start is not the predecessor's exit, write.
//
// Everything here is sequencing over the staged entry points of et_api.cpp; the exchange is a callback
// (plain host buffers: MPI, gloo, threads ...) or RCCL, which is loaded on first use (no link-time
// dependency: a host without librccl.so still loads the library).
#include "entreepy_hip.h"
#include "et_kernels.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

// ---- RCCL, bound at run time ---------------------------------------------------------------------
// (types as in <rccl/rccl.h>; only what is used)
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
static_assert(sizeof(ncclUniqueId) == ET_RCCL_ID_BYTES, "unique id size");
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1, ncclUint32 = 3, ncclUint64 = 5 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // A process that already holds an RCCL (PyTorch ships its own) keeps using that one; else the ROCm one.
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.why = "librccl.so not found";
            return;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(r.handle, name);
            if (!p && r.why.empty()) r.why = std::string("librccl.so lacks ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.why.empty();
    });
    return r;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct et_group {
    et_ctx *ctx = nullptr;
    int rank = 0, world = 1, device = 0;
    et_allgather_fn allgather = nullptr;
    void *user = nullptr;
    ncclComm_t comm = nullptr;
    std::string err;

    // exchange staging: device (RCCL) and pinned host
    void *d_send = nullptr, *d_recv = nullptr;  // 2 KiB / world x 2 KiB
    uint64_t gather_epoch = 0;                  // (the word behind h_send == gather_epoch: the gathered histograms are in h_recv)
    uint8_t *h_send = nullptr, *h_recv = nullptr;

    // the plan of the last et_encode_sharded
    bool have_plan = false, seams_merged = false;
    et_codebook cb = {};
    std::vector<uint64_t> starts;  // world + 1 file bit offsets
    std::vector<uint8_t> header;
    uint64_t text_len = 0;
    et_shard_info info = {};
};

namespace {

int fail(et_group *g, int status, const std::string &what) {
    if (g) g->err = what;
    return status;
}

#define ETG_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(g, ET_ERR_HIP, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)
#define ETG_TRY(expr)                                                             \
    do {                                                                          \
        int rc_ = (expr);                                                         \
        if (rc_ != ET_OK) {                                                       \
            if (g->err.empty()) g->err = et_last_error(g->ctx);                   \
            return rc_;                                                           \
        }                                                                         \
    } while (0)
#define ETG_NCCL(call)                                                                            \
    do {                                                                                          \
        int e_ = (call);                                                                          \
        if (e_ != ncclSuccess) return fail(g, ET_ERR_RCCL, std::string(#call ": ") + rccl().GetErrorString(e_)); \
    } while (0)

constexpr size_t EXCHANGE_MAX = 2048;  // bytes per rank of the largest exchange (the histogram)

hipStream_t stream_of(et_group *g) { return static_cast<hipStream_t>(et_ctx_stream(g->ctx)); }

// All-gather of `bytes` (<= EXCHANGE_MAX) host bytes per rank: send -> recv[world x bytes].  The stream is
// drained when this returns.
int exchange(et_group *g, const void *send, void *recv, size_t bytes) {
    if (g->world == 1) {
        std::memcpy(recv, send, bytes);
        return ET_OK;
    }
    if (g->comm) {
        hipStream_t s = stream_of(g);
        std::memcpy(g->h_send, send, bytes);
        ETG_HIP(hipMemcpyAsync(g->d_send, g->h_send, bytes, hipMemcpyHostToDevice, s));
        ETG_NCCL(rccl().AllGather(g->d_send, g->d_recv, bytes, ncclUint8, g->comm, s));
        ETG_HIP(hipMemcpyAsync(g->h_recv, g->d_recv, bytes * g->world, hipMemcpyDeviceToHost, s));
        ETG_HIP(hipStreamSynchronize(s));
        std::memcpy(recv, g->h_recv, bytes * g->world);
        return ET_OK;
    }
    if (g->allgather(g->user, send, recv, bytes) != 0) return fail(g, ET_ERR_RCCL, "the exchange callback failed");
    return ET_OK;
}

// File words of rank r's local buffer [piece) and the words it contributes to the image [owned): a word
// several ranks share belongs to the first of them.
void shard_words(const std::vector<uint64_t> &starts, int r, uint64_t *piece_lo, uint64_t *piece_hi, uint64_t *owned_lo, uint64_t *owned_hi) {
    const uint64_t s = starts[r], e = starts[r + 1];
    *piece_lo = r == 0 ? 0 : s / 32;
    *piece_hi = (e + 31) / 32;
    if (*piece_hi < *piece_lo) *piece_hi = *piece_lo;
    *owned_lo = r == 0 ? 0 : (s + 31) / 32;
    *owned_hi = (e + 31) / 32;
    if (*owned_hi < *owned_lo) *owned_hi = *owned_lo;
}

}  // namespace

extern "C" int et_shard_words(const uint64_t *start_bits, uint32_t world, uint32_t rank, uint64_t words[4]) {
    if (!start_bits || !words || rank >= world) return ET_ERR_ARG;
    const std::vector<uint64_t> starts(start_bits, start_bits + world + 1);
    shard_words(starts, static_cast<int>(rank), &words[0], &words[1], &words[2], &words[3]);
    return ET_OK;
}

// The word that closes rank `rank`'s owned range, with the bits of every later shard that begins in it.
// first_last: per rank {its first word, its last word} as they stand in its own buffer (own bits only).
extern "C" int et_seam_word(const uint64_t *start_bits, uint32_t world, uint32_t rank, const uint32_t *first_last, uint32_t *merged,
                            int *has_seam) {
    if (!start_bits || !first_last || !merged || !has_seam || rank >= world) return ET_ERR_ARG;
    *has_seam = 0;
    *merged = 0;
    const uint64_t s = start_bits[rank], e = start_bits[rank + 1];
    const bool holds = e > s || rank == 0;  // (the head shard holds the header even without body bits)
    if (!holds) return ET_OK;
    const uint64_t last_word = (e + 31) / 32;  // one past the last word this rank touches
    if (last_word == 0 || (e & 31) == 0) return ET_OK;  // ends on a word boundary: nothing is shared
    const uint64_t w = last_word - 1;
    // this rank owns w only if no earlier rank reaches into it
    if (rank > 0 && s / 32 == w && (s & 31) != 0) return ET_OK;
    uint32_t word = first_last[2 * rank + 1];
    for (uint32_t q = rank + 1; q < world; ++q) {
        if (start_bits[q] / 32 != w) break;                      // q begins in a later word (starts only grow)
        if (start_bits[q + 1] > start_bits[q]) word |= first_last[2 * q];  // a shard without bits has nothing to give
    }
    *merged = word;
    *has_seam = 1;
    return ET_OK;
}

// ---------------------------------------------------------------------------------------------------
extern "C" int et_rccl_unique_id(uint8_t id[ET_RCCL_ID_BYTES]) {
    if (!id) return ET_ERR_ARG;
    Rccl &r = rccl();
    if (!r.ok) return ET_ERR_RCCL;
    ncclUniqueId u;
    if (r.GetUniqueId(&u) != ncclSuccess) return ET_ERR_RCCL;
    std::memcpy(id, &u, sizeof u);
    return ET_OK;
}

namespace {

int group_alloc(et_group *g) {
    ETG_HIP(hipMalloc(&g->d_send, EXCHANGE_MAX));
    ETG_HIP(hipMalloc(&g->d_recv, EXCHANGE_MAX * g->world));
    ETG_HIP(hipHostMalloc(reinterpret_cast<void **>(&g->h_send), EXCHANGE_MAX + 8));  // (+ the gather's "done" word)
    std::memset(g->h_send, 0, EXCHANGE_MAX + 8);
    ETG_HIP(hipHostMalloc(reinterpret_cast<void **>(&g->h_recv), EXCHANGE_MAX * g->world));
    return ET_OK;
}

int group_new(et_ctx *ctx, int rank, int world, et_group **out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return ET_ERR_ARG;
    *out = nullptr;
    et_group *g = new (std::nothrow) et_group();
    if (!g) return ET_ERR_NOMEM;
    g->ctx = ctx;
    g->rank = rank;
    g->world = world;
    g->device = et_ctx_device(ctx);
    DeviceGuard guard(g->device);
    const int rc = group_alloc(g);
    if (rc != ET_OK) {
        et_group_destroy(g);
        return rc;
    }
    *out = g;
    return ET_OK;
}

}  // namespace

extern "C" int et_group_create(et_ctx *ctx, int rank, int world, et_allgather_fn allgather, void *user, et_group **out) {
    if (world > 1 && !allgather) return ET_ERR_ARG;
    const int rc = group_new(ctx, rank, world, out);
    if (rc != ET_OK) return rc;
    (*out)->allgather = allgather;
    (*out)->user = user;
    return ET_OK;
}

extern "C" int et_group_create_rccl(et_ctx *ctx, int rank, int world, const uint8_t id[ET_RCCL_ID_BYTES], et_group **out) {
    if (!id) return ET_ERR_ARG;
    Rccl &r = rccl();
    if (!r.ok) return ET_ERR_RCCL;
    const int rc = group_new(ctx, rank, world, out);
    if (rc != ET_OK) return rc;
    et_group *g = *out;
    DeviceGuard guard(g->device);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    if (r.CommInitRank(&g->comm, world, u, rank) != ncclSuccess) {
        et_group_destroy(g);
        *out = nullptr;
        return ET_ERR_RCCL;
    }
    return ET_OK;
}

extern "C" void et_group_destroy(et_group *g) {
    if (!g) return;
    DeviceGuard guard(g->device);
    if (g->ctx) (void)hipStreamSynchronize(stream_of(g));
    if (g->comm) (void)rccl().CommDestroy(g->comm);
    if (g->d_send) (void)hipFree(g->d_send);
    if (g->d_recv) (void)hipFree(g->d_recv);
    if (g->h_send) (void)hipHostFree(g->h_send);
    if (g->h_recv) (void)hipHostFree(g->h_recv);
    delete g;
}

extern "C" const char *et_group_last_error(const et_group *g) { return g ? g->err.c_str() : ""; }

extern "C" int et_group_codebook(const et_group *g, et_codebook *cb) {
    if (!g || !cb || !g->have_plan) return ET_ERR_ARG;
    *cb = g->cb;
    return ET_OK;
}

extern "C" int et_group_start_bits(const et_group *g, uint64_t *start_bits) {
    if (!g || !start_bits || !g->have_plan) return ET_ERR_ARG;
    std::memcpy(start_bits, g->starts.data(), g->starts.size() * sizeof(uint64_t));
    return ET_OK;
}

// ---------------------------------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------------------------------
extern "C" int et_encode_sharded(et_group *g, const void *d_text, size_t n, void *d_out, size_t cap, et_shard_info *info) {
    if (!g || !d_out || !info || (n && !d_text)) return ET_ERR_ARG;
    g->err.clear();
    g->have_plan = g->seams_merged = false;
    DeviceGuard guard(g->device);
    hipStream_t s = stream_of(g);
    const int world = g->world, r = g->rank;
    // (1) local histogram, (2) the one exchange
    std::vector<uint64_t> hists(static_cast<size_t>(world) * 256);
    ETG_TRY(et_histogram_device(g->ctx, d_text, n, nullptr));  // (the counts stay in the ctx: on the device, and on their way into pinned host memory)
    const double t0 = now_ms();
    if (g->comm && world > 1) {
        // RCCL gathers straight from the ctx's device copy; one small kernel stores the world x 2 KiB into pinned
        // memory and then a "done" word, which this thread polls (no copy command, no stream wait)
        const void *d_local = nullptr;
        ETG_TRY(et_histogram_device_ptr(g->ctx, &d_local));
        ETG_NCCL(rccl().AllGather(d_local, g->d_recv, 256, ncclUint64, g->comm, s));
        volatile uint64_t *done = reinterpret_cast<volatile uint64_t *>(g->h_send + EXCHANGE_MAX);  // (a word of its own behind the exchange buffer)
        const uint64_t epoch = ++g->gather_epoch;
        et::launch_words_to_host(s, g->d_recv, 512u * static_cast<uint32_t>(world), g->h_recv,
                                 const_cast<unsigned long long *>(reinterpret_cast<volatile unsigned long long *>(done)), epoch);
        ETG_HIP(hipGetLastError());
        const double w0 = now_ms();
        for (uint32_t spin = 0; *done != epoch; ++spin)
            if ((spin & 1023u) == 1023u && now_ms() - w0 > 2000.0) {  // (a collective: another rank may be late)
                ETG_HIP(hipStreamSynchronize(s));
                if (*done != epoch) return fail(g, ET_ERR_RCCL, "the gathered histograms never reached the host");
            }
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(hists.data(), g->h_recv, 2048 * static_cast<size_t>(world));
    } else {
        // the local counts are polled out of the ctx's pinned copy (no read-back)
        ETG_TRY(et_histogram_host(g->ctx, reinterpret_cast<uint64_t *>(g->h_send)));
        if (world == 1) std::memcpy(hists.data(), g->h_send, 2048);
        else if (g->allgather(g->user, g->h_send, hists.data(), 2048) != 0) return fail(g, ET_ERR_RCCL, "the exchange callback failed");
    }
    const double t1 = now_ms();
    // (3) the same plan on every rank
    g->starts.assign(world + 1, 0);
    g->header.assign(8192, 0);
    size_t header_len = 0;
    int rc = et_plan_shards(hists.data(), static_cast<uint32_t>(world), &g->cb, g->header.data(), g->header.size(), &header_len, g->starts.data());
    if (rc != ET_OK) return fail(g, rc, rc == ET_ERR_EMPTY ? "empty input" : "et_plan_shards");
    g->header.resize(header_len);
    g->text_len = 0;
    for (uint64_t c : hists) g->text_len += c;
    const double t2 = now_ms();
    // (4) this rank's shard at its bit offset; its row of the exchange spares the shard encode a read-back
    if (n) ETG_TRY(et_histogram_on_host(g->ctx, hists.data() + static_cast<size_t>(r) * 256));
    uint64_t end = 0, local_start = 0;
    if (r == 0) {
        local_start = g->starts[0];
        ETG_TRY(et_encode_head_shard_device(g->ctx, &g->cb, d_text, n, d_out, cap, g->header.data(), header_len, &end));
    } else {
        local_start = g->starts[r] % 32;
        ETG_TRY(et_encode_body_device(g->ctx, &g->cb, d_text, n, d_out, cap, local_start, &end));
    }
    if (end - local_start != g->starts[r + 1] - g->starts[r]) return fail(g, ET_ERR_HIP, "shard bit count differs from the plan");
    et_shard_info &o = g->info;
    o = et_shard_info{};
    o.start_bit = g->starts[r];
    o.end_bit = g->starts[r + 1];
    o.local_start_bit = local_start;
    o.header_len = r == 0 ? header_len : 0;
    o.file_bytes = (g->starts[world] + 7) / 8;
    o.text_len = g->text_len;
    shard_words(g->starts, r, &o.piece_word_lo, &o.piece_word_hi, &o.owned_word_lo, &o.owned_word_hi);
    o.exchange_ms = static_cast<float>(t1 - t0);
    o.plan_ms = static_cast<float>(t2 - t1);
    g->have_plan = true;
    *info = o;
    return ET_OK;
}

// ---------------------------------------------------------------------------------------------------
// concat
// ---------------------------------------------------------------------------------------------------
extern "C" int et_shard_merge_seams(et_group *g, void *d_out) {
    if (!g || !d_out) return ET_ERR_ARG;
    if (!g->have_plan) return fail(g, ET_ERR_ARG, "et_shard_merge_seams needs et_encode_sharded first");
    if (g->seams_merged) return ET_OK;
    DeviceGuard guard(g->device);
    hipStream_t s = stream_of(g);
    const int world = g->world, r = g->rank;
    const et_shard_info &o = g->info;
    const double t0 = now_ms();
    // this rank's first and last word, own bits only (a shard without bits gives zeros)
    uint32_t mine[2] = {0, 0};
    const bool holds = o.end_bit > o.start_bit || r == 0;
    const uint64_t n_words = o.piece_word_hi - o.piece_word_lo;
    uint32_t *h = reinterpret_cast<uint32_t *>(g->h_send);
    if (holds && n_words) {
        ETG_HIP(hipMemcpyAsync(h, d_out, 4, hipMemcpyDeviceToHost, s));
        ETG_HIP(hipMemcpyAsync(h + 1, static_cast<const uint8_t *>(d_out) + (n_words - 1) * 4, 4, hipMemcpyDeviceToHost, s));
        ETG_HIP(hipStreamSynchronize(s));
        mine[0] = h[0];
        mine[1] = h[1];
    } else {
        ETG_HIP(hipStreamSynchronize(s));
    }
    std::vector<uint32_t> all(2 * static_cast<size_t>(world));
    ETG_TRY(exchange(g, mine, all.data(), sizeof mine));
    uint32_t merged = 0;
    int has = 0;
    ETG_TRY(et_seam_word(g->starts.data(), static_cast<uint32_t>(world), static_cast<uint32_t>(r), all.data(), &merged, &has));
    if (has && merged != mine[1]) {
        h[2] = merged;
        ETG_HIP(hipMemcpyAsync(static_cast<uint8_t *>(d_out) + (n_words - 1) * 4, h + 2, 4, hipMemcpyHostToDevice, s));
        ETG_HIP(hipStreamSynchronize(s));  // (h is reused by the next exchange)
    }
    g->info.seam_ms = static_cast<float>(now_ms() - t0);
    g->seams_merged = true;
    return ET_OK;
}

namespace {

// Bytes [lo, hi) of the file this rank contributes, and where they sit in its buffer.
void owned_bytes(const et_group *g, uint64_t *file_lo, uint64_t *file_hi, uint64_t *local_off) {
    const et_shard_info &o = g->info;
    *file_lo = o.owned_word_lo * 4;
    *file_hi = o.owned_word_hi * 4;
    if (*file_hi > o.file_bytes) *file_hi = o.file_bytes;  // the image ends with the body's last byte, not its last word
    if (*file_hi < *file_lo) *file_hi = *file_lo;
    *local_off = (o.owned_word_lo - o.piece_word_lo) * 4;
}

}  // namespace

extern "C" int et_shard_write_fd(et_group *g, const void *d_out, int fd) {
    if (!g || !d_out || fd < 0) return ET_ERR_ARG;
    if (!g->have_plan || !g->seams_merged) return fail(g, ET_ERR_ARG, "et_shard_write_fd needs et_encode_sharded and et_shard_merge_seams first");
    uint64_t lo, hi, off;
    owned_bytes(g, &lo, &hi, &off);
    const double t0 = now_ms();
    ETG_TRY(et_device_to_fd(g->ctx, static_cast<const uint8_t *>(d_out) + off, static_cast<size_t>(hi - lo), fd, lo));
    g->info.concat_ms = static_cast<float>(now_ms() - t0);
    return ET_OK;
}

extern "C" int et_shard_place(et_group *g, const void *d_out, void *d_image, size_t cap) {
    if (!g || !d_out || !d_image) return ET_ERR_ARG;
    if (!g->have_plan || !g->seams_merged) return fail(g, ET_ERR_ARG, "et_shard_place needs et_encode_sharded and et_shard_merge_seams first");
    if (cap < g->info.file_bytes) return fail(g, ET_ERR_CAP, "image buffer too small");
    DeviceGuard guard(g->device);
    uint64_t lo, hi, off;
    owned_bytes(g, &lo, &hi, &off);
    if (hi > lo) ETG_HIP(hipMemcpyAsync(static_cast<uint8_t *>(d_image) + lo, static_cast<const uint8_t *>(d_out) + off, hi - lo, hipMemcpyDeviceToDevice, stream_of(g)));
    return ET_OK;
}

extern "C" int et_shard_gather(et_group *g, const void *d_out, void *d_image, size_t cap, int root) {
    if (!g || !d_out || root < 0 || root >= g->world) return ET_ERR_ARG;
    if (!g->have_plan || !g->seams_merged) return fail(g, ET_ERR_ARG, "et_shard_gather needs et_encode_sharded and et_shard_merge_seams first");
    if (g->rank == root && (!d_image || cap < ((g->info.file_bytes + 3) & ~static_cast<uint64_t>(3)))) return fail(g, ET_ERR_CAP, "image buffer too small (file bytes rounded up to a word)");
    if (g->world == 1) return et_shard_place(g, d_out, d_image, cap);
    if (!g->comm) return fail(g, ET_ERR_UNSUPPORTED, "et_shard_gather moves data with RCCL: create the group with et_group_create_rccl (or use et_shard_place / et_shard_write_fd)");
    DeviceGuard guard(g->device);
    hipStream_t s = stream_of(g);
    const double t0 = now_ms();
    // whole owned words travel (the image's last word may carry up to 3 pad bytes: cap is checked for them)
    ETG_NCCL(rccl().GroupStart());
    for (int q = 0; q < g->world; ++q) {
        uint64_t plo, phi, olo, ohi;
        shard_words(g->starts, q, &plo, &phi, &olo, &ohi);
        const size_t words = static_cast<size_t>(ohi - olo);
        if (!words) continue;
        if (q == g->rank) {
            const uint8_t *src = static_cast<const uint8_t *>(d_out) + (olo - plo) * 4;
            if (q == root) ETG_HIP(hipMemcpyAsync(static_cast<uint8_t *>(d_image) + olo * 4, src, words * 4, hipMemcpyDeviceToDevice, s));
            else ETG_NCCL(rccl().Send(src, words, ncclUint32, root, g->comm, s));
        } else if (g->rank == root) {
            ETG_NCCL(rccl().Recv(static_cast<uint8_t *>(d_image) + olo * 4, words, ncclUint32, q, g->comm, s));
        }
    }
    ETG_NCCL(rccl().GroupEnd());
    ETG_HIP(hipStreamSynchronize(s));
    g->info.concat_ms = static_cast<float>(now_ms() - t0);
    return ET_OK;
}

extern "C" int et_group_last_info(const et_group *g, et_shard_info *info) {
    if (!g || !info || !g->have_plan) return ET_ERR_ARG;
    *info = g->info;
    return ET_OK;
}

// ---------------------------------------------------------------------------------------------------
// decode of one cold stream (decode.zig:13-220 walks it serially; here every rank takes a range)
// ---------------------------------------------------------------------------------------------------
extern "C" int et_decode_sharded(et_group *g, const void *d_compressed, size_t len, void *d_out, size_t cap, size_t *written,
                                 uint64_t *first_index) {
    if (!g || !d_compressed || !written || !first_index) return ET_ERR_ARG;
    *written = 0;
    *first_index = 0;
    g->err.clear();
    if (len < 5) return fail(g, ET_ERR_FORMAT, "stream shorter than its header");
    DeviceGuard guard(g->device);
    hipStream_t s = stream_of(g);
    const int world = g->world, r = g->rank;
    // header and dictionary: parsed on the host, by every rank
    std::vector<uint8_t> head(len < 8192 ? len : 8192);
    ETG_HIP(hipMemcpyAsync(head.data(), d_compressed, head.size(), hipMemcpyDeviceToHost, s));
    ETG_HIP(hipStreamSynchronize(s));
    et_codebook cb;
    uint64_t n_symbols = 0;
    size_t body_off = 0;
    int rc = et_parse_header(head.data(), head.size(), &cb, &n_symbols, &body_off);
    if (rc != ET_OK) return fail(g, rc, "et_parse_header");
    if (body_off > len) return fail(g, ET_ERR_FORMAT, "dictionary runs past the end of the stream");
    // the body from its 4-byte aligned base, cut into blocks of 8 KiB; a last block shorter than the 16-byte
    // run-out a range needs after it is not a block of its own
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_compressed) + body_off;
    const uint8_t *base = reinterpret_cast<const uint8_t *>(a & ~static_cast<uintptr_t>(3));
    const uint32_t first_bit = static_cast<uint32_t>(a & 3) * 8;
    const uint64_t stream_bytes = (a & 3) + (len - body_off);
    uint64_t n_blocks = (stream_bytes + 8191) / 8192;
    if (n_blocks > 1 && stream_bytes - (n_blocks - 1) * 8192 < 16) --n_blocks;
    const uint64_t lo_b = static_cast<uint64_t>(r) * n_blocks / world, hi_b = static_cast<uint64_t>(r + 1) * n_blocks / world;
    const uint64_t begin = lo_b * 8192, end = hi_b == n_blocks ? stream_bytes : hi_b * 8192;
    const bool active = hi_b > lo_b && cb.n_coded > 0 && n_symbols > 0;
    const bool has_front = begin >= 16;  // (the 16 bytes before a later range are stream bytes)
    et_range_info info = {};
    const bool exhaustive = cb.n_coded > 2 && cb.max_length <= cb.min_length + 1;
    if (exhaustive) {
        // codes that do not self-synchronise: exit maps over every possible start, chained from the stream's start
        uint8_t mine_map[32];
        for (int i = 0; i < 32; ++i) mine_map[i] = static_cast<uint8_t>(i);  // a rank without blocks passes the start on
        uint32_t n_starts = 0;
        if (active) ETG_TRY(et_decode_range_maps(g->ctx, &cb, base + begin, end - begin, stream_bytes - end, lo_b == 0 ? static_cast<int32_t>(first_bit) : -1, mine_map, &n_starts));
        std::vector<uint8_t> maps(32 * static_cast<size_t>(world));
        ETG_TRY(exchange(g, mine_map, maps.data(), 32));
        uint32_t s_in = first_bit;
        for (int q = 0; q < r; ++q) s_in = maps[static_cast<size_t>(q) * 32 + s_in];
        if (active) ETG_TRY(et_decode_range_resolve(g->ctx, s_in, &info));
    } else if (active) {
        ETG_TRY(et_decode_range_sync(g->ctx, &cb, base + begin, end - begin, stream_bytes - end, has_front ? 1 : 0, lo_b == 0 ? static_cast<int32_t>(first_bit) : -1, &info));
    }
    // agree on the seams: every active rank's start must be the exit of the active rank before it
    std::vector<int64_t> table(3 * static_cast<size_t>(world));
    for (int round = 0;; ++round) {
        const int64_t mine[3] = {active ? static_cast<int64_t>(info.start_bit) : -1, active ? static_cast<int64_t>(info.exit_bit) : -1,
                                 static_cast<int64_t>(info.n_symbols)};
        ETG_TRY(exchange(g, mine, table.data(), sizeof mine));
        int64_t prev_exit = first_bit, want_mine = -1;
        bool any_wrong = false;
        for (int q = 0; q < world; ++q) {
            if (table[3 * q] < 0) continue;
            if (table[3 * q] != prev_exit) {
                any_wrong = true;
                if (q == r) want_mine = prev_exit;
            }
            prev_exit = table[3 * q + 1];
        }
        if (!any_wrong) break;
        if (round > world) return fail(g, ET_ERR_HIP, "cold decode did not settle");
        if (want_mine >= 0) ETG_TRY(et_decode_range_sync(g->ctx, &cb, base + begin, end - begin, stream_bytes - end, has_front ? 1 : 0, static_cast<int32_t>(want_mine), &info));
    }
    uint64_t first = 0;
    for (int q = 0; q < r; ++q) first += static_cast<uint64_t>(table[3 * q + 2]);
    const uint64_t mine_n = static_cast<uint64_t>(table[3 * r + 2]);
    const uint64_t take = first >= n_symbols ? 0 : (mine_n < n_symbols - first ? mine_n : n_symbols - first);
    *first_index = first;
    if (active && take) {
        if (!d_out) return ET_ERR_ARG;
        ETG_TRY(et_decode_range_write(g->ctx, take, d_out, cap, written));
    }
    return ET_OK;
}
