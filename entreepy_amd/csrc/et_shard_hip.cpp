// et_shard_hip.cpp -- the GPU side of a group (et_shard_seq.h): one rank's compute on its et_ctx, and RCCL over xGMI
// as the transport of the ranks' rows.  The sequence itself is et_shard_seq.cpp.
//
// RCCL is loaded on first use (no link-time dependency: a host without librccl.so still loads the library), and a
// process that already holds one -- PyTorch ships its own -- keeps using that one.
#include "et_kernels.h"
#include "et_shard_seq.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

namespace {

using et_shard::HistRow;
using et_shard::ROW_MAX;

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------
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
    int (*CommAbort)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why, bound;  // bound: which object the symbols come from
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // An RCCL that is already in the process first (RTLD_NOLOAD: by the sonames it is loaded under -- PyTorch's own
        // is librccl.so.1 --; a second, possibly different RCCL beside torch's communicators is what must not happen),
        // then the search path, then ROCm's.
        const char *loaded[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : loaded) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.handle) {
                r.bound = std::string(n) + " (already loaded)";
                break;
            }
        }
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (size_t i = 0; !r.handle && i < sizeof names / sizeof *names; ++i) {
            r.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
            if (r.handle) r.bound = names[i];
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
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
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

// ---- one rank's compute: the staged entry points of et_api.cpp on its et_ctx ------------------------------------------
struct HipBackend : et_shard::Backend {
    et_ctx *ctx;
    int device;
    uint32_t *h_words = nullptr;  // pinned: 4 words
    std::string err;

    explicit HipBackend(et_ctx *c) : ctx(c), device(et_ctx_device(c)) {
        DeviceGuard guard(device);
        if (hipHostMalloc(reinterpret_cast<void **>(&h_words), 16) != hipSuccess) h_words = nullptr;
    }
    ~HipBackend() override {
        if (h_words) (void)hipHostFree(h_words);  // (the ctx is the caller's, and may be gone already: not touched here)
    }
    hipStream_t stream() const { return static_cast<hipStream_t>(et_ctx_stream(ctx)); }
    const char *last_error() const override { return err.empty() ? et_last_error(ctx) : err.c_str(); }
    int of(int rc) {  // a ctx call's status: its text is the ctx's
        err.clear();
        return rc;
    }
    int hip(hipError_t e, const char *what) {
        if (e == hipSuccess) return ET_OK;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return ET_ERR_HIP;
    }

    int histogram_begin(const void *d_text, size_t n, void *d_row) override { return of(et_histogram_device(ctx, d_text, n, d_row)); }
    int histogram_host(uint64_t counts[256]) override { return of(et_histogram_host(ctx, counts)); }
    int histogram_known(const uint64_t counts[256]) override { return of(et_histogram_on_host(ctx, counts)); }
    int encode_head(const et_codebook *cb, const void *d_text, size_t n, void *d_out, size_t cap, const uint8_t *header, size_t header_len, uint64_t *end_bit) override {
        return of(et_encode_head_shard_device(ctx, cb, d_text, n, d_out, cap, header, header_len, end_bit));
    }
    int encode_body(const et_codebook *cb, const void *d_text, size_t n, void *d_out, size_t cap, uint64_t start_bit, uint64_t *end_bit) override {
        return of(et_encode_body_device(ctx, cb, d_text, n, d_out, cap, start_bit, end_bit));
    }
    int read_first_last(const void *d_out, uint64_t n_words, uint32_t fl[2]) override {
        if (!h_words) return hip(hipErrorOutOfMemory, "pinned words");
        DeviceGuard guard(device);
        hipStream_t s = stream();
        int rc = hip(hipMemcpyAsync(h_words, d_out, 4, hipMemcpyDeviceToHost, s), "hipMemcpyAsync");
        if (rc == ET_OK) rc = hip(hipMemcpyAsync(h_words + 1, static_cast<const uint8_t *>(d_out) + (n_words - 1) * 4, 4, hipMemcpyDeviceToHost, s), "hipMemcpyAsync");
        if (rc == ET_OK) rc = hip(hipStreamSynchronize(s), "hipStreamSynchronize");
        fl[0] = h_words[0];
        fl[1] = h_words[1];
        return rc;
    }
    int patch_word(void *d_out, uint64_t word, uint32_t value) override {
        if (!h_words) return hip(hipErrorOutOfMemory, "pinned words");
        DeviceGuard guard(device);
        hipStream_t s = stream();
        h_words[2] = value;
        int rc = hip(hipMemcpyAsync(static_cast<uint8_t *>(d_out) + word * 4, h_words + 2, 4, hipMemcpyHostToDevice, s), "hipMemcpyAsync");
        if (rc == ET_OK) rc = hip(hipStreamSynchronize(s), "hipStreamSynchronize");  // (h_words is used again by the next call)
        return rc;
    }
    int drain() override {
        DeviceGuard guard(device);
        return hip(hipStreamSynchronize(stream()), "hipStreamSynchronize");
    }
    int to_fd(const void *d_src, size_t len, int fd, uint64_t off) override { return of(et_device_to_fd(ctx, d_src, len, fd, off)); }
    int copy(void *d_dst, const void *d_src, size_t len) override {
        DeviceGuard guard(device);
        return hip(hipMemcpyAsync(d_dst, d_src, len, hipMemcpyDeviceToDevice, stream()), "hipMemcpyAsync");
    }
    int read_head(const void *d_src, size_t len, uint8_t *host) override {
        DeviceGuard guard(device);
        hipStream_t s = stream();
        int rc = hip(hipMemcpyAsync(host, d_src, len, hipMemcpyDeviceToHost, s), "hipMemcpyAsync");
        if (rc == ET_OK) rc = hip(hipStreamSynchronize(s), "hipStreamSynchronize");
        return rc;
    }
    int range_sync(const et_codebook *cb, const void *d_range, size_t range_bytes, size_t tail_bytes, int has_front, int32_t in_start_bit, et_range_info *info) override {
        return of(et_decode_range_sync(ctx, cb, d_range, range_bytes, tail_bytes, has_front, in_start_bit, info));
    }
    int range_maps(const et_codebook *cb, const void *d_range, size_t range_bytes, size_t tail_bytes, int32_t in_start_bit, uint8_t map[32], uint32_t *n_starts) override {
        return of(et_decode_range_maps(ctx, cb, d_range, range_bytes, tail_bytes, in_start_bit, map, n_starts));
    }
    int range_resolve(uint32_t in_start_bit, et_range_info *info) override { return of(et_decode_range_resolve(ctx, in_start_bit, info)); }
    int range_write(uint64_t max_symbols, void *d_out, size_t cap, size_t *out_len) override { return of(et_decode_range_write(ctx, max_symbols, d_out, cap, out_len)); }
};

// ---- RCCL over xGMI ---------------------------------------------------------------------------------------------------
// Rows are small (<= 2 KiB per rank): latency-bound, one collective each.  The histogram rows are gathered straight
// from device memory (K1's reduction stores the counts into d_send as it stores them everywhere else), and a
// one-workgroup kernel behind the collective stores the gathered rows into pinned memory and then a "done" word, which
// the calling thread polls: no copy command, no stream wait between K1 and the host's code construction.
struct RcclExchange : et_shard::Exchange {
    HipBackend *be;  // (not owned: the group's)
    int rank, world;
    ncclComm_t comm = nullptr;
    bool wedged = false;  // a collective was left unfinished: the communicator is aborted, not destroyed
    void *d_send = nullptr, *d_recv = nullptr;
    uint8_t *h_send = nullptr, *h_recv = nullptr;
    uint64_t epoch = 0;                        // the word behind h_send == epoch: the gathered rows are in h_recv
    uint64_t tail_on_device[2] = {~0ull, ~0ull};  // the {status, cap} words d_send's row holds
    double timeout_ms = 600000.0;
    std::string err;

    RcclExchange(HipBackend *b, int r, int w) : be(b), rank(r), world(w) {}
    ~RcclExchange() override {
        DeviceGuard guard(be->device);
        if (comm) (void)(wedged ? rccl().CommAbort(comm) : rccl().CommDestroy(comm));
        if (d_send) (void)hipFree(d_send);
        if (d_recv) (void)hipFree(d_recv);
        if (h_send) (void)hipHostFree(h_send);
        if (h_recv) (void)hipHostFree(h_recv);
    }
    const char *last_error() const override { return err.c_str(); }
    void set_timeout_ms(int64_t ms) override { timeout_ms = static_cast<double>(ms); }
    void *d_row() override { return d_send; }
    bool moves_bulk() const override { return true; }

    int hip(hipError_t e, const char *what) {
        if (e == hipSuccess) return ET_OK;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return ET_ERR_HIP;
    }
    int nccl(int e, const char *what) {
        if (e == ncclSuccess) return ET_OK;
        err = std::string(what) + ": " + rccl().GetErrorString(e);
        return ET_ERR_RCCL;
    }
    int init(const uint8_t id[ET_RCCL_ID_BYTES]) {
        DeviceGuard guard(be->device);
        int rc = hip(hipMalloc(&d_send, ROW_MAX), "hipMalloc");
        if (rc == ET_OK) rc = hip(hipMalloc(&d_recv, ROW_MAX * world), "hipMalloc");
        if (rc == ET_OK) rc = hip(hipHostMalloc(reinterpret_cast<void **>(&h_send), ROW_MAX + 16), "hipHostMalloc");  // (+ the gather's "done" word)
        if (rc == ET_OK) rc = hip(hipHostMalloc(reinterpret_cast<void **>(&h_recv), ROW_MAX * world), "hipHostMalloc");
        if (rc != ET_OK) return rc;
        std::memset(h_send, 0, ROW_MAX + 16);
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof u);
        return nccl(rccl().CommInitRank(&comm, world, u, rank), "ncclCommInitRank");
    }
    // The rows are behind a collective on the stream: poll the word the kernel behind it stores.  A peer that never
    // makes the call leaves the collective -- and the stream -- unfinished: after timeout_ms this rank gives up on the
    // communicator (the caller destroys the group).
    int wait_rows(hipStream_t s, volatile uint64_t *done, uint64_t want) {
        const double w0 = now_ms();
        double next_look = w0 + 2000.0;  // (the word normally appears within microseconds: the stream is only looked at when it does not)
        for (uint32_t spin = 0; *done != want; ++spin) {
            if ((spin & 1023u) != 1023u) continue;
            const double t = now_ms();
            if (t < next_look) continue;
            next_look = t + 50.0;
            if (hipStreamQuery(s) == hipSuccess && *done != want) {  // (the stream ran dry without the word: an error on it)
                err = "the gathered rows never reached the host";
                return ET_ERR_RCCL;
            }
            if (t - w0 > timeout_ms) {
                wedged = true;
                err = "a rank of the group never made this exchange (timeout)";
                return ET_ERR_RCCL;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return ET_OK;
    }
    // d_send[0 .. bytes) -> every rank's d_recv -> h_recv, polled
    int gather_from_device(size_t bytes) {
        hipStream_t s = be->stream();
        int rc = nccl(rccl().AllGather(d_send, d_recv, bytes / 8, ncclUint64, comm, s), "ncclAllGather");
        if (rc != ET_OK) return rc;
        volatile uint64_t *done = reinterpret_cast<volatile uint64_t *>(h_send + ROW_MAX);  // (a word of its own behind the row)
        const uint64_t want = ++epoch;
        et::launch_words_to_host(s, d_recv, static_cast<uint32_t>(bytes / 4) * static_cast<uint32_t>(world), h_recv,
                                 const_cast<unsigned long long *>(reinterpret_cast<volatile unsigned long long *>(done)), want);
        if ((rc = hip(hipGetLastError(), "k_words_to_host")) != ET_OK) return rc;
        return wait_rows(s, done, want);
    }
    int allgather(const void *send, void *recv, size_t bytes) override {
        if (bytes > ROW_MAX || (bytes & 7)) {
            err = "row size";
            return ET_ERR_ARG;
        }
        DeviceGuard guard(be->device);
        std::memcpy(h_send, send, bytes);
        int rc = hip(hipMemcpyAsync(d_send, h_send, bytes, hipMemcpyHostToDevice, be->stream()), "hipMemcpyAsync");
        tail_on_device[0] = tail_on_device[1] = ~0ull;  // (the row's words are somebody else's now)
        if (rc == ET_OK) rc = gather_from_device(bytes);
        if (rc == ET_OK) std::memcpy(recv, h_recv, bytes * world);
        return rc;
    }
    int gather_hist(et_shard::Backend *, uint64_t status, uint64_t cap, HistRow *rows, int) override {
        DeviceGuard guard(be->device);
        // the counts are in d_send already (histogram_begin's d_row, stream-ordered); status and cap behind them, when
        // they are not what the row holds from the step before
        if (tail_on_device[0] != status || tail_on_device[1] != cap) {
            uint64_t *h_tail = reinterpret_cast<uint64_t *>(h_send) + 256;
            h_tail[0] = status;
            h_tail[1] = cap;
            int rc = hip(hipMemcpyAsync(static_cast<uint8_t *>(d_send) + offsetof(HistRow, status), h_tail, 16, hipMemcpyHostToDevice, be->stream()), "hipMemcpyAsync");
            if (rc == ET_OK) rc = hip(hipStreamSynchronize(be->stream()), "hipStreamSynchronize");  // (h_tail may be written again at once; rare: the words seldom change)
            if (rc != ET_OK) return rc;
            tail_on_device[0] = status;
            tail_on_device[1] = cap;
        }
        const int rc = gather_from_device(sizeof(HistRow));
        if (rc == ET_OK) std::memcpy(rows, h_recv, sizeof(HistRow) * static_cast<size_t>(world));
        return rc;
    }
    int gather_words(const uint64_t (*words)[4], int me, int n_ranks, int root, const void *d_out, void *d_image, bool self) override {
        DeviceGuard guard(be->device);
        hipStream_t s = be->stream();
        // every send and receive is posted, whatever one of them says, and the group is closed: nobody is left inside it
        int first = ET_OK;
        auto keep = [&](int rc) {
            if (rc != ET_OK && first == ET_OK) first = rc;
        };
        keep(nccl(rccl().GroupStart(), "ncclGroupStart"));
        for (int q = 0; q < n_ranks; ++q) {
            const uint64_t plo = words[q][0], olo = words[q][2], ohi = words[q][3];
            const size_t n_words = static_cast<size_t>(ohi - olo);
            if (!n_words) continue;
            if (q == me) {
                const uint8_t *src = static_cast<const uint8_t *>(d_out) + (olo - plo) * 4;
                if (q == root && !self) keep(hip(hipMemcpyAsync(static_cast<uint8_t *>(d_image) + olo * 4, src, n_words * 4, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync"));
                else keep(nccl(rccl().Send(src, n_words, ncclUint32, root, comm, s), "ncclSend"));
            }
            if (me == root && (q != me || self)) keep(nccl(rccl().Recv(static_cast<uint8_t *>(d_image) + olo * 4, n_words, ncclUint32, q, comm, s), "ncclRecv"));
        }
        keep(nccl(rccl().GroupEnd(), "ncclGroupEnd"));
        if (first != ET_OK) {
            wedged = true;
            return first;
        }
        // (bulk: 0.6 GB per rank at 1 GiB of text; the wait is the stream's, bounded by the same patience)
        const double w0 = now_ms();
        for (;;) {
            const hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) return ET_OK;
            if (e != hipErrorNotReady) return hip(e, "hipStreamQuery");
            if (now_ms() - w0 > timeout_ms) {
                wedged = true;
                err = "a rank of the group never joined the gather (timeout)";
                return ET_ERR_RCCL;
            }
            std::this_thread::yield();
        }
    }
};

}  // namespace

// -----------------------------------------------------------------------------------------------------------------------
extern "C" int et_rccl_unique_id(uint8_t id[ET_RCCL_ID_BYTES]) {
    if (!id) return ET_ERR_ARG;
    Rccl &r = rccl();
    if (!r.ok) return ET_ERR_RCCL;
    ncclUniqueId u;
    if (r.GetUniqueId(&u) != ncclSuccess) return ET_ERR_RCCL;
    std::memcpy(id, &u, sizeof u);
    return ET_OK;
}

extern "C" const char *et_rccl_library(void) {
    Rccl &r = rccl();
    return r.ok ? r.bound.c_str() : r.why.c_str();
}

extern "C" int et_group_create(et_ctx *ctx, int rank, int world, et_allgather_fn allgather, void *user, et_group **out) {
    if (!ctx || !out || (world > 1 && !allgather)) return ET_ERR_ARG;
    return et_shard::group_new(new (std::nothrow) HipBackend(ctx), new (std::nothrow) et_shard::CallbackExchange(allgather, user), rank, world, out);
}

extern "C" int et_group_create_rccl(et_ctx *ctx, int rank, int world, const uint8_t id[ET_RCCL_ID_BYTES], et_group **out) {
    if (!ctx || !out || !id || world < 1 || rank < 0 || rank >= world) return ET_ERR_ARG;
    *out = nullptr;
    if (!rccl().ok) return ET_ERR_RCCL;
    HipBackend *be = new (std::nothrow) HipBackend(ctx);
    RcclExchange *xc = be ? new (std::nothrow) RcclExchange(be, rank, world) : nullptr;
    if (!xc) {
        delete be;
        return ET_ERR_NOMEM;
    }
    const int rc = xc->init(id);
    if (rc != ET_OK) {
        delete xc;
        delete be;
        return rc == ET_ERR_HIP ? ET_ERR_HIP : ET_ERR_RCCL;
    }
    return et_shard::group_new(be, xc, rank, world, out);
}
