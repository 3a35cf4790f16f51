// et_io.cpp -- see et_io.h.
#include "et_io.h"

#include <errno.h>
#include <string.h>
#include <unistd.h>

#include <atomic>

namespace et_io {

Pool::Pool(int n) {
    for (int i = 1; i < n; ++i) threads_.emplace_back([this, i] { worker(i); });
}

Pool::~Pool() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_start_.notify_all();
    for (auto &t : threads_) t.join();
}

void Pool::worker(int id) {
    uint64_t seen = 0;
    for (;;) {
        const std::function<void(int, int)> *job;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_start_.wait(lk, [&] { return stop_ || epoch_ != seen; });
            if (stop_) return;
            seen = epoch_;
            job = job_;
        }
        (*job)(id, size());
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) cv_done_.notify_one();
        }
    }
}

void Pool::run(const std::function<void(int, int)> &fn) {
    if (!threads_.empty()) {
        std::lock_guard<std::mutex> lk(mu_);
        job_ = &fn;
        pending_ = static_cast<int>(threads_.size());
        ++epoch_;
    }
    cv_start_.notify_all();
    fn(0, size());  // the caller is worker 0
    if (!threads_.empty()) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
    }
}

Pipe::~Pipe() {
    delete pool_;
    for (int i = 0; i < 2; ++i) {
        if (pin_[i]) (void)hipHostFree(pin_[i]);
        if (ev_[i]) (void)hipEventDestroy(ev_[i]);
    }
}

bool Pipe::init(size_t chunk_bytes, int threads) {
    chunk_ = chunk_bytes;
    for (int i = 0; i < 2; ++i) {
        if (hipHostMalloc(reinterpret_cast<void **>(&pin_[i]), chunk_) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming) != hipSuccess) return false;
    }
    pool_ = new Pool(threads < 1 ? 1 : threads);
    return true;
}

// One chunk between the host end and a pinned buffer, split over the pool (slices of
// whole 4 KiB pages).  Files: pread/pwrite per slice, looping over short transfers.
bool Pipe::move(const HostEnd &end, uint8_t *pinned, size_t off, size_t len, bool to_pinned) {
    std::atomic<bool> ok{true};
    pool_->run([&](int w, int nw) {
        size_t per = ((len + nw - 1) / nw + 4095) & ~static_cast<size_t>(4095);
        const size_t lo = static_cast<size_t>(w) * per;
        if (lo >= len) return;
        const size_t hi = lo + per < len ? lo + per : len;
        if (end.fd < 0) {
            if (to_pinned) memcpy(pinned + lo, end.ptr + off + lo, hi - lo);
            else memcpy(end.ptr + off + lo, pinned + lo, hi - lo);
            return;
        }
        size_t done = lo;
        while (done < hi) {
            const off_t at = static_cast<off_t>(end.offset + off + done);
            const ssize_t r = to_pinned ? pread(end.fd, pinned + done, hi - done, at) : pwrite(end.fd, pinned + done, hi - done, at);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) {  // error, or the file is shorter than it was when we sized it
                ok = false;
                return;
            }
            done += static_cast<size_t>(r);
        }
    });
    return ok;
}

int Pipe::upload(hipStream_t stream, void *d_dst, const HostEnd &src, size_t n, uint8_t *peek, size_t peek_cap) {
    size_t k = 0;
    for (size_t off = 0; off < n; off += chunk_, ++k) {
        const int b = static_cast<int>(k & 1);
        const size_t len = n - off < chunk_ ? n - off : chunk_;
        if (k >= 2 && (last_hip = hipEventSynchronize(ev_[b])) != hipSuccess) return -2;  // its last DMA has left the buffer
        if (!move(src, pin_[b], off, len, true)) return -1;
        if (off == 0 && peek && peek_cap) memcpy(peek, pin_[b], len < peek_cap ? len : peek_cap);
        if ((last_hip = hipMemcpyAsync(static_cast<uint8_t *>(d_dst) + off, pin_[b], len, hipMemcpyHostToDevice, stream)) != hipSuccess) return -2;
        if ((last_hip = hipEventRecord(ev_[b], stream)) != hipSuccess) return -2;
    }
    // the staging buffers may be refilled by the next transfer: make them safe to touch
    for (int b = 0; b < 2 && b < static_cast<int>(k); ++b)
        if ((last_hip = hipEventSynchronize(ev_[b])) != hipSuccess) return -2;
    return 0;
}

int Pipe::download(hipStream_t stream, const HostEnd &dst, const void *d_src, size_t n) {
    const size_t n_chunks = (n + chunk_ - 1) / chunk_;
    auto issue = [&](size_t k) -> bool {
        const size_t off = k * chunk_, len = n - off < chunk_ ? n - off : chunk_;
        const int b = static_cast<int>(k & 1);
        if ((last_hip = hipMemcpyAsync(pin_[b], static_cast<const uint8_t *>(d_src) + off, len, hipMemcpyDeviceToHost, stream)) != hipSuccess) return false;
        return (last_hip = hipEventRecord(ev_[b], stream)) == hipSuccess;
    };
    for (size_t k = 0; k < n_chunks && k < 2; ++k)
        if (!issue(k)) return -2;
    for (size_t k = 0; k < n_chunks; ++k) {
        const size_t off = k * chunk_, len = n - off < chunk_ ? n - off : chunk_;
        const int b = static_cast<int>(k & 1);
        if ((last_hip = hipEventSynchronize(ev_[b])) != hipSuccess) return -2;
        if (!move(dst, pin_[b], off, len, false)) return -1;
        if (k + 2 < n_chunks && !issue(k + 2)) return -2;
    }
    return 0;
}

}  // namespace et_io
