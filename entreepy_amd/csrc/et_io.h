// et_io.h -- chunked pinned-buffer pipeline between host memory / files and HBM
// (replaces the reference's read-all / write-all, main.zig:34-40,186,192-197
// and the single writeAll of encode.zig:319).
//
// A transfer moves through two pinned staging buffers: while the DMA engine copies chunk k
// over PCIe, a small thread pool fills (memcpy / pread) or drains (memcpy / pwrite) chunk
// k+1.  Why not hand the caller's pointer to hipMemcpy: from a COLD pageable buffer (a file
// just read, a fresh malloc) the runtime manages 5 GB/s host-to-device on this box, from
// pinned memory 57 GB/s; one CPU thread copies 21 GB/s into pinned memory, four 60 GB/s
// (tools/probe/pcie_probe.cpp, profiles/r01_pcie_probe.txt).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace et_io {

// Fork-join pool: run(fn) calls fn(worker, n_workers) on every worker and returns when all are done.
class Pool {
  public:
    explicit Pool(int n);
    ~Pool();
    int size() const { return static_cast<int>(threads_.size()) + 1; }
    void run(const std::function<void(int, int)> &fn);

  private:
    void worker(int id);
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_start_, cv_done_;
    const std::function<void(int, int)> *job_ = nullptr;
    uint64_t epoch_ = 0;
    int pending_ = 0;
    bool stop_ = false;
};

// Where the host side of a transfer lives: memory (ptr) or a file (fd, byte offset).
struct HostEnd {
    uint8_t *ptr = nullptr;
    int fd = -1;
    uint64_t offset = 0;
};

class Pipe {
  public:
    Pipe() = default;
    ~Pipe();
    bool init(size_t chunk_bytes, int threads);  // after hipSetDevice
    size_t chunk() const { return chunk_; }
    // host -> device, n bytes; enqueued on `stream`, returns when the last chunk has been
    // handed to the stream (NOT when it has arrived).  `peek`, if given, receives the first
    // min(n, peek_cap) bytes as they pass through the staging buffer.  0 ok, -1 I/O error, -2 HIP error.
    int upload(hipStream_t stream, void *d_dst, const HostEnd &src, size_t n, uint8_t *peek = nullptr, size_t peek_cap = 0);
    // device -> host, n bytes; returns when everything is in place on the host side.
    int download(hipStream_t stream, const HostEnd &dst, const void *d_src, size_t n);
    hipError_t last_hip = hipSuccess;

  private:
    bool move(const HostEnd &end, uint8_t *pinned, size_t off, size_t len, bool to_pinned);
    uint8_t *pin_[2] = {nullptr, nullptr};
    hipEvent_t ev_[2] = {nullptr, nullptr};
    size_t chunk_ = 0;
    Pool *pool_ = nullptr;
};

}  // namespace et_io
