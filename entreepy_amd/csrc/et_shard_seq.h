// et_shard_seq.h -- one stream over the ranks of a group (include/entreepy_hip.h, "groups"): the SEQUENCE of a
// sharded encode, of the bit-offset-adjusted concatenation and of a cold decode (et_shard_seq.cpp; plain C++, no
// HIP in it) and the two things that sequence is written against:
//   Backend   what ONE rank computes on its chunk -- the staged entry points of et_api.cpp on an et_ctx
//             (et_shard_hip.cpp); the CPU tests link the same et_shard_seq.cpp against a stand-in of their own;
//   Exchange  how the ranks' small rows travel: a callback of the caller's (threads, gloo, MPI) or RCCL over xGMI.
// The reference has one thread and one buffer (encode.zig:25-337, decode.zig:13-220); nothing of this has a
// counterpart there.
//
// Failure protocol: every row a rank contributes to an exchange carries its STATUS.  A rank whose local step
// failed still makes every exchange of the call (with nothing to contribute but that status), and when the rows
// are in, all ranks return the status of the first rank that failed -- nobody leaves a collective sequence early,
// nobody waits for a rank that has gone.  A failure BEHIND a call's last exchange (an enqueue that failed) is
// returned by that rank alone and poisons its group: its later calls only take part in their exchanges, carrying
// the status, so that the peers learn of it at their next call.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "entreepy_hip.h"

namespace et_shard {

// ---- rows --------------------------------------------------------------------------------------------------------
struct HistRow {  // et_encode_sharded's one exchange
    uint64_t counts[256];  // the rank's local histogram (encode.zig:43-47 over its chunk)
    uint64_t status;       // et_status of the rank's steps so far
    uint64_t cap;          // bytes its output buffer holds: every rank can tell whether every shard fits
};
struct SeamRow {  // et_shard_merge_seams
    uint32_t first, last;  // the piece's first and last word, own bits only
    uint32_t status, pad_;
};
struct ColdRow {  // et_decode_sharded: one per round
    int64_t start, exit;  // bit at which the range's first codeword begins / the next range's (-1: the rank holds no blocks)
    uint64_t n_symbols;
    uint64_t status;
    uint64_t cap;     // symbols its output buffer holds (~0: not known yet)
    uint8_t map[32];  // codes that do not self-synchronise: exit of the range for every start
};
constexpr size_t ROW_MAX = 2080;  // bytes per rank of the largest exchange
static_assert(sizeof(HistRow) <= ROW_MAX && sizeof(HistRow) % 8 == 0, "row size");

// ---- one rank's compute ------------------------------------------------------------------------------------------
// Pointers named d_* are in the backend's memory (HBM of the rank's GPU).  Every call returns an et_status;
// last_error() has the text of the last failure.
struct Backend {
    virtual ~Backend() {}
    virtual const char *last_error() const = 0;
    // K1 over the chunk, enqueued; d_row (optional, backend memory): the 256 counts go there as well
    virtual int histogram_begin(const void *d_text, size_t n, void *d_row) = 0;
    virtual int histogram_host(uint64_t counts[256]) = 0;         // the counts of histogram_begin, waited for
    virtual int histogram_known(const uint64_t counts[256]) = 0;  // "these are they": spares the shard encode a read-back
    virtual int encode_head(const et_codebook *cb, const void *d_text, size_t n, void *d_out, size_t cap, const uint8_t *header, size_t header_len,
                            uint64_t *end_bit) = 0;
    virtual int encode_body(const et_codebook *cb, const void *d_text, size_t n, void *d_out, size_t cap, uint64_t start_bit, uint64_t *end_bit) = 0;
    // concat: word 0 and word n_words - 1 of the piece (n_words >= 1; everything enqueued before is complete on return)
    virtual int read_first_last(const void *d_out, uint64_t n_words, uint32_t first_last[2]) = 0;
    virtual int patch_word(void *d_out, uint64_t word, uint32_t value) = 0;  // complete on return
    virtual int drain() = 0;                                                  // everything enqueued is complete
    virtual int to_fd(const void *d_src, size_t len, int fd, uint64_t file_offset) = 0;
    virtual int copy(void *d_dst, const void *d_src, size_t len) = 0;  // enqueued
    // cold decode
    virtual int read_head(const void *d_src, size_t len, uint8_t *host) = 0;  // complete on return
    virtual int range_sync(const et_codebook *cb, const void *d_range, size_t range_bytes, size_t tail_bytes, int has_front, int32_t in_start_bit,
                           et_range_info *info) = 0;
    virtual int range_maps(const et_codebook *cb, const void *d_range, size_t range_bytes, size_t tail_bytes, int32_t in_start_bit, uint8_t map[32],
                           uint32_t *n_starts) = 0;
    virtual int range_resolve(uint32_t in_start_bit, et_range_info *info) = 0;
    virtual int range_write(uint64_t max_symbols, void *d_out, size_t cap, size_t *out_len) = 0;
};

// ---- how rows travel ---------------------------------------------------------------------------------------------
struct Exchange {
    virtual ~Exchange() {}
    virtual const char *last_error() const = 0;
    // all-gather of `bytes` (<= ROW_MAX) host bytes per rank; ET_OK or ET_ERR_RCCL (then nothing more can be agreed on)
    virtual int allgather(const void *send, void *recv, size_t bytes) = 0;
    // The histogram rows.  The default waits for the backend's counts on the host and all-gathers the rows; a
    // transport that reads device memory (RCCL) gathers straight from d_row(), where histogram_begin left the counts.
    virtual void *d_row() { return nullptr; }
    virtual int gather_hist(Backend *be, uint64_t status, uint64_t cap, HistRow *rows, int world);
    // Bulk: every rank's owned words [olo, ohi) of the image to root's d_image (words[q] = {piece_lo, piece_hi, owned_lo,
    // owned_hi}); self: also send/receive this rank's own words to itself (test mode).  RCCL only.
    virtual bool moves_bulk() const { return false; }
    virtual int gather_words(const uint64_t (*words)[4], int rank, int world, int root, const void *d_out, void *d_image, bool self) { return ET_ERR_UNSUPPORTED; }
    virtual void set_timeout_ms(int64_t) {}
};

// The caller's transport (et_group_create).
struct CallbackExchange : Exchange {
    et_allgather_fn fn;
    void *user;
    std::string err;
    CallbackExchange(et_allgather_fn f, void *u) : fn(f), user(u) {}
    const char *last_error() const override { return err.c_str(); }
    int allgather(const void *send, void *recv, size_t bytes) override;
};

}  // namespace et_shard

struct et_group {
    et_shard::Backend *be = nullptr;  // owned
    et_shard::Exchange *xc = nullptr;  // owned
    int rank = 0, world = 1;
    bool force = false;    // ET_GROUP_FORCE_COLLECTIVES: a group of one takes the transport's path all the same
    int poison = ET_OK;    // see the failure protocol above
    std::string err;

    // the plan of the last et_encode_sharded
    bool have_plan = false, seams_merged = false;
    bool seams_exchanged = false;  // the merge's exchange of this plan has been made (the same on every rank, whatever a rank's own patch did afterwards)
    et_codebook cb = {};
    std::vector<uint64_t> starts;  // world + 1 file bit offsets
    std::vector<uint8_t> header;
    uint64_t text_len = 0;
    et_shard_info info = {};

    // et_decode_sharded_begin -> et_decode_sharded_write
    bool cold_ready = false;
    uint64_t cold_take = 0;
};

namespace et_shard {
// A new group around a backend and an exchange (both owned by it from here on, also on failure).
int group_new(Backend *be, Exchange *xc, int rank, int world, et_group **out);
}  // namespace et_shard
