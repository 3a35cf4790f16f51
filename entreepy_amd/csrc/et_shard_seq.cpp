// et_shard_seq.cpp -- one stream over several ranks: the sequence (see et_shard_seq.h for what it is written against
// and for the failure protocol).  Plain C++: compiles without HIP; the CPU tests build this very file against a
// stand-in backend (tests/support/shard_cpu.cpp).
//
// The reference encodes one text with one code table into one image (encode.zig:43-47 histogram, :54-214 table,
// :303-319 body + a single writeAll).  A group does the same for a text split into contiguous chunks, one per rank:
//   encode   K1 on the local chunk -> ONE exchange (all-gather of the 256 x u64 local histograms; their sum is the
//            histogram of encode.zig:43-47, each row gives a shard's bit count) -> the same code table and header on
//            every rank (et_plan_shards) -> K2 + K4 at the shard's bit offset.
//   concat   the bit-offset-adjusted concatenation of encode.zig:319's image: a 32-bit word two shards share belongs
//            to the first of them; et_shard_merge_seams hands that owner the bits of its successors (one exchange of
//            first/last words), after which the pieces are disjoint word ranges that go to a file (pwrite per shard)
//            or to one GPU's image (RCCL send/recv over xGMI, or a device copy within one address space).
//   decode   a cold .et stream: ranges cut at multiples of 8 KiB, every rank synchronises its range, one exchange of
//            (start, exit, symbols) -- or of the 32-byte exit maps for codes that do not self-synchronise --, repair
//            where a start is not the predecessor's exit, write.
#include "et_shard_seq.h"

#include <chrono>
#include <cstring>

using et_shard::Backend;
using et_shard::ColdRow;
using et_shard::Exchange;
using et_shard::HistRow;
using et_shard::SeamRow;

namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int fail(et_group *g, int status, const std::string &what) {
    if (g) g->err = what;
    return status;
}

// A step of this rank's backend failed: its text, the status.
int backend_failed(et_group *g, int status, const char *step) {
    g->err = std::string(step) + ": " + g->be->last_error();
    return status;
}

// The same failure, behind the call's last exchange: the peers cannot hear of it in this call any more.
int poisoned(et_group *g, int status, const char *step) {
    g->poison = status;
    return backend_failed(g, status, step);
}

// All-gather of one row per rank.  A group of one copies -- unless it was asked to take the transport's path.
int gather(et_group *g, const void *send, void *recv, size_t bytes) {
    if (g->world == 1 && !g->force) {
        std::memcpy(recv, send, bytes);
        return ET_OK;
    }
    const int rc = g->xc->allgather(send, recv, bytes);
    if (rc != ET_OK) g->err = std::string("exchange: ") + g->xc->last_error();
    return rc;
}

// What all ranks return once the rows are in: the status of the first rank that failed.  The failing rank keeps its
// own text; the others name it.
template <typename Row>
int settle(et_group *g, const Row *rows, const char *call) {
    for (int q = 0; q < g->world; ++q) {
        const int st = static_cast<int>(rows[q].status);
        if (st == ET_OK) continue;
        if (q != g->rank || g->err.empty()) g->err = std::string(call) + ": rank " + std::to_string(q) + " failed: " + et_strerror(st);
        return st;
    }
    return ET_OK;
}

// File words of rank r's local buffer [piece) and the words it contributes to the image [owned): a word several ranks
// share belongs to the first of them.
void shard_words(const uint64_t *starts, int r, uint64_t *piece_lo, uint64_t *piece_hi, uint64_t *owned_lo, uint64_t *owned_hi) {
    const uint64_t s = starts[r], e = starts[r + 1];
    *piece_lo = r == 0 ? 0 : s / 32;
    *piece_hi = (e + 31) / 32;
    if (*piece_hi < *piece_lo) *piece_hi = *piece_lo;
    *owned_lo = r == 0 ? 0 : (s + 31) / 32;
    *owned_hi = (e + 31) / 32;
    if (*owned_hi < *owned_lo) *owned_hi = *owned_lo;
}

// Bytes of rank q's output buffer its piece needs (what the shard encode checks before it packs): the head shard's
// header padded to words when it has no body bits, one zeroed word for any other shard without bits.
uint64_t piece_bytes(const uint64_t *starts, int q, uint64_t header_len) {
    const uint64_t bits = starts[q + 1] - starts[q];
    if (bits == 0) return q == 0 ? ((header_len + 3) / 4) * 4 : 4;
    const uint64_t local_start = q == 0 ? starts[0] : starts[q] % 32;
    return ((local_start + bits + 31) / 32) * 4;
}

}  // namespace

// ---- the default histogram exchange and the caller's transport ----------------------------------------------------
int et_shard::Exchange::gather_hist(Backend *be, uint64_t status, uint64_t cap, HistRow *rows, int world) {
    HistRow mine;
    std::memset(&mine, 0, sizeof mine);
    if (status == ET_OK) {
        const int rc = be->histogram_host(mine.counts);  // the local counts are polled out of pinned memory (no read-back)
        if (rc != ET_OK) {
            status = static_cast<uint64_t>(rc);
            std::memset(mine.counts, 0, sizeof mine.counts);
        }
    }
    mine.status = status;
    mine.cap = cap;
    (void)world;
    return allgather(&mine, rows, sizeof mine);
}

int et_shard::CallbackExchange::allgather(const void *send, void *recv, size_t bytes) {
    if (!fn) {
        err = "the group has no exchange callback";
        return ET_ERR_RCCL;
    }
    if (fn(user, send, recv, bytes) != 0) {
        err = "the exchange callback failed";
        return ET_ERR_RCCL;
    }
    return ET_OK;
}

int et_shard::group_new(Backend *be, Exchange *xc, int rank, int world, et_group **out) {
    et_group *g = (be && xc && out && world >= 1 && rank >= 0 && rank < world) ? new (std::nothrow) et_group() : nullptr;
    if (!g) {
        delete be;
        delete xc;
        if (out) *out = nullptr;
        return (be && xc && out) ? (world >= 1 && rank >= 0 && rank < world ? ET_ERR_NOMEM : ET_ERR_ARG) : ET_ERR_ARG;
    }
    g->be = be;
    g->xc = xc;
    g->rank = rank;
    g->world = world;
    *out = g;
    return ET_OK;
}

// ---- plain accessors ----------------------------------------------------------------------------------------------
extern "C" void et_group_destroy(et_group *g) {
    if (!g) return;
    delete g->xc;  // (before the backend: a transport may hold memory of the backend's device)
    delete g->be;
    delete g;
}

extern "C" const char *et_group_last_error(const et_group *g) { return g ? g->err.c_str() : ""; }

extern "C" int et_group_set_option(et_group *g, int option, int64_t value) {
    if (!g) return ET_ERR_ARG;
    switch (option) {
        case ET_GROUP_FORCE_COLLECTIVES: g->force = value != 0; return ET_OK;
        case ET_GROUP_TIMEOUT_MS:
            if (value < 1) return ET_ERR_ARG;
            g->xc->set_timeout_ms(value);
            return ET_OK;
        default: return ET_ERR_ARG;
    }
}

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

extern "C" int et_group_last_info(const et_group *g, et_shard_info *info) {
    if (!g || !info || !g->have_plan) return ET_ERR_ARG;
    *info = g->info;
    return ET_OK;
}

extern "C" int et_shard_words(const uint64_t *start_bits, uint32_t world, uint32_t rank, uint64_t words[4]) {
    if (!start_bits || !words || rank >= world) return ET_ERR_ARG;
    shard_words(start_bits, static_cast<int>(rank), &words[0], &words[1], &words[2], &words[3]);
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
        if (start_bits[q] / 32 != w) break;                                // q begins in a later word (starts only grow)
        if (start_bits[q + 1] > start_bits[q]) word |= first_last[2 * q];  // a shard without bits has nothing to give
    }
    *merged = word;
    *has_seam = 1;
    return ET_OK;
}

// -------------------------------------------------------------------------------------------------------------------
// encode
// -------------------------------------------------------------------------------------------------------------------
extern "C" int et_encode_sharded(et_group *g, const void *d_text, size_t n, void *d_out, size_t cap, et_shard_info *info) {
    if (!g || !info) return ET_ERR_ARG;
    g->err.clear();
    g->have_plan = g->seams_merged = g->seams_exchanged = false;
    const int world = g->world, r = g->rank;
    // (1) what can be checked here, then the local histogram -- whatever comes of it, the exchange is made
    int st = g->poison;
    if (st != ET_OK) g->err = "the group failed in an earlier call";
    if (st == ET_OK && (!d_out || (n && !d_text))) st = fail(g, ET_ERR_ARG, "et_encode_sharded: null buffer");
    if (st == ET_OK && (reinterpret_cast<uintptr_t>(d_out) & 3)) st = fail(g, ET_ERR_ARG, "et_encode_sharded: d_out must be 4-byte aligned");
    const bool device_rows = world > 1 || g->force;  // (a group of one has nobody to gather from)
    if (st == ET_OK) {
        st = g->be->histogram_begin(d_text, n, device_rows ? g->xc->d_row() : nullptr);  // (the counts stay with the backend: on the device, and on their way into pinned host memory)
        if (st != ET_OK) backend_failed(g, st, "histogram");
    }
    // (2) the one exchange
    std::vector<HistRow> rows(static_cast<size_t>(world));
    const double t0 = now_ms();
    int rc;
    if (device_rows) {
        rc = g->xc->gather_hist(g->be, static_cast<uint64_t>(st), cap, rows.data(), world);
        if (rc != ET_OK) return fail(g, rc, std::string("exchange: ") + g->xc->last_error());
    } else {
        std::memset(rows.data(), 0, sizeof(HistRow));
        if (st == ET_OK && (st = g->be->histogram_host(rows[0].counts)) != ET_OK) backend_failed(g, st, "histogram");
        rows[0].status = static_cast<uint64_t>(st);
        rows[0].cap = cap;
    }
    const double t1 = now_ms();
    if ((rc = settle(g, rows.data(), "et_encode_sharded")) != ET_OK) return rc;
    // (3) the same plan on every rank
    std::vector<uint64_t> hists(static_cast<size_t>(world) * 256);
    for (int q = 0; q < world; ++q) std::memcpy(hists.data() + static_cast<size_t>(q) * 256, rows[q].counts, sizeof rows[q].counts);
    g->starts.assign(world + 1, 0);
    g->header.assign(8192, 0);
    size_t header_len = 0;
    rc = et_plan_shards(hists.data(), static_cast<uint32_t>(world), &g->cb, g->header.data(), g->header.size(), &header_len, g->starts.data());
    if (rc != ET_OK) return fail(g, rc, rc == ET_ERR_EMPTY ? "empty input" : "et_plan_shards");  // (from the same rows: the same on every rank)
    g->header.resize(header_len);
    g->text_len = 0;
    for (uint64_t c : hists) g->text_len += c;
    // ... and every rank can tell whether every rank's piece fits its buffer (a shard full of symbols that are rare in
    // the whole text packs to MORE than its own bytes): all return, or none
    for (int q = 0; q < world; ++q)
        if (piece_bytes(g->starts.data(), q, header_len) > rows[q].cap)
            return fail(g, ET_ERR_CAP, "et_encode_sharded: rank " + std::to_string(q) + "'s piece (" + std::to_string(piece_bytes(g->starts.data(), q, header_len)) +
                                           " bytes) does not fit its buffer (" + std::to_string(rows[q].cap) + ")");
    const double t2 = now_ms();
    // (4) this rank's shard at its bit offset; its row of the exchange spares the shard encode a read-back
    if (n && (rc = g->be->histogram_known(rows[r].counts)) != ET_OK) return poisoned(g, rc, "histogram");
    uint64_t end = 0, local_start = 0;
    if (r == 0) {
        local_start = g->starts[0];
        rc = g->be->encode_head(&g->cb, d_text, n, d_out, cap, g->header.data(), header_len, &end);
    } else {
        local_start = g->starts[r] % 32;
        rc = g->be->encode_body(&g->cb, d_text, n, d_out, cap, local_start, &end);
    }
    if (rc != ET_OK) return poisoned(g, rc, "shard encode");
    if (end - local_start != g->starts[r + 1] - g->starts[r]) {
        g->poison = ET_ERR_HIP;
        return fail(g, ET_ERR_HIP, "shard bit count differs from the plan");
    }
    et_shard_info &o = g->info;
    o = et_shard_info{};
    o.start_bit = g->starts[r];
    o.end_bit = g->starts[r + 1];
    o.local_start_bit = local_start;
    o.header_len = r == 0 ? header_len : 0;
    o.file_bytes = (g->starts[world] + 7) / 8;
    o.text_len = g->text_len;
    shard_words(g->starts.data(), r, &o.piece_word_lo, &o.piece_word_hi, &o.owned_word_lo, &o.owned_word_hi);
    o.exchange_ms = static_cast<float>(t1 - t0);
    o.plan_ms = static_cast<float>(t2 - t1);
    g->have_plan = true;
    *info = o;
    return ET_OK;
}

// -------------------------------------------------------------------------------------------------------------------
// concat
// -------------------------------------------------------------------------------------------------------------------
extern "C" int et_shard_merge_seams(et_group *g, void *d_out) {
    if (!g) return ET_ERR_ARG;
    // A repeated call makes no exchange -- on ANY rank: what decides is whether this plan's exchange has been made, which all
    // ranks know alike, not whether this rank's own patch behind it went through (a rank whose patch failed is poisoned and
    // says so here; deciding by its own success it would walk into an all-gather that nobody else joins).
    if (g->have_plan && g->seams_exchanged) {
        if (g->seams_merged) return ET_OK;
        return fail(g, g->poison != ET_OK ? g->poison : ET_ERR_HIP, "the seam word was never patched: the group failed in an earlier call");
    }
    g->err.clear();
    const int world = g->world, r = g->rank;
    const et_shard_info &o = g->info;
    const double t0 = now_ms();
    int st = g->poison;
    if (st != ET_OK) g->err = "the group failed in an earlier call";
    if (st == ET_OK && !g->have_plan) st = fail(g, ET_ERR_ARG, "et_shard_merge_seams needs et_encode_sharded first");
    if (st == ET_OK && !d_out) st = fail(g, ET_ERR_ARG, "et_shard_merge_seams: null buffer");
    // this rank's first and last word, own bits only (a shard without bits gives zeros)
    SeamRow mine = {0, 0, 0, 0};
    const bool holds = st == ET_OK && (o.end_bit > o.start_bit || r == 0);
    const uint64_t n_words = st == ET_OK ? o.piece_word_hi - o.piece_word_lo : 0;
    if (st == ET_OK) {
        if (holds && n_words) {
            uint32_t fl[2] = {0, 0};
            if ((st = g->be->read_first_last(d_out, n_words, fl)) != ET_OK) backend_failed(g, st, "reading the piece's first and last word");
            mine.first = fl[0];
            mine.last = fl[1];
        } else if ((st = g->be->drain()) != ET_OK) {
            backend_failed(g, st, "waiting for the shard encode");
        }
    }
    mine.status = static_cast<uint32_t>(st);
    std::vector<SeamRow> rows(static_cast<size_t>(world));
    int rc = gather(g, &mine, rows.data(), sizeof mine);
    if (rc != ET_OK) return rc;
    if ((rc = settle(g, rows.data(), "et_shard_merge_seams")) != ET_OK) return rc;
    g->seams_exchanged = true;
    std::vector<uint32_t> all(2 * static_cast<size_t>(world));
    for (int q = 0; q < world; ++q) {
        all[2 * q] = rows[q].first;
        all[2 * q + 1] = rows[q].last;
    }
    uint32_t merged = 0;
    int has = 0;
    et_seam_word(g->starts.data(), static_cast<uint32_t>(world), static_cast<uint32_t>(r), all.data(), &merged, &has);
    if (has && merged != mine.last && (rc = g->be->patch_word(d_out, n_words - 1, merged)) != ET_OK) return poisoned(g, rc, "patching the seam word");
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

// (rank-local: no exchange)
extern "C" int et_shard_write_fd(et_group *g, const void *d_out, int fd) {
    if (!g || !d_out || fd < 0) return ET_ERR_ARG;
    if (!g->have_plan || !g->seams_merged) return fail(g, ET_ERR_ARG, "et_shard_write_fd needs et_encode_sharded and et_shard_merge_seams first");
    uint64_t lo, hi, off;
    owned_bytes(g, &lo, &hi, &off);
    const double t0 = now_ms();
    const int rc = g->be->to_fd(static_cast<const uint8_t *>(d_out) + off, static_cast<size_t>(hi - lo), fd, lo);
    if (rc != ET_OK) return backend_failed(g, rc, "writing the piece");
    g->info.concat_ms = static_cast<float>(now_ms() - t0);
    return ET_OK;
}

// (rank-local: no exchange)
extern "C" int et_shard_place(et_group *g, const void *d_out, void *d_image, size_t cap) {
    if (!g || !d_out || !d_image) return ET_ERR_ARG;
    if (!g->have_plan || !g->seams_merged) return fail(g, ET_ERR_ARG, "et_shard_place needs et_encode_sharded and et_shard_merge_seams first");
    if (cap < g->info.file_bytes) return fail(g, ET_ERR_CAP, "image buffer too small");
    uint64_t lo, hi, off;
    owned_bytes(g, &lo, &hi, &off);
    if (hi > lo) {
        const int rc = g->be->copy(static_cast<uint8_t *>(d_image) + lo, static_cast<const uint8_t *>(d_out) + off, hi - lo);
        if (rc != ET_OK) return backend_failed(g, rc, "placing the piece");
    }
    return ET_OK;
}

extern "C" int et_shard_gather(et_group *g, const void *d_out, void *d_image, size_t cap, int root) {
    if (!g || root < 0 || root >= g->world) return ET_ERR_ARG;  // (the same on every rank)
    g->err.clear();
    const bool collective = g->world > 1 || g->force;
    if (collective && !g->xc->moves_bulk())
        return fail(g, ET_ERR_UNSUPPORTED, "et_shard_gather moves data with RCCL: create the group with et_group_create_rccl (or use et_shard_place / et_shard_write_fd)");
    int st = g->poison;
    if (st != ET_OK) g->err = "the group failed in an earlier call";
    if (st == ET_OK && (!g->have_plan || !g->seams_merged)) st = fail(g, ET_ERR_ARG, "et_shard_gather needs et_encode_sharded and et_shard_merge_seams first");
    if (st == ET_OK && !d_out) st = fail(g, ET_ERR_ARG, "et_shard_gather: null buffer");
    if (st == ET_OK && g->rank == root && (!d_image || cap < ((g->info.file_bytes + 3) & ~static_cast<uint64_t>(3))))
        st = fail(g, ET_ERR_CAP, "image buffer too small (file bytes rounded up to a word)");
    if (!collective) return st != ET_OK ? st : et_shard_place(g, d_out, d_image, cap);
    // the ranks agree to move before anybody posts a send or a receive
    SeamRow mine = {0, 0, static_cast<uint32_t>(st), 0};
    std::vector<SeamRow> rows(static_cast<size_t>(g->world));
    int rc = gather(g, &mine, rows.data(), sizeof mine);
    if (rc != ET_OK) return rc;
    if ((rc = settle(g, rows.data(), "et_shard_gather")) != ET_OK) return rc;
    const double t0 = now_ms();
    std::vector<uint64_t> words(4 * static_cast<size_t>(g->world));
    for (int q = 0; q < g->world; ++q) shard_words(g->starts.data(), q, &words[4 * q], &words[4 * q + 1], &words[4 * q + 2], &words[4 * q + 3]);
    // whole owned words travel (the image's last word may carry up to 3 pad bytes: cap was checked for them)
    rc = g->xc->gather_words(reinterpret_cast<const uint64_t(*)[4]>(words.data()), g->rank, g->world, root, d_out, d_image, g->force && g->world == 1);
    if (rc != ET_OK) {
        g->poison = rc;
        return fail(g, rc, std::string("gather: ") + g->xc->last_error());
    }
    g->info.concat_ms = static_cast<float>(now_ms() - t0);
    return ET_OK;
}

// -------------------------------------------------------------------------------------------------------------------
// decode of one cold stream (decode.zig:13-220 walks it serially; here every rank takes a range)
// -------------------------------------------------------------------------------------------------------------------
namespace {

// How a cold stream is cut.  Offsets count from compressed[0] (the .et file minus its first 4 bytes), which is taken
// to sit on a 4-byte boundary of the rank's memory; the body is cut from its 4-byte aligned base into blocks of
// 8 KiB, and a last block shorter than the 16-byte run-out a range needs behind it is not a block of its own.
struct ColdPlan {
    et_codebook cb;
    uint64_t n_symbols = 0, base_off = 0, stream_bytes = 0, n_blocks = 0;
    uint32_t first_bit = 0;
    bool exhaustive = false;
};

int cold_plan(const uint8_t *head, size_t head_len, uint64_t len, ColdPlan *p) {
    if (len < 5 || head_len < 5) return ET_ERR_FORMAT;
    size_t body_off = 0;
    const int rc = et_parse_header(head, head_len, &p->cb, &p->n_symbols, &body_off);
    if (rc != ET_OK) return rc;
    if (body_off > len) return ET_ERR_FORMAT;
    p->base_off = body_off & ~static_cast<uint64_t>(3);
    p->first_bit = static_cast<uint32_t>(body_off & 3) * 8;
    p->stream_bytes = len - p->base_off;
    p->n_blocks = (p->stream_bytes + 8191) / 8192;
    if (p->n_blocks > 1 && p->stream_bytes - (p->n_blocks - 1) * 8192 < 16) --p->n_blocks;
    p->exhaustive = p->cb.n_coded > 2 && p->cb.max_length <= p->cb.min_length + 1;
    return ET_OK;
}

struct ColdRange {
    uint64_t begin = 0, end = 0;  // stream bytes, from the aligned base
    bool active = false, has_front = false, first = false;
};

ColdRange cold_range(const ColdPlan &p, int r, int world) {
    ColdRange c;
    const uint64_t lo_b = static_cast<uint64_t>(r) * p.n_blocks / world, hi_b = static_cast<uint64_t>(r + 1) * p.n_blocks / world;
    c.begin = lo_b * 8192;
    c.end = hi_b == p.n_blocks ? p.stream_bytes : hi_b * 8192;
    c.active = hi_b > lo_b && p.cb.n_coded > 0 && p.n_symbols > 0;
    c.has_front = c.begin >= 16;  // (the 16 bytes before a later range are stream bytes)
    c.first = lo_b == 0;
    return c;
}

constexpr uint64_t COLD_MARGIN = 16;  // bytes of the stream a range needs on either side (et_decode_range_sync)

}  // namespace

extern "C" int et_decode_shard_window(const uint8_t *head, size_t head_len, uint64_t len, int rank, int world, uint64_t *window_off, uint64_t *window_len) {
    if (!head || !window_off || !window_len || world < 1 || rank < 0 || rank >= world) return ET_ERR_ARG;
    *window_off = *window_len = 0;
    ColdPlan p;
    const int rc = cold_plan(head, head_len, len, &p);
    if (rc != ET_OK) return rc;
    const ColdRange c = cold_range(p, rank, world);
    if (!c.active) return ET_OK;
    const uint64_t lo = p.base_off + (c.has_front ? c.begin - COLD_MARGIN : c.begin);
    uint64_t hi = p.base_off + c.end + COLD_MARGIN;
    if (hi > len) hi = len;
    *window_off = lo;
    *window_len = hi - lo;
    return ET_OK;
}

extern "C" int et_decode_sharded_begin(et_group *g, const uint8_t *head, size_t head_len, uint64_t len, const void *d_window, uint64_t window_off, size_t window_len,
                                       uint64_t cap, uint64_t *n_mine, uint64_t *first_index) {
    if (!g || !n_mine || !first_index) return ET_ERR_ARG;
    *n_mine = *first_index = 0;
    g->err.clear();
    g->cold_ready = false;
    const int world = g->world, r = g->rank;
    int st = g->poison;
    if (st != ET_OK) g->err = "the group failed in an earlier call";
    ColdPlan p;
    ColdRange c;
    const uint8_t *d_range = nullptr;
    size_t tail = 0;
    if (st == ET_OK && !head) st = fail(g, ET_ERR_ARG, "et_decode_sharded: no header bytes");
    if (st == ET_OK && (st = cold_plan(head, head_len, len, &p)) != ET_OK) fail(g, st, st == ET_ERR_FORMAT ? "malformed header or dictionary" : "et_parse_header");
    if (st == ET_OK) {
        c = cold_range(p, r, world);
        if (c.active) {
            // the window must hold the range and its margins (what et_decode_shard_window names, or more)
            const uint64_t need_lo = p.base_off + (c.has_front ? c.begin - COLD_MARGIN : c.begin);
            uint64_t need_hi = p.base_off + c.end + COLD_MARGIN;
            if (need_hi > len) need_hi = len;
            if (!d_window || (window_off & 3) || (reinterpret_cast<uintptr_t>(d_window) & 3)) st = fail(g, ET_ERR_ARG, "et_decode_sharded: the window must be 4-byte aligned, in memory and in the stream");
            else if (window_off > need_lo || window_off + window_len < need_hi) st = fail(g, ET_ERR_ARG, "et_decode_sharded: the window does not hold the rank's range and its 16-byte margins");
            else {
                d_range = static_cast<const uint8_t *>(d_window) + (p.base_off + c.begin - window_off);
                const uint64_t readable = window_off + window_len - (p.base_off + c.end);  // stream bytes behind the range that the window holds
                tail = static_cast<size_t>(c.end == p.stream_bytes ? 0 : readable);
            }
        }
    }
    std::vector<ColdRow> rows(static_cast<size_t>(world));
    ColdRow mine;
    std::memset(&mine, 0, sizeof mine);
    mine.start = mine.exit = -1;
    mine.cap = cap;
    for (int i = 0; i < 32; ++i) mine.map[i] = static_cast<uint8_t>(i);  // a rank without blocks passes the start on
    et_range_info info = {};
    const bool active = st == ET_OK && c.active;
    int rc;
    if (st == ET_OK && p.exhaustive) {
        // codes that do not self-synchronise: exit maps over every possible start, chained from the stream's start
        uint32_t n_starts = 0;
        if (active && (st = g->be->range_maps(&p.cb, d_range, c.end - c.begin, tail, c.first ? static_cast<int32_t>(p.first_bit) : -1, mine.map, &n_starts)) != ET_OK)
            backend_failed(g, st, "et_decode_range_maps");
        mine.status = static_cast<uint64_t>(st);
        if ((rc = gather(g, &mine, rows.data(), sizeof mine)) != ET_OK) return rc;
        if ((rc = settle(g, rows.data(), "et_decode_sharded")) != ET_OK) return rc;
        uint32_t s_in = p.first_bit;
        for (int q = 0; q < r; ++q) s_in = rows[q].map[s_in & 31u];
        if (active && (st = g->be->range_resolve(s_in, &info)) != ET_OK) backend_failed(g, st, "et_decode_range_resolve");
    } else if (active) {
        if ((st = g->be->range_sync(&p.cb, d_range, c.end - c.begin, tail, c.has_front ? 1 : 0, c.first ? static_cast<int32_t>(p.first_bit) : -1, &info)) != ET_OK)
            backend_failed(g, st, "et_decode_range_sync");
    }
    // agree on the seams: every active rank's start must be the exit of the active rank before it.  (A rank that failed
    // above has said so in its row of the first exchange it reaches; all return there.)
    for (int round = 0;; ++round) {
        const bool on = st == ET_OK && c.active;
        mine.start = on ? static_cast<int64_t>(info.start_bit) : -1;
        mine.exit = on ? static_cast<int64_t>(info.exit_bit) : -1;
        mine.n_symbols = on ? info.n_symbols : 0;
        mine.status = static_cast<uint64_t>(st);
        if ((rc = gather(g, &mine, rows.data(), sizeof mine)) != ET_OK) return rc;
        if ((rc = settle(g, rows.data(), "et_decode_sharded")) != ET_OK) return rc;
        int64_t prev_exit = p.first_bit, want_mine = -1;
        bool any_wrong = false;
        for (int q = 0; q < world; ++q) {
            if (rows[q].start < 0) continue;
            if (rows[q].start != prev_exit) {
                any_wrong = true;
                if (q == r) want_mine = prev_exit;
            }
            prev_exit = rows[q].exit;
        }
        if (!any_wrong) break;
        if (round > world) return fail(g, ET_ERR_HIP, "cold decode did not settle");  // (from the same rows: every rank gives up here)
        if (want_mine >= 0 && (st = g->be->range_sync(&p.cb, d_range, c.end - c.begin, tail, c.has_front ? 1 : 0, static_cast<int32_t>(want_mine), &info)) != ET_OK)
            backend_failed(g, st, "et_decode_range_sync (repair)");
    }
    // who writes what -- and whether it fits, for every rank, from the same rows
    uint64_t first = 0, mine_first = 0, mine_take = 0;
    for (int q = 0; q < world; ++q) {
        const uint64_t have = rows[q].n_symbols;
        const uint64_t take = first >= p.n_symbols ? 0 : (have < p.n_symbols - first ? have : p.n_symbols - first);
        if (take > rows[q].cap)
            return fail(g, ET_ERR_CAP, "et_decode_sharded: rank " + std::to_string(q) + "'s " + std::to_string(take) + " symbols do not fit its buffer (" + std::to_string(rows[q].cap) + ")");
        if (q == r) {
            mine_first = first;
            mine_take = take;
        }
        first += have;
    }
    *first_index = mine_first;
    *n_mine = mine_take;
    g->cold_take = c.active ? mine_take : 0;
    g->cold_ready = true;
    return ET_OK;
}

// (rank-local: no exchange)
extern "C" int et_decode_sharded_write(et_group *g, void *d_out, size_t cap, size_t *written) {
    if (!g || !written) return ET_ERR_ARG;
    *written = 0;
    if (!g->cold_ready) return fail(g, ET_ERR_ARG, "et_decode_sharded_write needs et_decode_sharded_begin first");
    g->cold_ready = false;
    if (g->cold_take == 0) return ET_OK;
    if (!d_out) return fail(g, ET_ERR_ARG, "et_decode_sharded_write: null buffer");
    const int rc = g->be->range_write(g->cold_take, d_out, cap, written);
    if (rc != ET_OK) return backend_failed(g, rc, "et_decode_range_write");
    return ET_OK;
}

extern "C" int et_decode_sharded(et_group *g, const void *d_compressed, size_t len, void *d_out, size_t cap, size_t *written, uint64_t *first_index) {
    if (!g || !written || !first_index) return ET_ERR_ARG;
    *written = 0;
    *first_index = 0;
    g->err.clear();
    // header and dictionary: parsed on the host, by every rank.  Whatever goes wrong here travels in the first row.
    std::vector<uint8_t> head(len < 8192 ? len : 8192);
    static const uint8_t nothing = 0;
    const uint8_t *h = head.empty() ? &nothing : head.data();  // (an empty stream is a malformed one, not a missing argument)
    int st = ET_OK;
    if (!d_compressed) st = ET_ERR_ARG;
    else if (!head.empty() && (st = g->be->read_head(d_compressed, head.size(), head.data())) != ET_OK) backend_failed(g, st, "reading the header");
    if (st != ET_OK) h = nullptr;  // (begin reports it: as this rank's status, to everybody)
    const std::string why = g->err;
    uint64_t n_mine = 0;
    int rc = et_decode_sharded_begin(g, h, head.size(), len, d_compressed, 0, len, d_out ? cap : 0, &n_mine, first_index);
    if (rc != ET_OK) {
        if (st != ET_OK && !why.empty()) g->err = why;
        return rc;
    }
    return et_decode_sharded_write(g, d_out, cap, written);
}
