/*
 * entreepy_hip.h -- C ABI of libentreepy_hip.so: the MI355X (gfx950) Huffman
 * encode/decode path that drops in for typio/entreepy's src/encode.zig and
 * src/decode.zig.
 *
 * The reference has no FFI of its own; the seam is the pair of Zig functions
 *     pub fn encode(allocator, text, out_writer, std_out, flags) !usize   (src/encode.zig:25)
 *     pub fn decode(allocator, compressed_text, out_writer, std_out, flags) !usize (src/decode.zig:13)
 * called from src/main.zig:202,204 and src/test.zig:15,26.  Each entry point below
 * names the reference lines it replaces; INTEGRATION.md shows the Zig `extern fn`
 * declarations a maintainer adds to bind them.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  Every function
 * returns an et_status (0 == ET_OK); nothing aborts or prints.  An et_ctx owns one
 * GPU's stream, workspaces and pinned staging and is NOT thread-safe; use one per
 * thread/GPU.  Pointers named d_* are device (HBM) pointers on the ctx's GPU,
 * everything else is host memory.
 */
#ifndef ENTREEPY_HIP_H
#define ENTREEPY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum et_status {
    ET_OK = 0,
    ET_ERR_EMPTY = 1,       /* error.QueueEmpty: empty input (queue.zig:28-30 via encode.zig:137-138) */
    ET_ERR_NOMEM = 2,       /* error.OutOfMemory (encode.zig:254) */
    ET_ERR_CAP = 3,         /* output buffer too small (writer error / error.NoSpaceLeft) */
    ET_ERR_FORMAT = 4,      /* malformed .et stream (the reference performs no validation, main.zig:199) */
    ET_ERR_HIP = 5,         /* HIP runtime failure; et_last_error() has the text */
    ET_ERR_ARG = 6,         /* null/misaligned/out-of-range argument */
    ET_ERR_UNSUPPORTED = 7, /* stream needs a feature outside the decoder's domain (code length > 32) */
    ET_ERR_IO = 8,          /* read/write on a file descriptor failed (et_encode_fd / et_decode_fd) */
    ET_ERR_RCCL = 9         /* the exchange between the GPUs of a group failed (RCCL error, or the callback's) */
} et_status;

/* The reference's `dictionary: [256]Code` (encode.zig:141-146) plus what its -d dump
 * and header need.  Plain data; filled by et_build_codebook / et_parse_header. */
typedef struct et_codebook {
    uint32_t data[256];      /* Code.data: path bits, u32-truncated (encode.zig:142,181,195) */
    uint8_t length[256];     /* Code.length: 0 == symbol absent (or dropped, 256-symbol quirk) */
    uint8_t dfs_order[256];  /* leaf symbols in the order encode.zig:204-212 prints them */
    uint32_t n_coded;        /* symbols with length > 0 */
    uint32_t min_length;     /* shortest non-zero length (0 when n_coded == 0) */
    uint32_t max_length;     /* longest length */
} et_codebook;

/* Wall-clock split of the last whole call, in milliseconds.  Device phases come from HIP events on
 * the ctx stream that the four large kernels carry themselves (begin and end of the histogram, pack,
 * first synchronisation sweep and write kernels; no marker packets between the kernels).
 * Replaces the -d "time taken" line (encode.zig:26-28). */
typedef struct et_timings {
    float hist_ms;      /* encode: the byte histogram kernel */
    float host_ms;      /* encode: tree/code build on the host; decode: header parse + table plan */
    float scan_ms;      /* encode: between the histogram and the pack kernel (reduce, host code construction, tile
                           scan, uploads); decode: between the first sweep and the write kernel (repair sweep,
                           verification, block count scan) */
    float body_ms;      /* encode: code scatter (pack) kernel; decode: symbol write kernel */
    float sync_ms;      /* decode: everything in front of the write kernel (sweeps, verification, scan) */
    float total_ms;     /* begin of the first large kernel to the end of the last */
    uint32_t sync_iters;/* decode: synchronisation launches */
    uint32_t reserved;  /* decode: bit 0 = the exhaustive synchronisation path ran, bit 1 = the sweeps ran as a tree walk, bit 2 = the write pass used the chained tables, bit 3 = synchronised (and, unless switched off, written) by rows: a complete code of 7- and 8-bit codewords, et_row_code, bit 4 = a fixed-length code (2^L codewords of L bits): decoded by arithmetic, no synchronisation (csrc/et_rowsync.h, k_fixed_write), bit 5 = the write pass ran as its instantiation for streams with more than 128 symbols per 256-bit subsequence (quarters that overflow the stage walk once, into strips) */
    float sync_first_ms;/* decode: the first synchronisation sweep alone (k_dec_sync<first>) */
    uint32_t pad_;
} et_timings;

typedef struct et_ctx et_ctx;

/* ---- lifetime ---------------------------------------------------------------- */
/* The reference call is stateless; a ctx amortises hipMalloc, pinned staging and
 * stream creation across calls.  `device` is a HIP device ordinal. */
int et_ctx_create(int device, et_ctx **ctx);
void et_ctx_destroy(et_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the
 * ctx's own.  NULL is HIP's default (null) stream, as everywhere in HIP. */
int et_ctx_set_stream(et_ctx *ctx, void *hip_stream);
/* Back to the non-blocking stream the ctx created for itself. */
int et_ctx_use_own_stream(et_ctx *ctx);
/* The hipStream_t the ctx's calls are enqueued on, and its HIP device ordinal. */
void *et_ctx_stream(const et_ctx *ctx);
int et_ctx_device(const et_ctx *ctx);
/* Pre-size workspaces for inputs of up to max_text_bytes so that no allocation
 * happens inside a timed call. */
int et_ctx_reserve(et_ctx *ctx, size_t max_text_bytes);
/* Tuning/test knob: force the encode tile to `rounds` x 4 KiB (a power of two, 1 .. 128);
 * 0 restores the size-based choice.  Results never depend on it. */
int et_ctx_set_tile_rounds(et_ctx *ctx, uint32_t rounds);
/* Record per-phase HIP events (small overhead); off by default.  The calls stay as
 * asynchronous as they are without: the event arithmetic happens in et_last_timings[_of],
 * which waits for the call's last event.  et_last_timings = the last call's; _of: which =
 * 0 the last encode-side call, 1 the last decode (separate event sets, so the encode
 * figures can be fetched after the decode that followed has been enqueued).
 * on = ET_TIMING_DECODE_BODY: only the decode's write kernel carries its pair of events (a
 * dispatch that carries events costs ~5 us of queue time on either side of it: ~35 us of a
 * 1.4 ms step with all four large kernels timed); et_timings then holds body_ms, host_ms,
 * sync_iters and the flags, everything else 0. */
#define ET_TIMING_DECODE_BODY 2
int et_ctx_enable_timing(et_ctx *ctx, int on);
int et_last_timings(et_ctx *ctx, et_timings *out);
int et_last_timings_of(et_ctx *ctx, int which, et_timings *out);
const char *et_last_error(const et_ctx *ctx);
const char *et_strerror(int status);
const char *et_version(void);

/* ---- whole-call entry points (what the Zig shims bind) -------------------------- */
/* encode.zig:253-254: the reference sizes its scratch as 7200 + text.len; same bound
 * here (rounded up to a multiple of 16). */
size_t et_encode_bound(size_t text_len);

/* Replaces encode() (encode.zig:25-337) for write_output=true: histogram (:43-47),
 * code construction (:54-214), header+dictionary (:253-299), body pack (:303-318).
 * Writes the complete .et file image to out[0..*out_len).  ET_ERR_EMPTY for n == 0
 * (the reference raises error.QueueEmpty). */
int et_encode(et_ctx *ctx, const uint8_t *text, size_t n,
              uint8_t *out, size_t cap, size_t *out_len);

/* Replaces decode() (decode.zig:13-220).  `compressed` is the .et file MINUS its
 * first 4 bytes, exactly what main.zig:204 / test.zig:26 pass.  Emits body_len
 * symbols (header field, decode.zig:36-42), or fewer when the bitstream ends first. */
int et_decode(et_ctx *ctx, const uint8_t *compressed, size_t len,
              uint8_t *out, size_t cap, size_t *out_len);

/* The same two calls on open files: the reference reads the whole input
 * (main.zig:34-40,186) and writes the result with one writeAll (encode.zig:319,
 * main.zig:192-197); these move the file through two pinned staging buffers in chunks
 * (pread by a small thread pool while the previous chunk crosses PCIe; the result leaves
 * the same way with pwrite), so host memory stays bounded whatever the file size.
 * Regular files only (sized with fstat, positioned I/O).  out_fd < 0: code, write nothing
 * (main.zig's -t).  et_decode_fd skips `in_skip` bytes first: 4 = main.zig:204's
 * `text_in[4..]`.  *in_len = bytes consumed after the skip, *out_len = bytes produced.
 * et_encode / et_decode use the same pipeline on host memory. */
int et_encode_fd(et_ctx *ctx, int in_fd, int out_fd, size_t *in_len, size_t *out_len);
int et_decode_fd(et_ctx *ctx, int in_fd, size_t in_skip, int out_fd, size_t *in_len, size_t *out_len);

/* Code table of the ctx's most recent et_encode / et_encode_device (for the -d dump,
 * encode.zig:204-212, which the reference prints from inside encode()). */
int et_last_codebook(const et_ctx *ctx, et_codebook *out);

/* The -d self-check of encode.zig:221-247: every ordered pair (i, j) of coded symbols for which the
 * reference's loop finds one code a prefix of the other (pairs[2k] = i, pairs[2k+1] = j, in the reference's
 * i-then-j order; *n_pairs = how many there are, also beyond cap_pairs).  The reference prints "Found
 * colliding prefix codes for {i} {c} and {j} {c}" for each; a Huffman table never has any. */
int et_prefix_collisions(const et_codebook *cb, uint8_t *pairs, size_t cap_pairs, size_t *n_pairs);

/* The four bytes decode() never sees (main.zig:204 passes text_in[4..] unchecked, TODO at
 * main.zig:199): magic e7 c0 de and format version 01 (encode.zig:262-266).  ET_OK, or
 * ET_ERR_FORMAT with the reason in *why (static string; may be NULL).  The CLI refuses
 * files that fail this check instead of decoding whatever follows. */
int et_check_magic(const uint8_t first4[4], const char **why);

/* Diagnostics: build the decode lookup tables of `cb` both ways -- on the device (what every
 * decode call does, from the host's plan) and with the host-side reference builders -- and compare
 * them entry for entry.  ET_OK when identical; ET_ERR_FORMAT with *where = 1 first-level table,
 * 2 long list, 3 second-level tables, 4 code lengths, 5 step table, 6 write-step table. */
int et_selftest_decode_tables(et_ctx *ctx, const et_codebook *cb, int *where);

/* The synchronisation sweeps of a decode run as a fixed-rate walk over the code TREE when the dictionary is a
 * full binary tree (an encoder's always is): one table row per internal node, one byte of the stream per step
 * (csrc/et_treewalk.h).  et_treewalk_table: that table as the host fills it -- (*n_int + 7) x 256 entries of
 * next row | codewords completed in the byte << 8 | bit at which the first of them ends << 12 -- or
 * ET_ERR_UNSUPPORTED when the walk does not apply (table may be NULL to ask just that).
 * et_selftest_treewalk_table: build it on the device (what a decode does) and compare it with the host fill;
 * *first_diff = 1 + the first differing entry. */
int et_treewalk_table(const et_codebook *cb, uint16_t *table, size_t cap_entries, uint32_t *n_int);
int et_selftest_treewalk_table(et_ctx *ctx, const et_codebook *cb, uint32_t *first_diff);
/* Under the same condition the write pass (decode.zig:143-203 again, now emitting symbols) looks codewords up in
 * CHAINED tables: a root table indexed by the next 11 bits, and for every tree node such an index can end on
 * without completing a codeword a table of its own; each entry names the table of the NEXT lookup, so a long code
 * is one more step of its lane, not an exception (csrc/et_treewalk.h).  et_chain_tables: the tables as the host
 * fills them (u64 entries: lo = i16 (symbols << 10) - bits | first symbol << 16 | bits to the end of the first
 * symbol << 24; hi = byte offset of the next table | second symbol << 16 | (32 - its index bits) << 24), and where
 * each table begins / how many index bits it has.  et_selftest_treewalk_table compares the device fill of these
 * as well (*first_diff counts on behind the tree-walk table's entries). */
int et_chain_tables(const et_codebook *cb, uint64_t *table, size_t cap_entries, uint32_t *n_entries, uint32_t *table_first, uint8_t *table_bits,
                    size_t cap_tables, uint32_t *n_tables);

/* (Nearly) uniform bytes -- BASELINE.json's worst case -- give a complete code of 7- and 8-bit codewords that never
 * re-synchronises (decode.zig:143-203 has no trouble with it; a parallel decoder has).  When the 7-bit codewords are
 * the values 0 .. t-1, which is how encode.zig:82-138 hands them out, a decode synchronises such a stream in one pass by
 * rows (bytes) and columns (bit offsets): csrc/et_rowsync.h.  et_row_code: ET_OK and *t when `cb` is such a code,
 * ET_ERR_UNSUPPORTED otherwise (the exit maps for every start offset then, csrc/et_kernels_fallback.hip).  (A code of
 * 2^L codewords of L bits each -- t = 0 and t = 128 among them -- is decoded by arithmetic before either: symbol i is
 * the L bits at first_bit + i L.) */
int et_row_code(const et_codebook *cb, uint32_t *t);

/* Which synchronisation a one-GPU decode of a whole stream starts with for this code table (diagnostics; decode.zig:143-203
 * needs no such choice -- one thread walks the stream).  The stream itself can still overrule the first two: blocks that do
 * not settle under the tree walk go to the exit maps, or by rows if the code is a row code.
 *   ET_PATH_TREE_WALK  the tree walk (text, and codes of L and L + 1 bits whose mix of lengths settles quickly: csrc/et_rowsync_host.cpp)
 *   ET_PATH_EXIT_MAPS  exit maps for every start offset (near-fixed-length codes that do not settle; csrc/et_kernels_fallback.hip)
 *   ET_PATH_ROWS       by byte rows and bit columns (complete codes of 7 and 8 bits that do not settle: uniform bytes; et_row_code)
 *   ET_PATH_FIXED      2^L codewords of L bits: no synchronisation, symbol i is the L bits at first_bit + i L
 *   ET_PATH_WINDOWS    the round-1 window kernels (a dictionary whose completed tree has more than 255 internal nodes or is not prefix-free) */
enum { ET_PATH_TREE_WALK = 0, ET_PATH_EXIT_MAPS = 1, ET_PATH_ROWS = 2, ET_PATH_FIXED = 3, ET_PATH_WINDOWS = 4 };
int et_decode_path(const et_codebook *cb, uint32_t *path);

/* Header field "length of body" (decode.zig:36-42) so callers can size `out`. */
int et_decoded_size(const uint8_t *compressed, size_t len, size_t *n_symbols);

/* Same two calls with input and output resident in HBM (benchmarks, pipelines).
 * d_out must hold et_encode_bound(n) bytes / n_symbols (+16 slack) bytes.  Both are
 * stream-ordered: *out_len is final on return, the bytes in d_out are complete once the
 * ctx's stream has drained (enqueue consumers on it, or synchronise it). */
int et_encode_device(et_ctx *ctx, const void *d_text, size_t n,
                     void *d_out, size_t cap, size_t *out_len);
int et_decode_device(et_ctx *ctx, const void *d_compressed, size_t len,
                     void *d_out, size_t cap, size_t *out_len);

/* ---- staged entry points (sharded multi-GPU encode, tests) ------------------------ */
/* encode.zig:43-47 on the GPU: 256 x u64 counts of d_text[0..n) into d_hist (device).
 * Also leaves per-tile histograms in the ctx for a following et_encode_body_device
 * on the SAME (d_text, n). */
int et_histogram_device(et_ctx *ctx, const void *d_text, size_t n, void *d_hist256_u64);
/* The same counts on the HOST: the reduction stores them into pinned host memory as it stores them on the device, so
 * this only waits for that histogram (a poll, no copy command) and copies 2 KiB.  After it the shard encode that
 * follows needs no et_histogram_on_host.  et_histogram_device may be given d_hist = NULL when only the host wants
 * the counts; et_histogram_device_ptr: where the ctx itself keeps them on the device (valid until its next
 * histogram), e.g. as the send buffer of a collective.  ET_ERR_ARG unless a histogram of this ctx is current. */
int et_histogram_host(et_ctx *ctx, uint64_t counts[256]);
int et_histogram_device_ptr(et_ctx *ctx, const void **d_hist256_u64);
/* A caller that already holds the counts et_histogram_device produced for (d_text, n) on the host
 * (a sharded encode reads them back for the exchange anyway) hands them over, and the shard encode
 * that follows does not copy them from the device again.  ET_ERR_ARG unless a histogram of this ctx
 * is current.  The counts must be the ones the device holds. */
int et_histogram_on_host(et_ctx *ctx, const uint64_t counts[256]);

/* encode.zig:54-214 + queue.zig:9-43 on the host: sort, two-queue tree, codes.
 * Bit-exact including the u8 book_index saturation (256 distinct symbols), the u32
 * path truncation and single-symbol inputs.  ET_ERR_EMPTY when every count is 0. */
int et_build_codebook(const uint64_t hist[256], et_codebook *cb);

/* encode.zig:259-299: magic, version, D, low 32 bits of text_len, bit-packed
 * dictionary, zero pad to a byte.  At most 4631 bytes. */
int et_write_header(const et_codebook *cb, uint64_t text_len,
                    uint8_t *out, size_t cap, size_t *header_len);

/* Sum over s of hist[s] * length[s]: the body bit count of a shard, from its local
 * histogram alone (no data pass). */
int et_codebook_bits(const et_codebook *cb, const uint64_t hist[256], uint64_t *bits);

/* The host step of a sharded encode in one call: hists = world rows of 256 local counts (the
 * all-gathered histograms, row r = shard r).  Their sum is the stream's histogram (encode.zig:43-47)
 * -> code table, header (text_len = the sum of all counts), and start_bits[0..world]: shard r's body
 * occupies file bits [start_bits[r], start_bits[r+1]), start_bits[0] = 8 * header_len.  Every rank
 * computes the same plan from the same rows.  ET_ERR_EMPTY when every count is 0. */
int et_plan_shards(const uint64_t *hists, uint32_t world, et_codebook *cb, uint8_t *header, size_t header_cap,
                   size_t *header_len, uint64_t *start_bits);

/* encode.zig:303-315 on the GPU for one shard: pack the codes of d_text[0..n) into
 * d_out as a MSB-first bitstream whose first bit lands at bit `start_bit` of d_out
 * (d_out 4-byte aligned).  Every 32-bit word the shard touches is fully overwritten
 * (bits outside [start_bit, *end_bit) in the first/last word become 0), so adjacent
 * shards are concatenated by OR-ing their boundary bytes.  Requires the preceding
 * et_histogram_device on the same (d_text, n). */
int et_encode_body_device(et_ctx *ctx, const et_codebook *cb, const void *d_text, size_t n,
                          void *d_out, size_t cap_bytes, uint64_t start_bit, uint64_t *end_bit);

/* Same, for the shard that also carries the file header (rank 0): header[0..header_len)
 * is written to d_out[0..header_len) and the body starts right after it
 * (start bit = 8 * header_len, encode.zig:299-303). */
int et_encode_head_shard_device(et_ctx *ctx, const et_codebook *cb, const void *d_text, size_t n,
                                void *d_out, size_t cap_bytes, const uint8_t *header, size_t header_len,
                                uint64_t *end_bit);

/* decode.zig:34-141 on the host: D, body length, dictionary.  *body_offset is the
 * byte offset of the body inside `compressed` (decode.zig:156: 5 + global_pos). */
int et_parse_header(const uint8_t *compressed, size_t len, et_codebook *cb,
                    uint64_t *n_symbols, size_t *body_offset);

/* decode.zig:143-203 on the GPU: decode up to n_symbols symbols from the bitstream
 * d_body[0..body_bytes) beginning at bit `start_bit` (< 8) of d_body.  `cb` is any prefix-free table with codes of
 * up to 32 bits.  A bit pattern that is no symbol's code cannot occur in a stream of the table's own encoder; in a
 * corrupted one it decodes as byte 0 -- or, for tables too sparse for the tree walk (more than 255 internal nodes),
 * is passed over one bit at a time; the reference spins there (quirk Q6).  Only hand-made tables leave such patterns
 * at all: an encoder's tree is full. */
int et_decode_body_device(et_ctx *ctx, const et_codebook *cb, const void *d_body,
                          size_t body_bytes, uint32_t start_bit, uint64_t n_symbols,
                          void *d_out, size_t cap, size_t *out_len);

/* ---- one stream decoded on several GPUs (cold .et file, no side information) ---------- */
/* The body is split at multiples of 8192 bytes counted from its 4-byte aligned base; every
 * rank synchronises its own range, the ranks exchange (start, exit, symbols), a rank whose
 * start differs from its predecessor's exit calls et_decode_range_sync again with that
 * exit, and when all agree each rank writes its symbols (entreepy_amd/sharded.py
 * decode_cold is the reference sequence).  decode.zig:143-203 has no counterpart: the
 * reference walks the stream serially. */
typedef struct et_range_info {
    uint32_t start_bit;   /* where the first codeword of the range begins, bits from d_range */
    uint32_t exit_bit;    /* first codeword boundary at or after the range end, bits past it */
    uint64_t n_symbols;   /* codewords that begin inside the range */
    uint32_t sweeps;      /* synchronisation launches of this call */
    uint32_t reserved;
} et_range_info;

/* d_range: 4-byte aligned pointer to the range's first byte (range_bytes long; a multiple
 * of 8192 unless the stream ends with it); tail_bytes more stream bytes are readable after
 * it (0 only when the stream ends there, else >= 16); has_front: the 16 bytes before
 * d_range are stream bytes too.  in_start_bit >= 0: the range's first codeword begins at
 * that bit (rank 0: the body's start; later calls: the predecessor's exit_bit); -1: unknown,
 * run in from the bytes in front (requires has_front).  Calling again for the same d_range
 * with a corrected in_start_bit repairs the previous result instead of starting over. */
int et_decode_range_sync(et_ctx *ctx, const et_codebook *cb, const void *d_range, size_t range_bytes, size_t tail_bytes,
                         int has_front, int32_t in_start_bit, et_range_info *info);
/* Write the first max_symbols symbols of the range synchronised last into d_out. */
int et_decode_range_write(et_ctx *ctx, uint64_t max_symbols, void *d_out, size_t cap, size_t *out_len);

/* The same split for codes that do NOT self-synchronise (near-fixed-length ones; SURVEY
 * §8e's "fall back" case), where run-ins find nothing: et_decode_range_maps computes, for
 * every bit offset p < *n_starts (= the longest code length, <= 32) at which the range's
 * first codeword might begin, the offset map[p] at which decoding leaves the range
 * (exhaustive synchronisation: one walk per offset, maps composed per block and per 256
 * blocks on the GPU, the last level on the host).  in_start_bit >= 0: the start is known
 * (the stream's first range), every entry is that start's exit.  The ranks exchange their
 * 32-byte maps, chain them from the stream's start, and call et_decode_range_resolve with
 * the start that reaches them; et_decode_range_write then works as above. */
int et_decode_range_maps(et_ctx *ctx, const et_codebook *cb, const void *d_range, size_t range_bytes, size_t tail_bytes,
                         int32_t in_start_bit, uint8_t map[32], uint32_t *n_starts);
int et_decode_range_resolve(et_ctx *ctx, uint32_t in_start_bit, et_range_info *info);

/* ---- groups: one stream over the GPUs of a node ------------------------------------------------ */
/* The reference has one thread and one buffer (encode.zig:25-337); north_star shards the text by
 * contiguous chunk over the GPUs.  An et_group is one rank's membership: its ctx (one GPU), its rank,
 * and how the ranks exchange small host buffers -- a callback (MPI, gloo, threads: anything that can
 * all-gather `bytes_per_rank` bytes) or RCCL over xGMI (loaded with dlopen on first use; a process that
 * already holds an RCCL, e.g. PyTorch's, shares it).  All ranks make the same calls in the same order. */
typedef struct et_group et_group;
/* send: bytes_per_rank bytes of this rank; recv: world x bytes_per_rank, rank order.  0 = success. */
typedef int (*et_allgather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
#define ET_RCCL_ID_BYTES 128
int et_group_create(et_ctx *ctx, int rank, int world, et_allgather_fn allgather, void *user, et_group **group);
/* RCCL: rank 0 obtains an id (ncclGetUniqueId) and hands it to the others out of band (the launcher's
 * store, a file, MPI); every rank then calls et_group_create_rccl with it (ncclCommInitRank).
 * ET_ERR_RCCL when librccl.so cannot be loaded or the communicator cannot be made. */
int et_rccl_unique_id(uint8_t id[ET_RCCL_ID_BYTES]);
int et_group_create_rccl(et_ctx *ctx, int rank, int world, const uint8_t id[ET_RCCL_ID_BYTES], et_group **group);
void et_group_destroy(et_group *group);  /* (does not touch the group's ctx: either may be destroyed first) */
const char *et_group_last_error(const et_group *group);
/* Which RCCL the library bound (the object's name), or why it could not. */
const char *et_rccl_library(void);
/* FAILURES.  Every row a rank contributes to an exchange carries its status.  A rank whose local step failed (a
 * null or too small buffer, a HIP error, a corrupted range) still makes every exchange of the call, and once the
 * rows are in ALL ranks return the status of the first rank that failed: no rank leaves a collective sequence
 * early, nobody waits for a rank that has gone.  The reference's error union (encode.zig:25 `!usize`) reaches
 * every caller.  A failure BEHIND a call's last exchange (an enqueue that failed) is returned by that rank alone
 * and poisons its group: later calls of that rank only take part in their exchanges, carrying the status, so its
 * peers hear of it at their next call.  A transport failure (ET_ERR_RCCL: a peer never arrived within the
 * timeout, the callback failed) ends the call where it is; destroy the group.
 * Options: ET_GROUP_FORCE_COLLECTIVES (value != 0): a group of ONE takes the transport's path all the same --
 * the all-gather from device memory, the kernel that hands the rows to the polling host, send/receive to itself in
 * et_shard_gather -- where it otherwise just copies (one-GPU tests of the N > 1 path).  ET_GROUP_TIMEOUT_MS: how
 * long an RCCL exchange waits for the other ranks (default 600 000). */
enum { ET_GROUP_FORCE_COLLECTIVES = 1, ET_GROUP_TIMEOUT_MS = 2 };
int et_group_set_option(et_group *group, int option, int64_t value);

/* Where this rank's shard sits in the .et image.  Bits and words are counted from the image's first
 * byte; a "word" is 4 bytes.  The rank's buffer holds words [piece_word_lo, piece_word_hi) (word
 * piece_word_lo at d_out[0]); it CONTRIBUTES words [owned_word_lo, owned_word_hi): a word two shards
 * share belongs to the first of them. */
typedef struct et_shard_info {
    uint64_t start_bit, end_bit;      /* the shard's body: image bits [start_bit, end_bit) */
    uint64_t local_start_bit;         /* bit of d_out at which that body begins (rank 0: 8 x header_len; else start_bit % 32) */
    uint64_t header_len;              /* rank 0: bytes of header + dictionary in front of its body; else 0 */
    uint64_t file_bytes;              /* length of the whole image (encode.zig:318,336) */
    uint64_t text_len;                /* bytes of text over all ranks */
    uint64_t piece_word_lo, piece_word_hi, owned_word_lo, owned_word_hi;
    float exchange_ms, plan_ms;       /* host clock: the histogram all-gather (incl. the wait for K1); code table + header + offsets */
    float seam_ms, concat_ms;         /* host clock: et_shard_merge_seams; the last et_shard_write_fd / et_shard_gather */
} et_shard_info;

/* encode() (encode.zig:25-337) for one rank's chunk d_text[0..n) of the group's text: local histogram
 * (K1), ONE exchange (all-gather of the 256 x u64 local histograms: their sum is encode.zig:43-47's
 * histogram, each row gives a shard's bit count; the row also carries the rank's status and cap), the same
 * code table, header and offsets on every rank (et_plan_shards), then the shard's body at its bit offset
 * (K2 + K4).  d_out: 4-byte aligned, cap >= et_encode_bound(n) -- which holds any shard whose symbols are as
 * frequent in it as in the whole text; a shard of symbols that are rare elsewhere packs to more, and then ALL
 * ranks return ET_ERR_CAP (every rank knows every shard's size and cap from the rows).  A rank may hold no text
 * (n = 0).  ET_ERR_EMPTY when all ranks are empty. */
int et_encode_sharded(et_group *group, const void *d_text, size_t n, void *d_out, size_t cap, et_shard_info *info);
/* The bit-offset-adjusted concatenation (encode.zig:319 writes ONE image).  After et_encode_sharded:
 * et_shard_merge_seams -- the owner of a word that several shards share receives their bits (one exchange
 * of 8 bytes per rank; d_out's last word is patched on the device).  From then on the ranks' owned words
 * are disjoint ranges of the image and travel independently:
 * et_shard_write_fd   pwrite this rank's owned bytes at their offset of `fd` (any rank order; rank-local),
 * et_shard_place      copy them into an image in device memory this rank can address (same GPU, or a
 *                     peer-mapped pointer), on the ctx stream,
 * et_shard_gather     RCCL groups: every rank's owned words to `root`'s d_image over xGMI (send/recv;
 *                     d_image and cap are read on root only; cap >= file_bytes rounded up to 4). */
int et_shard_merge_seams(et_group *group, void *d_out);
int et_shard_write_fd(et_group *group, const void *d_out, int fd);
int et_shard_place(et_group *group, const void *d_out, void *d_image, size_t cap);
int et_shard_gather(et_group *group, const void *d_out, void *d_image, size_t cap, int root);
/* The plan of the group's last et_encode_sharded: code table, the world + 1 shard offsets, this rank's info
 * (with the timings of the calls made since). */
int et_group_codebook(const et_group *group, et_codebook *cb);
int et_group_start_bits(const et_group *group, uint64_t *start_bits);
int et_group_last_info(const et_group *group, et_shard_info *info);
/* The host arithmetic of the concatenation, on its own (tests, other hosts): words[4] = {piece_lo, piece_hi,
 * owned_lo, owned_hi} of `rank`; et_seam_word: the word that closes `rank`'s owned range with the bits of
 * every later shard that begins in it (first_last = per rank {first word, last word} of its buffer, own bits
 * only); *has_seam = 0 when the rank shares no word it owns. */
int et_shard_words(const uint64_t *start_bits, uint32_t world, uint32_t rank, uint64_t words[4]);
int et_seam_word(const uint64_t *start_bits, uint32_t world, uint32_t rank, const uint32_t *first_last, uint32_t *merged, int *has_seam);
/* d_src[0..len) -> fd at file_offset, and back, through the ctx's pinned staging (pwrite / pread in chunks,
 * overlapped with the copies; main.zig:34-40 and encode.zig:319 for one rank's part of a file). */
int et_device_to_fd(et_ctx *ctx, const void *d_src, size_t len, int fd, uint64_t file_offset);
int et_fd_to_device(et_ctx *ctx, int fd, uint64_t file_offset, size_t len, void *d_dst);

/* decode() (decode.zig:13-220) of ONE cold stream by the ranks of a group: d_compressed = the .et file
 * minus its first 4 bytes, resident on every rank (at least its own 8 KiB-block range with 16 bytes on
 * either side, and the first 8 KiB for the dictionary).  Rank r decodes blocks [r, r+1) x n_blocks / world:
 * *written symbols into d_out, the first of which is symbol *first_index of the text (the output stays
 * sharded).  One exchange of (start, exit, symbols) per round (expected: one round); codes that do not
 * self-synchronise exchange their 32-byte exit maps instead. */
int et_decode_sharded(et_group *group, const void *d_compressed, size_t len, void *d_out, size_t cap, size_t *written,
                      uint64_t *first_index);
/* The same in two steps, for a rank that holds only what it needs of the stream and sizes its output once it knows
 * its share.  Offsets count from compressed[0] (the .et file minus its first 4 bytes), which is taken to sit on a
 * 4-byte boundary; d_compressed above must be 4-byte aligned for the same reason.
 * et_decode_shard_window: which bytes of `compressed` rank `rank` of `world` needs besides the dictionary -- its
 *   8 KiB-block range with 16 bytes on either side (window_len 0: the rank holds no blocks); head = the first
 *   min(len, 8192) bytes of `compressed` in HOST memory (header + dictionary, decode.zig:34-141), len = all of it.
 * et_decode_sharded_begin (collective): d_window = bytes [window_off, window_off + window_len) of `compressed` in
 *   device memory (window_off a multiple of 4, d_window 4-byte aligned; at least the window above); cap = symbols
 *   the rank's output will hold, ~0 when it is sized afterwards.  *n_mine = the symbols this rank writes, the
 *   first of which is symbol *first_index of the text.
 * et_decode_sharded_write (rank-local): those symbols into d_out. */
int et_decode_shard_window(const uint8_t *head, size_t head_len, uint64_t len, int rank, int world, uint64_t *window_off, uint64_t *window_len);
int et_decode_sharded_begin(et_group *group, const uint8_t *head, size_t head_len, uint64_t len, const void *d_window, uint64_t window_off,
                            size_t window_len, uint64_t cap, uint64_t *n_mine, uint64_t *first_index);
int et_decode_sharded_write(et_group *group, void *d_out, size_t cap, size_t *written);

#ifdef __cplusplus
}
#endif
#endif /* ENTREEPY_HIP_H */
