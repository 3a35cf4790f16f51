/*
 * et_oracle.h -- CPU restatement of typio/entreepy's Huffman encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it, and
 * only as the checker.  The product path is entreepy_amd/csrc (HIP, gfx950).
 *
 * Parity pinning (see DESIGN.md "Oracle"): the reference is Zig and there is no Zig
 * toolchain in this image, so oracle/_ref cannot be built.  The reference's own
 * tests are round-trip only and hold no golden .et bytes; the single exact
 * known-answer it publishes is README.md:51 (res/nice.shakespeare.txt 477 B ->
 * 374 B), which this oracle reproduces, together with the three round-trip tests
 * of src/test.zig:35-72 run through the LITERAL decoder restatement below.
 * Byte-level content beyond that (pad-bit value, >= 2^32 length truncation) follows
 * Zig std.io.BitWriter semantics from recall: "parity unpinned" for those two.
 */
#ifndef ET_ORACLE_H
#define ET_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes returned (negated) by the int64 entry points */
enum {
    ET_ORACLE_OK = 0,
    ET_ORACLE_QUEUE_EMPTY = 1,  /* error.QueueEmpty: empty input (queue.zig:28-30 via encode.zig:137-138) */
    ET_ORACLE_NO_SPACE = 2,     /* out buffer too small (fixedBufferStream error.NoSpaceLeft) */
    ET_ORACLE_HANG = 3,         /* literal decoder would spin forever (decode.zig:166-200, quirk Q6/Q8) */
    ET_ORACLE_OOB = 4,          /* literal decoder would index out of bounds (decode.zig:124,182; Q9) */
    ET_ORACLE_FORMAT = 5        /* intended decoder: malformed stream */
};

/* Code table exactly as encode.zig:141-146: dictionary[256] of {data:u32, length:u8}. */
typedef struct {
    uint32_t data[256];
    uint8_t length[256];
} et_oracle_dict;

/* encode.zig:43-47 */
void et_oracle_histogram(const uint8_t *text, size_t n, uint64_t occ[256]);

/* encode.zig:54-214 (+ queue.zig:9-43).  Returns 0, or ET_ORACLE_QUEUE_EMPTY when no
 * symbol has a count >= 1.  dfs_order (optional, 256 entries) receives leaf symbols in
 * the order the -d dump prints them (encode.zig:204-212); *n_leaves their count. */
int et_oracle_build_dict(const uint64_t occ[256], et_oracle_dict *dict,
                         uint8_t *dfs_order, int *n_leaves);

/* encode.zig:253-299: magic, version, D, body length, bit-packed dictionary, zero pad.
 * text_len is the full usize; only its low 32 bits are emitted (quirk Q4).
 * Returns header byte count or -status. */
int64_t et_oracle_write_header(const et_oracle_dict *dict, uint64_t text_len,
                               uint8_t *out, size_t cap);

/* encode.zig:303-318 body pack for an ARBITRARY dictionary, starting at bit
 * position start_bit of a zeroed `out`.  Returns the end bit position or -status. */
int64_t et_oracle_pack_body(const et_oracle_dict *dict, const uint8_t *text, size_t n,
                            uint8_t *out, size_t cap, uint64_t start_bit);

/* encode.zig:25-337 whole call.  Returns bytes written (bits_written / 8) or -status. */
int64_t et_oracle_encode(const uint8_t *text, size_t n, uint8_t *out, size_t cap);

/* decode.zig:13-220 LITERAL restatement, quirks included (u32 window, 0 == absent,
 * no tail flush).  `compressed` is the .et file minus its first 4 bytes
 * (main.zig:204, test.zig:26).  Returns bytes written or -status. */
int64_t et_oracle_decode_ref(const uint8_t *compressed, size_t len, uint8_t *out, size_t cap);

/* Intended inverse of et_oracle_encode on well-formed streams: same header and
 * dictionary parse (decode.zig:34-141), then emits body_len symbols (or stops when
 * the bitstream holds no further complete code).  Same `compressed` convention. */
int64_t et_oracle_decode(const uint8_t *compressed, size_t len, uint8_t *out, size_t cap);

/* utils.zig:3-13 format_file_size (f32 argument).  Writes a NUL-terminated string. */
void et_oracle_format_file_size(float byte_count, char *buf, size_t cap);

/* decode.zig:34-141 as a code table: the stream's dictionary, where its body starts and the
 * header's length field.  Returns 0 or -ET_ORACLE_FORMAT. */
int64_t et_oracle_parse_dict(const uint8_t *compressed, size_t len, et_oracle_dict *dict,
                             uint64_t *body_start, uint32_t *body_length);

/* et_cpu_fast.c: the per-chunk pieces of a fast chunk-parallel CPU variant (bench.py's
 * all-cores baseline).  Same outputs as the restatement above. */
struct et_fast_tables;
size_t et_fast_tables_size(void);
int et_fast_build_tables(const et_oracle_dict *dict, struct et_fast_tables *t);
uint64_t et_fast_pack(const et_oracle_dict *dict, const uint8_t *text, size_t n, uint8_t *out, uint64_t start_bit);
uint64_t et_fast_walk(const struct et_fast_tables *t, const uint8_t *body, uint64_t total_bytes, uint64_t start_bit,
                      uint64_t end_bit, uint64_t max_syms, uint8_t *out, uint64_t *exit_bit, uint32_t *marks,
                      uint64_t mark_base, uint32_t n_marks);
int et_fast_merge(const struct et_fast_tables *t, const uint8_t *body, uint64_t total_bytes, uint64_t p,
                  const uint32_t *marks, uint64_t mark_base, uint32_t n_marks, uint64_t *extra, uint32_t *at);

#ifdef __cplusplus
}
#endif
#endif
