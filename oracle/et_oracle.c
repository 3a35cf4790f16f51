/*
 * et_oracle.c -- plain-C restatement of typio/entreepy src/encode.zig, src/queue.zig
 * and src/decode.zig (v1.1.0).  TEST INFRASTRUCTURE ONLY -- see et_oracle.h.
 *
 * Every function cites the reference lines it follows.  The restatement keeps the
 * reference's control structure (multi-pass sort, two ring queues, explicit-stack
 * DFS, one writeBits call per output bit, u32 sliding window) so that its quirks
 * fall out of the same arithmetic rather than being special-cased.
 */
#include "et_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* std.io.bitWriter(.big, fixedBufferStream) as used at encode.zig:256-257.   */
/* writeBits(v, k) appends the low k bits of v, most significant first;       */
/* flushBits() zero-fills the open byte (Zig std semantics, from recall).     */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint8_t *buf;
    size_t cap;
    size_t pos;       /* next byte to be written */
    uint8_t cur;      /* open byte, bits filled from the MSB side */
    unsigned cur_bits; /* 0..7 bits held in cur */
    int overflow;
} bitw;

static void bitw_init(bitw *w, uint8_t *buf, size_t cap)
{
    w->buf = buf; w->cap = cap; w->pos = 0; w->cur = 0; w->cur_bits = 0; w->overflow = 0;
}

static void bitw_write(bitw *w, uint64_t value, unsigned nbits)
{
    for (unsigned k = nbits; k > 0; k--) {
        unsigned bit = (unsigned)((value >> (k - 1)) & 1u);
        w->cur = (uint8_t)((w->cur << 1) | bit);
        w->cur_bits++;
        if (w->cur_bits == 8) {
            if (w->pos < w->cap) w->buf[w->pos] = w->cur; else w->overflow = 1;
            w->pos++;
            w->cur = 0; w->cur_bits = 0;
        }
    }
}

static void bitw_flush(bitw *w)
{
    if (w->cur_bits == 0) return;
    w->cur = (uint8_t)(w->cur << (8 - w->cur_bits));
    if (w->pos < w->cap) w->buf[w->pos] = w->cur; else w->overflow = 1;
    w->pos++;
    w->cur = 0; w->cur_bits = 0;
}

/* ------------------------------------------------------------------------- */
/* queue.zig:9-43  Queue(T, length): fixed ring, count/front/back.            */
/* ------------------------------------------------------------------------- */
#define QCAP 256
typedef struct {
    size_t count, front, back;
    int data[QCAP]; /* node indices (the reference stores *Node) */
} queue;

static void q_init(queue *q) { q->count = 0; q->front = 0; q->back = 0; }

static int q_enqueue(queue *q, int v) /* queue.zig:18-25 */
{
    if (q->count == QCAP) return -1; /* QueueFull */
    q->back = (q->back % QCAP) + 1;
    q->data[q->back - 1] = v;
    q->count += 1;
    return 0;
}

static int q_dequeue(queue *q, int *v) /* queue.zig:27-36 */
{
    if (q->count == 0) return -1; /* QueueEmpty */
    *v = q->data[q->front];
    q->front = (q->front + 1) % QCAP;
    q->count -= 1;
    return 0;
}

static int q_peek(const queue *q) { return q->data[q->front]; } /* queue.zig:38-41 */

/* ------------------------------------------------------------------------- */
void et_oracle_histogram(const uint8_t *text, size_t n, uint64_t occ[256]) /* encode.zig:43-47 */
{
    memset(occ, 0, 256 * sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) occ[text[i]] += 1;
}

typedef struct {
    int symbol; /* -1 == null */
    uint64_t weight;
    int left, right; /* -1 == null */
} node;

int et_oracle_build_dict(const uint64_t occ[256], et_oracle_dict *dict,
                         uint8_t *dfs_order, int *n_leaves)
{
    /* encode.zig:54-74: repeated min-selection passes; book_index is a u8 that
     * saturates at 255 (quirk Q1). */
    uint8_t sorted_letter_book[256];
    memset(sorted_letter_book, 0, sizeof sorted_letter_book);
    uint8_t book_index = 0;
    uint64_t min_value = 1;
    uint64_t next_min_value = 0;
    while (next_min_value != UINT64_MAX) {
        next_min_value = UINT64_MAX;
        for (int char_code = 0; char_code < 256; char_code++) {
            uint64_t occurences = occ[char_code];
            if (occurences < next_min_value && occurences > min_value) next_min_value = occurences;
            if (occurences == min_value) {
                sorted_letter_book[book_index] = (uint8_t)char_code;
                if (book_index < 255) book_index += 1;
            }
        }
        min_value = next_min_value;
    }
    const unsigned symbols_length = book_index; /* encode.zig:79 */

    /* encode.zig:82-100 */
    static const node null_node = { -1, 0, -1, -1 };
    node nodes[513];
    for (int i = 0; i < 513; i++) nodes[i] = null_node;
    unsigned nodes_index = 0;
    queue leaf_queue, sapling_queue;
    q_init(&leaf_queue); q_init(&sapling_queue);
    for (unsigned i = 0; i < symbols_length; i++) {
        uint8_t c = sorted_letter_book[i];
        nodes[i].symbol = c;
        nodes[i].weight = occ[c];
        q_enqueue(&leaf_queue, (int)i);
    }
    nodes_index = symbols_length;

    /* encode.zig:102-135 */
    while (leaf_queue.count + sapling_queue.count > 1) {
        int lowest[2] = { -1, -1 };
        for (int i = 0; i < 2; i++) {
            if (sapling_queue.count == 0) {
                if (q_dequeue(&leaf_queue, &lowest[i])) return ET_ORACLE_QUEUE_EMPTY;
            } else if (leaf_queue.count == 0) {
                if (q_dequeue(&sapling_queue, &lowest[i])) return ET_ORACLE_QUEUE_EMPTY;
            } else if (nodes[q_peek(&leaf_queue)].weight <= nodes[q_peek(&sapling_queue)].weight) {
                q_dequeue(&leaf_queue, &lowest[i]);
            } else {
                q_dequeue(&sapling_queue, &lowest[i]);
            }
        }
        nodes[nodes_index].symbol = -1;
        nodes[nodes_index].weight = nodes[lowest[0]].weight + nodes[lowest[1]].weight;
        nodes[nodes_index].left = lowest[0];
        nodes[nodes_index].right = lowest[1];
        q_enqueue(&sapling_queue, (int)nodes_index);
        nodes_index += 1;
    }

    /* encode.zig:137-138 */
    int root;
    if (leaf_queue.count > 0) {
        if (q_dequeue(&leaf_queue, &root)) return ET_ORACLE_QUEUE_EMPTY;
    } else {
        if (q_dequeue(&sapling_queue, &root)) return ET_ORACLE_QUEUE_EMPTY;
    }

    /* encode.zig:146 */
    memset(dict, 0, sizeof *dict);

    /* encode.zig:161-214: explicit stack; right child pushed first, so the left
     * subtree is popped first.  path.data is u32 (bits shifted out are lost, Q3),
     * path.length is u8. */
    struct { int node; uint32_t data; uint8_t length; } stack[513], t;
    size_t top = 1;
    stack[0].node = root; stack[0].data = 0; stack[0].length = 0;
    int leaves = 0;
    while (top > 0) {
        t = stack[top - 1];
        top -= 1;
        const node *nd = &nodes[t.node];
        if (nd->right != -1) {
            stack[top].node = nd->right;
            stack[top].data = (uint32_t)((t.data << 1) | 1u);
            stack[top].length = (uint8_t)(t.length + 1);
            top += 1;
        }
        if (nd->left != -1) {
            stack[top].node = nd->left;
            stack[top].data = (uint32_t)(t.data << 1);
            stack[top].length = (uint8_t)(t.length + 1);
            top += 1;
        }
        if (nd->right == -1 && nd->left == -1) {
            dict->data[nd->symbol] = t.data;
            dict->length[nd->symbol] = t.length;
            if (dfs_order) dfs_order[leaves] = (uint8_t)nd->symbol;
            leaves++;
        }
    }
    if (n_leaves) *n_leaves = leaves;
    return ET_ORACLE_OK;
}

/* One code, one writeBits(...,1) call per bit: encode.zig:291-295 and :309-313.
 * The shift amount is @truncate'd to u5. */
static void write_code_bits(bitw *w, uint32_t data, uint8_t length, uint64_t *bits_written)
{
    for (size_t j = length; j > 0; j--) {
        bitw_write(w, (data >> ((j - 1) & 31u)) & 1u, 1);
        *bits_written += 1;
    }
}

static uint64_t header_into(bitw *w, const et_oracle_dict *dict, uint64_t text_len)
{
    uint64_t bits_written = 0;
    bitw_write(w, 0xe7c0de, 24); bits_written += 24; /* encode.zig:262 */
    bitw_write(w, 0x01, 8); bits_written += 8;       /* encode.zig:266 */
    uint64_t dictionary_length = 0;                  /* encode.zig:270-276 */
    for (int i = 0; i < 256; i++) if (dict->length[i] > 0) dictionary_length += 1;
    if (dictionary_length > 0) dictionary_length -= 1;
    bitw_write(w, dictionary_length, 8); bits_written += 8;
    bitw_write(w, text_len, 32); bits_written += 32; /* encode.zig:279 (low 32 bits, Q4) */
    for (int i = 0; i < 256; i++) {                  /* encode.zig:285-297 */
        if (dict->length[i] > 0) {
            bitw_write(w, (uint64_t)i, 8); bits_written += 8;
            bitw_write(w, dict->length[i], 8); bits_written += 8;
            write_code_bits(w, dict->data[i], dict->length[i], &bits_written);
        }
    }
    bitw_flush(w);                                   /* encode.zig:298-299 */
    if (bits_written % 8 != 0) bits_written = (bits_written / 8 + 1) * 8;
    return bits_written;
}

int64_t et_oracle_write_header(const et_oracle_dict *dict, uint64_t text_len,
                               uint8_t *out, size_t cap)
{
    bitw w;
    bitw_init(&w, out, cap);
    uint64_t bits = header_into(&w, dict, text_len);
    if (w.overflow) return -ET_ORACLE_NO_SPACE;
    return (int64_t)(bits / 8);
}

int64_t et_oracle_pack_body(const et_oracle_dict *dict, const uint8_t *text, size_t n,
                            uint8_t *out, size_t cap, uint64_t start_bit)
{
    /* Same loop as encode.zig:307-314, but OR-ing into a zeroed buffer at an
     * arbitrary start bit so that shard concatenation can be checked. */
    uint64_t pos = start_bit;
    for (size_t i = 0; i < n; i++) {
        uint32_t data = dict->data[text[i]];
        uint8_t length = dict->length[text[i]];
        for (size_t j = length; j > 0; j--) {
            unsigned bit = (data >> ((j - 1) & 31u)) & 1u;
            if ((pos >> 3) >= cap) return -ET_ORACLE_NO_SPACE;
            out[pos >> 3] |= (uint8_t)(bit << (7 - (pos & 7)));
            pos++;
        }
    }
    return (int64_t)pos;
}

int64_t et_oracle_encode(const uint8_t *text, size_t n, uint8_t *out, size_t cap)
{
    uint64_t occ[256];
    et_oracle_histogram(text, n, occ);
    et_oracle_dict dict;
    int rc = et_oracle_build_dict(occ, &dict, NULL, NULL);
    if (rc) return -rc;

    bitw w;
    bitw_init(&w, out, cap);
    uint64_t bits_written = header_into(&w, &dict, (uint64_t)n);

    /* encode.zig:303-315.  The ten "writing_sections" slices are contiguous and
     * cover text[0..n) in order; they exist only to tick the progress bar. */
    const size_t writing_sections = 10;
    for (size_t s = 0; s < writing_sections; s++) {
        size_t lo = s * n / writing_sections, hi = (s + 1) * n / writing_sections;
        for (size_t i = lo; i < hi; i++)
            write_code_bits(&w, dict.data[text[i]], dict.length[text[i]], &bits_written);
    }
    bitw_flush(&w);                                  /* encode.zig:317-318 */
    if (bits_written % 8 != 0) bits_written = (bits_written / 8 + 1) * 8;
    if (w.overflow) return -ET_ORACLE_NO_SPACE;
    return (int64_t)(bits_written / 8);              /* encode.zig:336 */
}

/* ------------------------------------------------------------------------- */
/* decode.zig:34-141: header + bit-serial dictionary state machine.           */
/* The reference keeps an AutoHashMap(usize,[32]u8): key = code integer value, */
/* entry[len-1] = symbol, 0 == absent.  A hash map is a pure lookup, so a flat */
/* open table keyed by (value) restates it; `wide` lifts the [32] bound for    */
/* the intended decoder only.                                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint64_t key;
    uint8_t used;
    uint8_t entry[64]; /* reference: [32]u8 */
    uint8_t present[64]; /* intended decoder: distinguishes symbol 0 from absent */
} tbl_slot;

#define TBL_SLOTS 1024
typedef struct {
    tbl_slot slot[TBL_SLOTS];
} code_table;

static tbl_slot *tbl_find(code_table *t, uint64_t key, int create)
{
    size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 54) % TBL_SLOTS;
    for (size_t probe = 0; probe < TBL_SLOTS; probe++) {
        tbl_slot *s = &t->slot[(h + probe) % TBL_SLOTS];
        if (s->used) { if (s->key == key) return s; continue; }
        if (!create) return NULL;
        s->used = 1; s->key = key;
        return s;
    }
    return NULL;
}

typedef struct {
    unsigned dictionary_length; /* decode.zig:34 (u8 in the reference) */
    uint32_t body_length;       /* decode.zig:36-42 */
    unsigned longest_code;      /* decode.zig:44 (u8) */
    uint64_t shortest_code;     /* decode.zig:45 (usize, starts at maxInt) */
    size_t global_pos;          /* decode.zig:61: dictionary bytes consumed */
    int oob;                    /* a dictionary entry had length 0 or > entry bound */
} parsed_header;

static int parse_header(const uint8_t *ct, size_t len, code_table *tbl, parsed_header *ph,
                        unsigned entry_bound)
{
    memset(ph, 0, sizeof *ph);
    if (len < 5) return -1;
    ph->dictionary_length = (uint8_t)(ct[0] + 1);
    ph->body_length = ((uint32_t)ct[1] << 24) | ((uint32_t)ct[2] << 16) | ((uint32_t)ct[3] << 8) | ct[4];
    ph->longest_code = 0;
    ph->shortest_code = UINT64_MAX;

    int reading_dict_letter = 1, reading_dict_code_len = 0, reading_dict_code = 0;
    uint8_t current_letter = 0, current_code_length = 0;
    uint64_t build_bits = 0;
    size_t i = 0;
    uint8_t letters_read = 0;
    size_t global_pos = 0;

    for (size_t bi = 5; bi < len; bi++) {
        uint8_t byte = ct[bi];
        size_t pos = 0;
        for (;;) { /* read: while (true) */
            if (reading_dict_letter) {
                int brk = 0;
                while (i <= 7) {
                    if (pos > 7) { brk = 1; break; }
                    build_bits = (build_bits << 1) | ((byte >> (7 - pos)) & 1u);
                    pos += 1; i += 1;
                }
                if (brk) break;
                current_letter = (uint8_t)build_bits;
                reading_dict_letter = 0; reading_dict_code_len = 1;
                build_bits = 0; i = 0;
            }
            if (reading_dict_code_len) {
                int brk = 0;
                while (i <= 7) {
                    if (pos > 7) { brk = 1; break; }
                    build_bits = (build_bits << 1) | ((byte >> (7 - pos)) & 1u);
                    pos += 1; i += 1;
                }
                if (brk) break;
                current_code_length = (uint8_t)build_bits;
                if (current_code_length > ph->longest_code) ph->longest_code = current_code_length;
                if (current_code_length < ph->shortest_code) ph->shortest_code = current_code_length;
                reading_dict_code_len = 0; reading_dict_code = 1;
                build_bits = 0; i = 0;
            }
            if (reading_dict_code) {
                int brk = 0;
                while (i < current_code_length) {
                    if (pos > 7) { brk = 1; break; }
                    build_bits = (build_bits << 1) | ((byte >> (7 - pos)) & 1u);
                    pos += 1; i += 1;
                }
                if (brk) break;
                /* decode.zig:123-125 */
                if (current_code_length == 0 || current_code_length > entry_bound) {
                    ph->oob = 1;
                } else {
                    tbl_slot *s = tbl_find(tbl, build_bits, 1);
                    if (!s) return -1;
                    s->entry[current_code_length - 1] = current_letter;
                    s->present[current_code_length - 1] = 1;
                }
                letters_read += 1;
                reading_dict_code = 0; reading_dict_letter = 1;
                build_bits = 0; i = 0;
            }
        }
        global_pos += 1;
        if (letters_read == ph->dictionary_length) break; /* decode.zig:138-140 */
    }
    ph->global_pos = global_pos;
    return 0;
}

int64_t et_oracle_decode_ref(const uint8_t *ct, size_t len, uint8_t *out, size_t cap)
{
    code_table *tbl = (code_table *)calloc(1, sizeof *tbl);
    if (!tbl) return -ET_ORACLE_NO_SPACE;
    parsed_header ph;
    if (parse_header(ct, len, tbl, &ph, 32)) { free(tbl); return -ET_ORACLE_FORMAT; }
    if (ph.oob) { free(tbl); return -ET_ORACLE_OOB; }

    uint32_t bytes_written = 0;       /* decode.zig:14 */
    uint32_t window = 0;              /* decode.zig:143 */
    uint64_t window_len = 0;
    uint64_t checking_code_len = 2;
    uint64_t testing_code = 0;
    uint64_t decoded_letters_read = 0;
    int64_t rc = 0;

    const size_t decoding_sections = 30; /* decode.zig:152 */
    const size_t body_start = 5 + ph.global_pos;
    if (body_start > len) { free(tbl); return -ET_ORACLE_OOB; }
    const size_t body_length = len - body_start;
    for (size_t s = 0; s < decoding_sections && rc == 0; s++) {
        size_t lo = body_start + s * body_length / decoding_sections;
        size_t hi = body_start + (s + 1) * body_length / decoding_sections;
        for (size_t bi = lo; bi < hi && rc == 0; bi++) {
            uint8_t byte = ct[bi];
            window <<= 8;             /* decode.zig:161 (u32: high bits fall off, Q8) */
            window |= byte;
            window_len += 8;
            /* decode_text: */
            while (window_len >= ph.longest_code) {
                int progressed = 0, left_decode_text = 0;
                checking_code_len = ph.shortest_code;
                while (window_len >= checking_code_len) {
                    if (decoded_letters_read >= ph.body_length || window_len < checking_code_len) {
                        left_decode_text = 1;
                        break;
                    }
                    /* decode.zig:176-179, shift amounts truncated to u5 / u6 */
                    uint32_t mask = (uint32_t)((((uint32_t)1u << (checking_code_len & 31u)) - 1u)
                                               << ((window_len - checking_code_len) & 31u));
                    testing_code = window & mask;
                    testing_code >>= ((window_len - checking_code_len) & 63u);
                    tbl_slot *e = tbl_find(tbl, testing_code, 0);
                    if (e) {
                        if (checking_code_len - 1 >= 32) { rc = -ET_ORACLE_OOB; left_decode_text = 1; break; }
                        if (e->entry[checking_code_len - 1] > 0) {
                            uint8_t c = e->entry[checking_code_len - 1];
                            if (bytes_written >= cap) { rc = -ET_ORACLE_NO_SPACE; left_decode_text = 1; break; }
                            out[bytes_written] = c; /* decode.zig:186 writeByte */
                            bytes_written += 1;
                            decoded_letters_read += 1;
                            window = window & (((uint32_t)1u << ((window_len - checking_code_len) & 31u)) - 1u);
                            window_len -= checking_code_len;
                            checking_code_len = ph.shortest_code;
                            progressed = 1;
                        }
                    }
                    checking_code_len += 1;
                }
                if (left_decode_text) break;
                if (!progressed) {
                    /* No state changed and window_len >= longest_code still holds:
                     * the reference spins here forever (Q6, Q8). */
                    rc = -ET_ORACLE_HANG;
                    break;
                }
            }
        }
    }
    free(tbl);
    return rc ? rc : (int64_t)bytes_written; /* decode.zig:219 */
}

int64_t et_oracle_decode(const uint8_t *ct, size_t len, uint8_t *out, size_t cap)
{
    code_table *tbl = (code_table *)calloc(1, sizeof *tbl);
    if (!tbl) return -ET_ORACLE_NO_SPACE;
    parsed_header ph;
    if (parse_header(ct, len, tbl, &ph, 64)) { free(tbl); return -ET_ORACLE_FORMAT; }
    if (ph.oob) { free(tbl); return -ET_ORACLE_FORMAT; }
    const size_t body_start = 5 + ph.global_pos;
    if (body_start > len) { free(tbl); return -ET_ORACLE_FORMAT; }

    uint64_t written = 0;
    uint64_t value = 0;
    unsigned nbits = 0;
    int64_t rc = 0;
    const uint64_t total_bits = (uint64_t)(len - body_start) * 8;
    for (uint64_t p = 0; p < total_bits && written < ph.body_length; p++) {
        unsigned bit = (ct[body_start + (p >> 3)] >> (7 - (p & 7))) & 1u;
        value = (value << 1) | bit;
        nbits++;
        if (nbits > 64 || nbits > ph.longest_code) { rc = -ET_ORACLE_FORMAT; break; }
        tbl_slot *e = tbl_find(tbl, value, 0);
        if (e && e->present[nbits - 1]) {
            if (written >= cap) { rc = -ET_ORACLE_NO_SPACE; break; }
            out[written++] = e->entry[nbits - 1];
            value = 0; nbits = 0;
        }
    }
    free(tbl);
    return rc ? rc : (int64_t)written;
}

/* The dictionary of a stream as a code table (for et_cpu_fast.c's lookup decoder): the
 * parse is parse_header above (decode.zig:34-141).  Returns 0 or -ET_ORACLE_FORMAT (also
 * for a code longer than 32 bits, which the fast variant leaves to et_oracle_decode). */
int64_t et_oracle_parse_dict(const uint8_t *ct, size_t len, et_oracle_dict *dict, uint64_t *body_start, uint32_t *body_length)
{
    code_table *tbl = (code_table *)calloc(1, sizeof *tbl);
    if (!tbl) return -ET_ORACLE_NO_SPACE;
    parsed_header ph;
    int64_t rc = 0;
    if (parse_header(ct, len, tbl, &ph, 64) || ph.oob || 5 + ph.global_pos > len) rc = -ET_ORACLE_FORMAT;
    memset(dict, 0, sizeof *dict);
    for (size_t k = 0; k < TBL_SLOTS && !rc; k++) {
        const tbl_slot *s = &tbl->slot[k];
        if (!s->used) continue;
        for (unsigned l = 1; l <= 64; l++) {
            if (!s->present[l - 1]) continue;
            if (l > 32) { rc = -ET_ORACLE_FORMAT; break; }
            dict->data[s->entry[l - 1]] = (uint32_t)s->key;
            dict->length[s->entry[l - 1]] = (uint8_t)l;
        }
    }
    *body_start = 5 + ph.global_pos;
    *body_length = ph.body_length;
    free(tbl);
    return rc;
}

void et_oracle_format_file_size(float byte_count, char *buf, size_t cap) /* utils.zig:3-13 */
{
    if (byte_count < 1024.0f) snprintf(buf, cap, "%g B", (double)byte_count);
    else if (byte_count < 1024.0f * 1024.0f) snprintf(buf, cap, "%.2f KB", (double)(byte_count / 1024.0f));
    else if (byte_count < 1024.0f * 1024.0f * 1024.0f) snprintf(buf, cap, "%.2f MB", (double)(byte_count / (1024.0f * 1024.0f)));
    else snprintf(buf, cap, "%.2f GB", (double)(byte_count / (1024.0f * 1024.0f * 1024.0f)));
}
