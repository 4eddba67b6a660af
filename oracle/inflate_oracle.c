/*
 * inflate_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the reference's DEFLATE read
 * path (dsnet/compress `flate.Reader`), used as the parity oracle for the HIP
 * kernels and as the "port" CPU baseline in bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 * The product path (compress_amd/csrc) never links, imports or calls it.
 *
 * What it follows (reference file:line, all under /root/reference):
 *   flate/reader.go:58-88      Read loop: step, Flush -> InputOffset, flush-on-error
 *   flate/reader.go:100-134    readBlockHeader (BFINAL/BTYPE, stored LEN/NLEN check)
 *   flate/reader.go:137-159    readRawData
 *   flate/reader.go:162-250    readBlock (symbol loop, length/distance ranges, checks)
 *   flate/reader.go:253-259    finishBlock (ReadPads on the last block)
 *   flate/prefix.go:23-32      lenRanges / distRanges
 *   flate/prefix.go:35-63      fixed literal/distance code sets
 *   flate/prefix.go:67         clenLens order
 *   flate/prefix.go:83-194     ReadPrefixCodes
 *   flate/prefix.go:203-208    handleDegenerateCodes
 *   internal/prefix/prefix.go:326-384  GeneratePrefixes (canonical, bit-reversed)
 *   internal/prefix/decoder.go:51-136  Decoder.Init (two-level table, 9-bit first level)
 *   internal/prefix/reader.go:158-221  TryReadBits/ReadBits/TryReadSymbol/ReadSymbol
 *   internal/prefix/reader.go:113-119  ReadPads
 *   internal/prefix/reader.go:226-245  Flush: offset = ceil(consumed bits / 8)
 *   flate/dict_decoder.go:46-51,86-132 HistSize, forward overlap copy
 *   flate/common.go:34-42      errWrap: Invalid -> Corrupted
 *
 * Source model: the input is a memory buffer (the reference's *bytes.Reader /
 * compress.BufferedReader case, internal/prefix/reader.go:71-76), so
 * InputOffset after any step is ceil(consumed_bits / 8).
 *
 * Parity pinning: checked against every vector of flate/reader_test.go:54-833
 * (output bytes, inIdx, outIdx, error class) by tests/test_oracle_flate.py via
 * tests/golden/flate_reader.json, and against C zlib on valid streams (the
 * reference's own -zcheck idea, flate/reader_test.go:861-871).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define ORC_OK             0   /* io.EOF reached after a final block            */
#define ORC_CORRUPTED      1   /* errors.Corrupted                              */
#define ORC_UNEXPECTED_EOF 2   /* io.ErrUnexpectedEOF                           */
#define ORC_OUT_FULL       3   /* oracle-only: out_cap too small for the output */

#define MAX_HIST      32768    /* flate/common.go:16 */
#define END_BLOCK_SYM 256      /* flate/common.go:17 */
#define MAX_CLEN_SYMS 19       /* flate/prefix.go:17-19 */
#define MAX_LIT_SYMS  286
#define MAX_DIST_SYMS 30
#define COUNT_BITS    5        /* internal/prefix/prefix.go: countBits  */
#define COUNT_MASK    31
#define CHUNK_BITS_MAX 9       /* internal/prefix/decoder.go:83 */

typedef struct { uint32_t sym, len, val; } pcode;

/* internal/prefix/decoder.go:39-48 */
typedef struct {
    uint32_t chunks[1 << CHUNK_BITS_MAX];
    uint32_t links[512 * 64];          /* <= 2^9 link tables of <= 2^6 entries */
    uint32_t nchunks;                  /* len(pd.chunks); 0 = empty tree */
    uint32_t chunk_mask, link_mask, chunk_bits, num_links_each;
    uint32_t min_bits;
} pdecoder;

/* flate/prefix.go:23-32 */
static const uint16_t len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t  len_bits[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t  dist_bits[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
/* flate/prefix.go:67 */
static const uint8_t clen_order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

/* internal/common.go:43-45 ReverseUint32N */
static uint32_t reverse_n(uint32_t v, uint32_t n) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; i++) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

/* internal/prefix/prefix.go:326-384; returns 0 ok, -1 = errors.Invalid */
static int generate_prefixes(pcode *codes, int n) {
    if (n <= 1) {
        if (n == 1) { if (codes[0].len != 0) return -1; codes[0].val = 0; }
        return 0;
    }
    uint32_t bit_cnts[33] = {0}, next_codes[33] = {0};
    uint32_t min_bits = codes[0].len, max_bits = codes[0].len, sym_last = codes[0].sym;
    bit_cnts[codes[0].len]++;
    for (int i = 1; i < n; i++) {
        if (codes[i].sym <= sym_last) return -1;
        if (min_bits > codes[i].len) min_bits = codes[i].len;
        if (max_bits < codes[i].len) max_bits = codes[i].len;
        bit_cnts[codes[i].len]++;
        sym_last = codes[i].sym;
    }
    if (min_bits == 0) return -1;
    uint64_t code = 0;
    for (uint32_t i = min_bits; i <= max_bits; i++) {
        code <<= 1;
        next_codes[i] = (uint32_t)code;
        code += bit_cnts[i];
    }
    if (code != ((uint64_t)1 << max_bits)) return -1;   /* over- or under-subscribed */
    for (int i = 0; i < n; i++) {
        codes[i].val = reverse_n(next_codes[codes[i].len], codes[i].len);
        next_codes[codes[i].len]++;
    }
    return 0;
}

/* internal/prefix/decoder.go:51-136 */
static void decoder_init(pdecoder *pd, const pcode *codes, int n) {
    if (n == 0) { pd->nchunks = 0; return; }
    /* n == 1 with Len == 0 never happens on this path (handleDegenerateCodes). */
    uint32_t min_bits = 32, max_bits = 0;
    for (int i = 0; i < n; i++) {
        if (min_bits > codes[i].len) min_bits = codes[i].len;
        if (max_bits < codes[i].len) max_bits = codes[i].len;
    }
    pd->min_bits = min_bits;
    pd->chunk_bits = max_bits > CHUNK_BITS_MAX ? CHUNK_BITS_MAX : max_bits;
    pd->nchunks = 1u << pd->chunk_bits;
    pd->chunk_mask = pd->nchunks - 1;
    pd->link_mask = 0;
    pd->num_links_each = 0;
    if (pd->chunk_bits < max_bits) {
        uint32_t num_links = 1u << (max_bits - pd->chunk_bits);
        pd->link_mask = num_links - 1;
        pd->num_links_each = num_links;
        uint32_t link_idx = 0;
        memset(pd->chunks, 0, sizeof(uint32_t) * pd->nchunks);
        for (int i = 0; i < n; i++) {
            const pcode *c = &codes[i];
            if (c->len > pd->chunk_bits && pd->chunks[c->val & pd->chunk_mask] == 0) {
                pd->chunks[c->val & pd->chunk_mask] = (link_idx << COUNT_BITS) | (pd->chunk_bits + 1);
                link_idx++;
            }
        }
    }
    for (int i = 0; i < n; i++) {
        const pcode *c = &codes[i];
        uint32_t chunk = (c->sym << COUNT_BITS) | c->len;
        if (c->len <= pd->chunk_bits) {
            uint32_t skip = 1u << c->len;
            for (uint32_t j = c->val; j < pd->nchunks; j += skip) pd->chunks[j] = chunk;
        } else {
            uint32_t link_idx = pd->chunks[c->val & pd->chunk_mask] >> COUNT_BITS;
            uint32_t *links = pd->links + (size_t)link_idx * pd->num_links_each;
            uint32_t skip = 1u << (c->len - pd->chunk_bits);
            for (uint32_t j = c->val >> pd->chunk_bits; j < pd->num_links_each; j += skip) links[j] = chunk;
        }
    }
}

/* Bit reader over a memory buffer: the BufferedReader case of
 * internal/prefix/reader.go:252-316.  bits are LSB-first. */
typedef struct {
    const uint8_t *in;
    size_t in_len;
    size_t pos;        /* next byte to feed into buf */
    uint64_t buf;
    uint32_t nbits;
} bitrd;

static inline void br_fill(bitrd *br) {
    while (br->nbits <= 56 && br->pos < br->in_len) {
        br->buf |= (uint64_t)br->in[br->pos++] << br->nbits;
        br->nbits += 8;
    }
}
/* total bits still readable (in buf + not yet fed) */
static inline uint64_t br_avail(const bitrd *br) {
    return (uint64_t)br->nbits + 8ull * (br->in_len - br->pos);
}
/* internal/prefix/reader.go:226-245: Offset after Flush = ceil(consumed/8) */
static inline int64_t br_offset(const bitrd *br) {
    uint64_t consumed = 8ull * br->pos - br->nbits;
    return (int64_t)((consumed + 7) / 8);
}
/* ReadBits (reader.go:169-177): 0 ok, -1 = io.ErrUnexpectedEOF (nothing consumed) */
static inline int br_read_bits(bitrd *br, uint32_t nb, uint32_t *val) {
    if (br->nbits < nb) { br_fill(br); if (br->nbits < nb) return -1; }
    *val = (uint32_t)(br->buf & (((uint64_t)1 << nb) - 1));
    br->buf >>= nb;
    br->nbits -= nb;
    return 0;
}
/* ReadPads (reader.go:113-119) */
static inline void br_read_pads(bitrd *br) {
    uint32_t nb = br->nbits % 8;
    br->buf >>= nb;
    br->nbits -= nb;
}
/* ReadSymbol (reader.go:198-221): 0 ok, -1 unexpected EOF, -2 empty tree (Invalid->Corrupted) */
static inline int br_read_symbol(bitrd *br, const pdecoder *pd, uint32_t *sym) {
    if (pd->nchunks == 0) return -2;
    if (br->nbits < 32) br_fill(br);
    uint32_t chunk = pd->chunks[(uint32_t)br->buf & pd->chunk_mask];
    uint32_t nb = chunk & COUNT_MASK;
    if (nb > pd->chunk_bits) {
        uint32_t link_idx = chunk >> COUNT_BITS;
        chunk = pd->links[(size_t)link_idx * pd->num_links_each +
                          ((uint32_t)(br->buf >> pd->chunk_bits) & pd->link_mask)];
        nb = chunk & COUNT_MASK;
    }
    if (nb > br->nbits) return -1;       /* zero-extended lookup needs more bits than exist */
    br->buf >>= nb;
    br->nbits -= nb;
    *sym = chunk >> COUNT_BITS;
    return 0;
}

/* flate/prefix.go:203-208 */
static int handle_degenerate(pcode *codes, int n, uint32_t max_syms) {
    if (n != 1) return n;
    codes[1].sym = max_syms; codes[1].len = 1; codes[1].val = 0;
    return 2;
}

typedef struct {
    pdecoder clen, lit, dist, fixed_lit, fixed_dist;
    int fixed_ready;
} tables;

static void init_fixed(tables *t) {
    /* flate/prefix.go:35-63 */
    pcode lit[288], dist[32];
    for (int i = 0; i < 144; i++) { lit[i].sym = i; lit[i].len = 8; }
    for (int i = 144; i < 256; i++) { lit[i].sym = i; lit[i].len = 9; }
    for (int i = 256; i < 280; i++) { lit[i].sym = i; lit[i].len = 7; }
    for (int i = 280; i < 288; i++) { lit[i].sym = i; lit[i].len = 8; }
    generate_prefixes(lit, 288);
    decoder_init(&t->fixed_lit, lit, 288);
    for (int i = 0; i < 32; i++) { dist[i].sym = i; dist[i].len = 5; }
    generate_prefixes(dist, 32);
    decoder_init(&t->fixed_dist, dist, 32);
    t->fixed_ready = 1;
}

/* flate/prefix.go:83-194. 0 ok, 1 corrupted, 2 unexpected EOF.
 *
 * rfc_rep16: what code-length symbol 16 ("copy the previous length") repeats
 * after a 17/18 zero run.
 *   0 = the in-repo flate package: clenLast is only updated by literal lengths
 *       (flate/prefix.go:127-133,142), so it repeats the last LITERAL length;
 *   1 = RFC 1951 / Go stdlib compress/flate, which is the inflater
 *       xflate.Reader really calls (xflate/flate.go:9,43): it repeats the
 *       previous entry of the length array, i.e. 0 after a zero run.
 * zlib-style encoders never emit 16 after 17/18, but XFLATE meta blocks do
 * (meta/writer.go:236-245), and xflate/reader_test.go:261-274 only passes with
 * the stdlib rule ("raw chunk with final bit and bad size": its index block
 * has a repLast after a repZero). */
static int read_prefix_codes(bitrd *br, tables *t, int rfc_rep16) {
    uint32_t v;
    if (br_read_bits(br, 5, &v)) return ORC_UNEXPECTED_EOF;
    uint32_t num_lit = v + 257;
    if (br_read_bits(br, 5, &v)) return ORC_UNEXPECTED_EOF;
    uint32_t num_dist = v + 1;
    if (br_read_bits(br, 4, &v)) return ORC_UNEXPECTED_EOF;
    uint32_t num_clen = v + 4;
    if (num_lit > MAX_LIT_SYMS || num_dist > MAX_DIST_SYMS) return ORC_CORRUPTED;

    pcode clen_arr[MAX_CLEN_SYMS + 1];
    memset(clen_arr, 0, sizeof clen_arr);
    for (uint32_t i = 0; i < num_clen; i++) {
        if (br_read_bits(br, 3, &v)) return ORC_UNEXPECTED_EOF;
        if (v > 0) { clen_arr[clen_order[i]].sym = clen_order[i]; clen_arr[clen_order[i]].len = v; }
    }
    pcode code_clens[MAX_CLEN_SYMS + 1];
    int nclen = 0;
    for (int i = 0; i < MAX_CLEN_SYMS; i++)
        if (clen_arr[i].len > 0) code_clens[nclen++] = clen_arr[i];
    nclen = handle_degenerate(code_clens, nclen, MAX_CLEN_SYMS);
    if (generate_prefixes(code_clens, nclen)) return ORC_CORRUPTED;
    decoder_init(&t->clen, code_clens, nclen);

    pcode code_lits[MAX_LIT_SYMS + 1], code_dists[MAX_DIST_SYMS + 1];
    int nlit = 0, ndist = 0;
    uint32_t clen_last = 0;
    uint32_t max_syms = num_lit + num_dist;
#define APPEND_CODE(s, l) do { \
        if ((s) < num_lit) { code_lits[nlit].sym = (s); code_lits[nlit].len = (l); nlit++; } \
        else { code_dists[ndist].sym = (s) - num_lit; code_dists[ndist].len = (l); ndist++; } \
    } while (0)
    for (uint32_t sym = 0; sym < max_syms;) {
        uint32_t clen;
        int rc = br_read_symbol(br, &t->clen, &clen);
        if (rc == -1) return ORC_UNEXPECTED_EOF;
        if (rc == -2) return ORC_CORRUPTED;
        if (clen < 16) {
            if (clen > 0) APPEND_CODE(sym, clen);
            clen_last = clen;
            sym++;
        } else {
            uint32_t rep_cnt;
            switch (clen) {
            case 16:
                if (sym == 0) return ORC_CORRUPTED;
                clen = clen_last;
                if (br_read_bits(br, 2, &v)) return ORC_UNEXPECTED_EOF;
                rep_cnt = 3 + v;
                break;
            case 17:
                clen = 0;
                if (br_read_bits(br, 3, &v)) return ORC_UNEXPECTED_EOF;
                rep_cnt = 3 + v;
                if (rfc_rep16) clen_last = 0;
                break;
            case 18:
                clen = 0;
                if (br_read_bits(br, 7, &v)) return ORC_UNEXPECTED_EOF;
                rep_cnt = 11 + v;
                if (rfc_rep16) clen_last = 0;
                break;
            default:
                return ORC_CORRUPTED;      /* the degenerate filler symbol 19 */
            }
            if (clen > 0) {
                /* The reference appends first and checks after; an overrun past
                 * maxSyms would index beyond its fixed array for huge repeats, but
                 * sym+repCnt <= 316+6 always fits (flate/prefix.go:112,154-163). */
                uint32_t sym_end = sym + rep_cnt;
                for (; sym < sym_end; sym++)
                    if (sym < max_syms) APPEND_CODE(sym, clen);
            } else {
                sym += rep_cnt;
            }
            if (sym > max_syms) return ORC_CORRUPTED;
        }
    }
#undef APPEND_CODE
    nlit = handle_degenerate(code_lits, nlit, MAX_LIT_SYMS);
    if (generate_prefixes(code_lits, nlit)) return ORC_CORRUPTED;
    decoder_init(&t->lit, code_lits, nlit);
    ndist = handle_degenerate(code_dists, ndist, MAX_DIST_SYMS);
    if (generate_prefixes(code_dists, ndist)) return ORC_CORRUPTED;
    decoder_init(&t->dist, code_dists, ndist);
    return ORC_OK;
}

/*
 * Inflate one raw DEFLATE stream held in memory.
 *   in/in_len      compressed bytes (the whole virtual stream)
 *   out/out_cap    output buffer; every byte produced before an error is kept
 *   *in_off        <- flate.Reader.InputOffset  when Read returned its final error
 *   *out_off       <- flate.Reader.OutputOffset (bytes delivered)
 * returns ORC_*.
 */
int oracle_inflate_ex(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                      int64_t *in_off, int64_t *out_off, int rfc_rep16)
{
    static __thread tables t;     /* large; keep off the stack, one per thread */
    if (!t.fixed_ready) init_fixed(&t);
    bitrd br = { in, in_len, 0, 0, 0 };
    size_t op = 0;
    int status;
    uint32_t v;

    for (;;) {
        /* readBlockHeader, flate/reader.go:100-134 */
        if (br_read_bits(&br, 1, &v)) { status = ORC_UNEXPECTED_EOF; goto done; }
        int last = (v == 1);
        if (br_read_bits(&br, 2, &v)) { status = ORC_UNEXPECTED_EOF; goto done; }
        const pdecoder *lit_tree, *dist_tree;
        switch (v) {
        case 0: {
            br_read_pads(&br);
            uint32_t n, nn;
            if (br_read_bits(&br, 16, &n))  { status = ORC_UNEXPECTED_EOF; goto done; }
            if (br_read_bits(&br, 16, &nn)) { status = ORC_UNEXPECTED_EOF; goto done; }
            if ((n ^ nn) != 0xffff) { status = ORC_CORRUPTED; goto done; }
            /* readRawData, flate/reader.go:137-159: drain the bit buffer bytes
             * first (prefix/reader.go:124-137), then the underlying reader. */
            uint32_t blk = n;
            while (blk > 0) {
                if (op >= out_cap) { status = ORC_OUT_FULL; goto done; }
                if (br.nbits >= 8) {
                    out[op++] = (uint8_t)br.buf; br.buf >>= 8; br.nbits -= 8; blk--;
                    continue;
                }
                size_t avail = br.in_len - br.pos;
                size_t take = blk;
                if (take > avail) take = avail;
                if (take > out_cap - op) take = out_cap - op;
                if (take == 0) {
                    status = avail == 0 ? ORC_UNEXPECTED_EOF : ORC_OUT_FULL;
                    goto done;
                }
                memcpy(out + op, br.in + br.pos, take);
                op += take; br.pos += take; blk -= (uint32_t)take;
            }
            goto finish_block;
        }
        case 1:
            lit_tree = &t.fixed_lit; dist_tree = &t.fixed_dist;
            break;
        case 2:
            status = read_prefix_codes(&br, &t, rfc_rep16);
            if (status != ORC_OK) goto done;
            lit_tree = &t.lit; dist_tree = &t.dist;
            break;
        default:
            status = ORC_CORRUPTED;             /* reserved block type */
            goto done;
        }

        /* readBlock, flate/reader.go:162-250 */
        for (;;) {
            uint32_t lit_sym;
            int rc = br_read_symbol(&br, lit_tree, &lit_sym);
            if (rc == -1) { status = ORC_UNEXPECTED_EOF; goto done; }
            if (rc == -2) { status = ORC_CORRUPTED; goto done; }
            if (lit_sym < END_BLOCK_SYM) {
                if (op >= out_cap) { status = ORC_OUT_FULL; goto done; }
                out[op++] = (uint8_t)lit_sym;
                continue;
            }
            if (lit_sym == END_BLOCK_SYM) break;
            if (lit_sym >= MAX_LIT_SYMS) { status = ORC_CORRUPTED; goto done; }
            uint32_t extra;
            if (br_read_bits(&br, len_bits[lit_sym - 257], &extra)) { status = ORC_UNEXPECTED_EOF; goto done; }
            uint32_t cpy_len = len_base[lit_sym - 257] + extra;
            uint32_t dist_sym;
            rc = br_read_symbol(&br, dist_tree, &dist_sym);
            if (rc == -1) { status = ORC_UNEXPECTED_EOF; goto done; }
            if (rc == -2) { status = ORC_CORRUPTED; goto done; }
            if (dist_sym >= MAX_DIST_SYMS) { status = ORC_CORRUPTED; goto done; }
            if (br_read_bits(&br, dist_bits[dist_sym], &extra)) { status = ORC_UNEXPECTED_EOF; goto done; }
            uint32_t dist = dist_base[dist_sym] + extra;
            /* dictDecoder.HistSize, flate/dict_decoder.go:46-51 */
            size_t hist = op < MAX_HIST ? op : MAX_HIST;
            if (dist > hist) { status = ORC_CORRUPTED; goto done; }
            if (cpy_len > out_cap - op) {
                /* keep what fits so that out_off is exact up to the cap */
                for (; op < out_cap; op++) out[op] = out[op - dist];
                status = ORC_OUT_FULL; goto done;
            }
            /* forward byte-serial overlap copy, flate/dict_decoder.go:93-100 */
            if (dist >= cpy_len) { memcpy(out + op, out + op - dist, cpy_len); op += cpy_len; }
            else { for (uint32_t i = 0; i < cpy_len; i++, op++) out[op] = out[op - dist]; }
        }
    finish_block:
        /* finishBlock, flate/reader.go:253-259 */
        if (last) { br_read_pads(&br); status = ORC_OK; goto done; }
    }
done:
    *in_off = br_offset(&br);
    *out_off = (int64_t)op;
    return status;
}

/* The flate.Reader path (in-repo flate package semantics). */
int oracle_inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                   int64_t *in_off, int64_t *out_off)
{
    return oracle_inflate_ex(in, in_len, out, out_cap, in_off, out_off, 0);
}

/*
 * xflate.Reader's view of one chunk (xflate/reader.go:32-70,186-202): the
 * decompressor sees `chunk || 01 00 00 ff ff`.  tmp must hold comp_len + 5.
 */
int oracle_inflate_chunk(const uint8_t *comp, size_t comp_len, uint8_t *tmp,
                         uint8_t *out, size_t out_cap, int64_t *in_off, int64_t *out_off)
{
    static const uint8_t end_block[5] = {0x01, 0x00, 0x00, 0xff, 0xff};   /* xflate/common.go:57 */
    memcpy(tmp, comp, comp_len);
    memcpy(tmp + comp_len, end_block, 5);
    /* stdlib compress/flate is the inflater on this path: RFC rule for symbol 16 */
    return oracle_inflate_ex(tmp, comp_len + 5, out, out_cap, in_off, out_off, 1);
}

/*
 * Batch helper for the CPU baseline and the big parity runs: decode chunks
 * [first, first+count) of a descriptor table laid out exactly like the
 * product's xfg_chunk (5 x 8 bytes: comp_off, comp_size, raw_off, raw_size,
 * flags|reserved).  res = {int32 status, uint32 pad, int64 in, int64 out}.
 */
typedef struct { uint64_t comp_off, comp_size, raw_off, raw_size; uint32_t flags, reserved; } orc_chunk;
typedef struct { int32_t status; uint32_t reserved; int64_t in_consumed, out_produced; } orc_result;

int oracle_inflate_batch(const uint8_t *comp, const orc_chunk *chunks, size_t first, size_t count,
                         uint8_t *raw, orc_result *res, uint8_t *tmp /* >= max comp_size + 5 */)
{
    for (size_t i = first; i < first + count; i++) {
        const orc_chunk *c = &chunks[i];
        res[i].reserved = 0;
        res[i].status = oracle_inflate_chunk(comp + c->comp_off, c->comp_size, tmp,
                                             raw + c->raw_off, c->raw_size,
                                             &res[i].in_consumed, &res[i].out_produced);
    }
    return 0;
}
