// host_reader.cpp -- host-side mirror of the reference's XFLATE reader.
//
// The reference is Go and this image has no Go toolchain, so the host layer
// above the C ABI is C++ with the reference's names, argument meaning and
// error behaviour:
//
//   meta::reverse_search   xflate/internal/meta/meta.go:35-44      ReverseSearch
//   meta::Reader           xflate/internal/meta/reader.go:17-113   Reader / Read / Close
//   meta::Reader::decode_block  meta/reader.go:118-257             decodeBlock
//   Index                  xflate/index.go:14-139                  index / record
//   xfg_reader             xflate/reader.go:75-95                  Reader
//     reset                reader.go:124-157                       Reset
//     read                 reader.go:162-215                       Read
//     seek                 reader.go:221-279                       Seek
//     close                reader.go:282-291                       Close
//     decode_indexes       reader.go:296-347
//     decode_index         reader.go:354-424
//     decode_footer        reader.go:430-483
//
// What differs by design: the reference inflates one chunk at a time through
// Go's stdlib on the CPU (xflate/flate.go:43).  Here Read hands a run of
// consecutive index records to the GPU in one launch (xfg_decode) and serves
// bytes from that decoded window; Seek inside the window is a pointer move.
// The window starts at one chunk after a Seek and doubles while reads stay
// sequential, so random access keeps the reference's I/O budget
// (xflate/reader_test.go:392,504) and streaming reads get large launches.
// Meta blocks (index, footer) are decoded on the host: KiB-scale, once per stream.
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>
#include "xfg_internal.h"

#include "../../include/xflate_amd.h"

namespace {

typedef std::vector<uint8_t> Bytes;

// ------------------------------------------------------------------ meta ---
namespace meta {

const uint32_t kMagicVals = 0x05860004u, kMagicMask = 0xfffe3fc6u;    // meta.go:28-31
const int kMaxSyms = 257, kMinRepLast = 3, kMinRepZero = 11;          // meta.go:46-54
const int kMaxEncBytes = 64;                                          // meta.go:67
enum FinalMode { FinalNil = 0, FinalMeta = 1, FinalStream = 2 };      // meta.go:89-98

int reverse_search(const uint8_t *data, size_t n) {
    uint32_t magic = 0;
    for (size_t i = n; i-- > 0;) {
        magic = (magic << 8) | data[i];
        if ((magic & kMagicMask) == kMagicVals) return (int)i;
    }
    return -1;
}

// LSB-first bit source with prefix.Reader's contract: a read gets all its bits
// or fails with io.ErrUnexpectedEOF having consumed none
// (internal/prefix/reader.go:169-177); Offset after Flush = ceil(bits / 8) (:226-245).
struct Bits {
    const uint8_t *d; size_t n; uint64_t bit;
    uint64_t avail() const { return (uint64_t)n * 8 - bit; }
    bool read(unsigned nb, uint32_t *v) {
        if (nb > avail()) return false;
        uint32_t x = 0;
        for (unsigned k = 0; k < nb; k++) {
            uint64_t b = bit + k;
            x |= (uint32_t)((d[b >> 3] >> (b & 7)) & 1) << k;
        }
        bit += nb; *v = x; return true;
    }
    unsigned peek1(unsigned k) const {
        uint64_t b = bit + k;
        return b < (uint64_t)n * 8 ? (d[b >> 3] >> (b & 7)) & 1 : 0;
    }
    int64_t offset() const { return (int64_t)((bit + 7) / 8); }
};

struct Reader {
    int64_t input_offset, output_offset, num_blocks;
    int final_mode;
    Bits br;
    int final;
    Bytes buf;
    int err;                                  // XFG_ERR_*; XFG_ERR_EOF at a clean end

    void reset(const uint8_t *d, size_t n) {
        input_offset = output_offset = num_blocks = 0;
        final_mode = final = FinalNil;
        br.d = d; br.n = n; br.bit = 0;
        buf.clear();
        err = XFG_ERR_NONE;
    }

    // decodeBlock, meta/reader.go:118-257.  Returns an XFG_ERR_* class.
    int decode_block() {
        const int64_t off0 = br.offset();
        int e = decode_block_inner();
        input_offset += br.offset() - off0;
        return e;
    }

    int decode_block_inner() {
        uint32_t v;
        if (br.avail() < 1) return XFG_ERR_EOF;               // EOF is okay for the first bit
        uint32_t magic;
        if (!br.read(32, &magic)) return XFG_ERR_UNEXPECTED_EOF;
        if ((magic & kMagicMask) != kMagicVals) return XFG_ERR_CORRUPTED;
        const bool final_stream = (magic & 1) != 0;
        const unsigned pads = (magic >> 3) & 7;
        const unsigned num_hclen = 4 + ((magic >> 13) & 15);
        // `fail = fail || cond` short-circuits in the reference: once failed, no further bits are read.
        bool fail = num_hclen < 6;
        for (unsigned i = 5; i + 1 < num_hclen; i++) {
            if (fail) break;
            if (!br.read(3, &v)) return XFG_ERR_UNEXPECTED_EOF;
            fail = v != 0;
        }
        if (!fail) { if (!br.read(3, &v)) return XFG_ERR_UNEXPECTED_EOF; fail = v != 2; }
        if (!fail) { if (!br.read(1, &v)) return XFG_ERR_UNEXPECTED_EOF; fail = v != 0; }
        if (fail) return XFG_ERR_CORRUPTED;
        const unsigned huff_len = 8 - (num_hclen - 4) / 2;
        const unsigned huff_range = 1u << huff_len;

        // symbols: 0 -> zero, 10 -> one, 110 -> repeat last, 111 -> repeat zero (meta.go:100-137)
        uint8_t syms[33 + 32];
        memset(syms, 0, sizeof syms);
        unsigned nbits = 1;                                   // first symbol is symZero
        unsigned bit = 0, ones = 0;
        uint8_t fifo = 0xff;
        for (int idx = 0; idx < kMaxSyms - 1;) {
            unsigned cnt = 1;
            const uint64_t a = br.avail();
            if (a < 1) return XFG_ERR_UNEXPECTED_EOF;
            if (br.peek1(0) == 0) {
                br.read(1, &v);
                bit = 0;
                fifo = (uint8_t)(fifo >> 1);
            } else if (br.peek1(1) == 0) {
                if (a < 2) return XFG_ERR_UNEXPECTED_EOF;
                br.read(2, &v);
                bit = 1;
                fifo = (uint8_t)((fifo >> 2) | (1 << 6));
            } else {
                if (a < 3) return XFG_ERR_UNEXPECTED_EOF;
                const unsigned b2 = br.peek1(2);
                br.read(3, &v);
                if (b2 == 0) {                                // symRepLast
                    if (!br.read(2, &v)) return XFG_ERR_UNEXPECTED_EOF;
                    cnt = v + kMinRepLast;
                    fifo = (uint8_t)((fifo >> 3) | (3 << 5));
                    fifo = (uint8_t)((fifo >> 2) | (v << 6));
                } else {                                      // symRepZero
                    bit = 0;
                    if (!br.read(7, &v)) return XFG_ERR_UNEXPECTED_EOF;
                    cnt = v + kMinRepZero;
                    fifo = (uint8_t)((fifo >> 3) | (7 << 5));
                    fifo = (uint8_t)((fifo >> 7) | (v << 1));
                }
            }
            if (fifo == 0x00) return XFG_ERR_CORRUPTED;       // 8 zero bits in a row are forbidden
            for (unsigned i = 0; i < cnt; i++) {
                if (nbits < sizeof(syms) * 8 && bit) syms[nbits >> 3] |= (uint8_t)(1u << (nbits & 7));
                nbits++;
                ones += bit;
            }
            idx += (int)cnt;
        }
        if (nbits != (unsigned)kMaxSyms) return XFG_ERR_CORRUPTED;          // excessive number of meta symbols
        const int idx_eob = kMaxSyms - 1;
        if (ones != huff_range) return XFG_ERR_CORRUPTED;                   // degenerate meta prefix tree
        if ((syms[idx_eob / 8] & (1 << (idx_eob % 8))) == 0) return XFG_ERR_CORRUPTED;
        const uint8_t flags = syms[0];
        const bool final_meta = ((flags >> 1) & 1) != 0;
        const bool invert = ((flags >> 2) & 1) != 0;
        const unsigned size = (flags >> 3) & 31;
        Bytes out(syms + 1, syms + 1 + size);
        if (invert) for (size_t i = 0; i < out.size(); i++) out[i] = (uint8_t)~out[i];
        const int fin = (final_meta ? 1 : 0) + (final_stream ? 1 : 0);
        if (final_stream && !final_meta) return XFG_ERR_CORRUPTED;
        // footer
        fail = false;
        if (!br.read(pads, &v)) return XFG_ERR_UNEXPECTED_EOF;
        fail = v > 0;
        if (!fail) { if (!br.read(1, &v)) return XFG_ERR_UNEXPECTED_EOF; fail = v > 0; }
        if (!fail) { if (!br.read(huff_len, &v)) return XFG_ERR_UNEXPECTED_EOF; fail = v != huff_range - 1; }
        if (!fail) fail = (br.bit % 8) > 0;
        if (fail) return XFG_ERR_CORRUPTED;
        buf.swap(out);
        final = fin;
        num_blocks++;
        return XFG_ERR_NONE;
    }

    // io.Copy(dst, mr): appends to dst, returns XFG_ERR_NONE at io.EOF or the error.
    int read_all(Bytes &dst) {
        for (;;) {
            if (err != XFG_ERR_NONE) break;
            if (final != FinalNil) { final_mode = final; err = XFG_ERR_EOF; break; }
            int e = decode_block();
            if (e != XFG_ERR_NONE) { err = e; break; }
            dst.insert(dst.end(), buf.begin(), buf.end());
            output_offset += (int64_t)buf.size();
            buf.clear();
        }
        return err == XFG_ERR_EOF ? XFG_ERR_NONE : err;
    }
};

} // namespace meta

// ----------------------------------------------------------------- index ---
struct Index {
    std::vector<xfg_record> records;
    int64_t back_size, index_size;
    Index() : back_size(0), index_size(0) {}
    void reset() { records.clear(); back_size = index_size = 0; }
    xfg_record last_record() const {
        xfg_record z; memset(&z, 0, sizeof z);
        return records.empty() ? z : records.back();
    }
    // AppendRecord, index.go:40-56
    bool append_record(int64_t comp_size, int64_t raw_size, int typ) {
        if (raw_size < 0 || comp_size < 0) return false;
        const xfg_record last = last_record();
        xfg_record rec; memset(&rec, 0, sizeof rec);
        rec.comp_offset = (int64_t)((uint64_t)last.comp_offset + (uint64_t)comp_size);
        rec.raw_offset = (int64_t)((uint64_t)last.raw_offset + (uint64_t)raw_size);
        rec.type = typ;
        if (rec.comp_offset < last.comp_offset || rec.raw_offset < last.raw_offset) return false;   // overflow
        records.push_back(rec);
        return true;
    }
    // AppendIndex, index.go:60-71 (atomic)
    bool append_index(const Index &other) {
        int64_t pc = 0, pr = 0;
        for (size_t i = 0; i < other.records.size(); i++) {
            const xfg_record &r = other.records[i];
            if (!append_record(r.comp_offset - pc, r.raw_offset - pr, r.type)) {
                records.resize(records.size() - i);
                return false;
            }
            pc = r.comp_offset; pr = r.raw_offset;
        }
        return true;
    }
    // Search, index.go:93-109
    int64_t search(int64_t offset) const {
        int64_t i = -1, imin = 0, imax = (int64_t)records.size() - 1;
        while (imax >= imin && i == -1) {
            const int64_t imid = (imin + imax) / 2;
            const bool gte_curr = offset >= records[imid].raw_offset;
            const bool lt_next = imid + 1 >= (int64_t)records.size() || offset < records[imid + 1].raw_offset;
            if (gte_curr && lt_next) i = imid;
            else if (gte_curr) imin = imid + 1;
            else imax = imid - 1;
        }
        return i + 1;
    }
    // GetRecords, index.go:115-130
    void get_records(int64_t i, xfg_record *prev, xfg_record *curr) const {
        const int64_t n = (int64_t)records.size();
        memset(prev, 0, sizeof *prev);
        if (i > n) i = n;
        if (i - 1 >= 0 && i - 1 < n) *prev = records[i - 1];
        if (i >= 0 && i < n) *curr = records[i];
        else { *curr = *prev; curr->type = XFG_REC_UNKNOWN; }
    }
};

// encoding/binary.Uvarint: n > 0 ok, n == 0 buffer too small, n < 0 overflow.
int uvarint(const uint8_t *b, size_t len, uint64_t *out) {
    uint64_t x = 0; unsigned s = 0;
    for (size_t i = 0; i < len; i++) {
        if (i == 10) return -(int)(i + 1);
        if (b[i] < 0x80) {
            if (i == 9 && b[i] > 1) return -(int)(i + 1);
            *out = x | ((uint64_t)b[i] << s);
            return (int)i + 1;
        }
        x |= (uint64_t)(b[i] & 0x7f) << s;
        s += 7;
    }
    *out = 0;
    return 0;
}

// hash/crc32 ChecksumIEEE
struct Crc32Table {
    uint32_t t[256];
    Crc32Table() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            t[i] = c;
        }
    }
};
uint32_t crc32_ieee(const uint8_t *p, size_t n) {
    static const Crc32Table tab;                            // function-local static: initialised once, thread-safe (C++11)
    const uint32_t *table = tab.t;
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

const uint8_t kMagic[3] = {'X', 'F', 0x00};                 // xflate/common.go:52
const int64_t kMaxInt64 = 0x7fffffffffffffffLL;

} // namespace

// ---------------------------------------------------------------- reader ---
// bytes.NewReader over memory the caller keeps alive: the io.ReadSeeker of xfg_reader_new_mem
namespace {
struct MemSource { const uint8_t *p; int64_t len, pos; };
int64_t mem_read(void *user, uint8_t *buf, int64_t n) {
    MemSource *m = (MemSource *)user;
    const int64_t k = std::max<int64_t>(0, std::min<int64_t>(n, m->len - m->pos));
    if (k > 0) memcpy(buf, m->p + m->pos, (size_t)k);
    m->pos += k;
    return k;
}
int64_t mem_seek(void *user, int64_t off, int whence) {                // bytes.Reader.Seek
    MemSource *m = (MemSource *)user;
    int64_t base;
    switch (whence) {
    case 0: base = 0; break;
    case 1: base = m->pos; break;
    case 2: base = m->len; break;
    default: return -1;
    }
    if (off < -base) return -1;
    m->pos = base + off;
    return m->pos;
}
}

struct xfg_reader {
    xfg_ctx *ctx;
    xfg_read_fn rd_fn; xfg_seek_fn sk_fn; void *user;
    int64_t bytes_read;

    int64_t ri;            // current record number
    int64_t offset;        // current raw offset
    int64_t discard;       // bytes to drop to reach offset
    Index idx;
    int64_t chk_csize, chk_rsize; int chk_typ;
    int err;               // sticky XFG_ERR_* (XFG_ERR_EOF included)

    // "zr": the decoded state of the current chunk
    int64_t cur_rec;       // record index of the current chunk, -1 = none
    int64_t cur_comp_pos;
    int64_t cur_rd;        // bytes of the current chunk already handed out (== flateReader.OutputOffset)
    int64_t cur_in_off;    // flateReader.InputOffset, valid once the chunk hit its end

    // decoded window: records [win_first, win_first + win_res.size())
    int64_t win_first;
    std::vector<xfg_result> win_res;
    std::vector<xfg_chunk> win_chunks;
    // the window's compressed and decoded bytes: grown, never zero-filled (a std::vector's resize costs as much as the
    // decode), kept from window to window
    struct Span {
        uint8_t *p; size_t cap, len;
        Span() : p(nullptr), cap(0), len(0) {}
        ~Span() { free(p); }
        Span(const Span &) = delete;
        Span &operator=(const Span &) = delete;
        bool fit(size_t n) {                                    // contents are not kept
            len = 0;
            if (n <= cap) return true;
            free(p);
            cap = n + n / 2 + 4096;
            p = (uint8_t *)malloc(cap);
            if (!p) { cap = 0; return false; }
            return true;
        }
    };
    Span win_raw, win_comp;
    std::vector<uint8_t> win_sync_ok;
    int64_t max_window_raw;
    int64_t want_raw = 0;        // bytes the Read in progress asks for: the window covers them if it has to be decoded anyway
    uint8_t *direct_dst = nullptr;   // set by a large Read: decode the window straight into the caller's buffer ...
    int64_t direct_cap = 0;          // ... of this many bytes (whole chunks only)
    bool win_direct = false;         // the window's bytes lie in the caller's buffer: valid during that Read only
    void *owned_src = nullptr;   // the MemSource of xfg_reader_new_mem (freed with the reader; a Reset to callbacks leaves it unused)
    int64_t ahead;         // chunks to decode in the next launch (doubles while sequential)

    // ---- io.ReadSeeker plumbing
    int64_t rs_seek(int64_t off, int whence) { return sk_fn(user, off, whence); }
    // io.Copy(dst, LimitedReader{rd, n}): read up to n bytes, stop early at EOF.  <0 = error
    int64_t rs_read_n(Bytes &dst, int64_t n) {
        int64_t got = 0;
        uint8_t tmp[1 << 16];
        while (got < n) {
            int64_t want = std::min<int64_t>(n - got, (int64_t)sizeof tmp);
            int64_t k = rd_fn(user, tmp, want);
            if (k < 0) return -1;
            if (k == 0) break;
            dst.insert(dst.end(), tmp, tmp + k);
            got += k; bytes_read += k;
        }
        return got;
    }

    // the same into a Span's memory, in large pieces (one call of the callback per piece)
    int64_t rs_read_into(Span &dst, int64_t n) {
        if (!dst.fit((size_t)n + 16)) return -1;
        int64_t got = 0;
        while (got < n) {
            const int64_t want = std::min<int64_t>(n - got, 16 << 20);
            const int64_t k = rd_fn(user, dst.p + got, want);
            if (k < 0) return -1;
            if (k == 0) break;
            got += k; bytes_read += k;
        }
        dst.len = (size_t)got;
        return got;
    }

    int decode_footer(int64_t *back_size, int64_t *foot_size);
    int decode_indexes(int64_t back_size);
    int decode_index(Index *idx);
    int reset(xfg_read_fn rd, xfg_seek_fn sk, void *u);
    int64_t seek(int64_t off, int whence, int *e);
    int64_t read(uint8_t *buf, int64_t n, int *e);
    int close();
    int ensure_decoded();
    int64_t zr_read(uint8_t *buf, int64_t n, int *e);
};

// decodeFooter, xflate/reader.go:430-483
int xfg_reader::decode_footer(int64_t *back_size, int64_t *foot_size) {
    int64_t end = rs_seek(0, 2);
    if (end < 0) return XFG_ERR_IO;
    if (end > meta::kMaxEncBytes) end = meta::kMaxEncBytes;
    if (rs_seek(-end, 2) < 0) return XFG_ERR_IO;
    Bytes br;
    if (rs_read_n(br, end + 1) < 0) return XFG_ERR_IO;          // io.Copy reads to EOF
    const int i = meta::reverse_search(br.data(), br.size());
    if (i < 0) return XFG_ERR_CORRUPTED;
    meta::Reader mr;
    mr.reset(br.data() + i, br.size() - i);
    Bytes bw;
    int e = mr.read_all(bw);
    if (e != XFG_ERR_NONE) return e;
    if ((size_t)mr.br.offset() != br.size() - i || mr.num_blocks != 1) return XFG_ERR_CORRUPTED;
    if (mr.final_mode != meta::FinalStream) return XFG_ERR_CORRUPTED;
    if (rs_seek(-mr.input_offset, 1) < 0) return XFG_ERR_IO;
    if (bw.size() < 3 || memcmp(bw.data(), kMagic, 3) != 0) return XFG_ERR_CORRUPTED;
    uint64_t bs;
    const int cnt = uvarint(bw.data() + 3, bw.size() - 3, &bs);
    if (cnt <= 0) return XFG_ERR_CORRUPTED;
    if (bw.size() - 3 - (size_t)cnt > 0) return XFG_ERR_CORRUPTED;
    *back_size = (int64_t)bs;                                    // may wrap negative, as in the reference
    *foot_size = mr.input_offset;
    return XFG_ERR_NONE;
}

// decodeIndex, xflate/reader.go:354-424
int xfg_reader::decode_index(Index *ix) {
    Bytes br;
    const int64_t n = rs_read_n(br, ix->index_size);
    if (n < 0) return XFG_ERR_IO;
    if (rs_seek(-n, 1) < 0) return XFG_ERR_IO;
    meta::Reader mr;
    mr.reset(br.data(), br.size());
    Bytes bw;
    int e = mr.read_all(bw);
    if (e != XFG_ERR_NONE) return e;
    uint32_t crc = 0;
    if (bw.size() > 4) crc = crc32_ieee(bw.data(), bw.size() - 4);
    size_t pos = 0;
    bool err_vli = false;
    auto read_vli = [&]() -> int64_t {
        uint64_t x;
        const int k = uvarint(bw.data() + pos, bw.size() - pos, &x);
        if (k <= 0 || x > (uint64_t)kMaxInt64) { err_vli = true; return 0; }
        pos += (size_t)k;
        return (int64_t)x;
    };
    ix->back_size = read_vli();
    const int64_t num_recs = read_vli();
    const int64_t total_comp = read_vli();
    const int64_t total_raw = read_vli();
    if (err_vli) return XFG_ERR_CORRUPTED;
    std::vector<std::pair<int64_t, int64_t> > chunks;
    for (int64_t i = 0; i < num_recs; i++) {
        const int64_t c = read_vli();
        const int64_t r = read_vli();
        chunks.push_back(std::make_pair(c, r));
        // The reference keeps appending zero chunks after a VLI error and fails the
        // checks below; one zero chunk is enough to fail them the same way.
        if (err_vli) break;
    }
    if (bw.size() - pos != 4) return XFG_ERR_CORRUPTED;
    const uint32_t got = (uint32_t)bw[pos] | ((uint32_t)bw[pos + 1] << 8) | ((uint32_t)bw[pos + 2] << 16) | ((uint32_t)bw[pos + 3] << 24);
    if (got != crc) return XFG_ERR_CORRUPTED;
    if (mr.final_mode != meta::FinalMeta) return XFG_ERR_CORRUPTED;
    if (mr.input_offset != ix->index_size) return XFG_ERR_CORRUPTED;
    for (size_t i = 0; i < chunks.size(); i++) {
        if (chunks[i].first <= 4) return XFG_ERR_CORRUPTED;      // every chunk has a sync marker
        if (!ix->append_record(chunks[i].first, chunks[i].second, XFG_REC_DEFLATE)) return XFG_ERR_CORRUPTED;
    }
    const xfg_record last = ix->last_record();
    if (last.comp_offset != total_comp || last.raw_offset != total_raw) return XFG_ERR_CORRUPTED;
    return XFG_ERR_NONE;
}

// decodeIndexes, xflate/reader.go:296-347
int xfg_reader::decode_indexes(int64_t back_size) {
    int64_t pos = rs_seek(0, 1);
    if (pos < 0) return XFG_ERR_IO;
    int64_t comp_size = 0;
    std::vector<Index> idxs;
    for (;;) {
        // int64 arithmetic that may wrap, as in the reference
        const int64_t new_pos = (int64_t)((uint64_t)pos - ((uint64_t)back_size + (uint64_t)comp_size));
        if (new_pos < 0 || new_pos > pos) return XFG_ERR_CORRUPTED;
        pos = rs_seek(new_pos, 0);
        if (pos < 0) return XFG_ERR_IO;
        if (back_size == 0) break;
        idxs.push_back(Index());
        Index *ix = &idxs.back();
        ix->index_size = back_size;
        int e = decode_index(ix);
        if (e != XFG_ERR_NONE) return e;
        back_size = ix->back_size;
        comp_size = ix->last_record().comp_offset;
    }
    if (pos != 0) return XFG_ERR_CORRUPTED;
    for (size_t i = idxs.size(); i-- > 0;) {
        if (!idx.append_index(idxs[i])) return XFG_ERR_CORRUPTED;
        if (!idx.append_record(idxs[i].index_size, 0, XFG_REC_INDEX)) return XFG_ERR_CORRUPTED;
    }
    return XFG_ERR_NONE;
}

// Reset, xflate/reader.go:124-157
int xfg_reader::reset(xfg_read_fn rd, xfg_seek_fn sk, void *u) {
    rd_fn = rd; sk_fn = sk; user = u;
    bytes_read = 0;
    ri = 0; offset = 0; discard = 0;
    idx.reset();
    chk_csize = chk_rsize = 0; chk_typ = XFG_REC_UNKNOWN;
    err = XFG_ERR_NONE;
    cur_rec = -1; cur_comp_pos = 0; cur_rd = 0; cur_in_off = 0;
    win_first = 0; win_res.clear(); win_chunks.clear();
    ahead = 1;
    int64_t back_size = 0, foot_size = 0;
    if ((err = decode_footer(&back_size, &foot_size)) != XFG_ERR_NONE) return err;
    if ((err = decode_indexes(back_size)) != XFG_ERR_NONE) return err;
    if (!idx.append_record(foot_size, 0, XFG_REC_FOOTER)) { err = XFG_ERR_CORRUPTED; return err; }
    int e = XFG_ERR_NONE;
    seek(0, 0, &e);
    err = e;
    return err;
}

// Seek, xflate/reader.go:221-279
int64_t xfg_reader::seek(int64_t off, int whence, int *e) {
    *e = XFG_ERR_NONE;
    if (err != XFG_ERR_NONE && err != XFG_ERR_EOF) { *e = err; return 0; }
    int64_t pos;
    const int64_t end = idx.last_record().raw_offset;
    switch (whence) {
    case 0: pos = off; break;
    case 1: pos = (int64_t)((uint64_t)offset + (uint64_t)off); break;
    case 2: pos = (int64_t)((uint64_t)end + (uint64_t)off); break;
    default: *e = XFG_ERR_INVALID; return 0;                    // invalid whence
    }
    if (pos < 0) { *e = XFG_ERR_INVALID; return 0; }            // negative position

    // Still inside the current chunk and moving forward: only the discard count changes.
    const int64_t dis = pos - offset;
    const int64_t remain = chk_rsize - cur_rd;
    if (dis > 0 && remain > 0 && dis < remain) {
        offset = pos; discard = dis;
        return pos;
    }

    xfg_record prev, curr;
    idx.get_records(ri, &prev, &curr);
    if (!(prev.raw_offset <= pos && pos <= curr.raw_offset)) {
        ri = idx.search(pos);
        idx.get_records(ri, &prev, &curr);
        ahead = 1;                                              // not sequential: shrink the read-ahead
    }
    const int64_t rec = ri;
    ri++;
    if (ri > (int64_t)idx.records.size()) ri = (int64_t)idx.records.size();

    chk_csize = curr.comp_offset - prev.comp_offset;
    chk_rsize = curr.raw_offset - prev.raw_offset;
    chk_typ = curr.type;
    offset = pos; discard = pos - prev.raw_offset;
    if (pos > end) discard = end - prev.raw_offset;
    // cr.Reset + zr.Reset: a fresh, not yet decoded chunk starting at prev.CompOffset
    cur_rec = rec < (int64_t)idx.records.size() ? rec : -1;
    cur_comp_pos = prev.comp_offset;
    cur_rd = 0; cur_in_off = 0;
    err = XFG_ERR_NONE;
    return pos;
}

static uint32_t flags_for(int typ) {
    switch (typ) {
    case XFG_REC_DEFLATE: return XFG_CHUNK_XFLATE_DEFLATE;
    // index and footer records were decoded block by block when the index was parsed (decode_index, decode_footer:
    // every meta block valid, the record consumed to its last byte): the device does not inflate them a second time
    case XFG_REC_INDEX: return XFG_CHUNK_XFLATE_INDEX | XFG_CHUNK_HOST_META;
    default: return XFG_CHUNK_XFLATE_FOOTER | XFG_CHUNK_HOST_META;
    }
}

// Make sure record cur_rec is in the decoded window; launch the GPU on a run of records if not.
int xfg_reader::ensure_decoded() {
    if (cur_rec >= win_first && cur_rec < win_first + (int64_t)win_res.size()) return XFG_ERR_NONE;
    const int64_t nrec = (int64_t)idx.records.size();
    int64_t last = cur_rec;                                     // inclusive
    int64_t raw_total = 0, comp_total = 0;
    xfg_record p0, c0;
    idx.get_records(cur_rec, &p0, &c0);
    for (int64_t k = cur_rec; k < nrec && (k < cur_rec + ahead || raw_total < want_raw); k++) {
        xfg_record p, c;
        idx.get_records(k, &p, &c);
        const int64_t cs = c.comp_offset - p.comp_offset, rs = c.raw_offset - p.raw_offset;
        if (k > cur_rec && (raw_total + rs > max_window_raw || comp_total + cs > max_window_raw)) break;
        if (direct_dst && raw_total + rs > direct_cap) break;   // (the caller made sure that the first chunk fits)
        if ((uint64_t)cs > XFG_MAX_CHUNK_COMP || (uint64_t)rs > XFG_MAX_CHUNK_RAW) {
            if (k == cur_rec) return XFG_ERR_LIMIT;             // documented per-chunk limit of the kernels (the reference has none)
            break;
        }
        raw_total += rs; comp_total += cs; last = k;
    }
    const size_t n = (size_t)(last - cur_rec + 1);
    if (ahead < (1 << 20)) ahead = std::max<int64_t>(ahead * 2, std::min<int64_t>(2 * (int64_t)n, 1 << 20));

    const uint8_t *comp_ptr;
    int64_t got;
    if (rd_fn == mem_read) {
        // a stream in memory is decoded from where it lies
        MemSource *m = (MemSource *)user;
        got = std::max<int64_t>(0, std::min<int64_t>(comp_total, m->len - p0.comp_offset));
        comp_ptr = m->p + std::min<int64_t>(p0.comp_offset, m->len);
        m->pos = std::min<int64_t>(p0.comp_offset, m->len) + got;
        bytes_read += got;
    } else {
        if (rs_seek(p0.comp_offset, 0) < 0) return XFG_ERR_IO;
        got = rs_read_into(win_comp, comp_total);
        if (got < 0) return XFG_ERR_IO;
        comp_ptr = win_comp.p;
    }
    // A short read means the index promised more bytes than the stream has; the chunk
    // reader of the reference then ends the chunk early.  Shrink the affected chunks.
    win_chunks.resize(n);
    win_res.resize(n);
    win_sync_ok.resize(n);
    win_direct = direct_dst != nullptr;
    if (!win_direct && !win_raw.fit((size_t)raw_total + 16)) return XFG_ERR_INTERNAL;
    int64_t co = 0, ro = 0;
    for (size_t i = 0; i < n; i++) {
        xfg_record p, c;
        idx.get_records(cur_rec + (int64_t)i, &p, &c);
        int64_t cs = c.comp_offset - p.comp_offset;
        const int64_t rs = c.raw_offset - p.raw_offset;
        xfg_chunk &ch = win_chunks[i];
        memset(&ch, 0, sizeof ch);
        ch.comp_off = (uint64_t)co;
        const int64_t have = std::max<int64_t>(0, std::min<int64_t>(cs, got - co));
        ch.comp_size = (uint64_t)have;
        ch.raw_off = (uint64_t)ro;
        ch.raw_size = (uint64_t)rs;
        ch.flags = flags_for(c.type);
        if (have != cs) ch.flags &= ~(XFG_CHUNK_VERIFY | XFG_CHUNK_HOST_META);   // sizes are re-checked below against cs; a short record is inflated for real
        co += cs; ro += rs;
    }
    int rc = xfg_decode(ctx, comp_ptr, (size_t)got, win_chunks.data(), n,
                        win_direct ? direct_dst : win_raw.p, (size_t)raw_total, win_res.data());
    if (rc != XFG_OK) { win_res.clear(); win_direct = false; return XFG_ERR_DEVICE; }
    // restore the promised csize for the size check of truncated chunks
    co = 0;
    for (size_t i = 0; i < n; i++) {
        xfg_record p, c;
        idx.get_records(cur_rec + (int64_t)i, &p, &c);
        const int64_t cs = c.comp_offset - p.comp_offset;
        if ((int64_t)win_chunks[i].comp_size != cs && win_res[i].status == XFG_ST_OK) {
            const int64_t want = cs + (c.type == XFG_REC_FOOTER ? 0 : 5);
            if (win_res[i].in_consumed != want || win_res[i].out_produced != (int64_t)win_chunks[i].raw_size)
                win_res[i].status = XFG_ST_SIZE_MISMATCH;
        }
        if (win_res[i].status == XFG_ST_OUT_FULL) win_res[i].status = XFG_ST_SIZE_MISMATCH;
        co += cs;
    }
    win_first = cur_rec;
    return XFG_ERR_NONE;
}

// flateReader.Read over the current chunk: data first, then the chunk's final outcome.
int64_t xfg_reader::zr_read(uint8_t *buf, int64_t n, int *e) {
    *e = XFG_ERR_NONE;
    if (cur_rec < 0) { *e = XFG_ERR_EOF; return 0; }           // past the last record: an empty stream
    int rc = ensure_decoded();
    if (rc != XFG_ERR_NONE) { *e = rc; return 0; }
    const size_t wi = (size_t)(cur_rec - win_first);
    const xfg_result &res = win_res[wi];
    const int64_t have = res.out_produced - cur_rd;
    if (have > 0) {
        const int64_t k = std::min(have, n);
        if (buf) memcpy(buf, win_raw.p + win_chunks[wi].raw_off + cur_rd, (size_t)k);
        cur_rd += k;
        return k;
    }
    cur_in_off = res.in_consumed;
    switch (res.status) {
    case XFG_ST_OK: case XFG_ST_NO_SYNC: case XFG_ST_SIZE_MISMATCH: *e = XFG_ERR_EOF; break;   // inflate itself ended cleanly
    case XFG_ST_CORRUPTED: *e = XFG_ERR_CORRUPTED; break;
    case XFG_ST_UNEXPECTED_EOF: *e = XFG_ERR_UNEXPECTED_EOF; break;
    default: *e = XFG_ERR_INTERNAL; break;
    }
    return 0;
}

// Read, xflate/reader.go:162-215
int64_t xfg_reader::read(uint8_t *buf, int64_t n, int *e) {
    *e = XFG_ERR_NONE;
    if (err != XFG_ERR_NONE) { *e = err; return 0; }
    if (discard > 0) {
        int64_t got = 0;
        while (got < discard) {
            int ze;
            const int64_t k = zr_read(nullptr, discard - got, &ze);
            got += k;
            if (ze == XFG_ERR_EOF) break;
            if (ze != XFG_ERR_NONE) { err = ze; *e = err; return 0; }
        }
        if (got != discard) { err = XFG_ERR_CORRUPTED; *e = err; return 0; }
        discard = 0;
    }
    if (n <= 0) return 0;
    // Many whole chunks in one call.  What one Read per chunk would do, done at once for a run of chunks of the decoded
    // window that ended cleanly with every post-chunk check passed (status OK): their bytes lie side by side in the
    // window and leave in one copy on the context's helper threads; the state then steps over them record by record
    // exactly as the loop below does at each chunk's io.EOF.  Anything else -- one chunk, an error, a short request --
    // takes the loop below.
    if (n >= (1 << 20) && cur_rec >= 0 && ctx) {
        // A request of 32 MiB and more that needs a new window anyway has it decoded straight into the caller's buffer
        // (io.Reader: "it may use all of p as scratch space during the call"): no window copy, no copy out.  Such a
        // window is gone when the call returns; what was not delivered from it is decoded again.
        const bool in_win = cur_rec >= win_first && cur_rec < win_first + (int64_t)win_res.size();
        if (!in_win && n >= (32 << 20) && cur_rd == 0 && chk_rsize <= n) { direct_dst = buf; direct_cap = n; }
        want_raw = n;
        const int rc = ensure_decoded();
        want_raw = 0;
        direct_dst = nullptr;
        if (rc == XFG_ERR_NONE) {
            const size_t wi = (size_t)(cur_rec - win_first);
            size_t wj = wi;
            int64_t bytes = 0;
            while (wj < win_res.size()) {
                const xfg_result &r = win_res[wj];
                if (r.status != XFG_ST_OK || r.out_produced != (int64_t)win_chunks[wj].raw_size) break;
                const int64_t have = r.out_produced - (wj == wi ? cur_rd : 0);
                if (have < 0 || bytes + have > n) break;
                bytes += have;
                wj++;
            }
            if (bytes > 0 && (win_direct || wj - wi >= 2)) {
                if (!win_direct) xfg_host_copy(ctx, buf, win_raw.p + win_chunks[wi].raw_off + cur_rd, (size_t)bytes);
                for (size_t k = wi; k < wj && err == XFG_ERR_NONE; k++) {
                    offset += win_res[k].out_produced - (k == wi ? cur_rd : 0);
                    cur_in_off = win_res[k].in_consumed;
                    int se;
                    seek(offset, 0, &se);
                    if (se != XFG_ERR_NONE) { err = se; break; }
                    if (chk_typ == XFG_REC_UNKNOWN) err = XFG_ERR_EOF;
                }
                if (win_direct) { win_res.clear(); win_direct = false; }
                return bytes;                                   // (an error, if any, surfaces on the next call)
            }
        }
        if (win_direct) { win_res.clear(); win_direct = false; }   // nothing clean at its head: the loop below decodes into the window
    }
    int64_t cnt = 0;
    while (cnt == 0 && err == XFG_ERR_NONE) {
        int ze;
        cnt = zr_read(buf, n, &ze);
        offset += cnt;
        if (ze == XFG_ERR_EOF) {
            // the checks after a chunk's io.EOF (reader.go:186-202)
            const size_t wi = (size_t)(cur_rec - win_first);
            const int st = cur_rec >= 0 ? win_res[wi].status : XFG_ST_OK;
            if (chk_typ == XFG_REC_DEFLATE && st == XFG_ST_NO_SYNC) { err = XFG_ERR_CORRUPTED; break; }
            if (st == XFG_ST_SIZE_MISMATCH) { err = XFG_ERR_CORRUPTED; break; }
            int se;
            seek(offset, 0, &se);
            if (se != XFG_ERR_NONE) { err = se; break; }
            if (chk_typ == XFG_REC_UNKNOWN) err = XFG_ERR_EOF;
        } else if (ze != XFG_ERR_NONE) {
            err = ze;
        }
    }
    if (cnt > 0) return cnt;                                    // data first; the error surfaces on the next call
    *e = err;
    return 0;
}

// Close, xflate/reader.go:282-291
int xfg_reader::close() {
    if (err == XFG_ERR_CLOSED) return XFG_ERR_NONE;
    if (err != XFG_ERR_NONE && err != XFG_ERR_EOF) return err;
    err = XFG_ERR_CLOSED;
    return XFG_ERR_NONE;
}

// ------------------------------------------------------------------ C ABI ---
extern "C" int xfg_reader_new(xfg_ctx *ctx, xfg_read_fn rd, xfg_seek_fn sk, void *user, xfg_reader **out) {
    if (!out) return XFG_ERR_INVALID;
    *out = nullptr;
    if (!ctx || !rd || !sk) return XFG_ERR_INVALID;
    xfg_reader *r = new (std::nothrow) xfg_reader();
    if (!r) return XFG_ERR_INTERNAL;
    r->ctx = ctx;
    r->max_window_raw = 256ll << 20;
    *out = r;
    return r->reset(rd, sk, user);
}
extern "C" int xfg_reader_new_mem(xfg_ctx *ctx, const uint8_t *stream, size_t len, xfg_reader **out) {
    if (!out) return XFG_ERR_INVALID;
    *out = nullptr;
    if (!ctx || (!stream && len) || len > (size_t)INT64_MAX) return XFG_ERR_INVALID;
    MemSource *m = new (std::nothrow) MemSource{stream, (int64_t)len, 0};
    if (!m) return XFG_ERR_INTERNAL;
    const int rc = xfg_reader_new(ctx, mem_read, mem_seek, m, out);
    if (*out) (*out)->owned_src = m; else delete m;
    return rc;
}
extern "C" int xfg_reader_reset_mem(xfg_reader *r, const uint8_t *stream, size_t len) {
    if (!r || (!stream && len) || len > (size_t)INT64_MAX) return XFG_ERR_INVALID;
    MemSource *m = (MemSource *)r->owned_src;
    if (!m) {
        m = new (std::nothrow) MemSource();
        if (!m) return XFG_ERR_INTERNAL;
        r->owned_src = m;
    }
    m->p = stream; m->len = (int64_t)len; m->pos = 0;
    return r->reset(mem_read, mem_seek, m);
}
extern "C" int xfg_reader_reset(xfg_reader *r, xfg_read_fn rd, xfg_seek_fn sk, void *user) {
    if (!r || !rd || !sk) return XFG_ERR_INVALID;
    return r->reset(rd, sk, user);
}
extern "C" int64_t xfg_reader_read(xfg_reader *r, uint8_t *buf, int64_t n, int *err) {
    int e = XFG_ERR_NONE;
    int64_t k = 0;
    if (!r || (!buf && n > 0)) e = XFG_ERR_INVALID; else k = r->read(buf, n, &e);
    if (err) *err = e;
    return k;
}
extern "C" int64_t xfg_reader_seek(xfg_reader *r, int64_t offset, int whence, int *err) {
    int e = XFG_ERR_NONE;
    int64_t p = 0;
    if (!r) e = XFG_ERR_INVALID; else p = r->seek(offset, whence, &e);
    if (err) *err = e;
    return p;
}
extern "C" int xfg_reader_close(xfg_reader *r) { return r ? r->close() : XFG_ERR_INVALID; }
extern "C" void xfg_reader_free(xfg_reader *r) {
    if (!r) return;
    void *src = r->owned_src;
    delete r;
    delete (MemSource *)src;
}
extern "C" int64_t xfg_reader_num_records(const xfg_reader *r) { return r ? (int64_t)r->idx.records.size() : 0; }
extern "C" int xfg_reader_get_records(const xfg_reader *r, xfg_record *out, int64_t cap) {
    if (!r || (!out && cap > 0)) return XFG_ERR_INVALID;
    const int64_t n = std::min<int64_t>(cap, (int64_t)r->idx.records.size());
    for (int64_t i = 0; i < n; i++) out[i] = r->idx.records[(size_t)i];
    return XFG_ERR_NONE;
}
extern "C" int64_t xfg_reader_bytes_read(const xfg_reader *r) { return r ? r->bytes_read : 0; }
extern "C" int xfg_reader_set_window(xfg_reader *r, int64_t max_raw_bytes) {
    if (!r || max_raw_bytes <= 0) return XFG_ERR_INVALID;
    r->max_window_raw = max_raw_bytes;
    return XFG_ERR_NONE;
}

// Index-only use: parse an in-memory stream and emit the device chunk table.
extern "C" int xfg_index_parse(const uint8_t *stream, size_t len, xfg_chunk *chunks, size_t cap, size_t *n_out) {
    if ((!stream && len) || !n_out) return XFG_ERR_INVALID;
    MemSource src = { stream, (int64_t)len, 0 };
    xfg_reader r;
    r.ctx = nullptr;
    r.max_window_raw = 256ll << 20;
    int e = r.reset(mem_read, mem_seek, &src);
    if (e != XFG_ERR_NONE) { *n_out = 0; return e; }
    const size_t n = r.idx.records.size();
    *n_out = n;
    if (!chunks) return XFG_ERR_NONE;
    int64_t pc = 0, pr = 0;
    // the device path's per-chunk limits apply to what this table is for: a stream with a larger chunk is refused here
    {
        int64_t qc = 0, qr = 0;
        for (size_t i = 0; i < n; i++) {
            const xfg_record &rec = r.idx.records[i];
            if ((uint64_t)(rec.comp_offset - qc) > XFG_MAX_CHUNK_COMP || (uint64_t)(rec.raw_offset - qr) > XFG_MAX_CHUNK_RAW) {
                *n_out = 0;
                return XFG_ERR_LIMIT;
            }
            qc = rec.comp_offset; qr = rec.raw_offset;
        }
    }
    for (size_t i = 0; i < n && i < cap; i++) {
        const xfg_record &rec = r.idx.records[i];
        memset(&chunks[i], 0, sizeof chunks[i]);
        chunks[i].comp_off = (uint64_t)pc;
        chunks[i].comp_size = (uint64_t)(rec.comp_offset - pc);
        chunks[i].raw_off = (uint64_t)pr;
        chunks[i].raw_size = (uint64_t)(rec.raw_offset - pr);
        chunks[i].flags = flags_for(rec.type);
        pc = rec.comp_offset; pr = rec.raw_offset;
    }
    if (n > cap) *n_out = n;
    return XFG_ERR_NONE;
}

// ------------------------------------------------------------ test hooks ---
// include/xflate_amd_testhooks.h: the reference's unit-test vectors for meta.Reader and index reach the
// C++ above through these, not through whole streams.  Not part of libxflate_amd.so: `make` builds this file a
// second time with -DXFG_TEST_HOOKS into libxflate_amd_testhooks.so (linked against the library), which only
// tests/test_host_cpp_vectors.py loads.
#ifdef XFG_TEST_HOOKS
#include "../../include/xflate_amd_testhooks.h"

extern "C" int xfg_test_meta_block(const uint8_t *in, size_t len, uint8_t *out, size_t *out_len, int *final, int64_t *in_off) {
    meta::Reader mr;
    mr.reset(in, len);
    const int e = mr.decode_block();
    if (out && out_len) { memcpy(out, mr.buf.data(), mr.buf.size()); *out_len = mr.buf.size(); }
    if (final) *final = mr.final;
    if (in_off) *in_off = mr.input_offset;
    return e;
}

extern "C" int xfg_test_meta_stream(const uint8_t *in, size_t len, uint8_t *out, size_t out_cap, size_t *out_len,
                                    int64_t *in_off, int64_t *out_off, int64_t *num_blocks, int *final_mode) {
    meta::Reader mr;
    mr.reset(in, len);
    Bytes bw;
    const int e = mr.read_all(bw);
    if (mr.err == XFG_ERR_NONE || mr.err == XFG_ERR_EOF) mr.final_mode = mr.final;   // Close, meta/reader.go:101-113
    if (out_len) *out_len = bw.size();
    if (out && bw.size() <= out_cap) memcpy(out, bw.data(), bw.size());
    if (in_off) *in_off = mr.input_offset;
    if (out_off) *out_off = mr.output_offset;
    if (num_blocks) *num_blocks = mr.num_blocks;
    if (final_mode) *final_mode = mr.final_mode;
    return e;
}

extern "C" int xfg_test_meta_reverse_search(const uint8_t *data, size_t len) { return meta::reverse_search(data, len); }

extern "C" void *xfg_test_index_new(void) { return new (std::nothrow) Index(); }
extern "C" void xfg_test_index_free(void *idx) { delete (Index *)idx; }
extern "C" int xfg_test_index_append_record(void *idx, int64_t comp_size, int64_t raw_size, int type) {
    return ((Index *)idx)->append_record(comp_size, raw_size, type) ? 1 : 0;
}
extern "C" int xfg_test_index_append_index(void *idx, const void *other) {
    return ((Index *)idx)->append_index(*(const Index *)other) ? 1 : 0;
}
extern "C" void xfg_test_index_push_raw(void *idx, int64_t comp_offset, int64_t raw_offset, int type) {
    xfg_record rec; memset(&rec, 0, sizeof rec);
    rec.comp_offset = comp_offset; rec.raw_offset = raw_offset; rec.type = type;
    ((Index *)idx)->records.push_back(rec);
}
extern "C" int64_t xfg_test_index_len(const void *idx) { return (int64_t)((const Index *)idx)->records.size(); }
extern "C" void xfg_test_index_record(const void *idx, int64_t i, xfg_record *out) { *out = ((const Index *)idx)->records[(size_t)i]; }
extern "C" void xfg_test_index_last_record(const void *idx, xfg_record *out) { *out = ((const Index *)idx)->last_record(); }
extern "C" int64_t xfg_test_index_search(const void *idx, int64_t offset) { return ((const Index *)idx)->search(offset); }
extern "C" void xfg_test_index_get_records(const void *idx, int64_t i, xfg_record *prev, xfg_record *curr) {
    ((const Index *)idx)->get_records(i, prev, curr);
}
extern "C" int xfg_test_decode_index(const uint8_t *enc, size_t len, void *idx, int64_t *back_size) {
    MemSource src = { enc, (int64_t)len, 0 };
    xfg_reader r;
    r.ctx = nullptr;
    r.rd_fn = mem_read; r.sk_fn = mem_seek; r.user = &src;
    r.bytes_read = 0;
    Index *ix = (Index *)idx;
    ix->index_size = (int64_t)len;
    const int e = r.decode_index(ix);
    if (back_size) *back_size = ix->back_size;
    return e;
}
#endif  // XFG_TEST_HOOKS

// Index-range sharding: cut the chunk table into n_shards contiguous
// record ranges of about equal work, work = comp_size + raw_size (the bytes a GPU reads and
// writes for the chunk).  Shard k owns records [first[k], first[k+1]).  Records are never
// split, every record belongs to exactly one shard, and a shard may be empty when there are
// fewer records than shards.  Host only; deterministic, so every rank computes the same plan.
extern "C" int xfg_shard_plan(const xfg_chunk *chunks, size_t n, int n_shards, size_t *first) {
    if ((!chunks && n) || !first || n_shards <= 0) return XFG_E_ARG;
    unsigned __int128 total = 0;
    for (size_t i = 0; i < n; i++) total += (unsigned __int128)chunks[i].comp_size + chunks[i].raw_size;
    first[0] = 0;
    size_t i = 0;
    unsigned __int128 done = 0;
    for (int k = 1; k < n_shards; k++) {
        // the boundary after shard k-1: the first record whose start reaches k/n_shards of the work
        const unsigned __int128 want = total * (unsigned)k / (unsigned)n_shards;
        while (i < n && done < want) {
            const unsigned __int128 w = (unsigned __int128)chunks[i].comp_size + chunks[i].raw_size;
            if (done + w / 2 > want) break;                 // nearer to the boundary without this record
            done += w;
            i++;
        }
        first[k] = i;
    }
    first[n_shards] = n;
    return XFG_OK;
}
