// inflate_kernel.hip -- DEFLATE inflate of independent XFLATE chunks on gfx950.
//
// One wavefront owns one chunk at a time.  A
// workgroup is one wave.  The grid is persistent (as many workgroups as fit the
// CUs' LDS); every wave draws its next chunk from a counter, expensive chunks first.
//
// Inside a Huffman block the 64 lanes work on 64 consecutive segments of the bit stream
// at once (decode_huffman_block), one "batch" per iteration; the segment size (80, 128, 160 or
// 240 bits) is chosen per batch from the expansion of the batch before it (seg_bits_for):
//   stage 1  a. speculative walk: lane i starts at segment boundary i, which is almost
//               never a code-word boundary, and marks every bit position where one of
//               its tokens starts (sync_segment<false>).  Prefix codes self-synchronise,
//               so the tail of its path is usually the true one.
//            b. re-synchronisation walks: a lane whose true start (its predecessor's
//               end) differs walks from there until it stands on a marked position --
//               from then on it would repeat its speculative path (sync_segment<true>)
//               -- until the chain of starts is stable.
//            c. count pass along the true path (bytes and matches per lane); wave
//               prefix sums give output and token offsets.  Only a chain that closes
//               (every lane starts where its predecessor's counted path ended) is kept.
//               The late walks of b. do this counting themselves (XFG_COUNT_FROM): a lane whose
//               start moved walks its whole path and counts, every other lane counts in the
//               first such round, and the separate pass runs for what is left.
//            d. emit pass: literals to their final positions in an LDS output window,
//               match tokens to a list in LDS.
//            Every walk is one table lookup per step (literal/length root, distance
//            root and second-level tables share the instruction stream: a small
//            per-lane state machine), from a 64-bit register bit buffer that is topped
//            up from the LDS copy of the compressed window.  The verified lanes are then
//            emitted, resolved and flushed one window-full at a time (rounds); the token list and
//            the output window share their LDS, and every round sets the border between them.  Stretches
//            where one lane's tokens fill the window are decoded by lane 0 alone (serial batches).
//   stage 2  match resolution (resolve_tokens): sources before the window come from
//            HBM/L2, four rounds of 64 tokens in flight; what is left is resolved 64
//            tokens per round, a source inside the window being ready once it lies
//            below the destination of the first unresolved token.  Long matches are
//            copied by the whole wave.  The window is then flushed to HBM with
//            coalesced 16-byte stores.
//
// What this restates, in the reference's terms (all under /root/reference):
//   chunkReader            xflate/reader.go:32-70      -> win_load(): chunk || 01 00 00 ff ff is a virtual stream
//   readBlockHeader        flate/reader.go:100-134     -> the block loop of inflate_chunk()
//   readRawData            flate/reader.go:137-159     -> stored_copy()
//   ReadPrefixCodes        flate/prefix.go:83-194      -> read_dynamic_header()
//   GeneratePrefixes       internal/prefix/prefix.go:326-384 -> build_table() steps 1-3 (histogram, Kraft, canonical rank)
//   Decoder.Init           internal/prefix/decoder.go:51-136 -> build_table() steps 4-5 (root table + sub-tables in LDS)
//   readBlock              flate/reader.go:162-250     -> decode_huffman_block(): sync_segment / decode_segment_fast / _tail
//   dictDecoder            flate/dict_decoder.go:86-132 -> resolve_tokens(): the output buffer is the window
//   Read's post-chunk checks xflate/reader.go:186-202  -> epilogue of inflate_chunk()
//
// The block-header state (bit buffer, positions, table entries) is wave-uniform and
// kept in SGPRs via readfirstlane; the 64 lanes cooperate on everything that
// has width: coalesced refill of the compressed window into LDS, table
// construction, stored-block and match copies.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "../../include/xflate_amd.h"
#include "xfg_internal.h"

namespace xfg {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

constexpr int WAVES_PER_BLOCK = 1;
constexpr int MAX_BLOCKS_PER_CU = 32;    // hardware limit of waves per CU; the real number comes from the occupancy query
#ifndef XFG_LIT_ROOT
#define XFG_LIT_ROOT 9
#endif
constexpr int LIT_ROOT = XFG_LIT_ROOT;  // first-level bits, literal/length table
constexpr int DIST_ROOT = 8;            // first-level bits, distance table
constexpr int CLEN_ROOT = 7;            // code-length code: max length 7, never needs sub-tables
static_assert(LIT_ROOT == 9 || LIT_ROOT == 10, "LIT_TAB below knows these two");
constexpr int LIT_TAB = LIT_ROOT == 10 ? 1344 : 864;   // >= enough(286, 10, 15) = 1334, enough(286, 9, 15) = 852 (zlib's enough.c)
constexpr int DIST_TAB = 404;           // >= enough(30, 8, 15)  = 402 entries
constexpr int CLEN_TAB = 128;
constexpr int WIN_BYTES = 1024;         // header reader's window: two 512-byte halves
constexpr int MAX_HIST = 32768;         // flate/common.go:16
// Bits of compressed stream per lane and batch: chosen per batch from what the last batch did
// (seg_bits_for): literal-heavy data wants long segments (a lane's speculative path meets the true
// one more often before its segment ends: fewer re-synchronisation walks), match-heavy data short
// ones (a lane's tokens produce many bytes: more lanes fit one window-full).  The LDS windows are
// sized for the longest; XFG_SEG_BITS != 0 pins one size (tuning runs).
#ifndef XFG_SEG_BITS
#define XFG_SEG_BITS 0
#endif
#ifndef XFG_SEG_LONG
#define XFG_SEG_LONG 240
#endif
#ifndef XFG_SEG_MID
#define XFG_SEG_MID 160
#endif
#ifndef XFG_SEG_SHORT
#define XFG_SEG_SHORT 80
#endif
#ifndef XFG_EXP_LONG_X4
#define XFG_EXP_LONG_X4 6             // expansion (x4) below which a batch is followed by long segments
#endif
#ifndef XFG_EXP_SHORT_X4
#define XFG_EXP_SHORT_X4 16           // expansion (x4) from which on it is followed by short ones
#endif
#ifndef XFG_SEG_RUN
#define XFG_SEG_RUN 128               // segments of batches whose output is nearly all long matches or runs
#endif
#ifndef XFG_EXP_RUN_X4
#define XFG_EXP_RUN_X4 256            // expansion (x4) from which on it is followed by those
#endif
#ifndef XFG_SPEC_BACK_MID
#define XFG_SPEC_BACK_MID 0
#endif
#ifndef XFG_SPEC_BACK_LONG
#define XFG_SPEC_BACK_LONG 0
#endif
#ifndef XFG_SPEC_BACK_SHORT
#define XFG_SPEC_BACK_SHORT 96         // run-in of the speculative walk before a short (XFG_SEG_SHORT) segment, dynamic blocks
#endif
#ifndef XFG_SPEC_WALK_ON_DYNAMIC
#define XFG_SPEC_WALK_ON_DYNAMIC 1    // the same for dynamic blocks
#endif
#ifndef XFG_SPEC_WALK_ON
#define XFG_SPEC_WALK_ON 1            // fixed blocks: a speculative walk goes on over a code that would end it
#endif
#ifndef XFG_RESOLVE_EXACT
#define XFG_RESOLVE_EXACT 1           // resolve: a token waits for the pending tokens that write its source bytes, not for all before it
#endif
#ifndef XFG_OUT_LDS
#define XFG_OUT_LDS 2304
#endif
#ifndef XFG_TOK_MAX
#define XFG_TOK_MAX 240               // match tokens per round at the middle segment size; the list takes what the window leaves
#endif
#ifndef XFG_SPEC_BACK
#define XFG_SPEC_BACK 0               // run-in of the speculative walk, bits before its segment, dynamic blocks
#endif
#ifndef XFG_SPEC_BACK_FIXED
#define XFG_SPEC_BACK_FIXED 480       // the same for fixed blocks, whose near-uniform code lengths take long to synchronise
#endif
#ifndef XFG_SUBBATCH
#define XFG_SUBBATCH 1                // 1: a batch's live lanes are written out in as many window-fulls as it takes
#endif
#ifndef XFG_ACCEPT_FD
#define XFG_ACCEPT_FD 64              // settle for a verified prefix of this many lanes (64 = insist on all)
#endif
constexpr int SEG_MAX = XFG_SEG_BITS ? XFG_SEG_BITS : XFG_SEG_LONG;   // the longest segment: sizes the LDS windows
constexpr int SEG_MIN = XFG_SEG_BITS ? XFG_SEG_BITS : XFG_SEG_SHORT;
constexpr int BATCH_BYTES = 64 * SEG_MAX / 8;
constexpr int BWIN_BYTES = (BATCH_BYTES + 64) > 1024 ? (BATCH_BYTES + 64) : 1024;   // batch window: 16-byte alignment slack + over-read; >= the header ring
constexpr int NPF = (BWIN_BYTES + 1023) / 1024;     // 16-byte units per lane that cover the batch window
constexpr int OUT_LDS = XFG_OUT_LDS;    // bytes of batch output staged in LDS
constexpr int OUT_CAP = OUT_LDS - 258;  // a lane stops at the token that takes it to OUT_CAP: one lane always fits
// The compressed window and the match-token list share one LDS region: the window of a batch of
// seg_bits-bit segments is 8 * seg_bits + 64 bytes, the list (6 bytes per token) takes the rest.
constexpr int WT_BYTES = (8 * (XFG_SEG_BITS ? XFG_SEG_BITS : XFG_SEG_MID) + 64 + 6 * XFG_TOK_MAX + 7) / 8 * 8;
constexpr int TOK_MIN = (WT_BYTES - BWIN_BYTES) / 6;            // tokens per round at the longest segment size
constexpr int MAX_PASSES = 6;           // speculative passes before settling for the stable prefix (careful path)
#ifndef XFG_MAX_FIX
#define XFG_MAX_FIX 7
#endif
#ifndef XFG_STORED_DEEP
#define XFG_STORED_DEEP 1
#endif
#ifndef XFG_STORED_SYNC
#define XFG_STORED_SYNC 0
#endif
#ifndef XFG_COUNT_FROM
#define XFG_COUNT_FROM 3                // from this walk on (1 = the first walk from corrected starts) a lane that walks counts: dynamic blocks
#endif
#ifndef XFG_COUNT_FROM_LONG
#define XFG_COUNT_FROM_LONG 1           // the same for batches of long segments (literal-heavy data)
#endif
#ifndef XFG_COUNT_FROM_RUN
#define XFG_COUNT_FROM_RUN 99           // the same for batches of runs and long matches (their long codes with many extra bits synchronise badly: many lanes walk in every round, and merge walks end early)
#endif
#ifndef XFG_COUNT_FROM_FIXED
#define XFG_COUNT_FROM_FIXED 99         // the same for fixed blocks (99: never)
#endif
#ifndef XFG_MAX_FIX_FIXED
#define XFG_MAX_FIX_FIXED 12          // fixed blocks: every walk adds about two lanes to the chain, and costs less than a new batch
#endif
#ifndef XFG_LONG_MATCH
#define XFG_LONG_MATCH 32
#endif
constexpr int LONG_MATCH = XFG_LONG_MATCH;   // matches this long are copied by the whole wave, one at a time
constexpr int SERIAL_ENTER = 6;         // a parallel batch that keeps this few lanes switches the block to serial batches
#ifndef XFG_ACCEPT_PASS
#define XFG_ACCEPT_PASS 2
#endif
constexpr int ACCEPT_PASS = XFG_ACCEPT_PASS;   // after this many walks from corrected starts, a chain of XFG_ACCEPT_FD lanes is good enough
constexpr int MAX_FIX = XFG_MAX_FIX;    // re-synchronisation walks per batch before settling for the stable prefix
constexpr int MASK_WORDS = (SEG_MAX + 48 + 31) / 32;    // a path ends less than 48 bits after its segment
static_assert(MASK_WORDS * 64 * 4 <= OUT_LDS, "the token-start masks share the output window's LDS");
static_assert(OUT_CAP >= 1024, "output window too small");
static_assert(WT_BYTES > BWIN_BYTES && TOK_MIN >= SEG_MAX / 2, "one lane's tokens (>= 2 bits each) must fit the token list");
static_assert(XFG_SEG_RUN >= SEG_MIN && XFG_SEG_RUN <= SEG_MAX && SEG_MIN >= 64 && (XFG_SEG_BITS || (SEG_MIN <= XFG_SEG_MID && XFG_SEG_MID <= SEG_MAX)), "segment sizes");
static_assert(OUT_LDS <= 65536, "token destinations are 16-bit");

// Table entry (u32): what one decode step does with the bits at the head of the buffer.
//   [4:0]   bits to consume in this step: code word + extra bits (length / distance), the code
//           word (literal / end of block / invalid), or ROOT for a link to a second-level table;
//           second-level entries hold what is left of the code word after ROOT bits
//   [9:5]   number of extra bits (length / distance), or index bits of the second-level table (link)
//   [15:10] what the step is, one bit each: E_LIT E_LEN E_DIST E_SUB E_STOP (end of block or
//           invalid code: the lane is finished) E_TOK (a token is complete: with E_LIT and E_DIST)
//   [31:16] literal byte, length base, distance base, code-length symbol, or the second-level
//           table's LDS byte address; on E_STOP entries bit 16 tells the end-of-block symbol from
//           an invalid code, and bit 31 is set (on no other entry)
enum : u32 { K_LIT = 0, K_LEN = 1, K_DIST = 2, K_SUB = 3, K_EOB = 4, K_INVALID = 5 };
enum : u32 { E_LIT = 1u << 10, E_LEN = 1u << 11, E_DIST = 1u << 12, E_SUB = 1u << 13, E_STOP = 1u << 14, E_TOK = 1u << 15,
             E_EOB = 1u << 16,         // in the value field of an E_STOP entry: it is the end-of-block symbol
             E_SIGN = 1u << 31 };      // set with E_STOP (no other entry's value reaches bit 15): the exit test is a sign test
enum : int { T_LIT = 0, T_DIST = 1, T_CLEN = 2 };

__device__ __forceinline__ u32 mk_entry(u32 code_bits, u32 kind, u32 extra, u32 val) {
    const u32 take = (kind == K_LEN || kind == K_DIST) ? code_bits + extra : code_bits;
    const u32 flags = kind == K_LIT ? (E_LIT | E_TOK) : kind == K_LEN ? E_LEN : kind == K_DIST ? (E_DIST | E_TOK) :
                      kind == K_SUB ? E_SUB : kind == K_EOB ? (E_STOP | E_EOB | E_SIGN) : (E_STOP | E_SIGN);
    return take | (extra << 5) | flags | (val << 16);
}
__device__ __forceinline__ u32 e_take(u32 e) { return e & 31; }
__device__ __forceinline__ u32 e_extra(u32 e) { return (e >> 5) & 31; }
__device__ __forceinline__ u32 e_val(u32 e) { return e >> 16; }

// Symbol -> entry.  lenRanges / distRanges of flate/prefix.go:23-32 in closed form.
__device__ __forceinline__ u32 sym_entry(int type, u32 sym, u32 nbits) {
    if (type == T_LIT) {
        if (sym < 256) return mk_entry(nbits, K_LIT, 0, sym);
        if (sym == 256) return mk_entry(nbits, K_EOB, 0, 0);
        if (sym >= 286) return mk_entry(nbits, K_INVALID, 0, 0);      // flate/reader.go:228-229
        u32 s = sym - 257;
        if (s < 8) return mk_entry(nbits, K_LEN, 0, 3 + s);
        if (s == 28) return mk_entry(nbits, K_LEN, 0, 258);
        u32 e = (s - 4) >> 2;
        return mk_entry(nbits, K_LEN, e, 3 + ((4 + (s & 3)) << e));
    } else if (type == T_DIST) {
        if (sym >= 30) return mk_entry(nbits, K_INVALID, 0, 0);       // flate/reader.go:212-214
        if (sym < 4) return mk_entry(nbits, K_DIST, 0, 1 + sym);
        u32 e = (sym - 2) >> 1;
        return mk_entry(nbits, K_DIST, e, 1 + ((2 + (sym & 1)) << e));
    } else {
        if (sym >= 19) return mk_entry(nbits, K_INVALID, 0, 0);       // flate/prefix.go:150-152
        return mk_entry(nbits, K_LIT, 0, sym);
    }
}

// ---- optional in-kernel profile (diagnostic build only: make prof) -------------
// Cycle and event counters per phase, summed over all waves into a small global array
// that no other code reads.  The production build compiles all of this away.
enum { PF_HEADER = 0, PF_WINDOW, PF_COUNT, PF_EMIT, PF_RESOLVE, PF_STORED, PF_TOTAL,
       PF_N_BATCH, PF_N_PASS, PF_N_LIVE, PF_N_ROUNDS, PF_N_BLOCKS, PF_FLUSH, PF_N_TOKENS,
       PF_H_CLEN, PF_H_LENS, PF_H_LIT, PF_H_DIST, PF_R_SYNC, PF_R_SWEEP, PF_P_SPEC, PF_P_FIX, PF_P_COUNT, PF_N_FIX,
       PF_C_FIX_LOST, PF_C_FLAG_CUTS, PF_C_FLAG_LOST, PF_C_CAP_CUTS, PF_C_CAP_LOST, PF_T_SORT, PF_T_ROOT, PF_T_SUB,
       PF_N_WFULL, PF_N_WF_TOK, PF_N_WF_OUT, PF_N_LONG, PF_N_LONG_G, PF_T_LONG, PF_N_INDEP, PF_N_LONG_RDY,
       // the workgroup kernel: cycles spent waiting for the wave before (chain, count), the group's outcome, the water mark; events
       PF_W_CHAIN, PF_W_CNT, PF_W_OUT, PF_W_WATER, PF_W_GROUPS, PF_W_ABORT, PF_W_ALONE, PF_W_REFIX,
       // lanes that walk in the first, second, third and later re-synchronisation walks of a batch (the one-wave kernel)
       PF_N_DIRTY1, PF_N_DIRTY2, PF_N_DIRTY3, PF_N_DIRTY4, PF_COUNTERS = 56 };   // lanes a batch did not keep, by cause
// Per-wave LDS.  The header scratch is live only while a block header is parsed and its
// tables are built; the output window is live only while a block's tokens are decoded.
struct HeaderScratch {
    u32 clen[CLEN_TAB];     // decode table of the code-length code
    u16 sorted[320];        // symbols in canonical (length, symbol) order
    u8  lens[320];          // code lengths: [0,hlit) literal/length, [hlit, hlit+hdist) distance
    u8  clens[32];          // code lengths of the code-length code
    u32 cnt[16];            // histogram of code lengths
    u32 first[16];          // first canonical code of each length
    u32 offs[16];           // index in sorted[] of the first symbol of each length
    u32 endr[16];           // end of the length-l codes in ROOT-bit code space (l <= ROOT)
};
struct alignas(16) WaveLds {
    u32 tab[LIT_TAB + DIST_TAB];    // literal/length table at 0, distance table at LIT_TAB
    u64 win[WT_BYTES / 8];          // batch window, then the batch's match tokens (token_list()); the first WIN_BYTES
                                    // double as the header reader's ring
    union {
        u64 out[OUT_LDS / 8 + 10];  // output of the current batch at [woff, woff + n), woff = its HBM address mod 16, so that
                                    // the flush is 16-byte aligned on both sides (+ 64 bytes that nobody reads)
        HeaderScratch h;
        u32 mask[MASK_WORDS * 64];  // synchronisation passes only (before anything is emitted): token-start bits of
                                    // every lane's speculative path, word w of lane l at [w * 64 + l]
    };
#ifdef XFG_PROFILE
    unsigned long long prof[PF_COUNTERS];
#endif
    // bytes from the start of win that a round's token list and output window share, and where the bytes go that
    // nobody reads (DUMP_POS below); the workgroup kernel's waves may have more room (inflate_wg_kernel.inc)
    static constexpr bool pointers_known = true;               // a __shared__ object of the kernel itself: the compiler sees what every access is
    static constexpr u32 room_total = WT_BYTES + OUT_LDS;
    static constexpr u32 dump = WT_BYTES + OUT_LDS + 16;
};
static_assert(sizeof(HeaderScratch) <= OUT_LDS, "header scratch must fit the output window");

// The match tokens of one round, behind the compressed window of a batch of seg_bits-bit segments.
struct TokList {
    u32 *tok;           // stream order: (len-3) | (dist-1) << 8
    u16 *tdst;          // their destinations in the output window
    u32 max;            // tokens per round
};
// Positions in the output window (token destinations, orel, woff) count bytes from the start of L.win, which the
// union follows without a gap: the window is at OUT_POS unless a round moves the border between its token list
// and its output (decode_huffman_block).  The 64 bytes from DUMP_POS on are written and never read.
constexpr u32 OUT_POS = WT_BYTES;
constexpr u32 DUMP_POS = WT_BYTES + OUT_LDS + 16;
static_assert(offsetof(WaveLds, out) == offsetof(WaveLds, win) + WT_BYTES && WT_BYTES % 16 == 0 && offsetof(WaveLds, win) % 16 == 0,
              "the output window follows the token list, 16-byte aligned");
static_assert(DUMP_POS + 64 <= WT_BYTES + sizeof(((WaveLds *)nullptr)->out) && DUMP_POS == WaveLds::dump, "dump bytes lie inside the window's slack");
template <class LDS>
__device__ __forceinline__ TokList token_list_n(LDS &L, u32 seg_bits, u32 max) {
    const u32 bwin = 8 * seg_bits + 64;
    TokList t;
    t.max = max;
    t.tok = reinterpret_cast<u32 *>(reinterpret_cast<u8 *>(L.win) + bwin);
    t.tdst = reinterpret_cast<u16 *>(t.tok + t.max);
    return t;
}
template <class LDS>
__device__ __forceinline__ TokList token_list(LDS &L, u32 seg_bits) {
    const u32 bwin = 8 * seg_bits + 64;
    TokList t;
    t.max = ((u32)WT_BYTES - bwin) / 6;
    t.tok = reinterpret_cast<u32 *>(reinterpret_cast<u8 *>(L.win) + bwin);
    t.tdst = reinterpret_cast<u16 *>(t.tok + t.max);
    return t;
}

#ifndef XFG_PROF_TABLE
#define XFG_PROF_TABLE T_LIT            // which table's build the t_sort / t_root / t_sub timers follow (diagnostic build)
#endif
#ifdef XFG_PROFILE
// Accumulated per wave in LDS (WaveLds::prof) and added to the global array once per chunk,
// so that the instrumentation puts no memory traffic into the phases it times.
__device__ unsigned long long g_prof[PF_COUNTERS];
#define PROF_T0() const unsigned long long prof_t0_ = __builtin_amdgcn_s_memtime()
#define PROF_ADD(slot, lane_) do { if ((lane_) == 0) L.prof[slot] += __builtin_amdgcn_s_memtime() - prof_t0_; } while (0)
#define PROF_CNT(slot, v, lane_) do { if ((lane_) == 0) L.prof[slot] += (unsigned long long)(v); } while (0)
#else
#define PROF_T0() do {} while (0)
#define PROF_ADD(slot, lane_) do {} while (0)
#define PROF_CNT(slot, v, lane_) do {} while (0)
#endif

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler
    // from moving LDS accesses across the point where lanes exchange data.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ u32 uni(u32 v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64 uni64(u64 v) {
    u32 lo = __builtin_amdgcn_readfirstlane((u32)v);
    u32 hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// A pointer that is known to point into LDS, said so that the compiler can tell: through the address space and back.
// Where a function is not inlined (or its LDS comes through references, as in the workgroup kernel's WgLds) the
// compiler otherwise falls back to flat instructions for every access.
template <class T>
__device__ __forceinline__ T *in_lds(T *p) {
    typedef __attribute__((address_space(3))) T *lds_p;
    u32 a = (u32)(uintptr_t)(lds_p)p;
    asm("" : "+v"(a));                                          // (opaque, so that the round trip is not folded away)
    return (T *)(lds_p)(uintptr_t)a;
}

// The same for a pointer into global memory.
template <class T>
__device__ __forceinline__ T *in_global(T *p) {
    typedef __attribute__((address_space(1))) T *glb_p;
    u64 a = (u64)(uintptr_t)(glb_p)p;
    asm("" : "+v"(a));
    return (T *)(glb_p)(uintptr_t)a;
}

// ---------------------------------------------------------------------------
// Bit reader over the virtual stream chunk || endBlock, wave-uniform.
// ---------------------------------------------------------------------------
struct Reader {
    const u8 *comp;     // whole compressed buffer
    u64 base_al;        // comp_off rounded down to 512
    u32 lead;           // comp_off - base_al
    u32 csize;          // chunk bytes
    u32 total;          // csize + 5 if the endBlock is appended
    u32 bytepos;        // next stream byte to feed into bitbuf
    i32 blk_hi;         // highest 512-byte window block resident (blk_hi-1 and blk_hi are in LDS)
    u64 bitbuf;
    u32 bitcnt;
    u64 *win;
    u32 lane;
};

// Eight bytes of the virtual stream  chunk || 01 00 00 ff ff (xflate/common.go:57) || zeros  at stream
// index ci, where p, the address that index has, is 8-byte aligned.  An aligned word lies in one page:
// loading a word that holds at least one byte of the chunk cannot fault, whatever follows the chunk in
// memory, and what is not the chunk's is masked out.  Bytes before the chunk's start (ci < 0) come out as
// whatever memory holds (the previous chunk); nobody consumes them.
__device__ __forceinline__ u64 stream_word(const Reader &r, const u8 *p, i64 ci) {
    const i64 k = (i64)r.csize - ci;                            // bytes of the chunk from ci on: >= 8 all of the word, <= 0 none
    u64 w = 0;
    if (k > 0 && ci + 8 > 0) w = *reinterpret_cast<const u64 *>(p);
    if (k < 8) {
        const u64 eb = r.total > r.csize ? 0xffff000001ull : 0ull;
        if (k > 0) w = (w & ((1ull << (8 * k)) - 1)) | (eb << (8 * k));
        else w = k > -5 ? eb >> (8 * (-k)) : 0ull;
    }
    return w;
}

__device__ __forceinline__ void win_load(Reader &r, i32 k) {
    // Block k covers window positions [512k, 512k+512); window position q = lead + stream byte index.
    const u32 q0 = (u32)k * 512u + r.lane * 8u;
    const i64 ci = (i64)q0 - (i64)r.lead;                 // stream byte index of this lane's first byte
    u64 w;
    if (ci + 8 <= (i64)r.csize) {
        // Entirely at or before the chunk's last byte: one aligned 8-byte load.  Bytes
        // before the chunk start (ci < 0) belong to the previous chunk and are never consumed.
        w = *reinterpret_cast<const u64 *>(r.comp + r.base_al + q0);
    } else {
        w = stream_word(r, r.comp + r.base_al + q0, ci);
    }
    r.win[(k & 1) * 64 + r.lane] = w;
}

// Make sure the 8 bytes at stream position bytepos are in the window.
__device__ __forceinline__ void win_ensure(Reader &r) {
    const u32 q = r.bytepos + r.lead;
    const i32 need = (i32)((q + 7) >> 9);
    if (need > r.blk_hi) {
        if (need > r.blk_hi + 1) r.blk_hi = (i32)(q >> 9) - 1;     // jumped (after a stored block)
        while (r.blk_hi < need) {
            r.blk_hi++;
            win_load(r, r.blk_hi);
        }
        wave_sync();
    }
}

// Make sure the n (<= 512) bytes at stream position bytepos are in the window.
__device__ __forceinline__ void win_ensure_at(Reader &r, u32 bytepos, u32 n) {
    const u32 q = bytepos + r.lead;
    const i32 need = (i32)((q + n - 1) >> 9);
    if (need > r.blk_hi || (i32)(q >> 9) < r.blk_hi - 1) {
        if (need > r.blk_hi + 1 || (i32)(q >> 9) < r.blk_hi - 1) r.blk_hi = (i32)(q >> 9) - 1;
        while (r.blk_hi < need) {
            r.blk_hi++;
            win_load(r, r.blk_hi);
        }
        wave_sync();
    }
}

// Top up the bit buffer to >= 56 bits.  Bytes past the end of the stream read as zero;
// avail() tells how many of the buffered bits are real.
__device__ __forceinline__ void refill(Reader &r) {
    win_ensure(r);
    const u32 q = (r.bytepos + r.lead) & (WIN_BYTES - 1);
    const u32 i0 = q >> 3;
    const u64 lo = r.win[i0];
    const u64 hi = r.win[(i0 + 1) & (WIN_BYTES / 8 - 1)];
    const u32 sh = (q & 7) * 8;
    u64 v = lo >> sh;
    if (sh) v |= hi << (64 - sh);
    v = uni64(v);
    r.bitbuf |= v << r.bitcnt;
    r.bytepos += (63 - r.bitcnt) >> 3;
    r.bitcnt |= 56;
}
__device__ __forceinline__ i32 avail(const Reader &r) {
    return ((i32)r.total - (i32)r.bytepos) * 8 + (i32)r.bitcnt;
}
__device__ __forceinline__ void consume(Reader &r, u32 n) {
    r.bitbuf >>= n;
    r.bitcnt -= n;
}
__device__ __forceinline__ u32 peek(const Reader &r, u32 n) {
    return (u32)r.bitbuf & ((1u << n) - 1);
}
// InputOffset = ceil(consumed bits / 8)  (internal/prefix/reader.go:226-245)
__device__ __forceinline__ i64 in_offset(const Reader &r) {
    const i64 consumed = (i64)r.bytepos * 8 - (i64)r.bitcnt;
    return (consumed + 7) >> 3;
}

// Reposition the header reader at an absolute bit position of the stream.
__device__ __forceinline__ void reader_seek(Reader &r, u32 bitpos) {
    r.bytepos = bitpos >> 3;
    r.bitbuf = 0;
    r.bitcnt = 0;
    r.blk_hi = -4;                                              // nothing resident
    refill(r);
    consume(r, bitpos & 7);
}
__device__ __forceinline__ u32 reader_bitpos(const Reader &r) { return r.bytepos * 8 - r.bitcnt; }

// ---------------------------------------------------------------------------
// Canonical prefix-code table construction, wave-cooperative, directly in LDS.
// Returns 0 = ok, 1 = empty code (decode with it is an error), 2 = invalid
// (over/under-subscribed: GeneratePrefixes errors, prefix.go:355-369).
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 wave_incl_scan(u32 v);

template <int TYPE, int ROOT, int TAB_OFF, class LDS>
__device__ __attribute__((noinline)) int build_table(LDS &L_, u32 *tab_, const u8 *lens_, u32 n, u32 lane) {
    struct View { HeaderScratch &h; u32 *tab;
#ifdef XFG_PROFILE
                  unsigned long long *prof;
#endif
    };
    View L{*in_lds(&L_.h), in_lds(&L_.tab[0])
#ifdef XFG_PROFILE
           , in_lds(&L_.prof[0])
#endif
    };
    u32 *tab = in_lds(tab_);
    const u8 *lens = in_lds(lens_);
#ifdef XFG_PROFILE
    const unsigned long long bt0_ = __builtin_amdgcn_s_memtime();
#endif
    // 1. How many codes of each length, and every symbol's rank among the symbols of its length, in one sweep: a lane
    //    holds symbols lane, 64 + lane, ...; length q (1..15) is a 10-bit field ((q - 1) % 3) of packed word (q - 1) / 3
    //    (there are <= 320 symbols), a symbol is a one in its length's field, and an inclusive prefix sum over the
    //    wave (DPP, no LDS), carried from one 64-symbol piece to the next, counts the symbols of that length up to and
    //    including this one.  After the last piece the carry is the histogram.
    constexpr int NK = TYPE == T_LIT ? 5 : 1;                   // 64-symbol pieces: n <= 320 literal/length, <= 32 distance, 19 code-length symbols
    constexpr int QMAX = TYPE == T_CLEN ? 7 : 15;               // the code-length code's lengths are 3-bit values
    constexpr int NW = (QMAX - 1) / 3 + 1;
    u32 lk[NK], rk[NK];
    u32 carry[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) carry[w] = 0;
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const u32 i = (u32)k * 64u + lane;
        const u32 l = i < n ? (u32)lens[i] : 0u;
        lk[k] = l;
        const u32 t = l ? l - 1 : 0u;                           // (l = 0: no field)
        const u32 wsel = (t * 11u) >> 5;                        // t / 3 for t < 15
        const u32 sh = 10u * (t - 3u * wsel);
        const u32 f = l ? 1u << sh : 0u;
        u32 own = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const u32 incl = wave_incl_scan(wsel == (u32)w ? f : 0u) + carry[w];
            carry[w] = (u32)__builtin_amdgcn_readlane((int)incl, 63);
            own = wsel == (u32)w ? incl : own;
        }
        rk[k] = ((own >> sh) & 1023u) - 1u;
    }

    // 2. Kraft check and canonical first codes (uniform; every lane computes the same)
    u32 cnt[16];
    u32 nz = 0, maxlen = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) {
        cnt[l] = l <= QMAX ? (carry[(l - 1) / 3] >> (10 * ((l - 1) % 3))) & 1023u : 0u;
        nz += cnt[l];
        if (cnt[l]) maxlen = l;
    }
    if (nz == 0) return 1;
    bool degenerate = false;
    if (nz == 1) {
        // handleDegenerateCodes (flate/prefix.go:203-208): a lone 1-bit code gets a
        // sibling whose symbol is out of range; any other lone code is incomplete.
        if (cnt[1] != 1) return 2;
        degenerate = true;
        cnt[1] = 2;
    }
    {
        i32 left = 1;
#pragma unroll
        for (int l = 1; l <= 15; l++) {
            left = left * 2 - (i32)cnt[l];
            if (left < 0) return 2;
        }
        if (left != 0) return 2;
    }
    // first[l]: first canonical code of length l; endr[l]: end of the length-l codes in ROOT-bit code space
    // (l <= ROOT).  All wave-uniform (scalar registers); lane l puts first[l] into LDS for the lanes to look up by
    // a length of their own.
    u32 first[16], endr[ROOT + 1];
    {
        u32 code = 0;
        u32 v_first = 0;
#pragma unroll
        for (int l = 1; l <= 15; l++) {
            code <<= 1;
            first[l] = code;
            if (l <= ROOT) endr[l] = (code + cnt[l]) << (ROOT - l);
            v_first = lane == (u32)l ? code : v_first;
            code += cnt[l];
        }
        if (lane < 16) L.h.first[lane] = v_first;
    }
    u32 end_root = 0;                                           // ROOT-bit prefixes at or beyond it belong to longer codes
#pragma unroll
    for (int l = 1; l <= ROOT; l++)
        if ((u32)l == (maxlen < (u32)ROOT ? maxlen : (u32)ROOT)) end_root = endr[l];
#ifdef XFG_PROFILE
    const unsigned long long bt1_ = __builtin_amdgcn_s_memtime();
    if (TYPE == XFG_PROF_TABLE && lane == 0) L.prof[PF_T_SORT] += bt1_ - bt0_;
#endif

    // 3. links: one second-level table per ROOT-bit prefix u >= end_root, sized by the longest code under that
    //    prefix (as zlib's inftrees does); the root entry of the prefix says where it is and how many bits index it.
    const u32 tab_lds = (u32)(uintptr_t)(const __attribute__((address_space(3))) u32 *)L.tab;
    if (TYPE != T_CLEN && maxlen > (u32)ROOT) {
        u32 next = 1u << ROOT;                                  // running sub-table offset (uniform)
        for (u32 ub = end_root; ub < (1u << ROOT); ub += 64) {
            const u32 u = ub + lane;
            const bool act = u < (1u << ROOT);
            u32 sb = 0;
#pragma unroll
            for (int l = ROOT + 1; l <= 15; l++) {              // the longest length with a code under this prefix
                const u32 lo = u << (l - ROOT), hi = (u + 1) << (l - ROOT);
                if (cnt[l] && lo < first[l] + cnt[l] && hi > first[l]) sb = (u32)(l - ROOT);
            }
            if (!act) sb = 0;
            const u32 size = act ? (1u << sb) : 0;
            const u32 incl = wave_incl_scan(size);
            const u32 my_off = next + incl - size;
            next += (u32)__builtin_amdgcn_readlane((int)incl, 63);
            if (act) tab[__brev(u) >> (32 - ROOT)] = mk_entry(ROOT, K_SUB, sb, tab_lds + 4 * (my_off + TAB_OFF));   // LDS address of the sub-table
        }
    }
    wave_sync();
#ifdef XFG_PROFILE
    const unsigned long long bt2_ = __builtin_amdgcn_s_memtime();
    if (TYPE == XFG_PROF_TABLE && lane == 0) L.prof[PF_T_ROOT] += bt2_ - bt1_;
#endif

    // 4. every symbol writes its own entries.  Its code word is first[len] + its rank among the symbols of its length
    //    (the canonical order); the tables are indexed LSB-first.  A code of l <= ROOT bits owns the 2^(ROOT - l) root
    //    entries whose low l bits are the reversed code; a longer one the entries of its prefix's second-level table
    //    whose low l - ROOT bits are the rest of the code, reversed.
    u32 fl[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) fl[k] = L.h.first[lk[k]];
    //    Short codes have many entries (2^(ROOT - l)): those are written by the whole wave, one code after the other.
    constexpr u32 OWN_MIN = ROOT >= 4 ? ROOT - 3 : 1;           // codes of at least this many bits (<= 8 entries) are written by their own lane
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const u32 l = lk[k];
        const u32 code = fl[k] + rk[k];
        const u32 sym = (u32)k * 64u + lane;
        const bool in_root = l <= (u32)ROOT;
        const u32 rest = in_root ? l : l - ROOT;                // bits of the code in the table its entry goes to
        const u32 e = sym_entry(TYPE, sym, rest);
        const u32 i0 = l ? __brev(in_root ? code : code & ((1u << rest) - 1)) >> (32 - rest) : 0u;
        if (l > (u32)ROOT) {
            const u32 link = tab[__brev(code >> rest) >> (32 - ROOT)];
            const u32 sb = e_extra(link);
            u32 *sub = tab + ((e_val(link) - tab_lds) >> 2) - TAB_OFF;
            for (u32 j = 0; j < (1u << sb); j += 1u << rest) sub[i0 + j] = e;
        } else if (l >= OWN_MIN) {
#pragma unroll
            for (u32 j = 0; j < 8; j++)
                if (j < (1u << (ROOT - l))) tab[i0 + (j << l)] = e;
        }
        for (u64 m = __ballot(l != 0 && l < OWN_MIN); m; m &= m - 1) {
            const int src = __ffsll((long long)m) - 1;
            const u32 e_s = (u32)__builtin_amdgcn_readlane((int)e, src);
            const u32 i_s = (u32)__builtin_amdgcn_readlane((int)i0, src);
            const u32 l_s = (u32)__builtin_amdgcn_readlane((int)l, src);
            for (u32 j = lane; j < (1u << (ROOT - l_s)); j += 64) tab[i_s + (j << l_s)] = e_s;
        }
    }
    if (degenerate) {
        // the lone code's out-of-range sibling: code word 1
        const u32 e = sym_entry(TYPE, 0xffffu, 1);
        for (u32 j = lane; j < (1u << (ROOT - 1)); j += 64) tab[1u + 2u * j] = e;
    }
    wave_sync();
#ifdef XFG_PROFILE
    if (TYPE == XFG_PROF_TABLE && lane == 0) L.prof[PF_T_SUB] += __builtin_amdgcn_s_memtime() - bt2_;
#endif
    return 0;
}

// Lookup at the head of the bit buffer in the code-length code's table (codes <= 7 bits: one level).
__device__ __forceinline__ u32 lookup_clen(const u32 *tab, u64 bitbuf) {
    return uni(tab[(u32)bitbuf & ((1u << CLEN_ROOT) - 1)]);
}

struct Out {
    u8 *raw;            // this chunk's output
    u32 cap;            // raw_size
    u32 pos;            // bytes produced
};

// readRawData (flate/reader.go:137-159): n bytes from stream position sp to the output,
// 16 bytes per lane and step with the stores aligned (the loads may straddle).  lane counts over
// the NL lanes that copy together: one wave's 64, or all the waves of a workgroup (inflate_wg_kernel.inc).
template <u32 NLC = 64>
__device__ __forceinline__ void stored_copy(const Reader &r, Out &o, u32 sp, u32 n, u32 lane, u32 nl_rt = 64) {
    const u32 NL = NLC ? NLC : nl_rt;                           // compile-time for the one-wave kernel
    const u8 *src = r.comp + r.base_al + r.lead + sp;
    u8 *dst = o.raw + o.pos;
    u32 fast = 0;                                               // bytes inside the chunk proper
    if (sp < r.csize) fast = n < r.csize - sp ? n : r.csize - sp;
    u32 head = (u32)((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > fast) head = fast;
    if (lane < head) dst[lane] = src[lane];
    const u32 body = (fast - head) >> 4;
    u32 i = lane;
#if XFG_STORED_DEEP
    for (; i + 7 * NL < body; i += 8 * NL) {                    // eight loads in flight per lane
        uint4 v0, v1, v2, v3, v4, v5, v6, v7;
        __builtin_memcpy(&v0, src + head + (size_t)(i + 0 * NL) * 16, 16);
        __builtin_memcpy(&v1, src + head + (size_t)(i + 1 * NL) * 16, 16);
        __builtin_memcpy(&v2, src + head + (size_t)(i + 2 * NL) * 16, 16);
        __builtin_memcpy(&v3, src + head + (size_t)(i + 3 * NL) * 16, 16);
        __builtin_memcpy(&v4, src + head + (size_t)(i + 4 * NL) * 16, 16);
        __builtin_memcpy(&v5, src + head + (size_t)(i + 5 * NL) * 16, 16);
        __builtin_memcpy(&v6, src + head + (size_t)(i + 6 * NL) * 16, 16);
        __builtin_memcpy(&v7, src + head + (size_t)(i + 7 * NL) * 16, 16);
        asm volatile("" ::: "memory");                          // all loads are issued before the first store
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 0 * NL) * 16) = v0;
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 1 * NL) * 16) = v1;
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 2 * NL) * 16) = v2;
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 3 * NL) * 16) = v3;
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 4 * NL) * 16) = v4;
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 5 * NL) * 16) = v5;
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 6 * NL) * 16) = v6;
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 7 * NL) * 16) = v7;
    }
#endif
    for (; i + 3 * NL < body; i += 4 * NL) {                    // four
        uint4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, src + head + (size_t)(i + 0 * NL) * 16, 16);
        __builtin_memcpy(&v1, src + head + (size_t)(i + 1 * NL) * 16, 16);
        __builtin_memcpy(&v2, src + head + (size_t)(i + 2 * NL) * 16, 16);
        __builtin_memcpy(&v3, src + head + (size_t)(i + 3 * NL) * 16, 16);
        asm volatile("" ::: "memory");                          // all loads are issued before the first store
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 0 * NL) * 16) = v0;
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 1 * NL) * 16) = v1;
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 2 * NL) * 16) = v2;
        *reinterpret_cast<uint4 *>(dst + head + (size_t)(i + 3 * NL) * 16) = v3;
    }
    for (; i < body; i += NL) {
        uint4 v;
        __builtin_memcpy(&v, src + head + (size_t)i * 16, 16);
        *reinterpret_cast<uint4 *>(dst + head + (size_t)i * 16) = v;
    }
    const u32 done = head + body * 16;
    if (done + lane < fast) dst[done + lane] = src[done + lane];
    // a stored block that claims more than the chunk holds runs into the appended endBlock
    for (u32 k = fast + lane; k < n; k += NL) {
        const u32 t = sp + k - r.csize;
        dst[k] = (t == 0) ? 0x01 : (t >= 3 ? 0xff : 0x00);
    }
    o.pos += n;
}

__device__ __forceinline__ u32 wave_excl_scan(u32 v, u32 &total);

// ReadPrefixCodes (flate/prefix.go:83-194).  Returns an XFG_ST_* status.
template <class LDS>
__device__ __forceinline__ int read_dynamic_header_impl(Reader &r, LDS &L, u32 lane, bool rfc_rep16, int &lit_state, int &dist_state) {
    HeaderScratch &H = *in_lds(&L.h);                           // (address spaces spelled out: see in_lds)
    refill(r);
    // three ReadBits; each one either gets all its bits or fails consuming none
    if (avail(r) < 5) return XFG_ST_UNEXPECTED_EOF;
    const u32 hlit = peek(r, 5) + 257;  consume(r, 5);
    if (avail(r) < 5) return XFG_ST_UNEXPECTED_EOF;
    const u32 hdist = peek(r, 5) + 1;   consume(r, 5);
    if (avail(r) < 4) return XFG_ST_UNEXPECTED_EOF;
    const u32 hclen = peek(r, 4) + 4;   consume(r, 4);
    if (hlit > 286 || hdist > 30) return XFG_ST_CORRUPTED;      // prefix.go:87-89

    // code-length code lengths, 3 bits each, in clenLens order (prefix.go:67,92-98)
    if (lane < 19) H.clens[lane] = 0;
    wave_sync();
    bool clens_vec = false;                                     // (then r still stands before the 3-bit fields; hp_after is where they end)
    const u32 hp_clens = uni(reader_bitpos(r));
    const u32 hp_after = hp_clens + 3u * uni(hclen);
    if (hp_clens + 3u * 19u + 64u <= r.total * 8u) {
        // every field is there: lane i reads field i straight from the window
        clens_vec = true;
        win_ensure_at(r, hp_clens >> 3, 16);
        const u32 qb = hp_clens + 3u * lane;
        const u32 idx = (qb >> 3) + r.lead;
        const u32 *w32 = reinterpret_cast<const u32 *>(r.win);
        const u32 d0 = w32[(idx >> 2) & (WIN_BYTES / 4 - 1)], d1 = w32[((idx >> 2) + 1) & (WIN_BYTES / 4 - 1)];
        const u32 v = (__builtin_amdgcn_alignbyte(d1, d0, idx & 3) >> (qb & 7)) & 7u;
        // 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15
        const u32 i = lane;
        u32 sidx;
        if (i < 3) sidx = 16 + i;
        else if (i == 3) sidx = 0;
        else if ((i & 1) == 0) sidx = 8 + ((i - 4) >> 1);
        else sidx = 7 - ((i - 5) >> 1);
        if (i < hclen) H.clens[sidx] = (u8)v;
    } else {
        for (u32 i = 0; i < hclen; i++) {
            if (r.bitcnt < 3) refill(r);
            if (avail(r) < 3) return XFG_ST_UNEXPECTED_EOF;
            const u32 v = peek(r, 3);
            consume(r, 3);
            // 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15
            u32 s;
            if (i < 3) s = 16 + i;
            else if (i == 3) s = 0;
            else if ((i & 1) == 0) s = 8 + ((i - 4) >> 1);          // i=4,6,8,..,18 -> 8,9,10,..,15
            else s = 7 - ((i - 5) >> 1);                            // i=5,7,9,..,17 -> 7,6,5,..,1
            if (lane == 0) H.clens[s] = (u8)v;
        }
    }
    wave_sync();
    int cst;
    { PROF_T0(); cst = build_table<T_CLEN, CLEN_ROOT, 0>(L, H.clen, H.clens, 19, lane); PROF_ADD(PF_H_CLEN, lane); }
    if (cst == 2) { if (clens_vec) reader_seek(r, hp_after); return XFG_ST_CORRUPTED; }
    PROF_T0();

    // literal/length and distance code lengths with the 16/17/18 repeaters (prefix.go:125-165).
    // 64 bit positions at a time: lane i looks up the code word that would start at bit hp + i and
    // packs what it means; then a wave-uniform walk hops from symbol to symbol through those
    // answers with readlane (no memory access per symbol).
    for (u32 i = lane; i < sizeof(H.lens) / 4; i += 64) reinterpret_cast<u32 *>(H.lens)[i] = 0;
    // (uni(): the walk below must compile to scalar code, so everything it depends on is pinned to SGPRs)
    const u32 max_syms = uni(hlit + hdist);
    if (cst == 1) { if (clens_vec) reader_seek(r, hp_after); return XFG_ST_CORRUPTED; }   // "decode with empty prefix tree"
    const u32 total_bits = uni(r.total * 8);
    u32 hp = clens_vec ? hp_after : uni(reader_bitpos(r));
    u32 prev = 0;                                               // clenLast
    int st = XFG_ST_OK;
    for (u32 sym = 0; sym < max_syms && st == XFG_ST_OK;) {
        win_ensure_at(r, hp >> 3, 16);
        u32 info;
        {
            const u32 qb = hp + lane;
            const u32 idx = (qb >> 3) + r.lead;                 // window position of the lane's first byte
            const u32 *w32 = reinterpret_cast<const u32 *>(r.win);
            const u32 d0 = w32[(idx >> 2) & (WIN_BYTES / 4 - 1)], d1 = w32[((idx >> 2) + 1) & (WIN_BYTES / 4 - 1)];
            const u32 field = __builtin_amdgcn_alignbyte(d1, d0, idx & 3) >> (qb & 7);    // >= 25 bits; 14 are needed
            const u32 e = H.clen[field & ((1u << CLEN_ROOT) - 1)];
            const u32 take = e_take(e), c = e_val(e) & 31;
            const u32 nb = c == 16 ? 2u : c == 17 ? 3u : c == 18 ? 7u : 0u;
            const u32 xv = (field >> take) & ((1u << nb) - 1);
            info = take | (nb << 4) | (c << 8) | (xv << 16) | ((e & E_STOP) ? 0x80000000u : 0u);
        }
        u32 pos = 0;
        // Fast way: every symbol of the window is whole (no end of input near) -- hop through the window
        // keeping only positions and counts, then let the lanes that were hopped on store their lengths
        // at once.  Zero runs need no store (lens[] was cleared); "repeat previous" runs follow in order.
        bool fast = hp + 64 + 16 <= total_bits;
        if (fast) {
            const u32 c = (info >> 8) & 31;
            const u32 cnt = c < 16 ? 1u : (c == 18 ? 11u : 3u) + ((info >> 16) & 127);
            u32 step = (info & 15) + ((info >> 4) & 7);
            step = step ? step : 64u;                             // (a code word is at least one bit: the hop loop ends whatever the table holds)
            // the hops: positions only (a scalar loop of six instructions per symbol); the symbol counts come from a prefix
            // sum over the visited lanes afterwards, and the path is cut behind the lane whose run reaches max_syms
            u64 vis = 0;
            while (pos < 64) {
                const u32 st1 = (u32)__builtin_amdgcn_readlane((int)step, (int)pos);
                asm("s_bitset1_b64 %0, %1" : "+s"(vis) : "s"(pos));
                pos += st1;
            }
            bool my = (vis >> lane) & 1;
            u32 total;
            const u32 idx = sym + wave_excl_scan(my ? cnt : 0u, total);
            u32 s2 = sym + total;
            if (s2 >= max_syms) {
                const u64 reach = __ballot(my && idx + cnt >= max_syms);      // (not empty)
                const u32 f = (u32)__ffsll((long long)reach) - 1;
                vis &= (2ull << f) - 1;
                my = my && lane <= f;
                pos = f + (u32)__builtin_amdgcn_readlane((int)step, (int)f);
                s2 = (u32)__builtin_amdgcn_readlane((int)(idx + cnt), (int)f);
            }
            // an invalid code on the path, or a repeat with nothing before it (prefix.go:139-141): the exact way
            if ((__ballot((info >> 31) != 0) & vis) || (sym == 0 && ((u32)__builtin_amdgcn_readlane((int)c, 0)) == 16)) { fast = false; pos = 0; }
            else {
                if (my && c < 16 && idx < max_syms) H.lens[idx] = (u8)c;
                const u64 m16 = __ballot(my && c == 16);
                const u64 mexp = __ballot(my && c < 16);             // explicit lengths
                u32 v16 = 0;
                if (m16) {
                    // What a "repeat previous" run repeats comes from the nearest lane below it on the path that is not
                    // such a run itself: the previous array entry (an explicit length, or zero after a zero run), or in
                    // the in-repo reader's reading the last explicit length (clenLast); from before this window if there
                    // is no such lane.  Every run's lane stores its own 3 to 6 entries.
                    const u64 msrc = rfc_rep16 ? __ballot(my && c != 16) : mexp;
                    const u64 below = msrc & ((1ull << lane) - 1);
                    const u32 j = below ? 63u - (u32)__builtin_clzll(below) : 0u;
                    const u32 got = (u32)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)(c < 16 ? c : 0u));
                    v16 = below ? got : prev;
                    if (my && c == 16) {
#pragma unroll
                        for (u32 t = 0; t < 6; t++)
                            if (t < cnt && idx + t < max_syms) H.lens[idx + t] = (u8)v16;
                    }
                }
                if (rfc_rep16) {
                    // the last array entry written: what the last lane on the path stands for
                    const u32 jl = 63u - (u32)__builtin_clzll(vis);
                    prev = (u32)__builtin_amdgcn_readlane((int)(c < 16 ? c : c == 16 ? v16 : 0u), (int)jl);
                } else if (mexp) {
                    prev = (u32)__builtin_amdgcn_readlane((int)c, 63 - __builtin_clzll(mexp));
                }
                sym = s2;
                if (sym > max_syms) st = XFG_ST_CORRUPTED;          // prefix.go:161-163
            }
        }
        if (!fast) {
            // Exact way, symbol by symbol: every read of the reference that can fail is checked where it fails.
            wave_sync();
            if (rfc_rep16 && sym > 0) prev = uni(H.lens[sym - 1]);
            while (pos < 64 && sym < max_syms) {
                const u32 inf = (u32)__builtin_amdgcn_readlane((int)info, (int)pos);
                const i32 av = (i32)total_bits - (i32)(hp + pos);
                const u32 take = inf & 15;
                if ((i32)take > av) { st = XFG_ST_UNEXPECTED_EOF; break; }
                pos += take;
                if (inf >> 31) { st = XFG_ST_CORRUPTED; break; }      // the degenerate filler symbol
                const u32 c = (inf >> 8) & 31;
                if (c < 16) {
                    if (lane == 0) H.lens[sym] = (u8)c;
                    prev = c;
                    sym++;
                    continue;
                }
                if (c == 16 && sym == 0) { st = XFG_ST_CORRUPTED; break; }   // prefix.go:139-141
                const u32 nb = (inf >> 4) & 7;
                if ((i32)nb > av - (i32)take) { st = XFG_ST_UNEXPECTED_EOF; break; }
                pos += nb;
                const u32 val = c == 16 ? prev : 0u;
                const u32 rep = (c == 18 ? 11u : 3u) + ((inf >> 16) & 127);
                if (c != 16 && rfc_rep16) prev = 0;                 // stdlib: 16 repeats the previous array entry
                const u32 end = sym + rep;
                for (u32 k = sym + lane; k < end && k < max_syms; k += 64) H.lens[k] = (u8)val;
                sym = end;
                if (sym > max_syms) { st = XFG_ST_CORRUPTED; break; } // prefix.go:161-163
            }
        }
        hp += pos;
    }
    reader_seek(r, hp);
    if (st != XFG_ST_OK) return st;
    wave_sync();
    PROF_ADD(PF_H_LENS, lane);
    // lens[0 .. hlit) literal/length, lens[hlit .. hlit+hdist) distance
    { PROF_T0(); lit_state = build_table<T_LIT, LIT_ROOT, 0>(L, L.tab, H.lens, hlit, lane); PROF_ADD(PF_H_LIT, lane); }
    if (lit_state == 2) return XFG_ST_CORRUPTED;
    { PROF_T0(); dist_state = build_table<T_DIST, DIST_ROOT, LIT_TAB>(L, (L.tab + LIT_TAB), H.lens + hlit, hdist, lane); PROF_ADD(PF_H_DIST, lane); }
    if (dist_state == 2) return XFG_ST_CORRUPTED;
    return XFG_ST_OK;
}

// (The reader's pointers with their address spaces spelled out, for when this is not inlined: see in_lds.)
__device__ __forceinline__ Reader reader_known(const Reader &r) {
    Reader k = r;
    k.comp = in_global(r.comp);
    k.win = in_lds(r.win);
    return k;
}
template <class LDS>
__device__ int read_dynamic_header(Reader &r, LDS &L, u32 lane, bool rfc_rep16, int &lit_state, int &dist_state) {
    if (LDS::pointers_known) return read_dynamic_header_impl(r, L, lane, rfc_rep16, lit_state, dist_state);   // (the one-wave kernel: everything is inlined)
    Reader k = reader_known(r);
    const int st = read_dynamic_header_impl(k, L, lane, rfc_rep16, lit_state, dist_state);
    r = k;
    return st;
}

// Fixed code sets of RFC 1951 3.2.6 (flate/prefix.go:35-63).
template <class LDS>
__device__ void build_fixed(LDS &L, u32 lane) {
    HeaderScratch &H = *in_lds(&L.h);
    for (u32 i = lane; i < 288; i += 64)
        H.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
    wave_sync();
    build_table<T_LIT, LIT_ROOT, 0>(L, L.tab, H.lens, 288, lane);
    if (lane < 32) H.clens[lane] = 5;
    wave_sync();
    build_table<T_DIST, DIST_ROOT, LIT_TAB>(L, (L.tab + LIT_TAB), H.clens, 32, lane);
}

// ---------------------------------------------------------------------------
// Lane-parallel decode of one Huffman block: readBlock (flate/reader.go:162-250)
// and dictDecoder (flate/dict_decoder.go:86-132) for 64 segments at a time.
// ---------------------------------------------------------------------------
enum : u32 { F_NONE = 0, F_EOB = 1, F_CORRUPT = 2, F_EOF = 3, F_DIST = 4 };

struct Seg {
    u32 end;        // bit position after the last token decoded
    u32 nout;       // bytes those tokens produce
    u32 nmatch;     // match tokens among them
    u32 flag;       // F_*: why the lane stopped before its segment end
};

// LDS accesses at any byte alignment.  A 64-bit (or wider) DS access off its natural alignment
// is replayed at ~64 cycles per wave-instruction on gfx950, so the bit buffer is fed from
// aligned dwords put together with v_alignbyte, and byte-granular copies use 32-bit accesses.
__device__ __forceinline__ u32 lds_load32(const u8 *p) {
    u32 v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ void lds_store32(u8 *p, u32 v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void no_merge() { asm volatile("" ::: "memory"); }   // keeps adjacent 32-bit accesses apart
// The 8 bytes at byte position bytepos of the window (three aligned dwords, funnel-shifted).
__device__ __forceinline__ u64 win_load64(const u32 *w32, u32 bytepos) {
    const u32 i = bytepos >> 2;
    const u32 d0 = w32[i], d1 = w32[i + 1], d2 = w32[i + 2];
    const u32 lo = __builtin_amdgcn_alignbyte(d1, d0, bytepos & 3);
    const u32 hi = __builtin_amdgcn_alignbyte(d2, d1, bytepos & 3);
    return ((u64)hi << 32) | lo;
}

// The compressed bytes [wb0, wb0 + BWIN_BYTES) of the virtual stream, 16 bytes per lane and
// unit, into registers.  wb0 (may be negative by < 16) is chosen so that the global address
// is 16-byte aligned.  Bytes past the stream read as the endBlock, then zero.
__device__ __forceinline__ void fetch_batch_window(const Reader &r, i32 wb0, u32 lane, uint4 (&pf)[NPF]) {
    const u8 *src = r.comp + r.base_al + r.lead;                // stream byte 0
#pragma unroll
    for (int u = 0; u < NPF; u++) {
        const u32 o = (u32)u * 1024u + lane * 16u;
        const i32 ci = wb0 + (i32)o;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (o < (u32)BWIN_BYTES) {
            if (ci >= 0 && ci + 16 <= (i32)r.csize) {
                v = *reinterpret_cast<const uint4 *>(src + ci);
            } else if (ci + 16 > 0 && ci < (i32)r.total) {
                const u64 lo = stream_word(r, src + ci, ci), hi = stream_word(r, src + ci + 8, (i64)ci + 8);
                v = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
            }
        }
        pf[u] = v;
    }
}
template <class LDS>
__device__ __forceinline__ void store_batch_window(LDS &L, u32 lane, const uint4 (&pf)[NPF]) {
    u8 *dst = reinterpret_cast<u8 *>(L.win);
#pragma unroll
    for (int u = 0; u < NPF; u++) {
        const u32 o = (u32)u * 1024u + lane * 16u;
        if (o < (u32)BWIN_BYTES) *reinterpret_cast<uint4 *>(dst + o) = pf[u];
    }
}

// Decode the tokens that START in [start, seg_end), or until the lane's output cap.
// One table lookup per step, the same instruction stream for every lane whatever it is
// looking at: the entry says what the step is (literal, length, distance, link to a
// second-level table) and the lane's next lookup (tbase, twidth) follows from it by selects,
// not branches.  Only what ends the lane (end of block, invalid code, end of input, distance
// beyond the history) leaves the straight line.  The bit buffer is topped up to >= 56 valid
// bits every two steps; one step consumes <= 28.
//   EMIT = false: count only.  EMIT = true: literals go to the LDS output window at orel..
//   (obase = the same position counted from the start of the chunk), matches to tok/tdst at moff..
//   TAIL = true: the batch may run past the end of the stream; every read of the reference
//   that can fail with io.ErrUnexpectedEOF is checked at the same point.
template <bool EMIT, class LDS>
__device__ __forceinline__ Seg decode_segment_tail(LDS &L, const TokList tl, i32 wb0, u32 start, u32 seg_end, u32 total_bits,
                                                   bool dist_empty, u32 orel, u32 obase, u32 moff, u32 &err_out)
{
    const u32 *win = reinterpret_cast<const u32 *>(L.win);
    const u32 tab_lds = (u32)(uintptr_t)(const __attribute__((address_space(3))) u32 *)L.tab;
    u8 *outw = reinterpret_cast<u8 *>(L.win);               // positions count from the start of L.win
    u32 wp = (u32)((i32)(start >> 3) - wb0);
    u64 buf = win_load64(win, wp) >> (start & 7);
    u32 cnt = 56 - (start & 7);                                 // account 7 of the 8 bytes: cnt stays below 64
    wp += 7;
    u64 nextw = win_load64(win, wp);
    u32 pos = start, nout = 0, nm = 0, flag = F_NONE;
    u32 tbase = 0, twidth = LIT_ROOT, len = 0;                  // tbase: byte offset in L.tab
    for (;;) {
        buf |= nextw << cnt;
        wp += (63 - cnt) >> 3;
        cnt |= 56;
        nextw = win_load64(win, wp);
        const u32 lo = (u32)buf;
        const u32 e = L.tab[(tbase >> 2) + (lo & ((1u << twidth) - 1))];
        const u32 take = e_take(e), xb = e_extra(e), val = e_val(e);
        const u32 value = val + ((lo >> (take - xb)) & ((1u << xb) - 1));   // length or distance
        const bool is_lit = (e & E_LIT) != 0, is_len = (e & E_LEN) != 0, is_dist = (e & E_DIST) != 0, is_sub = (e & E_SUB) != 0;
        if (pos + take > total_bits) {
            // io.ErrUnexpectedEOF: which read of the reference came up short?
            const u32 code = take - ((is_len || is_dist) ? xb : 0);
            if (pos + code > total_bits) {
                // the code word: nothing of it is consumed, second-level steps included
                if (tbase >= 4u * (1 << LIT_ROOT) && tbase < 4u * LIT_TAB) pos -= LIT_ROOT;
                else if (tbase >= 4u * (LIT_TAB + (1 << DIST_ROOT))) pos -= DIST_ROOT;
            } else {
                pos += code;                                    // the extra bits
            }
            flag = F_EOF;
            break;
        }
        if ((e & E_STOP) || (dist_empty && is_len)) {
            pos += take;
            flag = ((e & E_STOP) && (e & E_EOB)) ? F_EOB : F_CORRUPT;   // 286/287, 30/31, the degenerate filler, empty distance tree
            break;
        }
        if (EMIT) {
            if (is_lit) outw[orel + nout] = (u8)val;
            if (is_dist) {
                const u32 dst = obase + nout;
                const u32 hist = dst < (u32)MAX_HIST ? dst : (u32)MAX_HIST;      // dictDecoder.HistSize
                if (value > hist) { pos += take; flag = F_DIST; err_out = nout; break; }   // flate/reader.go:223-225
                tl.tok[moff + nm] = (len - 3) | ((value - 1) << 8);
                tl.tdst[moff + nm] = (u16)(orel + nout);
            }
        }
        tbase = is_sub ? val - tab_lds : (is_len ? 4u * LIT_TAB : 0u);
        twidth = is_sub ? xb : (is_len ? (u32)DIST_ROOT : (u32)LIT_ROOT);
        if (is_len) len = value;
        if (is_lit) nout++;
        if (is_dist) { nout += len; nm++; }
        buf >>= take; cnt -= take; pos += take;
        if ((is_lit || is_dist) && (pos >= seg_end || nout >= (u32)OUT_CAP)) break;    // a token boundary
    }
    Seg r;
    r.end = pos; r.nout = nout; r.nmatch = nm; r.flag = flag;
    return r;
}

// The same for a batch that cannot run into the end of the stream (every batch of a chunk but
// its last one or two): no end-of-input checks, and the step is straight-line integer
// arithmetic -- the entry's flag bits become all-ones / all-zero masks that select the next
// lookup and what to add to the counters -- with one exit test per step and two steps per
// top-up of the bit buffer.
// (m & a) | (~m & b) in one v_bitop3
#define SEL(m, a, b) __builtin_amdgcn_bitop3_b32((m), (a), (b), 0xCA)
typedef const __attribute__((address_space(3))) u32 *lds_cu32p;
typedef const __attribute__((address_space(3))) u8 *lds_cu8p;
struct __attribute__((packed)) PackedU64 { u64 v; };             // 8 bytes at any alignment
typedef const __attribute__((address_space(3))) PackedU64 *lds_cpu64p;

// (NEED, with EMIT = false: err_out becomes the history the lane's matches need before the lane's first byte,
// max over its matches of distance - bytes the lane has produced by then, as an i32: the count pass of the
// workgroup kernel finds a distance beyond the history, flate/reader.go:223-225, before anything is written.)
template <bool EMIT, bool NEED = false, class LDS>
__device__ __forceinline__ Seg decode_segment_fast(LDS &L, const TokList tl, i32 wb0, u32 start, u32 seg_end,
                                                   u32 orel, u32 obase, u32 moff, u32 lane, u32 &err_out)
{
    // LDS byte addresses: the lookup state is the address of the table the next step indexes
    const u32 tab0 = (u32)(uintptr_t)(lds_cu32p)L.tab;
    const u32 tab1 = tab0 + 4u * LIT_TAB;
    const u32 win0 = (u32)(uintptr_t)(lds_cu8p)reinterpret_cast<const u8 *>(L.win);
    u8 *outw = reinterpret_cast<u8 *>(L.win);               // positions count from the start of L.win
    u32 wp = win0 + (u32)((i32)(start >> 3) - wb0);
    u64 nextw;
    nextw = ((lds_cpu64p)(uintptr_t)wp)->v;
    u64 buf = nextw >> (start & 7);
    u32 cnt = 56 - (start & 7);                                 // account 7 of the 8 bytes: cnt stays below 64
    wp += 7;
    nextw = ((lds_cpu64p)(uintptr_t)wp)->v;
    u32 rem1 = seg_end - 1 - start;                             // bits to the segment end, minus one: negative = at or past it
    u32 nout = 0, nm = 0, len = 0;
    u32 tb = tab0, tw = LIT_ROOT;
    u32 e = 0, bad = 0;
    i32 need = (i32)0x80000000;
    const u32 dump_at = L.dump + lane;
    for (;;) {
        // top up; the 8 bytes after it are already on their way for the next top-up
        buf |= nextw << cnt;
        wp += (63 - cnt) >> 3;
        cnt |= 56;
        nextw = ((lds_cpu64p)(uintptr_t)wp)->v;
        // two steps per top-up, one exit from the loop: a lane that finishes in the first step sits out the second
        // (written with `goto out` in both steps, the compiler keeps a state variable to tell the exits apart)
        auto one_step = [&]() -> bool {
            const u32 lo = (u32)buf;
            e = *(lds_cu32p)(uintptr_t)(tb + (__builtin_amdgcn_ubfe(lo, 0u, tw) << 2));
            const u32 take = e & 31, xb = __builtin_amdgcn_ubfe(e, 5u, 5u), val = e >> 16;
            const u32 value = val + __builtin_amdgcn_ubfe(lo, take - xb, xb);   // length or distance
            const u32 m_lit = (u32)__builtin_amdgcn_sbfe((int)e, 10u, 1u);      // all ones or zero
            const u32 m_len = (u32)__builtin_amdgcn_sbfe((int)e, 11u, 1u);
            const u32 m_dist = (u32)__builtin_amdgcn_sbfe((int)e, 12u, 1u);
            const u32 m_sub = (u32)__builtin_amdgcn_sbfe((int)e, 13u, 1u);
            const u32 m_tok = (u32)__builtin_amdgcn_sbfe((int)e, 15u, 1u);
            if (EMIT) {
                // a literal goes to its place; every other step writes a byte nobody reads
                outw[SEL(m_lit, orel + nout, dump_at)] = (u8)val;
                if (m_dist) {
                    const u32 dst = obase + nout;
                    const u32 hist = dst < (u32)MAX_HIST ? dst : (u32)MAX_HIST;  // dictDecoder.HistSize
                    bad = value > hist ? 0x80000000u : 0u;                       // flate/reader.go:223-225
                    tl.tok[moff + nm] = (len - 3) | ((value - 1) << 8);
                    tl.tdst[moff + nm] = (u16)(orel + nout);
                }
            }
            if (NEED) {
                const i32 cand = (i32)SEL(m_dist, value - nout, 0x80000000u);
                need = cand > need ? cand : need;
            }
            tb = SEL(m_sub, val, SEL(m_len, tab1, tab0));
            tw = SEL(m_sub, xb, SEL(m_len, (u32)DIST_ROOT, (u32)LIT_ROOT));
            len = SEL(m_len, value, len);
            nout += (len & m_dist) - m_lit;                     // a literal adds one, a finished match its length
            nm -= m_dist;
            buf >>= take; cnt -= take; rem1 -= take;
            // finished: a token boundary at or past the segment end or the output cap, or an entry that ends the lane
            const u32 over = rem1 | ((u32)(OUT_CAP - 1) - nout);
            u32 stop = (over & m_tok) | e;
            if (EMIT) stop |= bad;
            return (i32)stop < 0;
        };
        bool fin = one_step();
        if (!fin) fin = one_step();
        if (fin) break;
    }
    u32 flag = F_NONE;
    if (e & E_STOP) flag = (e & E_EOB) ? F_EOB : F_CORRUPT;     // 286/287, 30/31, the degenerate filler
    else if (EMIT && bad) { nout -= len; nm--; flag = F_DIST; err_out = nout; }
    if (NEED) err_out = (u32)need;
    Seg r;
    r.end = seg_end - 1 - rem1; r.nout = nout; r.nmatch = nm; r.flag = flag;
    return r;
}

// Synchronisation passes: the same walk as decode_segment_fast without the bookkeeping -- only
// where the tokens start matters.
//   CHECK = false  the speculative pass.  Lane decodes from its segment boundary and sets a bit in
//                  its row of L.mask for every position where a token starts (its start included).
//   CHECK = true   a lane whose true start became known walks from there until it stands on a
//                  marked position: from that point on it would repeat the speculative path, so
//                  that path's end is its end (returns merged = true), or until its own end.
// Returns the bit position after the last token; e_out = the entry that ended the walk.
template <bool CHECK, bool RUNIN, bool MARK = true, class LDS>
__device__ __forceinline__ u32 sync_segment(LDS &L, i32 wb0, u32 start, u32 seg_start, u32 seg_bits, u32 lane,
                                            u32 &e_out, bool &merged, u32 keep_stop = ~0u)
{
    const u32 tab0 = (u32)(uintptr_t)(lds_cu32p)L.tab;
    const u32 tab1 = tab0 + 4u * LIT_TAB;
    const u32 win0 = (u32)(uintptr_t)(lds_cu8p)reinterpret_cast<const u8 *>(L.win);
    // the lane's mask word for position rel, as an LDS byte address: word (rel >> 5) of lane l is at [w * 64 + l]
    typedef __attribute__((address_space(3))) u32 *lds_u32p;
    // (a shift and a shift-add; the compiler's own form of the same address is shift, and, add)
    const u32 mbase = (u32)(uintptr_t)(lds_cu32p)L.mask + (lane << 2);
    auto mrow_addr = [mbase](u32 rel_) { u32 a; asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a) : "v"(rel_ >> 5), "v"(mbase)); return a; };
#define MROW(rel_) ((lds_u32p)(uintptr_t)mrow_addr(rel_))
    // (opaque to the compiler: as a comparison of the lane number it would live in a scalar register pair,
    // which is spilled and reloaded in every step)
    asm volatile("" : "+v"(keep_stop));
    const u32 ks_sign = keep_stop | ~(u32)E_SIGN, ks_inv = ~keep_stop;
    u32 wp = win0 + (u32)((i32)(start >> 3) - wb0);
    u64 nextw;
    nextw = ((lds_cpu64p)(uintptr_t)wp)->v;
    u64 buf = nextw >> (start & 7);
    u32 cnt = 56 - (start & 7);
    wp += 7;
    nextw = ((lds_cpu64p)(uintptr_t)wp)->v;
    u32 rel = start - seg_start;                               // position in the lane's own segment
    u32 tb = tab0, tw = LIT_ROOT;
    u32 e = 0, hit = 0;
    u32 tokprev = ~0u;                                          // the walk starts at a token boundary
    if (!CHECK && MARK && (i32)rel >= 0) *MROW(rel) = 1u << (rel & 31);   // rows were cleared: a plain store
    for (;;) {
        buf |= nextw << cnt;
        wp += (63 - cnt) >> 3;
        cnt |= 56;
        nextw = ((lds_cpu64p)(uintptr_t)wp)->v;
        // (two steps per top-up and one exit from the loop, as in decode_segment_fast)
        auto one_step = [&]() -> bool {
            const u32 lo = (u32)buf;
            e = *(lds_cu32p)(uintptr_t)(tb + (__builtin_amdgcn_ubfe(lo, 0u, tw) << 2));
            if (CHECK) {
                const u32 w = *MROW(rel);
                hit = (w >> (rel & 31)) & tokprev & 1u;
            }
            const u32 take = e & 31, xb = __builtin_amdgcn_ubfe(e, 5u, 5u), val = e >> 16;
            const u32 m_len = (u32)__builtin_amdgcn_sbfe((int)e, 11u, 1u);
            const u32 m_sub = (u32)__builtin_amdgcn_sbfe((int)e, 13u, 1u);
            const u32 m_tok = (u32)__builtin_amdgcn_sbfe((int)e, 15u, 1u);
            tb = SEL(m_sub, val, SEL(m_len, tab1, tab0));
            tw = SEL(m_sub, xb, SEL(m_len, (u32)DIST_ROOT, (u32)LIT_ROOT));
            if (CHECK) {
                if (hit) return true;                           // nothing of this step counts
            }
            buf >>= take; cnt -= take; rel += take;
            if (!CHECK && MARK) {
                // a token starts at rel (words past the segment exist; nobody tests them); positions
                // before the segment (the run-in, rel < 0) are nobody's start and are not marked
                if (RUNIN) {
                    const u32 in_seg = ~(u32)((i32)rel >> 31);
                    const u32 rc = rel & in_seg;
                    __hip_atomic_fetch_or(MROW(rc), m_tok & in_seg & (1u << (rc & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                } else {
                    __hip_atomic_fetch_or(MROW(rel), m_tok & (1u << (rel & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            tokprev = m_tok;
            const u32 over = seg_bits - 1 - rel;                // negative = at or past the segment end
            if (!CHECK && (RUNIN || XFG_SPEC_WALK_ON_DYNAMIC)) {
                // keep_stop = 0: a guessed path walks on over a code that would end it (it is a guess: with the fixed codes
                // seven zero bits read as the end of the block) and such a code counts as a token boundary for the exit
                const u32 m_stop = (u32)__builtin_amdgcn_sbfe((int)e, 14u, 1u);
                const u32 m_end = m_tok | (m_stop & ks_inv);
                return (i32)((over & m_end) | (e & ks_sign)) < 0;
            } else {
                return (i32)((over & m_tok) | e) < 0;
            }
        };
        bool fin = one_step();
        if (!fin) fin = one_step();
        if (fin) break;
    }
#undef MROW
    e_out = e;
    merged = CHECK && hit != 0;
    return seg_start + rel;
}

// Inclusive prefix sum over the wave with DPP row shifts and row broadcasts (no LDS traffic).
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);    // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);    // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);    // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);    // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ u32 wave_excl_scan(u32 v, u32 &total) {
    const u32 incl = wave_incl_scan(v);
    total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
    return incl - v;
}
// The value of the lane below (lane 0 keeps its own): one DPP move, no LDS crossbar.
__device__ __forceinline__ u32 lane_prev(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);    // wave_shr:1
}
__device__ __forceinline__ u32 lane_get(u32 v, u32 l) {       // l is wave-uniform
    return (u32)__builtin_amdgcn_readlane((int)v, (int)l);
}

__device__ __forceinline__ void mem_sync() {
    // Make this wave's global stores visible to its own later loads: both go through the
    // CU's write-through L1, so waiting for the stores (workgroup-scope release) suffices.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// n <= 16 bytes LDS -> LDS, any alignment, ranges disjoint: all reads happen before the writes.
// Pieces overlap instead of branching on the exact length.
__device__ __forceinline__ void copy16_l2l(u8 *dst, const u8 *src, u32 n) {
    if (n >= 8) {
        const u32 a0 = lds_load32(src); no_merge();
        const u32 a1 = lds_load32(src + 4); no_merge();
        const u32 b0 = lds_load32(src + n - 8); no_merge();
        const u32 b1 = lds_load32(src + n - 4); no_merge();
        lds_store32(dst, a0); no_merge();
        lds_store32(dst + 4, a1); no_merge();
        lds_store32(dst + n - 8, b0); no_merge();
        lds_store32(dst + n - 4, b1);
    } else if (n >= 4) {
        const u32 a = lds_load32(src), b = lds_load32(src + n - 4);
        lds_store32(dst, a); no_merge();
        lds_store32(dst + n - 4, b);
    } else {                                                    // 1..3
        const u32 m = n > 1 ? 1u : 0u;
        const u8 a = src[0], b = src[m], c = src[n - 1];
        dst[0] = a; dst[m] = b; dst[n - 1] = c;
    }
}
// n <= 16 bytes global -> LDS, any alignment.
__device__ __forceinline__ void copy16_g2l(u8 *dst, const u8 *src, u32 n) {
    if (n >= 8) {
        u64 a, b;
        __builtin_memcpy(&a, src, 8);
        __builtin_memcpy(&b, src + n - 8, 8);
        lds_store32(dst, (u32)a); no_merge();
        lds_store32(dst + 4, (u32)(a >> 32)); no_merge();
        lds_store32(dst + n - 8, (u32)b); no_merge();
        lds_store32(dst + n - 4, (u32)(b >> 32));
    } else if (n >= 4) {
        u32 a, b;
        __builtin_memcpy(&a, src, 4);
        __builtin_memcpy(&b, src + n - 4, 4);
        lds_store32(dst, a); no_merge();
        lds_store32(dst + n - 4, b);
    } else {
        const u32 m = n > 1 ? 1u : 0u;
        const u8 a = src[0], b = src[m], c = src[n - 1];
        dst[0] = a; dst[m] = b; dst[n - 1] = c;
    }
}

// Where the bytes of a match source that lie BEFORE the window come from.  Window position p (< woff) is the chunk's
// output byte base - woff + p.  at(p) points at that byte and at least 48 more bytes that may be read (not
// necessarily meaningful ones); lo() is the window position of the chunk's first byte, below which nothing exists.
//   SrcHbm   the chunk's output in HBM, through L2: everything before the window has been flushed (the one-wave kernel)
//   SrcRing  the workgroup kernel's history ring in LDS (inflate_wg_kernel.inc): the last 32 KB of output before
//            the window, byte q of the chunk at ring offset (q + the output's HBM address mod 16) mod 32 KB,
//            with the ring's first 64 bytes repeated behind its end
constexpr u32 RING_BYTES = 32768, RING_MIRROR = 64;
struct SrcHbm {
    const u8 *gwin;     // HBM address of window position 0
    i32 pmin;
    static constexpr bool always_readable = false;              // the chunk may be shorter than the 8 bytes an idle lane reads
    __device__ __forceinline__ const u8 *at(i32 p) const { return gwin + p; }
    __device__ __forceinline__ i32 lo() const { return pmin; }
    __device__ __forceinline__ SrcHbm known() const { SrcHbm k; k.gwin = in_global(gwin); k.pmin = pmin; return k; }
};
struct SrcRing {
    u32 ring;           // LDS byte address of the ring
    u32 gbase;          // ring key of window position 0
    i32 pmin;
    static constexpr bool always_readable = true;
    __device__ __forceinline__ const u8 *at(i32 p) const {
        return (const u8 *)(lds_cu8p)(uintptr_t)(ring + ((gbase + (u32)p) & (RING_BYTES - 1)));
    }
    __device__ __forceinline__ i32 lo() const { return pmin; }
    __device__ __forceinline__ SrcRing known() const { return *this; }
};

// One long match, copied by the whole wave: byte j of the match is source byte j mod dist
// (flate/dict_decoder.go:93-100 in closed form).  The dist source bytes end where the destination
// starts and are final (window) or lie before the window (HBM/L2), so all reads precede all writes
// of the same bytes and the 64 lanes need no ordering among themselves.
template <class SRC>
__device__ __forceinline__ void copy_long_match(u8 *outw, const SRC &src, u32 woff, u32 d, u32 len, u32 dist, u32 lane) {
    const i32 s = (i32)d - (i32)dist;
    if (s >= (i32)woff) {                                       // wave-uniform
        // The whole source lies in the window (19 of 20 long matches on match-heavy data): LDS reads only.  Below, a
        // lane's source byte may be in the window or before it, and the choice between the two address spaces per
        // lane makes every load a flat one, at several times an LDS read's latency.
        if (dist >= len || dist == 1) {
            const u32 nw = len >> 2, nt = len & 3;
            const bool rep1 = dist == 1;
            u32 w = 0, t = 0;
            if (rep1) { w = outw[s] * 0x01010101u; t = w; }
            else {
                if (lane < nw) w = lds_load32(outw + s + 4 * lane);
                if (lane < nt) t = outw[s + (i32)((len & ~3u) + lane)];
            }
            if (lane < nw) lds_store32(outw + d + 4 * lane, w);
            if (lane < nt) outw[d + (len & ~3u) + lane] = (u8)t;
            return;
        }
        const u32 rcp = ((1u << 20) + dist - 1) / dist;
        u8 b[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const u32 j = lane + 64u * i;
            b[i] = 0;
            if (j < len) b[i] = outw[s + (i32)(j - ((j * rcp) >> 20) * dist)];
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const u32 j = lane + 64u * i;
            if (j < len) outw[d + j] = b[i];
        }
        return;
    }
    if (dist >= len || dist == 1) {                             // wave-uniform
        // No wrap inside the match (or one byte repeated): a dword per lane, then the last len mod 4 bytes.
        const u32 nw = len >> 2, nt = len & 3;
        const bool rep1 = dist == 1;
        const i32 p = rep1 ? s : s + (i32)(4 * lane);           // this lane's source dword, window position
        const i32 q = rep1 ? s : s + (i32)((len & ~3u) + lane); // and its tail byte
        u32 w = 0, t = 0;
        if (lane < nw || rep1) {
            if (rep1) w = (p < (i32)woff ? src.at(p)[0] : outw[p]) * 0x01010101u;
            else if (p + 4 <= (i32)woff) __builtin_memcpy(&w, src.at(p), 4);
            else if (p >= (i32)woff) w = lds_load32(outw + p);
            else {
                for (int k = 0; k < 4; k++) {
                    const i32 pk = p + k;
                    w |= (u32)(pk < (i32)woff ? src.at(pk)[0] : outw[pk]) << (8 * k);
                }
            }
        }
        if (rep1) t = w;
        else if (lane < nt) t = q < (i32)woff ? src.at(q)[0] : outw[q];
        if (lane < nw) lds_store32(outw + d + 4 * lane, w);
        if (lane < nt) outw[d + (len & ~3u) + lane] = (u8)t;
        return;
    }
    // j mod dist for j < 258 by a 20-bit reciprocal: exact because j * (dist - 1) < 2^20
    const u32 rcp = ((1u << 20) + dist - 1) / dist;
    u8 b[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const u32 j = lane + 64u * i;
        b[i] = 0;
        if (j < len) {
            const u32 o = j - ((j * rcp) >> 20) * dist;
            const i32 p = s + (i32)o;
            b[i] = p < (i32)woff ? src.at(p)[0] : outw[p];
        }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const u32 j = lane + 64u * i;
        if (j < len) outw[d + j] = b[i];
    }
}

// Long matches whose whole source lies before the window and does not overlap the destination
// depend on nothing in the batch: up to four of them are fetched together (one dword per lane
// and match, plus the last len mod 4 bytes), so their HBM/L2 latencies overlap.
//   mask  lanes of this round that hold such a token; d/len/dist are the lanes' tokens
template <class SRC>
__device__ __forceinline__ void copy_independent_long(u8 *outw, const SRC &src, u64 mask, u32 d, u32 len, u32 dist, u32 lane) {
    while (mask) {
        u32 dd[4], ll[4], w[4], t[4];
        i32 sp[4];                                              // the sources' window positions
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ll[k] = 0; dd[k] = 0; sp[k] = 0; w[k] = 0; t[k] = 0;
            if (mask) {
                const u32 l = (u32)__ffsll((long long)mask) - 1;
                mask &= mask - 1;
                dd[k] = lane_get(d, l);
                ll[k] = lane_get(len, l);
                sp[k] = (i32)dd[k] - (i32)lane_get(dist, l);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (4 * lane + 4 <= ll[k]) __builtin_memcpy(&w[k], src.at(sp[k] + (i32)(4 * lane)), 4);
            if (lane < (ll[k] & 3)) t[k] = src.at(sp[k] + (i32)((ll[k] & ~3u) + lane))[0];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (4 * lane + 4 <= ll[k]) lds_store32(outw + dd[k] + 4 * lane, w[k]);
            if (lane < (ll[k] & 3)) outw[dd[k] + (ll[k] & ~3u) + lane] = (u8)t[k];
        }
    }
}

// The part of a short match's source that lies before the window (HBM/L2), first 16 bytes at most,
// as two 8-byte loads that every lane issues whether it has such a token or not: a fixed number of
// loads per round lets several rounds' loads be in flight before the first store needs them.
struct PreLoad {
    u64 a, b;       // n >= 8: bytes [0, 8) and [n - 8, n); n < 8: the bytes are a >> sh
    u32 tok, dpos;  // the token and where it goes in the window
    u32 n_sh;       // bytes fetched (0 = nothing to do) | sh << 8
    u32 g;          // bytes of the match that come from before the window (all of them, not just the first 16)
};
template <class SRC>
__device__ __forceinline__ PreLoad pre_load(const SRC &src, u32 woff, const TokList tl, u32 m, u32 M) {
    PreLoad p;
    p.tok = 0; p.dpos = 0; p.n_sh = 0; p.g = 0;
    i32 qa = src.lo(), qb = qa;                                 // a lane without work reads the chunk's first bytes
    if (m < M) {
        const u32 t = tl.tok[m];
        const u32 d = tl.tdst[m];
        p.tok = t; p.dpos = d;
        const u32 len = (t & 255) + 3, dist = (t >> 8) + 1;
        const i32 s = (i32)d - (i32)dist;
        if (len < (u32)LONG_MATCH && s < (i32)woff) {
            u32 g = (u32)((i32)woff - s);
            if (g > len) g = len;
            p.g = g;
            const u32 n = g < 16 ? g : 16;
            u32 sh = 0;                                         // (s >= lo(): dist <= history was checked by the emitter)
            if (n >= 8) { qa = s; qb = s + (i32)n - 8; }
            else {
                // 8 bytes that END with the wanted ones (never reading past them), clamped to the chunk's start
                qa = s + (i32)n - 8;
                if (qa < src.lo()) qa = src.lo();
                qb = qa;
                sh = (u32)(s - qa) * 8;
            }
            p.n_sh = n | (sh << 8);
        }
    }
    __builtin_memcpy(&p.a, src.at(qa), 8);
    __builtin_memcpy(&p.b, src.at(qb), 8);
    return p;
}
template <class SRC>
__device__ __forceinline__ void pre_store(u8 *outw, const SRC &src, const PreLoad &p) {
    u8 *dst = outw + p.dpos;
    const u32 n = p.n_sh & 255, sh = p.n_sh >> 8;
    if (n >= 8) {
        lds_store32(dst, (u32)p.a); no_merge();
        lds_store32(dst + 4, (u32)(p.a >> 32)); no_merge();
        lds_store32(dst + n - 8, (u32)p.b); no_merge();
        lds_store32(dst + n - 4, (u32)(p.b >> 32));
        if (p.g > 16) {                                         // the rest of a longer one, the plain way (rare)
            const i32 s = (i32)p.dpos - (i32)((p.tok >> 8) + 1);
            for (u32 off = 16; off < p.g;) {
                const u32 k = p.g - off < 16 ? p.g - off : 16;
                copy16_g2l(dst + off, src.at(s + (i32)off), k);
                off += k;
            }
        }
    } else if (n >= 4) {
        const u64 v = p.a >> sh;
        lds_store32(dst, (u32)v); no_merge();
        lds_store32(dst + n - 4, (u32)(v >> (8 * (n - 4))));
    } else if (n) {
        const u32 v = (u32)(p.a >> sh);
        const u32 mid = n > 1 ? 1u : 0u;
        dst[0] = (u8)v; dst[mid] = (u8)(v >> (8 * mid)); dst[n - 1] = (u8)(v >> (8 * (n - 1)));
    }
}
// After its bytes from before the window are in place: does the token still need bytes from inside the
// window (or is it a long one, which the wave copies whole)?  Those move up to the front of the list.
__device__ __forceinline__ u32 pre_compact(const TokList tl, const PreLoad &p, bool have, u32 K, u32 lane) {
    const u32 len = (p.tok & 255) + 3;
    const bool need = have && (len >= (u32)LONG_MATCH || p.g < len);
    const u64 nm = __ballot(need);
    if (need) {
        const u32 k = K + (u32)__popcll(nm & ((1ull << lane) - 1));
        tl.tok[k] = p.tok;
        tl.tdst[k] = (u16)p.dpos;
    }
    return K + (u32)__popcll(nm);
}

// Stage 2: resolve the M match tokens of one batch inside the LDS output window.
//   raw/base   the chunk's output so far: bytes [0, base) are final in HBM
//   outw       the batch's bytes [base, base + total) live at outw[0, total); literals are in place
// First sweep, 64 tokens per round and one per lane: source bytes before the window come from
// HBM/L2 and depend on nothing in the batch; four rounds' loads are in flight together.  Tokens
// that are complete after that leave the list; the others are compacted to its front (in
// stream order).  Second sweep over what is left, round-major: source bytes inside the window
// are final once they lie below the destination of the first unresolved token of the round
// (all earlier tokens are complete, and every literal is in place), so every pass of the
// inner loop retires at least that token.  A started copy never waits again: what it reads
// beyond its foreign source bytes it has written itself.  Matches of LONG_MATCH bytes and more
// are not copied by their lane: when such a token is the first unresolved one, the whole wave
// copies it (copy_long_match).
template <class SRC, class LDS>
__device__ void resolve_tokens(const SRC &src_, u32 cap, u32 woff, LDS &L, const TokList tl_, u32 M, u32 lane) {
    const SRC src = src_.known();                               // (address spaces spelled out: see in_lds)
    u8 *outw = in_lds(reinterpret_cast<u8 *>(L.win));       // positions count from the start of L.win
    TokList tl;
    tl.tok = in_lds(tl_.tok); tl.tdst = in_lds(tl_.tdst); tl.max = tl_.max;
    u32 rounds = 0;
    const bool swept = SRC::always_readable || cap >= 8;        // pre_load may read the chunk's first 8 bytes
    PROF_T0();
    if (swept) {
        u32 K = 0;
        for (u32 r0 = 0; r0 < M; r0 += 256) {
            // (rounds beyond the list's end are skipped whole: literal-heavy batches have fewer than 64 tokens)
            const bool h1 = r0 + 64 < M, h2 = r0 + 128 < M, h3 = r0 + 192 < M;     // wave-uniform
            PreLoad p1, p2, p3;
            p1.a = p1.b = 0; p1.tok = p1.dpos = p1.n_sh = p1.g = 0;
            p2 = p1; p3 = p1;
            const PreLoad p0 = pre_load(src, woff, tl, r0 + lane, M);
            if (h1) p1 = pre_load(src, woff, tl, r0 + 64 + lane, M);
            if (h2) p2 = pre_load(src, woff, tl, r0 + 128 + lane, M);
            if (h3) p3 = pre_load(src, woff, tl, r0 + 192 + lane, M);
            wave_sync();                                        // all four rounds' tokens are read before the list is compacted
            pre_store(outw, src, p0);
            K = pre_compact(tl, p0, r0 + lane < M, K, lane);
            if (h1) { pre_store(outw, src, p1); K = pre_compact(tl, p1, r0 + 64 + lane < M, K, lane); }
            if (h2) { pre_store(outw, src, p2); K = pre_compact(tl, p2, r0 + 128 + lane < M, K, lane); }
            if (h3) { pre_store(outw, src, p3); K = pre_compact(tl, p3, r0 + 192 + lane < M, K, lane); }
        }
        wave_sync();
        M = K;
    }
    PROF_ADD(PF_R_SWEEP, lane);
    for (u32 r0 = 0; r0 < M; r0 += 64) {
        const u32 m = r0 + lane;
        const bool have = m < M;
        u32 d = 0, len = 0, dist = 1;
        if (have) {
            const u32 t = tl.tok[m];
            d = tl.tdst[m];
            len = (t & 255) + 3;
            dist = (t >> 8) + 1;
        }
        const bool is_long = len >= (u32)LONG_MATCH;
        u32 cur = d, rem = len;
        const i32 s = (i32)d - (i32)dist;                       // source start, window position
        // ---- bytes before the window: in place already, unless the sweep was skipped
        if (have && !is_long && s < (i32)woff) {
            u32 g = (u32)((i32)woff - s);
            if (g > rem) g = rem;
            if (!swept) {
                for (u32 off = 0; off < g;) {                   // (s >= lo(): dist <= history was checked by the emitter)
                    const u32 n = g - off < 16 ? g - off : 16;
                    copy16_g2l(outw + cur + off, src.at(s + (i32)off), n);
                    off += n;
                }
            }
            cur += g; rem -= g;
        }
        // ---- long matches that depend on nothing in the batch, four at a time
        const bool indep = have && is_long && dist >= len && s + (i32)len <= (i32)woff;
        {
            const u64 im = __ballot(indep);
            if (im) {
                PROF_CNT(PF_N_INDEP, (u32)__popcll(im), lane);
                copy_independent_long(outw, src, im, d, len, dist, lane);
                wave_sync();
            }
        }
        // ---- bytes inside the window
        const i32 e = s + (i32)(len < dist ? len : dist);       // end of the source bytes other tokens may own
        bool pending = rem > 0 && !indep;
        u32 span = dist;
#if XFG_RESOLVE_EXACT
        // Which tokens of the round can write into this one's source bytes [s, e): destinations are disjoint and in
        // stream order over the lanes, so they are the lanes from the first whose destination ends after s to the last
        // whose destination starts before e (both below this lane).  Two binary searches over the lanes' values; a
        // token is ready when none of those lanes is pending -- not only when everything before its source is done,
        // which on match-heavy data took 2 to 2.5 times as many passes of the loop below.
        u64 blockers = 0;
        {
            const u32 dst_lo = have ? d : 0x7fffffffu, dst_hi = have ? d + len : 0x7fffffffu;
            u32 n_end = 0, n_beg = 0;                           // lanes whose destination ends at or before s / starts before e
#pragma unroll
            for (u32 step = 32; step >= 1; step >>= 1) {
                const i32 pe = __builtin_amdgcn_ds_bpermute((int)((n_end + step - 1) << 2), (int)dst_hi);
                const i32 pb = __builtin_amdgcn_ds_bpermute((int)((n_beg + step - 1) << 2), (int)dst_lo);
                if (pe <= s) n_end += step;
                if (pb < e) n_beg += step;
            }
            if (have && n_beg > n_end) blockers = (~0ull >> (64 - n_beg)) & ~((1ull << n_end) - 1);   // lanes n_end .. n_beg - 1
        }
#endif
        for (;;) {
            const u64 pm = __ballot(pending);
            if (pm == 0) break;
            rounds++;
            const u32 first = (u32)__ffsll((long long)pm) - 1;
            const i32 water = (i32)lane_get(d, first);          // everything below is final
            if (lane_get((u32)is_long, first)) {                 // wave-uniform
#ifdef XFG_PROFILE
                const unsigned long long tl0_ = __builtin_amdgcn_s_memtime();
#endif
                copy_long_match(outw, src, woff, (u32)water, lane_get(len, first), lane_get(dist, first), lane);
                if (lane == first) pending = false;
                wave_sync();
#ifdef XFG_PROFILE
                {
                    const u64 rdy = __ballot(pending && is_long && (pm & blockers) == 0);   // other long matches that could go now
                    if (lane == 0) L.prof[PF_N_LONG_RDY] += (unsigned long long)__popcll(rdy);
                }
                if (lane == 0) {
                    L.prof[PF_T_LONG] += __builtin_amdgcn_s_memtime() - tl0_;
                    L.prof[PF_N_LONG] += 1;
                    if (water - (i32)lane_get(dist, first) < (i32)woff) L.prof[PF_N_LONG_G] += 1;
                }
#endif
                continue;
            }
#if XFG_RESOLVE_EXACT
            if (pending && !is_long && (pm & blockers) == 0) {
#else
            if (pending && !is_long && e <= water) {
#endif
                while (rem) {
                    u32 n = span < rem ? span : rem;
                    if (n > 16) n = 16;
                    copy16_l2l(outw + cur, outw + cur - span, n);
                    cur += n; rem -= n;
                    if (n == span) span *= 2;                   // [cur - 2 span, cur) now has period dist
                }
                pending = false;
            }
            wave_sync();
        }
        wave_sync();
    }
    PROF_CNT(PF_N_ROUNDS, rounds, lane);
    (void)rounds;
}

// Flush window positions [woff, woff + n) to raw[base, base + n): woff is the HBM address mod 16,
// so the 16-byte loads from LDS and the 16-byte stores to HBM are both aligned.
__device__ void flush_output(u8 *raw, u32 base, const u8 *win0, u32 woff, u32 n, u32 lane) {
    u8 *g = raw + base;
    const u8 *w = win0 + woff;
    u32 head = (16 - woff) & 15;
    if (head > n) head = n;
    if (lane < head) g[lane] = w[lane];
    const u32 body = (n - head) >> 4;
    for (u32 i = lane; i < body; i += 64)
        *reinterpret_cast<uint4 *>(g + head + i * 16) = *reinterpret_cast<const uint4 *>(w + head + i * 16);
    const u32 done = head + body * 16;
    if (done + lane < n) g[done + lane] = w[done + lane];
}

struct BlockEnd {
    int status;         // XFG_ST_OK = end-of-block symbol reached, anything else ends the chunk
    u32 bitpos;         // consumed bits of the stream
};

// Decode one Huffman block that starts at bit position p0.  Output goes to o.raw at o.pos.
// Segment size for the next batch from the last one's expansion (output bytes per compressed byte).
__device__ __forceinline__ u32 seg_bits_for(u32 out_bytes, u32 in_bits) {
#if XFG_SEG_BITS
    return XFG_SEG_BITS;
#else
    const u64 out_x4 = (u64)out_bytes * 32;                     // output bits times four (a whole chunk's sizes come through here too)
    if (out_x4 < (u64)XFG_EXP_LONG_X4 * in_bits) return XFG_SEG_LONG;          // below 1.5: literals dominate
    if (out_x4 < (u64)XFG_EXP_SHORT_X4 * in_bits) return XFG_SEG_MID;
    if (out_x4 < (u64)XFG_EXP_RUN_X4 * in_bits) return XFG_SEG_SHORT;          // 4 and more: matches dominate
    return XFG_SEG_RUN;                                         // 32 and more: a few tokens fill the window
#endif
}

// (WG = true: called by wave 0 of the workgroup-per-chunk kernel for the stretches its waves do not share -- batches
// near the end of the stream, blocks without a distance code, serial batches, anything that ended a group with an
// error: decodes at least one batch and returns XFG_ST_AGAIN at the first batch that is an ordinary parallel one.)
#ifndef XFG_WG_KEEP_SHORT
#define XFG_WG_KEEP_SHORT 1             // match-heavy batches (short segments) stay with wave 0 as well: repeats kind 65 -> 77 GB/s at 768 chunks
#endif
#ifndef XFG_WG_SERIAL
#define XFG_WG_SERIAL 1                 // serial batches are shared out over the workgroup's waves (wg_serial_group)
#endif
constexpr int XFG_ST_AGAIN = -100;                              // never leaves the kernel
// Segments of XFG_SEG_RUN bits follow a batch whose output was nearly all long matches or runs (seg_bits_for).
__device__ __forceinline__ bool wg_keeps(u32 seg_bits) {
    return !XFG_SEG_BITS && (u32)XFG_SEG_RUN != (u32)XFG_SEG_LONG && (u32)XFG_SEG_RUN != (u32)XFG_SEG_MID && (u32)XFG_SEG_RUN != (u32)XFG_SEG_SHORT &&
           (seg_bits == (u32)XFG_SEG_RUN || (XFG_WG_KEEP_SHORT && seg_bits == (u32)XFG_SEG_SHORT));
}
template <bool WG = false, class LDS>
__device__ BlockEnd decode_huffman_block(const Reader &r, Out &o, LDS &L, u32 lane, u32 p0,
                                         int lit_state, int dist_state, u32 spec_back, u32 max_fix, u32 &seg_bits,
                                         bool serial_init = false, bool *serial_out = nullptr)
{
    BlockEnd be;
    if (lit_state == 1) { be.status = XFG_ST_CORRUPTED; be.bitpos = p0; return be; }   // empty literal tree
    const u32 total_bits = r.total * 8;
    const bool dist_empty = dist_state == 1;
    const i64 stream0 = (i64)(r.base_al + r.lead);              // address of stream byte 0, relative to comp
    uint4 pf[NPF];
    bool have_pf = false, force_tail = false;
    u32 want = 64;                                              // lanes worth synchronising: what the last batch could keep, and some
    bool serial = WG && serial_init;                            // output-bound stretch (long matches): lane 0 decodes alone
    for (u32 nbatch = 0;; nbatch++) {
        if (nbatch > total_bits + 4) { be.status = XFG_ST_INTERNAL; be.bitpos = p0; return be; }   // every batch consumes >= 1 bit
        // (batches of runs and long matches stay with wave 0: their write-out is all there is to them, and a group's write-outs are serial)
        // (and serial batches go back to the workgroup, whose waves take them in turns: wg_serial_group)
        if (WG && nbatch > 0 && !dist_empty && !force_tail && p0 + 64u <= total_bits && (serial ? XFG_WG_SERIAL != 0 : !wg_keeps(seg_bits))) {
            if (serial_out) *serial_out = serial;
            be.status = XFG_ST_AGAIN; be.bitpos = p0; return be;
        }
        // ---- compressed window of this batch (fetched during the previous batch's stage 2 when there was one)
        const i32 wb0 = (i32)(((stream0 + (p0 >> 3)) & ~(i64)15) - stream0);
        {
            PROF_T0();
            if (!have_pf) fetch_batch_window(r, wb0, lane, pf);
            wave_sync();
            store_batch_window(L, lane, pf);
            wave_sync();
            PROF_ADD(PF_WINDOW, lane);
        }
        have_pf = false;
        PROF_CNT(PF_N_BATCH, 1, lane);
        PROF_T0();
        // wave-uniform: blocks without a distance code, and batches in which a token ran past the end of
        // the stream (seen after the fact, below), go the careful way
        const bool tail = dist_empty || force_tail || p0 + 64u > total_bits;
        force_tail = false;

        const u32 woff = (u32)((uintptr_t)(o.raw + o.pos) & 15);  // window position of the batch's first byte
        u32 err_out = 0;
        bool emitted = false;
        u32 start = p0 + lane * seg_bits;
        const u32 seg_start = start;
        u32 seg_end = p0 + (lane + 1) * seg_bits;
        const u32 pos0 = o.pos;
        const TokList tl = token_list(L, seg_bits);
        bool dirty = true, dead = false;
        Seg sg;
        sg.end = start; sg.nout = 0; sg.nmatch = 0; sg.flag = F_NONE;
        u32 dummy = 0;
        u32 verified = 64;
        if (!tail && serial) {
            // ---- a few tokens fill the whole output window (long matches): speculation would be thrown
            // away, so lane 0 walks the batch's bits alone until the window is full
            seg_end = p0 + 2 * tl.max;                          // a match token is >= 2 bits: the list cannot overflow
            dead = lane != 0;
            start = p0;
            // (alone, its offsets are known: this walk is the emit pass already)
            if (!dead) sg = decode_segment_fast<true>(L, tl, wb0, start, seg_end, OUT_POS + woff, o.pos, 0, lane, err_out);
            emitted = true;
            wave_sync();
            PROF_CNT(PF_N_PASS, 1, lane);
            verified = 1;
            const u32 e0 = lane_get(sg.end, 0);
            if (e0 > total_bits) { force_tail = true; have_pf = false; continue; }
            if (lane_get(sg.nout, 0) < (u32)OUT_CAP && lane_get(sg.flag, 0) == F_NONE) serial = false;   // the window did not fill: less compressible again
        } else if (!tail) {
            // ---- stage 1a: find every lane's true start.  One speculative walk per lane from its segment
            // boundary, then walks from the predecessor's end that stop as soon as they meet the lane's
            // own speculative path (prefix codes self-synchronise), until the chain of starts is stable.
            // (where the first walk from corrected starts counts already -- literal-heavy batches -- nobody ever reads the
            // marks: the speculative walk leaves them out, five VALU and an LDS atomic per step less)
            const u32 count_from = max_fix > (u32)MAX_FIX ? (u32)XFG_COUNT_FROM_FIXED :
                                   seg_bits == (u32)XFG_SEG_LONG ? (u32)XFG_COUNT_FROM_LONG :
                                   seg_bits == (u32)XFG_SEG_RUN ? (u32)XFG_COUNT_FROM_RUN : (u32)XFG_COUNT_FROM;
            const bool marks = count_from > 1;
            if (marks) {
#pragma unroll
                for (int w = 0; w < MASK_WORDS; w++) L.mask[w * 64 + lane] = 0;
            }
            u32 eA, e1;
            bool mg;
            u32 endA;
            {
                PROF_T0();
                // (dynamic blocks: a run-in for the short segments of match-heavy stretches, where half of the guessed paths
                // would not meet the true one before their segment ends)
                const u32 back = spec_back ? spec_back : seg_bits <= (u32)XFG_SEG_SHORT ? (u32)XFG_SPEC_BACK_SHORT :
                                 seg_bits == (u32)XFG_SEG_MID ? (u32)XFG_SPEC_BACK_MID :
                                 seg_bits == (u32)XFG_SEG_LONG ? (u32)XFG_SPEC_BACK_LONG : 0u;
                if (back) {
                    // the walk starts before its segment: by the segment's first bit it has usually met the true
                    // path.  The run-in never reaches back beyond p0, a true start.
                    start = seg_start - p0 > back ? seg_start - back : p0;
#if XFG_SPEC_WALK_ON
                    // A walk from a guessed position often runs into a code that ends it: with the fixed codes the end of
                    // block is seven zero bits, and a third of all guessed starts meet that pattern (or 286 / 287 / a
                    // distance of 30 / 31) before they meet the true path, and then say nothing.  A guessed path walks on
                    // over such codes (measured on fixed-code text: synchronised after 177 bits on average, 549 at the 95th
                    // percentile, against never for one start in three).  If the true path goes through the same code, the
                    // count pass below finds the end of the block there and the chain is cut at that lane.  Lane 0 starts
                    // at p0, a true start, and stops where its path stops.
                    endA = sync_segment<false, true>(L, wb0, start, seg_start, seg_bits, lane, eA, mg, lane == 0 ? ~0u : 0u);
#else
                    endA = sync_segment<false, true>(L, wb0, start, seg_start, seg_bits, lane, eA, mg);
#endif
                } else {
#if XFG_SPEC_WALK_ON_DYNAMIC
                    if (marks) endA = sync_segment<false, false>(L, wb0, start, seg_start, seg_bits, lane, eA, mg, lane == 0 ? ~0u : 0u);
                    else       endA = sync_segment<false, false, false>(L, wb0, start, seg_start, seg_bits, lane, eA, mg, lane == 0 ? ~0u : 0u);
#else
                    endA = sync_segment<false, false>(L, wb0, start, seg_start, seg_bits, lane, eA, mg);
#endif
                }
                PROF_ADD(PF_P_SPEC, lane);
            }
            const u32 stopA = eA & E_STOP;
            u32 end = endA, stop = stopA;
            PROF_CNT(PF_N_PASS, 1, lane);
            // The first XFG_COUNT_FROM - 1 walks from corrected starts are merge walks (sync_segment<true>: many lanes, most
            // of them done after a few tokens).  From then on a lane whose start has moved walks its whole path and COUNTS
            // (bytes and matches: stage 1a'), and so does, in the first such round, every lane that has not moved: the later
            // re-synchronisation walks have a handful of lanes and last as long as a whole segment anyway, so the count
            // pass that used to follow them rides along (round 3: one pass per batch less).
            bool counted = false;                               // sg belongs to this lane's (start, dead)
            // (literal-heavy batches, long segments: one lane in twenty fails to merge, so the first walk from corrected starts
            // counts already; fixed blocks, where a walk adds two lanes to the chain: merge walks only, and one count pass)
            for (int pass = 1;; pass++) {
                const u32 pe = lane_prev(end);
                const u32 pst = lane_prev(stop);
                const u32 pd = lane_prev((u32)dead);
                u32 ns = p0;
                bool nd = false;
                if (lane > 0) { nd = pd != 0 || pst != 0; ns = pe; }
                dirty = (nd != dead) || (!nd && ns != start);
                start = ns; dead = nd;
                if (dirty) counted = false;
                const bool cround = (u32)pass >= count_from;
                const u64 dm = __ballot(dirty);
                const u64 needm = __ballot(!dead && !counted);
                if (dm == 0 && (!cround || needm == 0)) { verified = 64; break; }
                const u32 fd = dm ? (u32)__ffsll((long long)dm) - 1 : 64u;   // lanes below fd are chained from p0
                if (dm && ((u32)pass > max_fix || fd >= want || (pass > ACCEPT_PASS && fd >= (u32)XFG_ACCEPT_FD))) { verified = fd; break; }
                if (!cround) {
                    PROF_T0();
#ifdef XFG_PROFILE
                    { const u32 nwalk_ = (u32)__popcll(__ballot(dirty && !dead)); PROF_CNT(PF_N_DIRTY1 + (pass > 4 ? 3 : pass - 1), nwalk_, lane); }
#endif
                    if (dirty && !dead) {
                        const u32 en = sync_segment<true, false>(L, wb0, start, seg_start, seg_bits, lane, e1, mg);
                        end = mg ? endA : en;
                        stop = mg ? stopA : (e1 & E_STOP);
                    }
                    PROF_ADD(PF_P_FIX, lane);
                    PROF_CNT(PF_N_FIX, 1, lane);
                } else {
                    PROF_T0();
#ifdef XFG_PROFILE
                    { const u32 nwalk_ = (u32)__popcll(needm); PROF_CNT(PF_N_DIRTY1 + (pass > 4 ? 3 : pass - 1), nwalk_, lane); }
#endif
                    if (!dead && !counted) {
                        sg = decode_segment_fast<false>(L, tl, wb0, start, seg_end, 0, 0, 0, lane, dummy);
                        // A lane that stopped at its output cap hands its end to its successor only if it got through
                        // most of its segment: where a few tokens fill the cap (runs), every round would add one lane
                        // to the chain; there the chain is cut behind this lane (the closure test below), as before.
                        if (sg.flag != F_NONE || sg.end + seg_bits / 2 >= seg_end) {
                            end = sg.end;
                            stop = sg.flag != F_NONE ? 1u : 0u;
                        }
                        counted = true;
                    }
                    PROF_ADD(PF_P_COUNT, lane);
                    PROF_CNT(PF_N_FIX, 1, lane);
                }
                PROF_CNT(PF_N_PASS, 1, lane);
            }
            wave_sync();                                        // the masks share LDS with the output window
            // ---- stage 1a': count bytes and matches along the true path, for the lanes that have not counted yet
            if (__ballot(!dead && lane < verified && !counted)) {
                PROF_T0();
                if (!dead && lane < verified && !counted) sg = decode_segment_fast<false>(L, tl, wb0, start, seg_end, 0, 0, 0, lane, dummy);
                PROF_ADD(PF_P_COUNT, lane);
                PROF_CNT(PF_N_PASS, 1, lane);
            }
            // Whatever the walks above concluded, only a chain that closes is kept: every lane must start
            // where its predecessor's counted path ended (a lane that stopped at the output cap ends early).
            const u32 pe2 = lane_prev(sg.end);
            const u32 pf2 = lane_prev(sg.flag);
            const bool okc = !dead && lane < verified && (lane == 0 || (pe2 == start && pf2 == F_NONE));
            const u64 bad = ~__ballot(okc);
            PROF_CNT(PF_C_FIX_LOST, 64 - verified, lane);       // the chain of starts was not settled beyond this lane (or it ended)
            {
                const u32 v1 = verified;
                verified = bad ? (u32)__ffsll((long long)bad) - 1 : 64u;
                if (verified < v1 && verified > 0) {
                    const bool flagged = lane_get(sg.flag, verified - 1) != F_NONE;      // the lane before the cut ended the block ...
                    (void)flagged;
                    PROF_CNT(flagged ? PF_C_FLAG_CUTS : PF_C_CAP_CUTS, 1, lane);         // ... or stopped at its output cap
                    PROF_CNT(flagged ? PF_C_FLAG_LOST : PF_C_CAP_LOST, v1 - verified, lane);
                }
            }
            (void)bad;
            // a token that runs past the end of the stream: decode this batch again with the end-of-input checks
            if (__ballot(lane < verified && sg.end > total_bits)) { force_tail = true; have_pf = false; continue; }
        } else {
            // ---- stage 1a (careful): speculative passes until the chain of segment starts is stable
            for (int pass = 1;; pass++) {
                if (dirty && !dead)
                    sg = decode_segment_tail<false>(L, tl, wb0, start, seg_end, total_bits, dist_empty, 0, 0, 0, dummy);
                PROF_CNT(PF_N_PASS, 1, lane);
                const u32 pe = lane_prev(sg.end);
                const u32 pfl = lane_prev(sg.flag);
                const u32 pd = lane_prev((u32)dead);
                u32 ns = p0;
                bool nd = false;
                if (lane > 0) { nd = pd != 0 || pfl != F_NONE; ns = pe; }
                dirty = (nd != dead) || (!nd && ns != start);
                start = ns; dead = nd;
                if (dead) { sg.end = 0; sg.nout = 0; sg.nmatch = 0; sg.flag = F_NONE; }
                const u64 dm = __ballot(dirty);
                if (dm == 0) { verified = 64; break; }
                const u32 fd = (u32)__ffsll((long long)dm) - 1;       // lanes below fd are chained from p0
                if ((pass >= 2 && fd >= (u32)XFG_ACCEPT_FD) || pass >= MAX_PASSES) { verified = fd; break; }
            }
        }
        PROF_ADD(PF_COUNT, lane);
        const bool live = !dead && lane < verified;              // live lanes are 0 .. nlive-1

        // ---- output and token offsets.  The live lanes are emitted, resolved and flushed one window-full at a
        // time: each round takes the longest run of lanes, from the first one not yet written, that fits the
        // output window and the token list.  The walks above are done once for all of them.
        u32 tot_o, tot_m;
        const u32 rel0 = wave_excl_scan(live ? sg.nout : 0, tot_o);
        const u32 moff0 = wave_excl_scan(live ? sg.nmatch : 0, tot_m);
        const u32 nlive = (u32)__popcll(__ballot(live));        // >= 1: lane 0 is live
        u32 lflag = F_NONE, lend = p0;
        for (u32 b0 = 0;;) {
            const u32 rel = rel0 - lane_get(rel0, b0), moff = moff0 - lane_get(moff0, b0);   // (wraps below b0: those lanes are not kept)
            // The token list and the output window share the LDS behind the compressed window: a round of many short
            // matches moves the border up, a round of long ones down (six bytes per token, and the list's length is
            // rounded up to eight tokens so that the window stays 16-byte aligned).  One lane always fits:
            // <= SEG_MAX / 2 tokens and < OUT_LDS bytes.  (A batch that lane 0 has emitted already used the fixed border.)
            const u32 room = L.room_total - (8 * seg_bits + 64);
            const bool keep = live && lane >= b0 &&
                              (emitted ? rel + sg.nout <= (u32)OUT_LDS && moff + sg.nmatch <= tl.max
                                       : 6 * (moff + sg.nmatch + 7) + rel + sg.nout <= room);
            u32 nkeep = (u32)__popcll(__ballot(keep));          // >= 1
#if !XFG_SUBBATCH
            const bool final = true;
#else
            const bool final = b0 + nkeep == nlive;
#endif
            const u32 lastl = b0 + nkeep - 1;
            u32 total_out = lane_get(rel + sg.nout, lastl);
            u32 M = lane_get(moff + sg.nmatch, lastl);
            lflag = lane_get(sg.flag, lastl);
            lend = lane_get(sg.end, lastl);
            const u32 wo = (u32)((uintptr_t)(o.raw + o.pos) & 15);    // the round's first byte in the window: its HBM address mod 16
            const u32 tmax = emitted ? tl.max : (M + 7) & ~7u;
            const TokList tlr = token_list_n(L, seg_bits, tmax);
            const u32 ob = emitted ? OUT_POS : 8 * seg_bits + 64 + 6 * tmax;   // where the window starts, from L.win
            PROF_CNT(PF_N_LIVE, nkeep, lane);
            PROF_CNT(PF_N_WFULL, 1, lane);
            if (!final) {                                       // what kept the next lane out: the token list or the output window
                const bool by_tok = 6 * lane_get(moff + sg.nmatch, b0 + nkeep) > lane_get(rel + sg.nout, b0 + nkeep);
                (void)by_tok;
                PROF_CNT(by_tok ? PF_N_WF_TOK : PF_N_WF_OUT, 1, lane);
            }
            if (b0 == 0) {
#if XFG_SUBBATCH
                want = 64;
                // one lane's tokens fill the window: long matches
                if (!tail && nlive <= (u32)SERIAL_ENTER && lane_get(sg.flag, nlive - 1) == F_NONE &&
                    lane_get(sg.nout, nlive - 1) >= (u32)OUT_CAP) serial = true;
#else
                want = nkeep >= 48 ? 64u : 2 * nkeep + 8;           // output-bound batches keep few lanes: do not chase the rest
                // the window (or the token list) cut the batch short after very few lanes: long matches
                if (!tail && nkeep <= (u32)SERIAL_ENTER && lflag == F_NONE &&
                    (nkeep < nlive || lane_get(sg.nout, lastl) >= (u32)OUT_CAP)) serial = true;
#endif
            }

            // ---- the next batch's compressed window: on its way while the last round is emitted and resolved
            if (final && lflag == F_NONE && lend > p0) {
                const i32 nwb0 = (i32)(((stream0 + (lend >> 3)) & ~(i64)15) - stream0);
                fetch_batch_window(r, nwb0, lane, pf);
                have_pf = true;
            }

            // ---- stage 1b: final pass, literals into the output window, matches to the token list
            Seg fin = sg;
            {
                PROF_T0();
                if (keep && !emitted) {
                    if (tail) fin = decode_segment_tail<true>(L, tlr, wb0, start, seg_end, total_bits, dist_empty, ob + wo + rel, o.pos + rel, moff, err_out);
                    else      fin = decode_segment_fast<true>(L, tlr, wb0, start, seg_end, ob + wo + rel, o.pos + rel, moff, lane, err_out);
                }
                wave_sync();
                PROF_ADD(PF_EMIT, lane);
            }

            // A distance beyond the history ends the chunk at that token (flate/reader.go:223-225).
            const u64 em = __ballot(keep && fin.flag == F_DIST);
            u32 elane = 64;
            if (em) {
                elane = (u32)__ffsll((long long)em) - 1;
                total_out = lane_get(rel + err_out, elane);     // tokens after the error do not exist
                M = lane_get(moff + fin.nmatch, elane);
            }
            PROF_CNT(PF_N_TOKENS, M, lane);
            // ---- stage 2
            if (M) {
                PROF_T0();
                mem_sync();                                     // earlier rounds' flushed bytes are read back here
                PROF_ADD(PF_R_SYNC, lane);
                const SrcHbm hbm{o.raw + o.pos - (ob + wo), (i32)(ob + wo) - (i32)o.pos};
                resolve_tokens(hbm, o.cap, ob + wo, L, tlr, M, lane);
                PROF_ADD(PF_RESOLVE, lane);
            }
            // ---- flush to HBM (never beyond the chunk's capacity)
            {
                PROF_T0();
                u32 nflush = total_out;
                if (o.pos >= o.cap) nflush = 0; else if (nflush > o.cap - o.pos) nflush = o.cap - o.pos;
                wave_sync();
                flush_output(o.raw, o.pos, reinterpret_cast<const u8 *>(L.win), ob + wo, nflush, lane);
                wave_sync();
                PROF_ADD(PF_FLUSH, lane);
            }

            if (em) {
                be.status = XFG_ST_CORRUPTED;
                be.bitpos = lane_get(fin.end, elane);
                o.pos += total_out;
                if (o.pos > o.cap) { o.pos = o.cap; be.status = XFG_ST_OUT_FULL; }
                return be;
            }
            if (total_out > o.cap - o.pos) {                     // more output than the chunk may hold
                o.pos = o.cap;
                be.status = XFG_ST_OUT_FULL;
                be.bitpos = lend;
                return be;
            }
            o.pos += total_out;
            if (final) break;
            b0 += nkeep;
        }
        if (lflag == F_EOB) { be.status = XFG_ST_OK; be.bitpos = lend; return be; }
        if (lflag == F_CORRUPT) { be.status = XFG_ST_CORRUPTED; be.bitpos = lend; return be; }
        if (lflag == F_EOF) { be.status = XFG_ST_UNEXPECTED_EOF; be.bitpos = lend; return be; }
        if (lflag != F_NONE || lend <= p0) { be.status = XFG_ST_INTERNAL; be.bitpos = lend; return be; }
        if (!tail && !serial) seg_bits = seg_bits_for(o.pos - pos0, lend - p0);
        p0 = lend;                                                 // next batch
    }
}

// Inflate one chunk; returns through *out.  Everything here is wave-uniform control flow.
__device__ void inflate_chunk(const u8 *__restrict__ comp, u64 comp_len, const xfg_chunk *__restrict__ chunks, u32 ci,
                              u8 *__restrict__ raw, u64 raw_len, xfg_result *__restrict__ res,
                              WaveLds &L, u32 lane)
{
    PROF_T0();
    xfg_chunk c = chunks[ci];
    c.comp_off = uni64(c.comp_off);
    c.raw_off = uni64(c.raw_off);
    c.comp_size = uni64(c.comp_size);
    c.raw_size = uni64(c.raw_size);
    const u32 flags = uni(c.flags);
    // The table may come from an untrusted index (xfg_index_parse) and is never seen by the host on the
    // device-resident path: a chunk whose sizes exceed the limits, or whose ranges leave the buffers, is
    // refused here -- nothing read, nothing written.
    if (c.comp_size > (u64)XFG_MAX_CHUNK_COMP || c.raw_size > (u64)XFG_MAX_CHUNK_RAW ||
        c.comp_off > comp_len || c.comp_size > comp_len - c.comp_off ||
        c.raw_off > raw_len || c.raw_size > raw_len - c.raw_off || uni(c.reserved) != 0) {
        if (lane == 0) {
            xfg_result out;
            out.status = XFG_ST_INTERNAL;
            out.reserved = 0;
            out.in_consumed = 0;
            out.out_produced = 0;
            res[ci] = out;
        }
        return;
    }

    if (flags & XFG_CHUNK_HOST_META) {
        // An index or footer record that the host decoded when it parsed the index: nothing to inflate
        // (include/xflate_amd.h).  What inflating it yields: no output, every byte consumed (and the appended endBlock).
        if (lane == 0) {
            xfg_result out;
            const i64 consumed = (i64)c.comp_size + ((flags & XFG_CHUNK_FOOTER) ? 0 : ((flags & XFG_CHUNK_END_BLOCK) ? 5 : 0));
            out.status = ((flags & XFG_CHUNK_VERIFY) && c.raw_size != 0) ? XFG_ST_SIZE_MISMATCH : XFG_ST_OK;
            out.reserved = 0;
            out.in_consumed = consumed;
            out.out_produced = 0;
            res[ci] = out;
        }
        return;
    }

    Reader r;
    r.comp = comp;
    r.base_al = c.comp_off & ~(u64)511;
    r.lead = uni((u32)(c.comp_off & 511));
    r.csize = uni((u32)c.comp_size);
    r.total = r.csize + ((flags & XFG_CHUNK_END_BLOCK) ? 5u : 0u);
    r.bytepos = 0;
    r.blk_hi = -4;
    r.bitbuf = 0;
    r.bitcnt = 0;
    r.win = L.win;
    r.lane = lane;

    Out o;
    o.raw = raw + c.raw_off;
    o.cap = uni((u32)c.raw_size);
    o.pos = 0;

    const bool rfc_rep16 = (flags & XFG_CHUNK_RFC_REP16) != 0;
    // the first batch's segment size from the chunk's own sizes, where the index vouches for them
    u32 seg_bits = (flags & XFG_CHUNK_VERIFY) ? seg_bits_for(o.cap, r.csize * 8u) : (u32)XFG_SEG_MID;
    int status;
    int tables = 0;                                             // 0 none, 1 fixed, 2 dynamic

    for (u32 nblk = 0;; nblk++) {
        if (nblk > r.total * 8 + 4) { status = XFG_ST_INTERNAL; break; }   // every block consumes >= 3 bits
        // readBlockHeader (flate/reader.go:100-134)
        refill(r);
        if (avail(r) < 1) { status = XFG_ST_UNEXPECTED_EOF; break; }
        const u32 last = peek(r, 1);
        consume(r, 1);
        if (avail(r) < 2) { status = XFG_ST_UNEXPECTED_EOF; break; }
        const u32 btype = peek(r, 2);
        consume(r, 2);
        if (btype == 0) {
            consume(r, r.bitcnt & 7);                           // ReadPads
            if (avail(r) < 16) { status = XFG_ST_UNEXPECTED_EOF; break; }
            const u32 len = peek(r, 16);
            consume(r, 16);
            if (avail(r) < 16) { status = XFG_ST_UNEXPECTED_EOF; break; }
            const u32 nlen = peek(r, 16);
            consume(r, 16);
            if ((len ^ nlen) != 0xffff) { status = XFG_ST_CORRUPTED; break; }
            // stream position of the first data byte; drop the look-ahead bits
            const u32 sp = r.bytepos - (r.bitcnt >> 3);
            u32 take = len;
            bool eof = false, full = false;
            if (take > r.total - sp) { take = r.total - sp; eof = true; }
            if (take > o.cap - o.pos) { take = o.cap - o.pos; full = true; eof = false; }
            {
                PROF_T0();
                stored_copy(r, o, sp, take, lane);
#if XFG_STORED_SYNC
                mem_sync();
#endif
                // (no wait for the copy's stores: the one reader of earlier output, match resolution,
                // waits for this wave's stores itself)
                PROF_ADD(PF_STORED, lane);
            }
            r.bytepos = sp + take;
            r.bitbuf = 0;
            r.bitcnt = 0;
            if (full) { status = XFG_ST_OUT_FULL; break; }
            if (eof) { status = XFG_ST_UNEXPECTED_EOF; break; }
        } else if (btype == 3) {
            status = XFG_ST_CORRUPTED;                          // reserved block type
            break;
        } else {
            int lit_state = 0, dist_state = 0;
            if (btype == 1) {
                if (tables != 1) build_fixed(L, lane);
                tables = 1;
            } else {
                tables = 2;
                PROF_T0();
                status = read_dynamic_header(r, L, lane, rfc_rep16, lit_state, dist_state);
                PROF_ADD(PF_HEADER, lane);
                if (status != XFG_ST_OK) break;
            }
            PROF_CNT(PF_N_BLOCKS, 1, lane);
            const BlockEnd be = decode_huffman_block(r, o, L, lane, reader_bitpos(r), lit_state, dist_state,
                                                     btype == 1 ? (u32)XFG_SPEC_BACK_FIXED : (u32)XFG_SPEC_BACK,
                                                     btype == 1 ? (u32)XFG_MAX_FIX_FIXED : (u32)MAX_FIX, seg_bits);
            wave_sync();
            reader_seek(r, be.bitpos);                          // the window was reused by the block decoder
            status = be.status;
            if (status != XFG_ST_OK) break;
        }
        // finishBlock (flate/reader.go:253-259)
        if (last) {
            consume(r, r.bitcnt & 7);                           // ReadPads
            status = XFG_ST_OK;
            break;
        }
    }

    const i64 in_off = in_offset(r);
    // xflate.Reader's checks after a chunk's io.EOF (xflate/reader.go:186-202)
    if (status == XFG_ST_OK) {
        if (flags & XFG_CHUNK_SYNC) {
            bool ok = r.csize >= 4;
            if (ok) {
                const u8 *p = comp + c.comp_off + r.csize - 4;
                ok = p[0] == 0x00 && p[1] == 0x00 && p[2] == 0xff && p[3] == 0xff;
            }
            if (!ok) status = XFG_ST_NO_SYNC;
        }
        if (status == XFG_ST_OK && (flags & XFG_CHUNK_VERIFY)) {
            const i64 want_in = (i64)r.csize + ((flags & XFG_CHUNK_FOOTER) ? 0 : ((flags & XFG_CHUNK_END_BLOCK) ? 5 : 0));
            if (in_off != want_in || o.pos != o.cap) status = XFG_ST_SIZE_MISMATCH;
        }
    } else if (status == XFG_ST_OUT_FULL && (flags & XFG_CHUNK_VERIFY)) {
        status = XFG_ST_SIZE_MISMATCH;                          // more output than the index promised
    }
    PROF_ADD(PF_TOTAL, lane);
#ifdef XFG_PROFILE
    wave_sync();
    if (lane < PF_COUNTERS) { atomicAdd(&g_prof[lane], L.prof[lane]); L.prof[lane] = 0; }
    wave_sync();
#endif
    if (lane == 0) {
        xfg_result out;
        out.status = status;
        out.reserved = 0;
        out.in_consumed = in_off;
        out.out_produced = o.pos;
        res[ci] = out;
    }
}

// The next chunk of the table for this wave: one atomic per wave, the value broadcast to all lanes.
// Kept out of line so that the chunk loop below stays a plain loop with a wave-uniform trip count.
__device__ __attribute__((noinline)) u32 draw_chunk(u32 *counter)
{
    u32 v = 0;
    if ((threadIdx.x & 63) == 0) v = atomicAdd(counter, 1u);
    return uni(v);
}

// Persistent grid: every wave draws chunks from shared counters until the table is exhausted
// (n <= 2^30: a counter runs to 3 n and beyond by one per wave).
#ifndef XFG_DRAW_SLICES
#define XFG_DRAW_SLICES 32
#endif
constexpr int DRAW_SLICES = XFG_DRAW_SLICES;
#ifndef XFG_WAVES_PER_EU
#define XFG_WAVES_PER_EU 4            // waves per SIMD the register allocation aims at (the LDS footprint must allow as many)
#endif
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(XFG_WAVES_PER_EU)))
inflate_chunks_kernel(const u8 *__restrict__ comp, u64 comp_len, const xfg_chunk *__restrict__ chunks, u32 n,
                      u8 *__restrict__ raw, u64 raw_len, xfg_result *__restrict__ res,
                      u32 *__restrict__ scratch)
{
    __shared__ WaveLds lds[WAVES_PER_BLOCK];
    // (one wave per workgroup: every LDS address is then a compile-time constant, which saves an add or a move per
    // table lookup and a scalar register)
    const u32 wave = WAVES_PER_BLOCK == 1 ? 0u : uni(threadIdx.x >> 6);
    const u32 lane = threadIdx.x & 63;
    WaveLds &L = lds[wave];
#ifdef XFG_PROFILE
    if (lane < PF_COUNTERS) L.prof[lane] = 0;
#endif
    // Dynamic schedule: every wave of the persistent grid takes the next chunk of the table from a
    // counter in scratch[0] (zeroed by the host before each launch).  Chunks differ a lot in cost
    // (stored vs Huffman-heavy), so a static stride leaves waves idle at the end.  The trip count
    // is wave-uniform: lane 0 draws, everybody reads its value.
    // The counter runs three times over the table: chunks that compress least (the most bits to
    // decode per output byte: the expensive ones) are taken in the first round, the cheapest in the
    // last, so that the grid's last wave finishes on a short chunk.
    // One counter serves about 10^8 draws a second (they are memory-side atomics on one address); a grid
    // of cheap chunks (stored blocks) draws faster than that.  So the table is cut into DRAW_SLICES slices
    // with a counter each, 64 bytes apart; a wave starts on the slice its workgroup index names and moves
    // on to the next one when a slice is used up, which is also how the grid's tail is shared out.
    const u32 home = uni(blockIdx.x) % (u32)DRAW_SLICES;
    for (u32 t = 0; t < (u32)DRAW_SLICES; t++) {
        const u32 k = (home + t) % (u32)DRAW_SLICES;
        const u32 lo = (u32)(((u64)n * k) / DRAW_SLICES), hi = (u32)(((u64)n * (k + 1)) / DRAW_SLICES);
        const u32 m = hi - lo;
        if (m == 0) continue;
        u32 *counter = scratch + k * 16u;
        const u32 rounds_m = 3u * m;
        for (u32 v = draw_chunk(counter); v < rounds_m; v = draw_chunk(counter)) {
            const u32 round = v < m ? 0u : (v < 2u * m ? 1u : 2u);
            const u32 ci = lo + (v - round * m);
            const u64 cs = uni64(chunks[ci].comp_size), rs = uni64(chunks[ci].raw_size);
#ifndef XFG_CLASS_MERGE
#define XFG_CLASS_MERGE 0               // 1: the two expensive classes are drawn in one round, interleaved as the table has them (tuning: match-source over-fetch)
#endif
            u32 cls = cs * 8 >= rs * 5 ? 0u : (cs * 4 >= rs ? 1u : 2u);
            if (XFG_CLASS_MERGE && cls == 1) cls = 0;
            if (cls == round) inflate_chunk(comp, comp_len, chunks, ci, raw, raw_len, res, L, lane);
        }
    }
}

#include "inflate_wg_kernel.inc"

} // namespace xfg

#ifdef XFG_PROFILE
// Diagnostic build only: read and clear the in-kernel profile counters.
extern "C" int xfg_debug_profile(unsigned long long *out, int n)
{
    unsigned long long h[xfg::PF_COUNTERS];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(xfg::g_prof), sizeof h) != hipSuccess) return -1;
    for (int i = 0; i < n && i < xfg::PF_COUNTERS; i++) out[i] = h[i];
    unsigned long long z[xfg::PF_COUNTERS] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(xfg::g_prof), z, sizeof z) != hipSuccess) return -1;
    return xfg::PF_COUNTERS;
}
#endif

static int blocks_per_cu()
{
    static const int cached = [] {                              // initialised once, thread-safe (C++11)
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, xfg::inflate_chunks_kernel, 64 * xfg::WAVES_PER_BLOCK, 0) != hipSuccess || nb < 1)
            nb = 1;
        if (nb > xfg::MAX_BLOCKS_PER_CU) nb = xfg::MAX_BLOCKS_PER_CU;
        return nb;
    }();
    return cached;
}

extern "C" uint32_t xfg_grid_blocks(uint32_t n, int num_cu)
{
    // Persistent grid: as many single-wave workgroups as the CUs can hold (LDS-bound).
    uint32_t want = (n + xfg::WAVES_PER_BLOCK - 1) / xfg::WAVES_PER_BLOCK;
    uint32_t cap = (uint32_t)num_cu * (uint32_t)blocks_per_cu();
    return want < cap ? want : cap;
}

extern "C" size_t xfg_scratch_bytes(int num_cu)
{
    (void)num_cu;
    return 64 * xfg::DRAW_SLICES;                               // the chunk counters; the token lists live in LDS, nothing is staged in HBM
}

// ---------------------------------------------------------------------------
// xfg_decode_device_noalloc (include/xflate_amd.h): the same launch for callers that cannot allocate.
// The only scratch buffer of the decode path is the chunk counters of the persistent grid (64 bytes per table
// slice, 2 KB); here they live in a __device__ array of this code object instead of the context's hipMalloc'd
// block.  Every GPU the library runs on holds its own copy of the array at its own address: the address is looked
// up once per device, with that device current, and only that pointer -- never the array's name, which in host
// code is a stand-in and no device address -- is handed to hipMemsetAsync and to the kernel.
// ---------------------------------------------------------------------------
namespace xfg {
static __device__ __align__(256) unsigned char scratch[64 * DRAW_SLICES];   // (no attribute "used": with it clang keeps the symbol local and the runtime cannot find it)
}

namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
struct NoAllocDevice {
    void *ptr;          // device address of xfg::scratch on this device
    int num_cu;
};
NoAllocDevice g_noalloc[NOALLOC_MAX_DEVICES];
bool g_noalloc_done[NOALLOC_MAX_DEVICES];                       // guarded by g_noalloc_mutex
hipError_t g_noalloc_err[NOALLOC_MAX_DEVICES];
std::mutex g_noalloc_mutex;

// once per device, with that device current
void noalloc_setup(int device)
{
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(xfg::scratch));
    int ncu = 0;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
    if (e == hipSuccess && (p == nullptr || ncu <= 0)) e = hipErrorInvalidValue;
    g_noalloc[device].ptr = p;
    g_noalloc[device].num_cu = ncu;
    g_noalloc_err[device] = e;
}
}

extern "C" int xfg_decode_device_noalloc(int device, const void *d_comp, size_t comp_len,
                                         const void *d_chunks, size_t n,
                                         void *d_raw, size_t raw_len, void *d_res, void *hip_stream)
{
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) { xfg_set_last_error("xfg_decode_device_noalloc: device index out of range"); return XFG_E_ARG; }
    if (n > (1ull << 30)) {
        xfg_set_last_error("xfg_decode_device_noalloc: more than 2^30 chunks in one call (the chunk counters count to 3 n in 32 bits)");
        return XFG_E_LIMIT;
    }
    if (n != 0 && (!d_comp || !d_chunks || !d_res || (!d_raw && raw_len) || ((uintptr_t)d_comp & 15) != 0)) {
        xfg_set_last_error("xfg_decode_device_noalloc: null or misaligned device pointer");
        return XFG_E_ARG;
    }
    // the caller's current device is put back on every way out
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) {
        const hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { xfg_set_last_error(hipGetErrorString(e)); return XFG_E_NODEVICE; }
    }
    int rc = XFG_OK;
    {
        std::lock_guard<std::mutex> lock(g_noalloc_mutex);
        if (!g_noalloc_done[device]) { noalloc_setup(device); g_noalloc_done[device] = true; }
    }
    if (g_noalloc_err[device] != hipSuccess) {
        xfg_set_last_error(hipGetErrorString(g_noalloc_err[device]));
        rc = XFG_E_HIP;
    } else if (n != 0) {
        void *ptr = g_noalloc[device].ptr;                      // the device address, never the symbol
        const hipError_t e = xfg_launch_inflate(d_comp, comp_len, d_chunks, (uint32_t)n, d_raw, raw_len, d_res, ptr,
                                                g_noalloc[device].num_cu, (hipStream_t)hip_stream);
        if (e != hipSuccess) { xfg_set_last_error(hipGetErrorString(e)); rc = XFG_E_HIP; }
    }
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    return rc;
}

// ---------------------------------------------------------------------------
// One wave per chunk, or a workgroup per chunk (inflate_wg_kernel.inc)?  A table with fewer chunks than half the
// chip's wave slots is decoded by workgroups of slots / n waves (at most 16): the same bytes, statuses and offsets,
// sooner.  xfg_set_decode_waves() / XFG_DECODE_WAVES pin the choice (tests, measurements).
// ---------------------------------------------------------------------------
namespace {
std::atomic<int> g_decode_waves{-1};                            // -1: not set yet (the environment decides), 0: by chunk count
int decode_waves_setting()
{
    int w = g_decode_waves.load(std::memory_order_relaxed);
    if (w < 0) {
        const char *e = getenv("XFG_DECODE_WAVES");
        w = e && *e ? atoi(e) : 0;
        if (w < 0 || w > xfg::WG_MAX_WAVES) w = 0;
        g_decode_waves.store(w, std::memory_order_relaxed);
    }
    return w;
}
// workgroups of `waves` waves that one CU holds (LDS- and register-bound), per wave count; 0 = not asked yet
std::atomic<int> g_wg_per_cu[2][xfg::WG_MAX_WAVES + 1];
// 9 to 12 waves per chunk are one workgroup per CU, three waves per SIMD: the kernel compiled for that register budget
#ifndef XFG_WG_WIDE
#define XFG_WG_WIDE 1
#endif
// ... and so are fewer waves per chunk when all the launch's workgroups fit the chip at three waves per SIMD (768 chunks
// of four waves: 12 waves on each of 256 CUs): the 128-register build spills inside its group loop
#ifndef XFG_WG_WIDE_SMALL
#define XFG_WG_WIDE_SMALL 1
#endif
bool wg_wide(int waves, uint32_t n, int num_cu)
{
    if (!XFG_WG_WIDE || waves > xfg::WG_WIDE_WAVES) return false;
    if (waves > 8) return true;
    return XFG_WG_WIDE_SMALL && (uint64_t)n * (uint64_t)waves <= (uint64_t)num_cu * (12 / waves) * waves;
}
int wg_blocks_per_cu(int waves, bool wide)
{
    int nb = g_wg_per_cu[wide ? 1 : 0][waves].load(std::memory_order_relaxed);
    if (nb > 0) return nb;
    const size_t lds = xfg::wg_lds_bytes((uint32_t)waves, xfg::wg_extra_bytes((uint32_t)waves), xfg::wg_has_ring((uint32_t)waves));
    // more than 64 KB of dynamic LDS has to be asked for
    (void)hipFuncSetAttribute((const void *)xfg::inflate_chunks_wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xfg::WG_LDS_MAX);
    (void)hipFuncSetAttribute((const void *)xfg::inflate_chunks_wg12_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xfg::WG_LDS_MAX);
    const hipError_t oe = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, xfg::inflate_chunks_wg12_kernel, 64 * waves, lds)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, xfg::inflate_chunks_wg_kernel, 64 * waves, lds);
    if (oe != hipSuccess || nb < 1) nb = 1;
    g_wg_per_cu[wide ? 1 : 0][waves].store(nb, std::memory_order_relaxed);
    return nb;
}
}

extern "C" int xfg_set_decode_waves(int waves)
{
    if (waves < 0 || waves > xfg::WG_MAX_WAVES) return XFG_E_ARG;
    g_decode_waves.store(waves, std::memory_order_relaxed);
    return XFG_OK;
}

// waves per chunk for a table of n chunks (1 = inflate_chunks_kernel)
extern "C" int xfg_decode_waves_for(uint32_t n, int num_cu)
{
    const int forced = decode_waves_setting();
    if (forced) return forced;
    const uint64_t slots = (uint64_t)num_cu * (uint64_t)blocks_per_cu();     // waves the chip holds
    const uint64_t w = n ? slots / n : 1;
    // Measured (profiles/r3/wg_wave_sweep.txt): 12 waves (one workgroup per CU, the history ring in LDS) where the
    // chunks are few enough for that; otherwise 4 (four workgroups per CU) -- 5 to 8 waves per chunk lose to 4 on
    // every shape; 3 or 2 when there are more chunks than that.
    // Two waves per chunk still win with somewhat more chunks than wave slots (the workgroups that do not fit wait their
    // turn): 4096 chunks 144 -> 156 GB/s, 3072 chunks 126 -> 147; from about 1.5 chunks per slot on one wave per chunk
    // is faster (6144 chunks: 188 against 167; profiles/r3/wg_wave_policy.txt).
    if (w >= 12) return 12;
    if (w >= 4) return 4;
    if (w >= 2) return (int)w;
    return (uint64_t)n * 4 <= slots * 5 ? 2 : 1;
}

extern "C" hipError_t xfg_launch_inflate(const void *d_comp, size_t comp_len, const void *d_chunks, uint32_t n,
                                         void *d_raw, size_t raw_len, void *d_res, void *d_scratch,
                                         int num_cu, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_scratch, 0, 64 * xfg::DRAW_SLICES, stream);      // the chunk counters
    if (e != hipSuccess) return e;
    const int waves = xfg_decode_waves_for(n, num_cu);
    if (waves >= 2) {
        const bool wide = wg_wide(waves, n, num_cu);
        uint32_t blocks = (uint32_t)num_cu * (uint32_t)wg_blocks_per_cu(waves, wide);
        if (blocks > n) blocks = n;
        const uint32_t extra = xfg::wg_extra_bytes((uint32_t)waves);
        const bool ring = xfg::wg_has_ring((uint32_t)waves);
        if (wide)
            hipLaunchKernelGGL(xfg::inflate_chunks_wg12_kernel, dim3(blocks), dim3(64 * waves), xfg::wg_lds_bytes((uint32_t)waves, extra, ring), stream,
                               (const uint8_t *)d_comp, (uint64_t)comp_len, (const xfg_chunk *)d_chunks, n,
                               (uint8_t *)d_raw, (uint64_t)raw_len, (xfg_result *)d_res, (uint32_t *)d_scratch, extra, ring ? 1u : 0u);
        else
            hipLaunchKernelGGL(xfg::inflate_chunks_wg_kernel, dim3(blocks), dim3(64 * waves), xfg::wg_lds_bytes((uint32_t)waves, extra, ring), stream,
                               (const uint8_t *)d_comp, (uint64_t)comp_len, (const xfg_chunk *)d_chunks, n,
                               (uint8_t *)d_raw, (uint64_t)raw_len, (xfg_result *)d_res, (uint32_t *)d_scratch, extra, ring ? 1u : 0u);
        return hipGetLastError();
    }
    const uint32_t blocks = xfg_grid_blocks(n, num_cu);
    hipLaunchKernelGGL(xfg::inflate_chunks_kernel, dim3(blocks), dim3(64 * xfg::WAVES_PER_BLOCK), 0, stream,
                       (const uint8_t *)d_comp, (uint64_t)comp_len, (const xfg_chunk *)d_chunks, n,
                       (uint8_t *)d_raw, (uint64_t)raw_len, (xfg_result *)d_res, (uint32_t *)d_scratch);
    return hipGetLastError();
}
