// iss_deflate.hip.h -- gzip members built on the device (`--compress`, iss/util.py:255-268).
//
// The reference gzips the finished FASTQ files on the host (gzip.open + copyfileobj).  At the kernel's rate the FASTQ
// text is the bottleneck twice over (PCIe, then the file system), so with compression on the text never leaves the
// device: every batch of FASTQ text becomes one gzip member made of DEFLATE blocks with a dynamic Huffman code whose
// only matches are runs (distance 1) and copies of the previous record (distance = the batch's record length), both
// found inside 32-byte chunks, so there is no search and no dependency between lanes.  FASTQ is bases (2 bits of
// entropy), phreds (runs of the top quality) and headers; this reaches 4.0x on NovaSeq text (zlib level 1: 3.9x,
// level 6: 5x).  The decompressed bytes are exactly the text k_fastq_format wrote (the compressed bytes
// differ from the reference's, like any two gzip implementations' do).
//
// Per batch and mate:  k_deflate_hist (token histogram of the text) -> k_deflate_build (ONE lane: length-limited
// Huffman code, canonical codes, the dynamic-block header bits -- the same code any host would build, here without a
// round trip) -> k_deflate_len (bits and raw CRC-32 of every 32 KB block) -> k_deflate_scan (byte offsets) ->
// k_deflate_encode (bit packing through LDS).  Every block ends with an empty stored block (the "sync flush" of
// zlib), which byte-aligns the next one, so blocks are written independently.  The host adds the 10-byte member
// header, the final empty block, CRC-32 and ISIZE (RFC 1952), combining the per-block CRCs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iss {

constexpr int DEFLATE_SYMS = 273;      // literals 0..255, end of block, lengths 3..32 (codes 257..272)
constexpr int DEFLATE_CHUNK = 32;      // bytes of text a lane tokenizes on its own
constexpr int DEFLATE_NQ = DEFLATE_CHUNK / 8;
constexpr int DEFLATE_BLOCK = 32768;   // text bytes per DEFLATE block
constexpr int DEFLATE_HDR_WORDS = 64;  // room for the dynamic-block header (<= 3 + 14 + 57 + 258 * 7 + ... bits)
constexpr int DEFLATE_THREADS = 256;

struct DeflateCode {
    uint32_t entry[DEFLATE_SYMS];      // bit-reversed code | length << 16
    uint32_t hdr_bits;                 // length of the block header: BFINAL, BTYPE, HLIT, HDIST, HCLEN, both code tables
    uint32_t hdr[DEFLATE_HDR_WORDS];   // its bits, least significant first
    uint32_t crc_shift[8][32];         // GF(2) operators "append 128 << k zero bytes" for the in-block CRC tree
};

// ---------------------------------------------------------------- code construction (host and device)
// Huffman code lengths of n <= 288 symbols, limited to maxbits; symbols with count 0 get length 0.  The code is
// complete (Kraft sum exactly 1), as inflate requires, whenever at least two symbols are in use.
struct DeflateWork {  // scratch of the code builder: in LDS on the device (private arrays would live in scratch memory)
    uint64_t heap[2 * 288];
    int16_t parent[2 * 288];
    uint32_t cnt[288];
    uint8_t len[320], sym[320], extra[320];
    uint16_t code[288];
    uint32_t prop[64], ccnt[19], ns;   // per-lane proposals of a repair step; code-length alphabet counts; RLE symbols
    uint8_t clen[19];
    uint16_t ccode[19];
    uint32_t entry[288];               // results, copied to the DeflateCode by the caller (all lanes on the device)
    uint32_t hdr[64];
    uint32_t hdr_bits;
};

// The builder is written for `nl` cooperating lanes (lane = 0 .. nl-1; `sync` separates the phases): the host runs it
// with one, the device with one wavefront.  Every choice is made on a total order (weight, then symbol / node number),
// so the result does not depend on nl.
struct DeflateNoSync { __host__ __device__ void operator()() const {} };

// Huffman code lengths of n <= 288 symbols, limited to maxbits; symbols with count 0 get length 0.  The code is
// complete (Kraft sum exactly 1), as inflate requires, whenever at least two symbols are in use.
//   1. keys (count << 16 | symbol) are rank-sorted -- each lane counts the smaller keys of its symbols;
//   2. lane 0 merges with two queues (sorted leaves, internal nodes in creation order: both ascending in the key);
//   3. depths: every lane walks its leaves up to the root;  4. the 15-bit limit: lengths are clamped and the Kraft
//   sum repaired one step at a time, the candidate of each step found by all lanes.
template <typename Sync>
__host__ __device__ inline void deflate_lengths(const uint32_t *cnt, int n, int maxbits, uint8_t *len, DeflateWork *ws,
                                                int lane, int nl, Sync sync) {
    uint64_t *key = ws->heap, *sorted = ws->heap + 288;
    int16_t *parent = ws->parent;
    const uint64_t none = ~0ull;
    for (int s = lane; s < n; s += nl) {
        key[s] = cnt[s] ? ((uint64_t)cnt[s] << 16) | (uint64_t)s : none;
        len[s] = 0;
    }
    sync();
    int used = 0;
    for (int t = 0; t < n; ++t) used += cnt[t] != 0;
    for (int s = lane; s < n; s += nl) {
        const uint64_t k = key[s];
        if (k == none) continue;
        int r = 0;
        for (int t = 0; t < n; ++t) r += key[t] < k;
        sorted[r] = k;
    }
    sync();
    if (used == 0) return;
    if (used == 1) {  // one symbol: one bit, paired with a second symbol so the code is complete
        if (lane == 0) {
            const int s = (int)(sorted[0] & 0xffffu);
            len[s] = 1;
            len[s == 0 ? 1 : 0] = 1;
        }
        sync();
        return;
    }
    if (lane == 0) {
        uint64_t *iq = key;  // internal nodes, in creation order (the keys are not needed any more)
        int li = 0, qh = 0, qt = 0, next = n;
        uint64_t lh = sorted[0], ih = none;
        for (int m = 0; m + 1 < used; ++m) {
            uint64_t ab[2];
            for (int k = 0; k < 2; ++k) {
                if (lh < ih) { ab[k] = lh; ++li; lh = li < used ? sorted[li] : none; }
                else { ab[k] = ih; ++qh; ih = qh < qt ? iq[qh] : none; }
            }
            parent[ab[0] & 0xffffu] = (int16_t)next;
            parent[ab[1] & 0xffffu] = (int16_t)next;
            const uint64_t nk = (((ab[0] >> 16) + (ab[1] >> 16)) << 16) | (uint64_t)next;
            iq[qt++] = nk;
            if (ih == none) ih = iq[qh];
            ++next;
        }
        parent[next - 1] = -1;
    }
    sync();
    for (int s = lane; s < n; s += nl) {
        if (!cnt[s]) continue;
        int d = 0;
        for (int v = s; parent[v] >= 0; v = parent[v]) ++d;
        len[s] = (uint8_t)(d > maxbits ? maxbits : d);
    }
    sync();
    // Kraft sum in units of 2^-maxbits; clamping may have pushed it over 1
    const uint32_t one = 1u << maxbits;
    uint32_t k = 0;
    for (int s = 0; s < n; ++s) if (len[s]) k += one >> len[s];
    // candidate of a repair step: the longest code that qualifies; among equals the rarer (lengthen) / more frequent
    // (shorten) symbol, then the lower symbol number.  Every lane proposes its best, lane order breaks no tie.
    uint32_t *prop = ws->prop;
    auto better = [&](int a, int b, bool lengthen) {  // is symbol a a better candidate than b (b may be -1)?
        if (b < 0) return true;
        if (len[a] != len[b]) return len[a] > len[b];
        if (cnt[a] != cnt[b]) return lengthen ? cnt[a] < cnt[b] : cnt[a] > cnt[b];
        return a < b;
    };
    auto pick = [&](bool lengthen, uint32_t room) {
        int best = -1;
        for (int s = lane; s < n; s += nl) {
            const bool ok = lengthen ? (len[s] && len[s] < maxbits) : (len[s] > 1 && (one >> len[s]) <= room);
            if (ok && better(s, best, lengthen)) best = s;
        }
        if (nl == 1) return best;
        prop[lane] = (uint32_t)best;
        sync();
        best = -1;
        for (int l = 0; l < nl; ++l) {
            const int c = (int)prop[l];
            if (c >= 0 && better(c, best, lengthen)) best = c;
        }
        sync();
        return best;
    };
    while (k > one) {  // lengthen the longest code that still can be lengthened (cheapest in expected bits)
        const int best = pick(true, 0);
        k -= one >> (len[best] + 1);
        sync();
        if (lane == 0) ++len[best];
        sync();
    }
    while (k < one) {  // shorten the longest code that fits into what is left
        const int best = pick(false, one - k);
        if (best < 0) break;
        k += one >> len[best];
        sync();
        if (lane == 0) --len[best];
        sync();
    }
}

// canonical codes (RFC 1951 3.2.2), bit-reversed for a least-significant-bit-first stream
__host__ __device__ inline void deflate_codes(const uint8_t *len, int n, uint16_t *code) {
    uint32_t bl_count[16] = {0}, next_code[16] = {0};
    for (int s = 0; s < n; ++s) ++bl_count[len[s]];
    bl_count[0] = 0;
    uint32_t c = 0;
    for (int b = 1; b < 16; ++b) { c = (c + bl_count[b - 1]) << 1; next_code[b] = c; }
    for (int s = 0; s < n; ++s) {
        uint32_t v = 0;
        if (len[s]) {
            const uint32_t x = next_code[len[s]]++;
            for (int b = 0; b < len[s]; ++b) v |= ((x >> b) & 1u) << (len[s] - 1 - b);
        }
        code[s] = (uint16_t)v;
    }
}

struct BitSink {  // least-significant-bit-first stream, whole words written as they fill
    uint32_t *w;
    uint32_t n;    // bits so far
    uint64_t acc;  // bits not yet written (below n & 31)
    __host__ __device__ void put(uint32_t v, int bits) {  // bits <= 16
        acc |= (uint64_t)(v & ((1u << bits) - 1u)) << (n & 31u);
        const uint32_t before = n >> 5;
        n += (uint32_t)bits;
        if ((n >> 5) != before) { w[before] = (uint32_t)acc; acc >>= 32; }
    }
    __host__ __device__ void finish() { if (n & 31u) w[n >> 5] = (uint32_t)acc; }
};

// hist[s]: token counts of the text (literals, hist[256] = number of blocks, match lengths 3..8 at 257..262).  Every
// symbol gets a code (count + 1, and at least 2^-15 of the total so that the tree stays shallow): a batch is
// compressed with the code of its own text, but nothing breaks if a symbol shows up that the histogram missed.
// dist_sym: distance symbol of the batch's record length (0: distance 1 is the only distance in use).
// Called by all nl lanes; results in ws->entry, ws->hdr, ws->hdr_bits (see deflate_store_code).
template <typename Sync>
__host__ __device__ inline void deflate_build_code(const uint32_t *hist, DeflateWork *ws, uint32_t dist_sym, int lane, int nl,
                                                   Sync sync) {
    uint32_t *cnt = ws->cnt;
    uint64_t total = 0;
    for (int s = 0; s < DEFLATE_SYMS; ++s) total += hist[s];
    const uint32_t floor_cnt = (uint32_t)(total >> 15);
    for (int s = lane; s < DEFLATE_SYMS; s += nl) cnt[s] = hist[s] + 1u > floor_cnt ? hist[s] + 1u : floor_cnt;
    sync();
    uint8_t *len = ws->len;
    deflate_lengths(cnt, DEFLATE_SYMS, 15, len, ws, lane, nl, sync);
    sync();
    if (lane == 0) {
        uint16_t *code = ws->code;
        deflate_codes(len, DEFLATE_SYMS, code);
        for (int s = 0; s < DEFLATE_SYMS; ++s) ws->entry[s] = (uint32_t)code[s] | ((uint32_t)len[s] << 16);
        // ---- header: the literal/length code lengths, then the distance code lengths -- distance 1 ("0") and, with
        // dist_sym, the record distance ("1"), one bit each -- run-length coded together (3.2.7)
        const int n_all = DEFLATE_SYMS + 1 + (int)dist_sym;
        for (int i = DEFLATE_SYMS; i < n_all; ++i) len[i] = 0;
        len[DEFLATE_SYMS] = 1;
        len[n_all - 1] = 1;
        uint8_t *sym = ws->sym, *extra = ws->extra;
        int ns = 0;
        for (int i = 0; i < n_all;) {
            int r = 1;
            while (i + r < n_all && len[i + r] == len[i]) ++r;
            sym[ns] = len[i]; extra[ns] = 0; ++ns;  // the value itself
            int rem = r - 1;
            if (len[i] != 0)
                while (rem >= 3) { const int t = rem > 6 ? 6 : rem; sym[ns] = 16; extra[ns] = (uint8_t)(t - 3); ++ns; rem -= t; }
            for (; rem > 0; --rem) { sym[ns] = len[i]; extra[ns] = 0; ++ns; }
            i += r;
        }
        ws->ns = (uint32_t)ns;
        for (int i = 0; i < 19; ++i) ws->ccnt[i] = 0;
        for (int i = 0; i < ns; ++i) ++ws->ccnt[sym[i]];
    }
    sync();
    deflate_lengths(ws->ccnt, 19, 7, ws->clen, ws, lane, nl, sync);
    sync();
    if (lane == 0) {
        const uint8_t *clen = ws->clen, *sym = ws->sym, *extra = ws->extra;
        uint16_t *ccode = ws->ccode;
        deflate_codes(clen, 19, ccode);
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = 19;
        while (hclen > 4 && clen[order[hclen - 1]] == 0) --hclen;
        for (int i = 0; i < DEFLATE_HDR_WORDS; ++i) ws->hdr[i] = 0;
        BitSink bs{ws->hdr, 0, 0};
        bs.put(0, 1);  // BFINAL = 0 (the member is closed by an empty final block)
        bs.put(2, 2);  // BTYPE = 10: dynamic Huffman codes
        bs.put(DEFLATE_SYMS - 257, 5);  // HLIT
        bs.put(dist_sym, 5);            // HDIST: distance codes 0 .. dist_sym
        bs.put((uint32_t)(hclen - 4), 4);
        for (int i = 0; i < hclen; ++i) bs.put(clen[order[i]], 3);
        const int ns = (int)ws->ns;
        for (int i = 0; i < ns; ++i) {
            bs.put(ccode[sym[i]], clen[sym[i]]);
            if (sym[i] == 16) bs.put(extra[i], 2);
        }
        bs.finish();
        ws->hdr_bits = bs.n;
    }
    sync();
}

// the part of a DeflateCode the builder makes; lanes `lane`, `lane + n_lanes`, ... of the caller copy
__host__ __device__ inline void deflate_store_code(const DeflateWork *ws, DeflateCode *out, int lane, int n_lanes) {
    for (int s = lane; s < DEFLATE_SYMS; s += n_lanes) out->entry[s] = ws->entry[s];
    for (int i = lane; i < DEFLATE_HDR_WORDS; i += n_lanes) out->hdr[i] = ws->hdr[i];
    if (lane == 0) out->hdr_bits = ws->hdr_bits;
}

// ---------------------------------------------------------------- CRC-32 (reflected 0xEDB88320), raw: initial value 0, no final xor
__host__ __device__ inline uint32_t crc_table_entry(uint32_t i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    return c;
}
__host__ __device__ inline uint32_t gf2_times(const uint32_t *mat, uint32_t vec) {
    uint32_t sum = 0;
    for (int i = 0; vec; vec >>= 1, ++i) if (vec & 1u) sum ^= mat[i];
    return sum;
}
inline void gf2_square(uint32_t *sq, const uint32_t *mat) { for (int n = 0; n < 32; ++n) sq[n] = gf2_times(mat, mat[n]); }
// operator "append n zero bytes" (zlib's crc32_combine construction), n >= 1
inline void crc_shift_operator(uint64_t n_bytes, uint32_t *op) {
    uint32_t even[32], odd[32];
    odd[0] = 0xEDB88320u;  // one zero bit
    for (int n = 1; n < 32; ++n) odd[n] = 1u << (n - 1);
    gf2_square(even, odd);   // two bits
    gf2_square(odd, even);   // four bits
    uint32_t acc[32];
    for (int n = 0; n < 32; ++n) acc[n] = 1u << n;  // identity
    // the first squaring below yields the one-byte operator
    uint32_t *cur = odd, *nxt = even;
    for (uint64_t n = n_bytes; n; n >>= 1) {
        gf2_square(nxt, cur);
        uint32_t *t = cur; cur = nxt; nxt = t;  // cur = operator for 2^k bytes
        if (n & 1u) {
            uint32_t tmp[32];
            for (int i = 0; i < 32; ++i) tmp[i] = gf2_times(cur, acc[i]);
            for (int i = 0; i < 32; ++i) acc[i] = tmp[i];
        }
    }
    for (int n = 0; n < 32; ++n) op[n] = acc[n];
}

// ---------------------------------------------------------------- tokens
// The text is cut into 32-byte chunks (aligned to the text's start; blocks are multiples of 32).  Inside a chunk, at
// every position two matches are tried: the run (the byte repeats its predecessor: distance 1) and the previous
// record (the same bytes `dist` earlier -- a batch's records have one length but for the digits of the pair number,
// so the constant parts of the header line, "+" and most of a top-quality phred line are found there); the longer
// one wins if it has >= 3 (run) / >= 4 (previous record) bytes (lengths 3..32: codes 257..272, RFC 1951 3.2.5),
// otherwise the byte is a literal.  The predecessor of a chunk's first byte and the bytes `dist` earlier may belong to
// the previous block: DEFLATE's window does not care.  Match lengths come from byte-difference masks and a count of
// trailing zeros, so a chunk costs the same whatever it holds.
// f(symbol, kind, extra bits of the length code, their value) is called per token: kind 0 literal, 1 run, 2 previous record.
struct DeflateChunk {
    uint64_t raw[DEFLATE_NQ];  // the bytes, little endian
    uint64_t src[DEFLATE_NQ];  // the bytes `dist` earlier (has_src)
    uint32_t m;                // bytes in the chunk (< DEFLATE_CHUNK only at the end of the text)
    int prev;                  // the byte before the chunk, -1: none
    bool has_src;
};

// bit k of the result <=> byte k of v is not zero
__host__ __device__ inline uint32_t deflate_nonzero_bytes(uint64_t v) {
    const uint64_t low7 = 0x7f7f7f7f7f7f7f7full;
    const uint64_t top = (((v & low7) + low7) | v) & ~low7;  // bit 7 of every non-zero byte
    return (uint32_t)((top * 0x0002040810204081ull) >> 56);   // the eight bits 7, 15, ... 63 gathered
}

__host__ __device__ inline void deflate_length_code(uint32_t len, uint32_t *sym, uint32_t *xbits, uint32_t *xval) {
    if (len <= 10u) { *sym = 254u + len; *xbits = 0; *xval = 0; return; }
    const uint32_t k = len - 11u;
    if (k < 8u) { *sym = 265u + (k >> 1); *xbits = 1; *xval = k & 1u; return; }    // 11 .. 18
    *sym = 269u + ((k - 8u) >> 2); *xbits = 2; *xval = (k - 8u) & 3u;              // 19 .. 34 (here <= 32)
}

template <typename F>
__host__ __device__ inline void deflate_tokens(const DeflateChunk &C, F &&f) {
    // differs[k]: byte k differs from the byte before it (run) / from the byte `dist` earlier (previous record);
    // a stop bit at position m ends every match at the chunk's end
    uint64_t diff_run = 0, diff_rec = 0;
    for (int q = 0; q < DEFLATE_NQ; ++q) {
        const uint64_t before = (C.raw[q] << 8) | (q ? C.raw[q - 1] >> 56 : (uint64_t)(C.prev & 0xff));
        diff_run |= (uint64_t)deflate_nonzero_bytes(C.raw[q] ^ before) << (8 * q);
        diff_rec |= (uint64_t)(C.has_src ? deflate_nonzero_bytes(C.raw[q] ^ C.src[q]) : 0xffu) << (8 * q);
    }
    if (C.prev < 0) diff_run |= 1u;
    diff_run |= 1ull << C.m;
    diff_rec |= 1ull << C.m;
    uint32_t i = 0;
    while (i < C.m) {
        const uint32_t r1 = (uint32_t)__builtin_ctzll(diff_run >> i), rd = (uint32_t)__builtin_ctzll(diff_rec >> i);
        uint32_t step = 1, sym = (uint32_t)((C.raw[i >> 3] >> (8 * (i & 7u))) & 0xffu), xbits = 0, xval = 0;
        int kind = 0;
        if (r1 >= 3u && r1 >= rd) { step = r1; kind = 1; }   // (a run is the cheaper match)
        else if (rd >= 4u) { step = rd; kind = 2; }          // (its distance costs 8-14 bits: three bytes are not worth it)
        if (kind) deflate_length_code(step, &sym, &xbits, &xval);
        f(sym, kind, xbits, xval);
        i += step;
    }
}

// DEFLATE distance code of `dist` (RFC 1951 3.2.5): symbol, number of extra bits, their value
__host__ __device__ inline void deflate_dist_code(uint32_t dist, uint32_t *sym, uint32_t *ebits, uint32_t *eval) {
    if (dist <= 4u) { *sym = dist - 1u; *ebits = 0; *eval = 0; return; }
    uint32_t d = dist - 1u, hb = 31u;
    while (!((d >> hb) & 1u)) --hb;           // highest set bit of dist - 1 (>= 2)
    const uint32_t second = (d >> (hb - 1u)) & 1u;
    *sym = 2u * hb + second;
    *ebits = hb - 1u;
    *eval = d & ((1u << (hb - 1u)) - 1u);
}

// ---------------------------------------------------------------- kernels
struct DeflateArgs {
    const uint8_t *text[2];     // [mate]
    uint64_t n_bytes;           // per mate
    uint32_t n_blocks;          // ceil(n_bytes / DEFLATE_BLOCK)
    uint32_t *hist[2];          // [mate][DEFLATE_SYMS]
    DeflateCode *code[2];
    uint32_t *block_bytes[2];   // [mate][n_blocks] compressed size of each block
    uint32_t *block_crc[2];     // [mate][n_blocks] raw CRC-32 of each block's text
    uint64_t *block_off[2];     // [mate][n_blocks + 1] byte offsets in `out` (exclusive scan), [n_blocks] = total
    uint8_t *out[2];
    uint64_t out_cap;
    uint32_t dist;              // record length of the batch (0: only runs are matched)
    uint32_t dist_sym, dist_ebits, dist_eval;
};

// text bytes of block b.  (Written with 32-bit block numbers on purpose: hipcc 7.2 lowers the obvious
// min(DEFLATE_BLOCK, n_bytes - start) on uniform 64-bit values to a v_cmp followed by an s_cselect on a stale SCC.)
__device__ __forceinline__ uint32_t deflate_block_len(uint64_t n_bytes, uint32_t b) {
    const uint32_t full = (uint32_t)(n_bytes / DEFLATE_BLOCK);
    return b < full ? (uint32_t)DEFLATE_BLOCK : (uint32_t)(n_bytes - (uint64_t)full * DEFLATE_BLOCK);
}

// the eight bytes that stand `dist` before text offset `at` (at >= dist; the buffer has 16 bytes of padding)
__device__ __forceinline__ uint64_t deflate_source(const uint8_t *t, uint64_t at, uint32_t dist) {
    const uint64_t p = at - dist, a = p & ~7ull;
    const uint32_t sh = (uint32_t)(p & 7ull) * 8u;
    const uint64_t lo = *reinterpret_cast<const uint64_t *>(t + a);
    if (!sh) return lo;
    const uint64_t hi = *reinterpret_cast<const uint64_t *>(t + a + 8);
    return (lo >> sh) | (hi << (64u - sh));
}

// chunk `c` of the text
__device__ __forceinline__ void deflate_chunk(const uint8_t *t, uint64_t n_bytes, uint64_t c, uint32_t dist, DeflateChunk &C) {
    const uint64_t at = c * DEFLATE_CHUNK;
    C.m = (uint32_t)min((uint64_t)DEFLATE_CHUNK, n_bytes - at);
    C.prev = at ? (int)t[at - 1] : -1;
    C.has_src = dist && at >= dist;
    for (int q = 0; q < DEFLATE_NQ; ++q) C.src[q] = C.has_src ? deflate_source(t, at + 8u * q, dist) : 0;
    if (C.m == (uint32_t)DEFLATE_CHUNK) {  // (the text buffer is 16-byte aligned)
        for (int q = 0; q < DEFLATE_NQ; q += 2) {
            const uint4 v = *reinterpret_cast<const uint4 *>(t + at + 8 * q);
            C.raw[q] = (uint64_t)v.x | ((uint64_t)v.y << 32);
            C.raw[q + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
        }
    } else {
        for (int q = 0; q < DEFLATE_NQ; ++q) C.raw[q] = 0;
        for (uint32_t k = 0; k < C.m; ++k) C.raw[k >> 3] |= (uint64_t)t[at + k] << (8 * (k & 7u));
    }
}

__global__ __launch_bounds__(DEFLATE_THREADS) void k_deflate_hist(DeflateArgs A) {
    __shared__ uint32_t h[DEFLATE_SYMS];
    const int mate = blockIdx.y;
    for (int s = threadIdx.x; s < DEFLATE_SYMS; s += DEFLATE_THREADS) h[s] = 0;
    __syncthreads();
    const uint8_t *t = A.text[mate];
    const uint64_t n_chunks = (A.n_bytes + DEFLATE_CHUNK - 1) / DEFLATE_CHUNK;
    for (uint64_t c = (uint64_t)blockIdx.x * DEFLATE_THREADS + threadIdx.x; c < n_chunks; c += (uint64_t)gridDim.x * DEFLATE_THREADS) {
        DeflateChunk C;
        deflate_chunk(t, A.n_bytes, c, A.dist, C);
        deflate_tokens(C, [&](uint32_t sym, int, uint32_t, uint32_t) { atomicAdd(&h[sym], 1u); });
    }
    __syncthreads();
    for (int s = threadIdx.x; s < DEFLATE_SYMS; s += DEFLATE_THREADS)
        if (h[s]) atomicAdd(&A.hist[mate][s], h[s]);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&A.hist[mate][256], A.n_blocks);
}

struct DeflateBlockSync { __device__ void operator()() const { __syncthreads(); } };

// one wavefront per mate
__global__ __launch_bounds__(64) void k_deflate_build(DeflateArgs A) {
    __shared__ DeflateWork ws;
    __shared__ uint32_t hist[DEFLATE_SYMS];
    for (int s = threadIdx.x; s < DEFLATE_SYMS; s += blockDim.x) hist[s] = A.hist[blockIdx.x][s];
    __syncthreads();
    deflate_build_code(hist, &ws, A.dist ? A.dist_sym : 0u, (int)threadIdx.x, (int)blockDim.x, DeflateBlockSync());
    deflate_store_code(&ws, A.code[blockIdx.x], (int)threadIdx.x, (int)blockDim.x);
}

// One workgroup per block: compressed size in bytes and the raw CRC-32 of the block's text.  For the CRC, lane t of
// the workgroup owns the 128 bytes that END (256 - t) * 128 bytes before the end of the block (a short block is
// padded with zeros in FRONT, which a raw CRC does not see), so the tree of "append 128 << k bytes" operators is the
// same for every block.  A full block is staged in LDS with coalesced loads first (pieces of 32 words at a stride of
// 33, so that the lanes' sequential walks over their pieces hit distinct banks); the last, shorter block of a text
// takes the plain path.
__global__ __launch_bounds__(DEFLATE_THREADS) void k_deflate_len(DeflateArgs A) {
    __shared__ uint32_t tab[4][256];  // slicing-by-4 tables
    __shared__ uint32_t lens[DEFLATE_SYMS];
    __shared__ uint32_t red[DEFLATE_THREADS];
    __shared__ uint32_t crcs[DEFLATE_THREADS];
    __shared__ uint32_t stage[DEFLATE_THREADS * 33];
    __shared__ uint32_t shift[8][32];
    const int mate = blockIdx.y;
    const uint32_t b = blockIdx.x;
    const DeflateCode *C = A.code[mate];
    shift[threadIdx.x >> 5][threadIdx.x & 31] = C->crc_shift[threadIdx.x >> 5][threadIdx.x & 31];
    tab[0][threadIdx.x] = crc_table_entry(threadIdx.x);
    for (int s = threadIdx.x; s < DEFLATE_SYMS; s += DEFLATE_THREADS) lens[s] = C->entry[s] >> 16;
    __syncthreads();
    for (int k = 1; k < 4; ++k) {
        const uint32_t v = tab[k - 1][threadIdx.x];
        tab[k][threadIdx.x] = (v >> 8) ^ tab[0][v & 0xffu];
        __syncthreads();
    }
    const uint64_t start = (uint64_t)b * DEFLATE_BLOCK;
    const uint32_t n = deflate_block_len(A.n_bytes, b);
    const uint8_t *t = A.text[mate] + start;
    uint32_t bits = 0, crc = 0;
    // bits after the length code: a run's distance code (1 bit), the record distance's code (1 bit) + extra bits
    const uint32_t kind_bits[3] = {0u, 1u, 1u + A.dist_ebits};
    if (n == (uint32_t)DEFLATE_BLOCK) {
        const uint4 *src = reinterpret_cast<const uint4 *>(t);
        for (int j = 0; j < DEFLATE_BLOCK / 16 / DEFLATE_THREADS; ++j) {
            const uint32_t q = threadIdx.x + j * DEFLATE_THREADS;  // 16-byte unit
            const uint4 v = src[q];
            uint32_t *d = stage + (q >> 3) * 33u + (q & 7u) * 4u;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        const int before = start ? (int)t[-1] : -1;
        for (uint32_t c = threadIdx.x; c < (uint32_t)(DEFLATE_BLOCK / DEFLATE_CHUNK); c += DEFLATE_THREADS) {
            const uint32_t w = (c >> 2) * 33u + (c & 3u) * 8u;  // four chunks per 128-byte piece
            DeflateChunk C;
            for (int q = 0; q < DEFLATE_NQ; ++q) C.raw[q] = (uint64_t)stage[w + 2 * q] | ((uint64_t)stage[w + 2 * q + 1] << 32);
            C.m = DEFLATE_CHUNK;
            C.prev = c ? (int)(stage[(c & 3u) ? w - 1 : w - 2] >> 24) : before;
            const uint64_t at = start + (uint64_t)c * DEFLATE_CHUNK;
            C.has_src = A.dist && at >= A.dist;
            for (int q = 0; q < DEFLATE_NQ; ++q) C.src[q] = C.has_src ? deflate_source(A.text[mate], at + 8u * q, A.dist) : 0;
            deflate_tokens(C, [&](uint32_t sym, int kind, uint32_t xbits, uint32_t) { bits += lens[sym] + xbits + kind_bits[kind]; });
        }
        const uint32_t *mine = stage + threadIdx.x * 33u;
        for (int i = 0; i < 32; ++i) {
            const uint32_t x = crc ^ mine[i];
            crc = tab[3][x & 0xffu] ^ tab[2][(x >> 8) & 0xffu] ^ tab[1][(x >> 16) & 0xffu] ^ tab[0][x >> 24];
        }
    } else {
        for (uint32_t c = threadIdx.x; c * DEFLATE_CHUNK < n; c += DEFLATE_THREADS) {
            DeflateChunk C;
            deflate_chunk(A.text[mate], A.n_bytes, start / DEFLATE_CHUNK + c, A.dist, C);
            deflate_tokens(C, [&](uint32_t sym, int kind, uint32_t xbits, uint32_t) { bits += lens[sym] + xbits + kind_bits[kind]; });
        }
        const int64_t lo = (int64_t)n - (int64_t)(DEFLATE_THREADS - threadIdx.x) * 128;  // may be negative: zeros in front
        for (int64_t i = lo < 0 ? 0 : lo; i < lo + 128; ++i) crc = tab[0][(crc ^ t[i]) & 0xffu] ^ (crc >> 8);
    }
    red[threadIdx.x] = bits;
    crcs[threadIdx.x] = crc;
    __syncthreads();
    for (int k = 0, s = 1; s < DEFLATE_THREADS; s <<= 1, ++k) {
        if ((threadIdx.x & (2 * s - 1)) == 0) {
            red[threadIdx.x] += red[threadIdx.x + s];
            crcs[threadIdx.x] = gf2_times(shift[k], crcs[threadIdx.x]) ^ crcs[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t total = C->hdr_bits + red[0] + lens[256] + 3u;  // + end of block + header of the empty stored block
        A.block_bytes[mate][b] = (total + 7u) / 8u + 4u;                 // + LEN = 0, NLEN = 0xffff
        A.block_crc[mate][b] = crcs[0];
    }
}

// exclusive scan of the block sizes (one workgroup per mate)
__global__ __launch_bounds__(1024) void k_deflate_scan(DeflateArgs A) {
    __shared__ uint64_t part[1024];
    const int mate = blockIdx.x;
    const uint32_t per = (A.n_blocks + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per, hi = min(A.n_blocks, lo + per);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += A.block_bytes[mate][i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint64_t v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (uint32_t i = lo; i < hi; ++i) { A.block_off[mate][i] = run; run += A.block_bytes[mate][i]; }
    if (threadIdx.x == 1023) A.block_off[mate][A.n_blocks] = part[1023];
}

// One workgroup per block.  Tiles of 256 lanes x 32 bytes: each lane counts the bits of its chunk's tokens (<= 480), a
// workgroup scan places them, then the lane walks its tokens again and ORs their bits into an LDS window, 32 at a
// time; whole words of the window go out and the unfinished last word starts the next tile.  The block starts on a
// byte of `out`, not on a word: the window is bit-shifted by the misalignment and the first / last word are merged
// with atomicOr (the buffer is zeroed).
constexpr int DEFLATE_WIN_WORDS = DEFLATE_THREADS * DEFLATE_CHUNK * 15 / 32 + 4;

__global__ __launch_bounds__(DEFLATE_THREADS) void k_deflate_encode(DeflateArgs A) {
    __shared__ uint32_t ent[DEFLATE_SYMS];
    __shared__ uint32_t win[DEFLATE_WIN_WORDS];
    __shared__ uint32_t wsum[DEFLATE_THREADS / 64];
    const int mate = blockIdx.y;
    const uint32_t b = blockIdx.x;
    const DeflateCode *C = A.code[mate];
    for (int s = threadIdx.x; s < DEFLATE_SYMS; s += DEFLATE_THREADS) ent[s] = C->entry[s];
    for (int i = threadIdx.x; i < DEFLATE_WIN_WORDS; i += DEFLATE_THREADS) win[i] = 0;
    __syncthreads();
    const uint64_t start = (uint64_t)b * DEFLATE_BLOCK;
    const uint32_t n = deflate_block_len(A.n_bytes, b);
    const uint64_t off = A.block_off[mate][b];
    if (off + A.block_bytes[mate][b] > A.out_cap) return;  // (the host reports the overflow from block_off[n_blocks])
    uint32_t *outw = reinterpret_cast<uint32_t *>(A.out[mate] + (off & ~3ull));
    uint32_t wpos = 0;                            // words of this block already written
    uint32_t fill = (uint32_t)(off & 3ull) * 8u;  // bits in the window so far (the first tile starts misaligned)
    auto or_bits = [&](uint32_t at, uint64_t v) {  // OR <= 64 bits at bit `at` of the window
        if (!v) return;
        const uint32_t w = at >> 5, sh = at & 31u;
        atomicOr(&win[w], (uint32_t)(v << sh));
        const uint64_t hi = sh ? v >> (32 - sh) : v >> 32;
        if (hi) {
            atomicOr(&win[w + 1], (uint32_t)hi);
            if (hi >> 32) atomicOr(&win[w + 2], (uint32_t)(hi >> 32));
        }
    };
    auto flush = [&](bool last) {  // whole words of the window -> out; the partial last word moves to the front
        __syncthreads();
        const uint32_t nw = last ? (fill + 31u) >> 5 : fill >> 5;
        for (uint32_t i = threadIdx.x; i < nw; i += DEFLATE_THREADS) {
            const uint32_t v = win[i];
            if ((wpos + i == 0) || (last && i == nw - 1)) { if (v) atomicOr(&outw[wpos + i], v); }
            else outw[wpos + i] = v;
        }
        __syncthreads();
        const uint32_t keep = last ? 0u : win[nw];
        __syncthreads();
        for (uint32_t i = threadIdx.x; i <= nw + 4 && i < DEFLATE_WIN_WORDS; i += DEFLATE_THREADS) win[i] = 0;
        __syncthreads();
        if (threadIdx.x == 0) win[0] = keep;
        wpos += nw;
        fill &= last ? 0u : 31u;
        __syncthreads();
    };
    // ---- block header
    for (uint32_t i = threadIdx.x; i * 32u < C->hdr_bits; i += DEFLATE_THREADS) {
        const uint32_t left = C->hdr_bits - i * 32u;
        const uint32_t v = left >= 32u ? C->hdr[i] : (C->hdr[i] & ((1u << left) - 1u));
        or_bits(fill + i * 32u, v);
    }
    fill += C->hdr_bits;
    flush(false);
    // ---- tokens
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t kind_bits[3] = {0u, 1u, 1u + A.dist_ebits};
    for (uint32_t base = 0; base < n; base += DEFLATE_THREADS * DEFLATE_CHUNK) {
        const uint32_t at = base + threadIdx.x * DEFLATE_CHUNK;
        DeflateChunk K;
        uint32_t nb = 0;
        if (at < n) {
            deflate_chunk(A.text[mate], A.n_bytes, (start + at) / DEFLATE_CHUNK, A.dist, K);
            deflate_tokens(K, [&](uint32_t sym, int kind, uint32_t xbits, uint32_t) { nb += (ent[sym] >> 16) + xbits + kind_bits[kind]; });
        }
        // exclusive scan of nb over the workgroup
        uint32_t x = nb;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t pre = x - nb, tile_bits = 0;
        for (int w = 0; w < DEFLATE_THREADS / 64; ++w) {
            if (w < wave) pre += wsum[w];
            tile_bits += wsum[w];
        }
        if (at < n) {
            uint32_t pos = fill + pre, have = 0;
            uint64_t acc = 0;  // bits not yet in the window (< 32 of them between tokens)
            deflate_tokens(K, [&](uint32_t sym, int kind, uint32_t xbits, uint32_t xval) {
                const uint32_t e = ent[sym];
                uint32_t v = e & 0xffffu, l = e >> 16;                     // <= 15 bits
                v |= xval << l; l += xbits;                                // extra bits of the length code
                // the distance code: "0" for distance 1, "1" + extra bits for the record distance (l <= 31)
                if (kind == 1) l += 1u;
                if (kind == 2) { v |= (1u | (A.dist_eval << 1)) << l; l += 1u + A.dist_ebits; }
                acc |= (uint64_t)v << have;
                have += l;
                if (have >= 32u) { or_bits(pos, acc & 0xffffffffull); pos += 32u; acc >>= 32; have -= 32u; }
            });
            or_bits(pos, acc);
        }
        fill += tile_bits;
        flush(false);
    }
    // ---- end of block, then an empty stored block: 3 header bits, padding to a byte, LEN = 0, NLEN = 0xffff
    if (threadIdx.x == 0) or_bits(fill, ent[256] & 0xffffu);
    fill += ent[256] >> 16;
    fill += 3u;
    fill = (fill + 7u) & ~7u;
    if (threadIdx.x == 0) or_bits(fill + 16u, 0xffffull);
    fill += 32u;
    flush(true);
}

}  // namespace iss
