// iss_fastq.hip.h -- FASTQ text built on the device.
//
// The reference writes one record at a time through Biopython (`SeqIO.write(record, handle, "fastq-sanger")`,
// iss/generator.py:64-65): "@{id}_{i}_{cpu}/{1|2}\n" SEQ "\n+\n" QUAL "\n" with QUAL = chr(33 + q) and the ids of
// iss/generator.py:150, 181.  At the kernel's rate the host cannot even format that text, so the text itself is
// produced here: a record's length is C + digits(i), hence its byte offset in the file is a closed form of i
// (k_fastq_format needs no scan), one wavefront writes one record, and the host only copies bytes to the file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "iss_kernels.hip.h"  // xp(): layout of the output rows

namespace iss {

// characters of the decimal numbers 0 .. x-1 written one after the other
__host__ __device__ inline uint64_t digits_before(uint64_t x) {
    if (x == 0) return 0;
    uint64_t p = 1, cum = 1;  // p = 10^(d-1); cum = characters of all numbers with fewer than d digits (0 counts once)
    int d = 1;
    while (x / 10 >= p) {     // x has more than d digits
        cum += (uint64_t)d * (d == 1 ? 9 : p * 9);
        p *= 10;
        ++d;
    }
    return cum + (uint64_t)d * (x - (d == 1 ? 1 : p));
}

// One emit call formats the rows of several work items (records): item k = rows [first_pair, +n_pairs) with ids
// "{id_k}_{first_i + j}_{cpu_k}" and its text at byte `text_off` of the call's text.
struct FastqItem {
    uint64_t first_i;        // pair id of the item's first row
    uint64_t before_first;   // digits_before(first_i)
    uint64_t text_off;       // byte offset of the item's first record in the text
    int64_t first_pair;      // first output row
    int64_t rec_first;       // records of the items before this one
    uint32_t id_off;         // of the record id in `ids`
    int32_t id_len;
    int32_t cpu_len;         // the worker's number (the items of one call may belong to different workers: iss_fastq_emit_scatter)
    char cpu[12];            // ... in decimal
};

struct FastqArgs {
    const uint8_t *base[2], *qual[2];  // output rows (row 0): [mate]; position p of row i at [i * row + xp(p)]
    uint8_t *text[2];                  // [mate] output text
    const FastqItem *items;            // [n_items]
    const char *ids;                   // record ids, back to back
    int32_t n_items, row, RL;
    int64_t n_records;                 // of all items
};

constexpr int FASTQ_WAVES = 4;

// grid = (ceil(n_records / FASTQ_WAVES), 2 mates), block = 64 * FASTQ_WAVES: one wavefront per record
__global__ __launch_bounds__(64 * FASTQ_WAVES) void k_fastq_format(FastqArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * FASTQ_WAVES + (threadIdx.x >> 6);
    if (r >= A.n_records) return;
    int lo = 0, hi = A.n_items;  // the item of record r: largest k with rec_first[k] <= r
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (A.items[mid].rec_first <= r) lo = mid; else hi = mid;
    }
    const FastqItem it = A.items[lo];
    const int64_t i = r - it.rec_first;
    const int mate = blockIdx.y;
    const uint64_t g = it.first_i + (uint64_t)i;
    int dg = 1;
    for (uint64_t p = 10; dg < 20 && g >= p; p *= 10) ++dg;
    const uint64_t C = (uint64_t)it.id_len + (uint64_t)it.cpu_len + 2ull * (uint64_t)A.RL + 10ull;
    uint8_t *w = A.text[mate] + it.text_off + (uint64_t)i * C + (digits_before(g) - it.before_first);
    const char *id = A.ids + it.id_off;
    // ---- "@id_i_cpu/m\n"
    const int h1 = 1 + it.id_len;      // '@' id
    const int h2 = h1 + 1 + dg;        // '_' digits
    const int hlen = h2 + 1 + it.cpu_len + 3;
    for (int k = lane; k < hlen; k += 64) {
        char c;
        if (k == 0) c = '@';
        else if (k < h1) c = id[k - 1];
        else if (k == h1) c = '_';
        else if (k < h2) {
            uint64_t v = g;
            for (int z = h2 - 1 - k; z > 0; --z) v /= 10;  // digit (h2 - 1 - k) from the right
            c = (char)('0' + (int)(v % 10));
        } else if (k == h2) c = '_';
        else if (k < h2 + 1 + it.cpu_len) c = it.cpu[k - h2 - 1];
        else if (k == hlen - 3) c = '/';
        else if (k == hlen - 2) c = (char)('1' + mate);
        else c = '\n';
        w[k] = (uint8_t)c;
    }
    w += hlen;
    const uint8_t *b = A.base[mate] + (size_t)(it.first_pair + i) * A.row;
    const uint8_t *q = A.qual[mate] + (size_t)(it.first_pair + i) * A.row;
    for (int k = lane; k < A.RL; k += 64) w[k] = b[xp(k)];
    if (lane < 3) w[A.RL + lane] = lane == 1 ? '+' : '\n';
    uint8_t *wq = w + A.RL + 3;
    for (int k = lane; k < A.RL; k += 64) wq[k] = (uint8_t)(33 + q[xp(k)]);
    if (lane == 0) wq[A.RL] = '\n';
}

}  // namespace iss
