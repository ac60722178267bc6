// iss_units.hip.h -- the inner plugin surface on the device: batched, function-level entry points behind the methods of
// the reference's ErrorModel duck type (iss/error_models/__init__.py:52-112, 158-228; kde.py:52-98), one read per lane,
// every uniform addressed by (worker seed, ordinal, kind, index) exactly as in the generation kernels (DESIGN.md
// section 4).  These are the reference's per-read methods, not the hot path: straight-line exact code (full 53-bit
// thresholds, sequential list edits), one lane per read.
#pragma once
#include "iss_kernels.hip.h"

namespace iss {

struct UnitArgs {
    uint64_t seed, first_ordinal;  // read i of the call draws at ordinal first_ordinal + i, attempt 0
    int32_t orientation;           // 0 forward, 1 reverse
    int32_t n;
};

// full quality draw of position p: m = h16 << 37 | l37 (K_QM / K_QM_LO)
__device__ __forceinline__ uint64_t unit_quality_draw(const Addr &a, int o, int p) {
    const int c = p & 7, half = c >> 2, cc = c & 3;
    const uint32_t h = hot_h16(draw_block(a, K_QM, (uint32_t)p >> 3, (uint32_t)(2 * half)), o, cc);
    return mk_digit(h, lo37(draw_block(a, K_QM_LO, (uint32_t)p, (uint32_t)o), 0));
}

// KDErrorModel.gen_phred_scores (kde.py:52-86): out[i][0 .. RL)
__global__ __launch_bounds__(64) void k_unit_phred(DevModel M, UnitArgs U, uint8_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U.n) return;
    const int o = U.orientation, RL = M.RL;
    const Addr a = make_addr(U.seed, U.first_ordinal + (uint64_t)i, 0u);
    const u32x4 w0 = draw_block(a, K_PAIR, 0, 0), w1 = draw_block(a, K_PAIR, 0, 1);
    int bin = count_le(M.bin_thr + 4 * o, 4, o ? mk53(w0.z, w1.z) : mk53(w0.y, w1.y));  // kde.py:74
    bin = bin > 3 ? 3 : bin;                                                                // kde.py:77-78
    const uint64_t *rows = M.q_thr + ((size_t)(o * 4 + bin) * RL) * M.n_q;
    for (int p = 0; p < RL; ++p)                                                            // kde.py:83-85
        out[(size_t)i * RL + p] = (uint8_t)count_lt(rows + (size_t)p * M.n_q, M.n_q, unit_quality_draw(a, o, p));
}

// ErrorModel.mut_sequence (__init__.py:69-112): seq[i][0 .. RL) in place, given its phred scores; status[i] = 2 for a
// letter outside the model's substitution table (the reference's KeyError)
__global__ __launch_bounds__(64) void k_unit_mut(DevModel M, UnitArgs U, uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                                 int32_t *__restrict__ status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U.n) return;
    const int o = U.orientation, RL = M.RL;
    const Addr a = make_addr(U.seed, U.first_ordinal + (uint64_t)i, 0u);
    int rc = 0;
    for (int p = 0; p < RL && !rc; ++p) {
        const int c = p & 7;
        const uint32_t e8 = hot_e8(draw_block(a, K_QM, (uint32_t)p >> 3, 1u), c >> 2, o, c & 3);
        const u32x4 sb = draw_block(a, K_SUB, (uint32_t)p, (uint32_t)o);
        const int ch = seq[(size_t)i * RL + p], cu = ch & ~0x20;
        const bool ambiguous = cu == 'R' || cu == 'Y' || cu == 'W' || cu == 'S' || cu == 'M' || cu == 'K' || cu == 'H' || cu == 'B' ||
                               cu == 'V' || cu == 'D' || cu == 'N';
        if (error_test_draw(e8, sb) > M.mut_thr[qual[(size_t)i * RL + p]] && !ambiguous) {  // :94
            const int bi = base_index(cu);
            if (bi < 0) { rc = 2; break; }
            const uint64_t m = mk53(sb.x, sb.y);
            const size_t row = ((size_t)(o * RL + p) * 4 + bi) * 3;
            const int k = (m >= M.subst_thr[row]) + (m >= M.subst_thr[row + 1]);
            seq[(size_t)i * RL + p] = M.subst_alt[row + k];
        }
    }
    status[i] = rc;
}

// KDErrorModel.random_insert_size (kde.py:88-98)
__global__ __launch_bounds__(64) void k_unit_isize(DevModel M, UnitArgs U, int64_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U.n) return;
    const Addr a = make_addr(U.seed, U.first_ordinal + (uint64_t)i, 0u);
    out[i] = count_lt(M.isize_thr, M.n_isize, mk53(draw_block(a, K_PAIR, 0, 0).x, draw_block(a, K_PAIR, 0, 1).x));
}

// One draw of the indel event process (ev_step, iss_kernels.hip.h) for n independent (state, uniform) inputs -- the
// function the scan / fix-up kernels loop over, exported so that the tests can compare it with the CPU oracle's twin, whose
// interval structure the CPU tests pin to the reference's per-test probabilities exactly.
__global__ __launch_bounds__(64) void k_unit_ev_step(DevModel M, int32_t o, int32_t n, const int32_t *__restrict__ cur,
                                                     const uint64_t *__restrict__ m53, const uint64_t *__restrict__ v53,
                                                     int32_t *__restrict__ next, int32_t *__restrict__ slot, uint8_t *__restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int sl;
    uint32_t m8;
    next[i] = ev_step(M.ev_S + (size_t)o * M.ev_ns, M.ev_E + (size_t)o * M.ev_ns, M.ev_T + (size_t)o * M.ev_ns, M.del_thr + (size_t)o * M.RL * 4,
                      cur[i], m53[i], v53[i], sl, m8);
    slot[i] = sl;
    mask[i] = (uint8_t)m8;
}

// ErrorModel.introduce_indels + adjust_seq_length (__init__.py:158-228, 114-156): read i = seq[i][0 .. len[i]) (the
// perfect read, <= RL letters, already in read direction), bounds[i] = (start, end) in `genome` (length L) for the
// padding; out[i][0 .. RL).  work: [n][cap] scratch letters + event masks, cap = 6 * RL + 8.  status: 0 ok, 2 KeyError, 3 IndexError.
__global__ __launch_bounds__(64) void k_unit_indels(DevModel M, UnitArgs U, const uint8_t *__restrict__ seq, const int32_t *__restrict__ len,
                                                    const uint8_t *__restrict__ genome, int64_t L, const int64_t *__restrict__ bounds,
                                                    uint8_t *__restrict__ work, int32_t cap, uint8_t *__restrict__ out,
                                                    int32_t *__restrict__ status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U.n) return;
    const int o = U.orientation, RL = M.RL;
    const Addr a = make_addr(U.seed, U.first_ordinal + (uint64_t)i, 0u);
    uint8_t *s = work + (size_t)i * cap;
    int n_s = len[i];
    for (int k = 0; k < n_s; ++k) s[k] = seq[(size_t)i * RL + k];
    const int64_t start = bounds[2 * i], end = bounds[2 * i + 1];
    int position = 0, rc = 0;
    // which tests fire at every step (the read's event process, iss_kernels.hip.h indel_events): bits 0-3 insertion of
    // letter slot x, bits 4-7 the deletion if the token is base b
    uint8_t *evm = s + cap - RL;  // (the tail of the read's scratch row)
    for (int k = 0; k < RL; ++k) evm[k] = 0;
    indel_events(M.ev_S + (size_t)o * M.ev_ns, M.ev_E + (size_t)o * M.ev_ns, M.ev_T + (size_t)o * M.ev_ns, M.del_thr + (size_t)o * RL * 4, M.ev_ns,
                 a, o, [&](int n, uint32_t mask) { evm[n] |= (uint8_t)mask; });
    for (int nucl = 0; nucl < RL - 1; ++nucl) {
        if (nucl >= n_s) continue;  // IndexError swallowed, :223-224 (position not advanced)
        const int cu = s[nucl] & ~0x20;
        if (cu == 'R' || cu == 'Y' || cu == 'W' || cu == 'S' || cu == 'M' || cu == 'K' || cu == 'H' || cu == 'B' || cu == 'V' ||
            cu == 'D' || cu == 'N') { ++position; continue; }  // :190-192
        const size_t en = (size_t)o * RL + position;
        for (int x = 0; x < 4; ++x) {  // :193-196, dict order
            if ((evm[position] >> x) & 1) {
                for (int z = n_s; z > position + 1; --z) s[z] = s[z - 1];  // insert after the base read
                s[position + 1] = M.ins_letter[en * 4 + x];
                ++n_s;
            }
        }
        const int bi = base_index(cu);
        if (bi < 0) { rc = 2; break; }  // deletions[position][X]: KeyError (:209)
        if ((evm[position] >> (4 + bi)) & 1) {
            for (int z = position; z + 1 < n_s; ++z) s[z] = s[z + 1];
            --n_s;
        }
        ++position;
    }
    if (!rc) {  // adjust_seq_length
        uint8_t *dst = out + (size_t)i * RL;
        for (int k = 0; k < RL && k < n_s; ++k) dst[k] = s[k];
        for (int t = 0; n_s + t < RL && !rc; ++t) {
            int c;
            if (o == 0) {
                c = end + t >= L ? 'A' : genome[end + t];
            } else {
                const int64_t idx = start - 1 - t;
                if (idx < 0) c = 'A';
                else if (idx >= L) { rc = 3; break; }
                else c = complement_ascii(genome[idx]);
            }
            dst[n_s + t] = (uint8_t)c;
        }
    }
    status[i] = rc;
}

}  // namespace iss
