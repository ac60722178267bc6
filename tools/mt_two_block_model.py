#!/usr/bin/env python3
"""The index arithmetic of mt_fill_two (tools/patches/mt_fill_two_blocks.patch: two MT19937 blocks per barrier, both written in
terms of the OLD block -- built and measured in round 5, slower than one block per barrier, not merged) lane by lane in Python,
against numpy's own generator: the tempered words of 2 k blocks from random states.  Runs on the CPU.
    python tools/mt_two_block_model.py"""
import numpy as np

U, L, MAG = 0x80000000, 0x7FFFFFFF, 0x9908B0DF


def T(a, b):
    y = (a & U) | (b & L)
    return (y >> 1) ^ (MAG if (y & 1) else 0)


def temper(y):
    y ^= y >> 11
    y ^= (y << 7) & 0x9D2C5680
    y ^= (y << 15) & 0xEFC60000
    y ^= y >> 18
    return y & 0xFFFFFFFF


def new_of_old(o, k):  # mt_new_of_old
    v = 0
    while k >= 227:
        v ^= T(o[k], o[k + 1])
        k -= 227
    return v ^ o[k + 397] ^ T(o[k], o[k + 1])


def fill_two(o):
    """-> (m, tempered words of the two blocks); every lane as the kernel computes it (clamped reads included)"""
    m = [None] * 624
    ob0 = [None] * 624
    ob1 = [None] * 624
    for tid in range(320):
        if tid < 227:
            j = tid
            o0, o1, o2 = o[j], o[j + 1], o[j + 2]
            p0, p1, p2 = o[227 + j], o[228 + j], o[229 + j]
            je = j if j < 169 else 168
            je1 = j if j < 168 else 167
            q0, q1, q2 = o[454 + je], o[455 + je], o[456 + je1]
            a = o[j + 397] ^ T(o0, o1)
            c = a ^ T(p0, p1)
            e = c ^ T(q0, q1)
            a1 = o[j + 398 if j < 226 else 623] ^ T(o1, o2)
            c1 = a1 ^ T(p1, p2)
            e1 = c1 ^ T(q1, q2)
            b = j + 170 if j < 57 else j - 57
            b3 = 0 if j < 57 else j + 397
            b3c = b3 if b3 < 623 else 622
            x = o[b + 397] ^ T(o[b], o[b + 1]) ^ T(o[b + 227], o[b + 228])
            f3 = T(o[b3c], o[b3c + 1])
            x ^= 0 if j < 57 else f3
            ma = x ^ T(a, a1)
            mc = ma ^ T(c, c1)
            me = mc ^ T(e, e1)
            ob0[j] = temper(a)
            ob0[227 + j] = temper(c)
            if j < 169:
                ob0[454 + j] = temper(e)
            if j < 226:
                m[j], m[227 + j] = ma, mc
                ob1[j], ob1[227 + j] = temper(ma), temper(mc)
                if j < 168:
                    m[454 + j] = me
                    ob1[454 + j] = temper(me)
        elif tid == 256:  # (the fifth wavefront: fifteen lanes' values, combined -- here by one)
            K = [0, 1, 169, 170, 396, 397, 566, 168, 395, 565, 622, 226, 227, 453, 454]
            n0, n1, n169, n170, n396, n397, n566, n168, n395, n565, n622, n226, n227, n453, n454 = [new_of_old(o, k) for k in K]
            n623 = n396 ^ T(o[623], n0)
            m0 = n397 ^ T(n0, n1)
            m396 = n566 ^ T(n169, n170) ^ T(n396, n397)
            m623 = m396 ^ T(n623, m0)
            m395 = n565 ^ T(n168, n169) ^ T(n395, n396)
            m622 = m395 ^ T(n622, n623)
            m226 = n623 ^ T(n226, n227)
            m453 = m226 ^ T(n453, n454)
            ob0[623] = temper(n623)
            for at, w in ((226, m226), (453, m453), (622, m622), (623, m623)):
                m[at] = w
                ob1[at] = temper(w)
    assert None not in m and None not in ob0 and None not in ob1
    return m, ob0 + ob1


def main():
    for seed in (1, 42, 2**32 - 1, 987654321):
        rs = np.random.RandomState(seed)
        st = rs.get_state()
        assert st[2] == 624  # (at a block boundary: the next output needs a twist)
        o = [int(x) for x in st[1]]
        want = rs.randint(0, 2**32, size=6 * 624, dtype=np.uint64)
        got = []
        for _ in range(3):
            o, words = fill_two(o)
            got += words
        assert [int(x) for x in want] == got, seed
    print("mt_fill_two's index arithmetic == numpy's MT19937 (4 seeds x 6 blocks)")


if __name__ == "__main__":
    main()
