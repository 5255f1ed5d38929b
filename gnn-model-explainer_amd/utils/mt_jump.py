"""Jump-ahead for mt19937 (the engine of torch's CPU generator, whose n x n normal_ draw seeds every target's edge mask: explain.py:645-652).

The engine walk of a target's mask stream is a serial chain - one 624-word block update after the other, n^2 / 624 of them: 3.7 million updates
(1.7 s of one workgroup) for the 47 913-node sub-graph of BA-House x100k.  mt19937 is LINEAR over GF(2): the raw word sequence x_k satisfies

    x_{m + J} = XOR over the set coefficients g_i of x_{m + i},      g(x) = x^J mod phi(x),   m >= 1,

phi = the characteristic polynomial of the one-word transition (degree 19 937; Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer, "Efficient
Jump Ahead for F2-Linear Random Number Generators", INFORMS J. Comput. 20(3), 2008 - the published algorithm restated, no code of theirs).  So the
state 2^23 draws ahead is 10 k XORs of sliding 624-word windows over 33 blocks of the plain sequence - 0.1 ms on one compute unit - and a target's
stream is cut into segments of J draws that walk IN PARALLEL (k_mt_segment_starts / k_mt_edge_words_seg, csrc/gnnx_xl.hpp).

This module is the host side: phi by Berlekamp-Massey on one bit plane of the sequence, g = x^J mod phi by square-and-multiply on Python integers
(bit i = coefficient of x^i), a NumPy reference of the jump (tests), and the committed polynomial of the stride the kernels use
(mt_jump_poly_13440_blocks.npy, 624 uint32 words; tests/test_mt_jump.py recomputes it).  Nothing here runs per target or per batch."""
import os

import numpy as np

MT_N, MT_M = 624, 397
DEG = 19937
JUMP = 13440 * MT_N                    # stride of the segmented walk: 8 386 560 draws (~2^23) = 13 440 whole blocks (segments start on block boundaries)
_HERE = os.path.dirname(os.path.abspath(__file__))
POLY_FILE = os.path.join(_HERE, "mt_jump_poly_13440_blocks.npy")


def mt_words(seed, count):
    """x_0 .. x_{count-1}: the seeded state (at::mt19937::init_with_uint32) followed by the recurrence's raw (untempered) words"""
    x = np.zeros(max(count, MT_N), np.uint64)
    x[0] = seed & 0xffffffff
    for j in range(1, MT_N):
        x[j] = (1812433253 * (int(x[j - 1]) ^ (int(x[j - 1]) >> 30)) + j) & 0xffffffff
    xs = [int(v) for v in x[:MT_N]] + [0] * max(0, count - MT_N)
    for k in range(MT_N, count):
        u, v = xs[k - MT_N], xs[k - MT_N + 1]
        y = (u & 0x80000000) | (v & 0x7fffffff)
        xs[k] = xs[k - MT_N + MT_M] ^ (y >> 1) ^ (0x9908b0df if (v & 1) else 0)
    return np.asarray(xs[:count], np.uint64).astype(np.uint32)


def charpoly():
    """phi(x) as a Python int (bit i = coefficient of x^i), degree 19 937: Berlekamp-Massey over GF(2) on bit 0 of x_1, x_2, ..."""
    n = 2 * DEG + 64
    bits = [int(w) & 1 for w in mt_words(5489, n + 1)[1:]]
    C, B, L, m, R = 1, 1, 0, 1, 0          # R: bit i = s_{N - i} (the reversed history)
    for N in range(n):
        R = (R << 1) | bits[N]
        d = (C & R).bit_count() & 1
        if d:
            T = C
            C ^= B << m
            if 2 * L <= N:
                L, B, m = N + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    assert L == DEG, L
    # C(x) = 1 + c_1 x + ... + c_L x^L is the connection polynomial (s_N = sum c_i s_{N-i}); phi(x) = x^L C(1 / x)
    return sum(((C >> i) & 1) << (L - i) for i in range(L + 1))


def _square(a):
    """a(x)^2 over GF(2): every bit moves to twice its position"""
    return int("0".join(bin(a)[2:]), 2)


def _mod(a, phi):
    d = phi.bit_length() - 1
    while a.bit_length() - 1 >= d:
        a ^= phi << (a.bit_length() - 1 - d)
    return a


def jump_poly(J, phi=None):
    """g(x) = x^J mod phi(x) as an int"""
    phi = charpoly() if phi is None else phi
    r = 1
    for b in bin(J)[2:]:
        r = _mod(_square(r), phi)
        if b == "1":
            r = _mod(r << 1, phi)
    return r


def poly_words(g):
    """-> uint32 [624]: bit i of the polynomial = bit (i & 31) of word i >> 5 (what the device kernel walks)"""
    return np.frombuffer(g.to_bytes(MT_N * 4, "little"), np.uint32).copy()


LEVELS, RADIX = 3, 4      # the segment starts of a target form a radix-4 tree of jumps with the strides J, 4 J, 16 J (csrc/gnnx_xl.hpp: k_mt_segment_starts)


def jump_polys(poly_words_J, levels=LEVELS, phi=None):
    """uint32 [levels][624]: x^(RADIX^l J) mod phi for l = 0 .. levels - 1, from the words of g = x^J mod phi (g^4, g^16: two squarings each)"""
    phi = charpoly() if (phi is None and levels > 1) else phi
    g = int.from_bytes(np.ascontiguousarray(poly_words_J, np.uint32).tobytes(), "little")
    out = [poly_words(g)]
    for _ in range(1, levels):
        for _ in range(RADIX.bit_length() - 1):
            g = _mod(_square(g), phi)
        out.append(poly_words(g))
    return np.stack(out)


def load_jump_poly():
    """The committed polynomial of stride JUMP (recomputed and compared by tests/test_mt_jump.py)."""
    return np.load(POLY_FILE)


def jump_apply(window, poly):
    """NumPy reference of the device jump: window = x_m .. x_{m+623} (m >= 1) -> x_{m+J} .. x_{m+J+623}"""
    need = DEG + MT_N
    y = [int(v) for v in window] + [0] * (need - MT_N)
    for k in range(MT_N, need):
        u, v = y[k - MT_N], y[k - MT_N + 1]
        yy = (u & 0x80000000) | (v & 0x7fffffff)
        y[k] = y[k - MT_N + MT_M] ^ (yy >> 1) ^ (0x9908b0df if (v & 1) else 0)
    y = np.asarray(y, np.uint32)
    out = np.zeros(MT_N, np.uint32)
    bits = np.unpackbits(poly.view(np.uint8), bitorder="little")[:DEG]
    for i in np.nonzero(bits)[0]:
        out ^= y[i:i + MT_N]
    return out


if __name__ == "__main__":
    import time
    t0 = time.time()
    phi = charpoly()
    print("phi: degree", phi.bit_length() - 1, "terms", phi.bit_count(), f"{time.time() - t0:.1f} s")
    g = jump_poly(JUMP, phi)
    print("g: terms", g.bit_count(), f"{time.time() - t0:.1f} s")
    np.save(POLY_FILE, poly_words(g))
    print("wrote", POLY_FILE)
