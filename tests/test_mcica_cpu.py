"""CPU tests of the host side of the McICA path: the KISS jump-ahead table the kernels use to enter the reference's
sequential random stream (kissvec, SW:1900-1932) at an arbitrary draw, checked against stepping the generator."""
import ctypes as C

import numpy as np
import pytest

M = 0xFFFFFFFF
P3, P4 = 18000 * 65536 - 1, 30903 * 65536 - 1


def step(s):
    s1, s2, s3, s4 = s
    s1 = (69069 * s1 + 1327217885) & M
    s2 ^= (s2 << 13) & M
    s2 ^= s2 >> 17
    s2 ^= (s2 << 5) & M
    s3 = (18000 * (s3 & 65535) + (s3 >> 16)) & M
    s4 = (30903 * (s4 & 65535) + (s4 >> 16)) & M
    return s1, s2, s3, s4


def jump(s, J):
    s1, s2, s3, s4 = s
    r = 0
    for j in range(32):
        if (s2 >> j) & 1:
            r ^= int(J[2 + j])
    return (int(J[0]) * s1 + int(J[1])) & M, r, (s3 * int(J[34])) % P3, (s4 * int(J[35])) % P4


def table(lib, nlay, change_seed, nsub):
    f = lib._ZN5rrtmg15kiss_jump_tableEiiiPj      # rrtmg::kiss_jump_table(int, int, int, unsigned*), host code
    t = np.zeros((nsub, 36), dtype=np.uint32)
    f(C.c_int(nlay), C.c_int(change_seed), C.c_int(nsub), t.ctypes.data_as(C.c_void_p))
    return t


@pytest.mark.parametrize("nlay,change_seed,nsub", [(41, 1, 112), (53, 150, 140), (4, 1, 112), (201, 150, 140)])
def test_kiss_jump_table_equals_sequential_stepping(wa, nlay, change_seed, nsub):
    lib = wa.engine.load_library()
    tab = table(lib, nlay, change_seed, nsub)
    rng = np.random.default_rng(nlay)
    for trial in range(3):
        # seeds as the generator forms them: 9-digit fractions of a pressure in Pa (SW:1696-1699); 0 is possible
        s0 = tuple(int(x) for x in rng.integers(0, 10**9, size=4))
        if trial == 2:
            s0 = (0, s0[1], 0, 0)
        s, n = s0, 0
        for g in range(nsub):
            target = change_seed + g * nlay
            while n < target:
                s = step(s)
                n += 1
            assert jump(s0, tab[g]) == s, (g, s0)


def test_jump_table_first_subcolumn_is_change_seed_steps(wa):
    lib = wa.engine.load_library()
    tab = table(lib, 41, 1, 2)
    s0 = (123456789, 987654321, 555555555, 111111111)
    assert jump(s0, tab[0]) == step(s0)
