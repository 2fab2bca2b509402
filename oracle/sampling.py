"""Bit-for-bit numpy restatement of the index bijection in nextbestpath_amd/csrc/common.h
(perm_bits / perm_index).  It stands in for torch.randperm(n)[:k] of the reference
(macarons/utility/macarons_utils.py:2837, next_best_path/utility/long_term_utils.py:446): the
reference draws from the unseeded global CPU generator, so there is no reference stream to
match -- what is pinned is "an exact-size subset, each index at most once"."""
import numpy as np

M32 = 0xFFFFFFFF


def perm_bits(n: int) -> int:
    b = 2
    while b < 32 and (1 << b) < n:
        b += 1
    return b


def perm_index(j, n: int, seed: int):
    """Vectorised: j uint array in [0, n) -> permuted indices in [0, n)."""
    j = np.asarray(j, dtype=np.uint64)
    if n == 0:
        return j.astype(np.int64)
    b = perm_bits(n)
    mask = np.uint64(M32 if b >= 32 else (1 << b) - 1)
    sh = np.uint64((b + 1) >> 1)
    v = j.copy()
    todo = np.ones(v.shape, dtype=bool)
    while todo.any():
        w = v[todo]
        for r in range(3):
            w = ((w * np.uint64(0x9E3779B1) + np.uint64((seed + r * 0x7F4A7C15) & M32)) & np.uint64(M32)) & mask
            w = w ^ (w >> sh)
        v[todo] = w
        todo[todo] = w >= n
    return v.astype(np.int64)
