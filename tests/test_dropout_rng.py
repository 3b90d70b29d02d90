"""Joint statistics of the counter-based dropout generator (csrc/vb_rt.h: vb_dropout_bits8), on the independent numpy
statement of it that tests/test_kernels.py pins the kernels' masks to bit for bit.  Words 2 and 3 of a group are derived from
words 0 and 1 with one 24-bit multiply each, so pairwise independence does not by itself give joint independence of the eight
keep bits of a group: here the 3- and 4-way keep patterns WITHIN a group -- including the triples and quadruples
that combine a derived word with the words it was derived from -- are held against the product of Bernoulli marginals."""
import itertools

import numpy as np

from test_kernels import generator_keep


def _chi2_patterns(keep, idx, p_keep):
    """chi-square of the 2^k keep patterns of the elements `idx` of each group against independent Bernoulli(p_keep)"""
    k = len(idx)
    code = np.zeros(keep.shape[0], dtype=np.int64)
    for b, e in enumerate(idx):
        code |= keep[:, e].astype(np.int64) << b
    obs = np.bincount(code, minlength=1 << k).astype(np.float64)
    n = keep.shape[0]
    exp = np.array([n * np.prod([p_keep if (c >> b) & 1 else 1 - p_keep for b in range(k)]) for c in range(1 << k)])
    return float(((obs - exp) ** 2 / exp).sum()), (1 << k) - 1


def test_keep_patterns_within_a_group_are_jointly_bernoulli():
    n = 1 << 20
    p = 0.1
    thresh = min(int(p * 65536.0 + 0.5), 65535)
    p_keep = 1.0 - thresh / 65536.0
    keep = generator_keep(np.arange(n), p, (11 << 32) | 12345, 9)            # [n, 8] bool
    assert abs(keep.mean() - p_keep) < 3e-4
    # element e sits in 16-bit lane e & 1 of word e >> 1; words 2, 3 are derived from words 0, 1
    worst = 0.0
    for k in (3, 4):
        for idx in itertools.combinations(range(8), k):
            chi2, dof = _chi2_patterns(keep, idx, p_keep)
            # chi-square with dof degrees of freedom: mean dof, sd sqrt(2 dof); 126 subsets tested -> allow 5.5 sd
            z = (chi2 - dof) / np.sqrt(2.0 * dof)
            worst = max(worst, z)
            assert z < 5.5, (idx, chi2, dof)
    # all eight bits at once: 256 patterns
    chi2, dof = _chi2_patterns(keep, tuple(range(8)), p_keep)
    assert (chi2 - dof) / np.sqrt(2.0 * dof) < 5.5, (chi2, dof)


def test_keep_patterns_across_neighbouring_groups():
    """the same element of consecutive groups (consecutive counters of the keyed mixer): 4-way patterns over groups g .. g+3"""
    n = 1 << 20
    p = 0.1
    thresh = min(int(p * 65536.0 + 0.5), 65535)
    p_keep = 1.0 - thresh / 65536.0
    keep = generator_keep(np.arange(n + 3), p, (3 << 32) | 777, 21)
    for e in range(8):
        stacked = np.stack([keep[i:n + i, e] for i in range(4)], 1)
        chi2, dof = _chi2_patterns(stacked, (0, 1, 2, 3), p_keep)
        assert (chi2 - dof) / np.sqrt(2.0 * dof) < 5.5, (e, chi2)
