"""GPU unit test of the lane-parallel 6x6 LDLT (csrc/klt_common.h: ldlt6_factor / ldlt6_solve) against the oracle's
restatement of Eigen's pivoted LDLT, bit for bit — including the matrices a tracker rarely produces."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def device_solve(ftk, a, b):
    from feature_tracker_amd import _native as N
    ctx = ftk.default_context()
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 36)
    b = np.ascontiguousarray(b, np.float32).reshape(-1, 6)
    x = np.zeros_like(b)
    N.check(N.lib().ftk_ldlt6_solve(ctx.handle, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), a.shape[0]),
            ctx.handle)
    return x


def compare(ftk, oracle, mats, rhs):
    with np.errstate(all="ignore"):
        got = device_solve(ftk, mats, rhs)
        want = np.stack([oracle.ldlt_solve(m.reshape(6, 6), r) for m, r in zip(mats.reshape(-1, 36), rhs.reshape(-1, 6))])
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    bad = np.flatnonzero(~same.all(axis=1))
    assert bad.size == 0, (bad[:5], got[bad[:1]], want[bad[:1]])


def sym(rs, n, scale):
    j = (rs.standard_normal((n, 6, 9)) * scale[None, :, None]).astype(np.float32)
    return np.einsum("nik,njk->nij", j, j).astype(np.float32)


def test_random_spd_systems_with_wide_dynamic_range(ftk, oracle):
    rs = np.random.RandomState(0)
    # affine-like conditioning: rows scaled like (x^2, xy, x, ...) with x ~ 1e2-1e3 -> pivoting on every system
    for scale in (np.ones(6), np.float64([1e5, 1e5, 3e2, 3e2, 1, 1]), np.float64([1, 1e3, 1e-3, 1e6, 1e-6, 10])):
        a = sym(rs, 400, scale)
        a = ((a + a.transpose(0, 2, 1)) * np.float32(0.5)).astype(np.float32)
        compare(ftk, oracle, a, rs.standard_normal((400, 6)).astype(np.float32))


def test_ties_zero_pivots_rank_deficiency_and_non_finite(ftk, oracle):
    rs = np.random.RandomState(1)
    cases, rhs = [], []

    def add(m, r=None):
        m = np.asarray(m, np.float32).reshape(6, 6)
        cases.append(((m + m.T) * np.float32(0.5)).astype(np.float32) if not np.isnan(m).any() else m)
        rhs.append(rs.standard_normal(6).astype(np.float32) if r is None else np.asarray(r, np.float32))

    add(np.zeros((6, 6)))                                   # degenerate: every pivot zero
    add(np.eye(6))                                          # all diagonals tie: first maximum wins, no swaps
    add(np.diag([2, 2, 3, 3, 1, 1]))                        # ties among the maxima
    add(np.diag([1, 0, 2, 0, 3, 0]))                        # zero pivots after the non-zero ones
    for _ in range(200):                                    # rank-deficient: J^T J with fewer rows than columns
        k = rs.randint(1, 6)
        j = rs.standard_normal((k, 6)).astype(np.float32)
        add(j.T @ j)
    for _ in range(100):                                    # equal diagonal magnitudes with random off-diagonals (indefinite)
        m = rs.standard_normal((6, 6)).astype(np.float32)
        m = m + m.T
        np.fill_diagonal(m, rs.choice([-2.0, 2.0], 6))
        add(m)
    for _ in range(100):                                    # duplicated rows / columns -> exact zero pivots mid-way
        j = rs.standard_normal((8, 6)).astype(np.float32)
        j[:, rs.randint(6)] = j[:, rs.randint(6)]
        add(j.T @ j)
    m = sym(rs, 1, np.ones(6))[0]
    m[2, 2] = np.nan
    add(m)                                                  # NaN on the diagonal
    m = sym(rs, 1, np.ones(6))[0]
    m[0, 0] = np.nan
    add(m)                                                  # NaN in the first pivot candidate
    m = sym(rs, 1, np.ones(6))[0]
    m[1, 4] = m[4, 1] = np.inf
    add(m)                                                  # infinite off-diagonal
    add(sym(rs, 1, np.ones(6))[0], [np.nan, 1, 2, 3, 4, 5])  # NaN right-hand side
    compare(ftk, oracle, np.stack(cases), np.stack(rhs))
