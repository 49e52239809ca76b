"""CPU property tests (hypothesis) on the oracle's small-matrix primitives and on the two mathematical facts the HIP fast path
relies on: (1) ten sequential scalar updates equal the batch update when R is diagonal; (2) the posterior covariance stays
symmetric, so storing its upper triangle loses nothing beyond rounding."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import c_oracle as orc

SEL = [0, 1, 2, 5, 6, 7, 8, 9, 10, 11]
angles = st.floats(min_value=-3.1, max_value=3.1, allow_nan=False)


@settings(max_examples=60, deadline=None)
@given(angles, angles, angles)
def test_rotation_is_orthonormal_with_unit_determinant(a, b, c):
    R = orc.rotation(a, b, c)
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-13
    assert abs(np.linalg.det(R) - 1.0) < 1e-13


def _spd(rng, n, scale):
    A = rng.normal(size=(n, n))
    return A @ A.T * scale + np.eye(n) * scale * 0.1


@settings(max_examples=40, deadline=None)
@given(st.integers(min_value=0, max_value=10 ** 6))
def test_sequential_scalar_updates_equal_batch_update_for_diagonal_R(seed):
    rng = np.random.default_rng(seed)
    P = _spd(rng, 12, 0.05)
    x = rng.normal(size=12)
    z = rng.normal(size=10)
    Rd = rng.uniform(1e-4, 50.0, size=10)
    xb, Pb, K, kg, st_ = orc.update(x, P, z, np.diag(Rd))
    assert st_ == 0
    xs, Ps = x.copy(), P.copy()
    for a, s in enumerate(SEL):
        S = Ps[s, s] + Rd[a]
        k = Ps[:, s] / S
        xs = xs + k * (z[a] - xs[s])
        Ps = Ps - np.outer(k, Ps[s, :])
    assert np.abs(xs - xb).max() < 1e-9 * max(1.0, np.abs(xb).max())
    assert np.abs(Ps - Pb).max() < 1e-9 * np.abs(Pb).max()
    # the reference's (I - KH)P form stays symmetric to rounding, so the upper triangle carries all of it
    assert np.abs(Pb - Pb.T).max() < 1e-12 * np.abs(Pb).max()


@settings(max_examples=30, deadline=None)
@given(st.integers(min_value=0, max_value=10 ** 6))
def test_update_shrinks_measured_variances_and_gain_trace_matches(seed):
    rng = np.random.default_rng(seed)
    P = _spd(rng, 12, 0.02)
    xb, Pb, K, kg, st_ = orc.update(rng.normal(size=12), P, rng.normal(size=10), np.diag(rng.uniform(1e-3, 1.0, 10)))
    assert st_ == 0
    assert all(Pb[s, s] <= P[s, s] + 1e-12 for s in SEL)
    assert abs(kg - sum(K[a, a] for a in range(10))) < 1e-12         # np.trace of the 12x10 gain (kalman_filter.py:174)


@settings(max_examples=30, deadline=None)
@given(st.integers(min_value=0, max_value=10 ** 6))
def test_next_state_rotates_feet_and_conserves_linear_structure(seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=12) * 0.3
    p, f = rng.normal(size=12) * 0.3, rng.normal(size=12) * 10
    xn, prot = orc.next_state(x, p, f)
    R = orc.rotation(*x[0:3])
    assert np.abs(prot.reshape(4, 3) - p.reshape(4, 3) @ R.T).max() < 1e-13      # feet in the world frame, in place
    assert np.abs(xn[3:6] - (x[3:6] + 0.01 * x[9:12])).max() < 1e-14             # position integrates the prior velocity
    g = np.array([0, 0, -9.81])
    assert np.abs(xn[9:12] - (x[9:12] + 0.01 * (f.reshape(4, 3).sum(0) / 8.8 + g))).max() < 1e-12
    assert np.abs(xn[0:3] - x[0:3]).max() == 0.0                                  # generic angles: int64 A block is zero
