"""The tile kernel's scanline solver replaces integer division by one fp32 reciprocal multiply (geograster.hip,
`edge_floor`): floor((E + 0.5) * rcp(m)) == floor(E / m) for integer E and 1 <= m <= GR_FLOOR_NOCORR_MAX wherever the result
lies in the clamp range [-66, 65].  Checked here on the CPU, in fp32 arithmetic, against exact integer division -- with
the reciprocal perturbed by up to 3 ulp on either side of the correctly rounded value (the hardware's v_rcp_f32 is
specified to 1 ulp), on the values of E that sit closest to a change of the quotient (multiples of m, their neighbours,
and the value just below the next multiple)."""
import numpy as np

GR_FLOOR_NOCORR_MAX = 16000  # geograypher_amd/csrc/geograster.hip


def _floor_fp32(E, m, ulps, sign):
    mf = m.astype(np.float32)
    r = (np.float32(1.0) / mf).astype(np.float32)
    for _ in range(ulps):
        r = np.nextafter(r, np.float32(np.inf * sign)).astype(np.float32)
    g = ((E.astype(np.float32) + np.float32(0.5)).astype(np.float32) * r).astype(np.float32)
    return np.floor(np.clip(g, np.float32(-66), np.float32(65))).astype(np.int64)


def test_float_floor_equals_integer_floor_division():
    ks = np.arange(-68, 69, dtype=np.int64)
    off = np.array([-1, 0, 1], dtype=np.int64)[None, None, :]
    checked = 0
    for m0 in range(1, GR_FLOOR_NOCORR_MAX + 1, 1000):
        m = np.arange(m0, min(m0 + 1000, GR_FLOOR_NOCORR_MAX + 1), dtype=np.int64)[:, None, None]
        E = np.concatenate([ks[None, :, None] * m + off, ks[None, :, None] * m + (m - 1) + 0 * off], axis=2)
        want = np.floor_divide(E, m)
        inside = (want >= -66) & (want <= 65)
        for ulps in (0, 1, 3):
            for sign in (-1, 1):
                got = _floor_fp32(E, m, ulps, sign)
                assert np.array_equal(got[inside], want[inside]), (m0, ulps, sign)
                # beyond the clamp range the proposal must stay on its side
                assert np.all(got[want > 65] == 65) and np.all(got[want < -66] == -66)
        checked += int(inside.sum())
    assert checked > 10_000_000


def test_float_floor_random_numerators():
    rng = np.random.default_rng(0)
    m = rng.integers(1, GR_FLOOR_NOCORR_MAX + 1, size=4_000_000)
    E = (rng.uniform(-67, 67, size=m.size) * m).astype(np.int64)
    want = np.floor_divide(E, m)
    inside = (want >= -66) & (want <= 65)
    for ulps, sign in ((0, 1), (3, -1), (3, 1)):
        got = _floor_fp32(E, m, ulps, sign)
        assert np.array_equal(got[inside], want[inside])


def test_parallel_edge_is_unconstrained_or_empty():
    """m == 0 (an edge parallel to the scanline): +-inf, clamped by the sign of E; never NaN."""
    E = np.array([-5, -1, 0, 1, 7], dtype=np.int64)
    with np.errstate(divide="ignore"):
        r = np.float32(1.0) / np.float32(0.0)
    g = (E.astype(np.float32) + np.float32(0.5)) * r
    assert not np.any(np.isnan(g))
    fl = np.floor(np.clip(g, -66, 65)).astype(int)
    assert list(fl) == [-66, -66, 65, 65, 65]


# ---- the INTERCEPT form of a 48-byte entry (binning.hip, pq_form; raster_tile.hip, span_solve_pq) ---------------------------
# p = (C + 0.5) * rcp(m), q = b * rcp(m) once per entry; per scanline floor(fma(q, y, p)) must be floor((C + b y) / m) for
# every row y of a tile (|y| <= 32) and slopes up to GR_FLOOR_NOCORR_MAX, wherever the quotient lies inside the clamp.

def _floor_pq(C, b, m, y, ulps, sign):
    r = (np.float32(1.0) / m.astype(np.float32)).astype(np.float32)
    for _ in range(ulps):
        r = np.nextafter(r, np.float32(np.inf * sign)).astype(np.float32)
    p = ((C.astype(np.float32) + np.float32(0.5)).astype(np.float32) * r).astype(np.float32)
    q = (b.astype(np.float32) * r).astype(np.float32)
    # v_fma_f32: the exact sum (64-bit mantissa: a 24-bit p and a 30-bit product at most 2^37 apart), rounded once
    g = (q.astype(np.longdouble) * y.astype(np.longdouble) + p.astype(np.longdouble)).astype(np.float32)
    return np.floor(np.clip(g, np.float32(-34), np.float32(33))).astype(np.int64)


def _check_pq(E, b, m, y, label):
    C = E - b * y
    assert np.abs(C).max() < 2 ** 23  # C + 0.5 is exact in fp32 (build_entry keeps the constants of such faces below 2^23)
    want = np.floor_divide(E, m)
    inside = (want >= -34) & (want <= 32)
    for ulps, sign in ((0, 1), (1, -1), (1, 1), (3, -1), (3, 1)):
        got = _floor_pq(*np.broadcast_arrays(C, b, m, y), ulps, sign)
        w = np.broadcast_to(want, got.shape)
        ins = np.broadcast_to(inside, got.shape)
        assert np.array_equal(got[ins], w[ins]), (label, ulps, sign)
        assert np.all(got[w > 32] >= 33) and np.all(got[w < -34] == -34), (label, ulps, sign)
    return int(np.broadcast_to(inside, np.broadcast_shapes(E.shape, b.shape, m.shape, y.shape)).sum())


def test_intercept_form_floor_on_the_quotient_boundaries():
    ks = np.arange(-36, 36, dtype=np.int64)[None, :, None, None]
    rng = np.random.default_rng(1)
    checked = 0
    for m0 in range(1, GR_FLOOR_NOCORR_MAX + 1, 2000):
        m = np.arange(m0, min(m0 + 2000, GR_FLOOR_NOCORR_MAX + 1), dtype=np.int64)[:, None, None, None]
        # numerators next to a change of the quotient: k m - 1, k m, k m + m - 1
        E = np.concatenate([ks * m - 1, ks * m, ks * m + (m - 1)], axis=2)
        # rows and row slopes: the extremes (largest |q y|: the largest cancellation against p) and random ones
        y = np.array([-32, 31, -32, 31, -17, 9], dtype=np.int64)[None, None, None, :]
        b = np.concatenate([np.array([GR_FLOOR_NOCORR_MAX, GR_FLOOR_NOCORR_MAX, -GR_FLOOR_NOCORR_MAX, -GR_FLOOR_NOCORR_MAX]),
                            rng.integers(-GR_FLOOR_NOCORR_MAX, GR_FLOOR_NOCORR_MAX + 1, size=2)])[None, None, None, :]
        checked += _check_pq(E, b, m, y, m0)
    assert checked > 15_000_000


def test_intercept_form_floor_random():
    rng = np.random.default_rng(2)
    n = 4_000_000
    m = rng.integers(1, GR_FLOOR_NOCORR_MAX + 1, size=n)
    b = rng.integers(-GR_FLOOR_NOCORR_MAX, GR_FLOOR_NOCORR_MAX + 1, size=n)
    y = rng.integers(-32, 32, size=n)
    E = (rng.uniform(-36, 36, size=n) * m).astype(np.int64)
    assert _check_pq(E, b, m, y, "random") > 3_000_000


def test_intercept_form_parallel_edge():
    """a = 0 (only the middle edge can be): p, q = (C + 0.5, b) * 2^20 -- g has the sign of E = C + b y and is cut by the clamp."""
    rng = np.random.default_rng(3)
    C = rng.integers(-(2 ** 23) + 1, 2 ** 23, size=200_000)
    b = rng.integers(-32767, 32768, size=C.size)
    y = rng.integers(-32, 32, size=C.size)
    C[:1000] = -b[:1000] * y[:1000] + rng.integers(-1, 1, size=1000)  # E = -1 or 0: either side of the edge
    s = np.float32(1048576.0)
    p = ((C.astype(np.float32) + np.float32(0.5)) * s).astype(np.float32)
    q = (b.astype(np.float32) * s).astype(np.float32)
    g = (q.astype(np.longdouble) * y + p.astype(np.longdouble)).astype(np.float32)
    assert not np.any(np.isnan(g))
    fl = np.floor(np.clip(g, np.float32(-34), np.float32(33))).astype(np.int64)
    E = C + b * y
    assert np.all(fl[E >= 0] == 33) and np.all(fl[E < 0] == -34)
