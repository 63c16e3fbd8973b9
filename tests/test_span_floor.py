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
