"""CPU: the integer identities the HIP kernels rely on, exhaustively."""
import numpy as np


def test_unorm8_roundtrip_exact():
    k = np.arange(256, dtype=np.float32)
    assert np.all(((k / np.float32(255.0)).astype(np.float32) * np.float32(255.0)).astype(np.float32) == k)


def test_floor_times_point4_equals_floor_2v_over_5():
    v = np.arange(-(1 << 22), (1 << 22) + 1, dtype=np.int64)
    a = np.floor(v.astype(np.float32) * np.float32(0.4)).astype(np.int64)
    assert np.array_equal(a, np.floor_divide(2 * v, 5))


def test_trunc_div_point4_equals_trunc_5w_over_2_and_times_2_5():
    w = np.arange(-32768, 32768, dtype=np.int64)
    c = np.trunc(w.astype(np.float32) / np.float32(0.4)).astype(np.int64)
    d = np.sign(w) * ((5 * np.abs(w)) // 2)
    assert np.array_equal(c, d)
    # the kernel's form: trunc(float(w) * 2.5f)
    e = np.trunc(w.astype(np.float32) * np.float32(2.5)).astype(np.int64)
    assert np.array_equal(e, d)
    # and the integer form of its rare path
    f = (5 * w + (w < 0)) >> 1
    assert np.array_equal(f, d)


def test_half_pel_bias_equals_integer_rounding():
    a = np.arange(256)
    A, B = np.meshgrid(a, a)
    f = ((A.astype(np.float32) / np.float32(255) + B.astype(np.float32) / np.float32(255) + np.float32(1 / 512)) / np.float32(2))
    got = np.rint(np.clip(f, 0, 1).astype(np.float64) * 255).astype(int)
    assert np.array_equal(got, (A + B + 1) >> 1)


def test_avg4_via_three_rounded_averages():
    """(a+b+c+d+2)>>2 == avg(avg(a,b),avg(c,d)) - (((a^b)|(c^d)) & (x^y) & 1), avg rounding up."""
    rng = np.random.default_rng(0)
    a, b, c, d = (rng.integers(0, 256, size=4_000_000) for _ in range(4))
    x, y = (a + b + 1) >> 1, (c + d + 1) >> 1
    r = ((x + y + 1) >> 1) - (((a ^ b) | (c ^ d)) & (x ^ y) & 1)
    assert np.array_equal(r, (a + b + c + d + 2) >> 2)
    # degenerate operand duplication used for the non-half-pel cases
    assert np.array_equal(((a + a + 1) >> 1), a)


def test_trunc_div256_as_biased_shift():
    t = np.arange(-(1 << 21), 1 << 21, dtype=np.int64)
    assert np.array_equal((t + ((t >> 63) & 255)) >> 8, np.sign(t) * (np.abs(t) // 256))


def test_dequant_closed_form_matches_shader_steps():
    """(f - (f>0)) | 1 == 'if even: f -= (f>0) ? 1 : -1' including f == 0 -> +1."""
    f = np.arange(-5000, 5000)
    ref = f.copy()
    even = (ref % 2) == 0
    ref[even] -= np.where(ref[even] > 0, 1, -1)
    assert np.array_equal((f - (f > 0)) | 1, ref)
