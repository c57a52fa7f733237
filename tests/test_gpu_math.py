"""The kernels' powf / atanf against the oracle's, element by element, through rt_eval_math.

The frames of the parity tests only visit the arguments their scenes produce; here the two functions are compared
directly on a few million arguments: the renderer's operating range, the whole binary32 range, arguments around the
table-interval and rounding boundaries of the algorithm, subnormals and every special case.  Bar: identical bits
(NaN compared as NaN).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def assert_same_floats(got, want, what):
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan), f"{what}: NaN positions differ"
    gb, wb = got.view(np.uint32)[~nan], want.view(np.uint32)[~nan]
    bad = np.nonzero(gb != wb)[0]
    assert bad.size == 0, f"{what}: {bad.size} of {gb.size} differ, first at {bad[:5]}: got {got[~nan][bad[:5]]}, want {want[~nan][bad[:5]]}"


def oracle_powf(orc, x, y):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty_like(x)
    orc.load().orc_powf_det_array(x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size)
    return out


def oracle_atanf(orc, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    orc.load().orc_atanf_det_array(x.ctypes.data, out.ctypes.data, x.size)
    return out


SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-40, 1.17549435e-38,
                    3.4028235e38, -3.4028235e38, 16777216.0, 16777217.0, -16777216.0, 0.4166, 0.4545, 0.8, 1.0000001, 0.99999994,
                    1.4609375, 1.4609374, 0.73046875, 127.0, -149.0, -150.0, 1.0 / 3.0], dtype=np.float32)


def test_powf_on_the_renderer_range(pkg, orc, ctx):
    """Bases as light sums and clamped cosines produce them, exponents of the reference's scenes (gamma 0.4166 / 0.4545,
    tri-planar 0.8, its shininess values)."""
    rng = np.random.default_rng(11)
    n = 1 << 21
    x = np.concatenate([rng.uniform(0, 1, n), rng.uniform(0, 8, n), 10 ** rng.uniform(-8, 3, n)]).astype(np.float32)
    y = rng.choice(np.array([0.4166, 0.4545, 0.8, 5, 12, 15, 40, 50, 60, 70, 80, 84, 200], dtype=np.float32), x.size)
    assert_same_floats(ctx.eval_math(pkg.RT_MATH_POWF, x, y), oracle_powf(orc, x, y), "powf, renderer range")


def test_powf_on_random_bit_patterns(pkg, orc, ctx):
    """Every binary32 value is a possible base or exponent: uniformly random bit patterns (NaNs, infinities, subnormals and
    negative bases included) and a band of moderate exponents where results stay finite."""
    rng = np.random.default_rng(12)
    n = 1 << 21
    xb = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    yb = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    assert_same_floats(ctx.eval_math(pkg.RT_MATH_POWF, xb, yb), oracle_powf(orc, xb, yb), "powf, random bits")
    ym = rng.uniform(-40, 40, n).astype(np.float32)
    assert_same_floats(ctx.eval_math(pkg.RT_MATH_POWF, np.abs(xb), ym), oracle_powf(orc, np.abs(xb), ym), "powf, |random bits| ^ moderate")
    yi = rng.integers(-9, 10, n).astype(np.float32)   # integer exponents: the sign rules for negative bases
    xs = rng.uniform(-4, 4, n).astype(np.float32)
    assert_same_floats(ctx.eval_math(pkg.RT_MATH_POWF, xs, yi), oracle_powf(orc, xs, yi), "powf, integer exponents")


def test_powf_special_values_and_interval_boundaries(pkg, orc, ctx):
    x, y = np.meshgrid(SPECIAL, SPECIAL)
    assert_same_floats(ctx.eval_math(pkg.RT_MATH_POWF, x.ravel(), y.ravel()), oracle_powf(orc, x.ravel(), y.ravel()), "powf, special pairs")
    # the 47 table intervals of the logarithm: every boundary (47 + i - 1/2)/64, one binary32 step either side, at several scales
    edges = ((np.arange(48) + 46.5) / 64.0).astype(np.float32)
    near = np.concatenate([edges, np.nextafter(edges, np.float32(0)), np.nextafter(edges, np.float32(4))])
    bases = np.concatenate([near * s for s in (1.0, 2.0, 0.5, 2.0 ** -20, 2.0 ** 20, 2.0 ** -130)]).astype(np.float32)
    for e in (0.4166, 0.4545, 0.8, 2.0, 50.0, -3.5, 1e-3):
        assert_same_floats(ctx.eval_math(pkg.RT_MATH_POWF, bases, np.full_like(bases, e)), oracle_powf(orc, bases, np.full_like(bases, e)), f"powf, interval edges ^ {e}")
    # results at the overflow / underflow thresholds: y log2 x around +-128, -126 and -149
    lg = np.linspace(-151, 129, 200001)
    for base in (2.0, 0.5, 10.0, 1.0001):
        yy = (lg / np.log2(base)).astype(np.float32)
        xx = np.full_like(yy, base)
        assert_same_floats(ctx.eval_math(pkg.RT_MATH_POWF, xx, yy), oracle_powf(orc, xx, yy), f"powf, range thresholds, base {base}")


def test_atanf(pkg, orc, ctx):
    rng = np.random.default_rng(13)
    n = 1 << 21
    x = np.concatenate([rng.uniform(-50, 50, n).astype(np.float32), rng.normal(0, 1, n).astype(np.float32), (10 ** rng.uniform(-12, 12, n // 2)).astype(np.float32), (-(10 ** rng.uniform(-12, 12, n // 2))).astype(np.float32),
                        rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32), SPECIAL,
                        np.array([0.4375, 0.6875, 1.1875, 2.4375, -0.4375, -0.6875, -1.1875, -2.4375], dtype=np.float32)])
    assert_same_floats(ctx.eval_math(pkg.RT_MATH_ATANF, x), oracle_atanf(orc, x), "atanf")


def test_eval_math_rejects_bad_arguments(pkg, ctx):
    with pytest.raises(Exception):
        ctx.eval_math(7, np.ones(4, np.float32), np.ones(4, np.float32))
    assert ctx.eval_math(pkg.RT_MATH_POWF, np.zeros(0, np.float32), np.zeros(0, np.float32)).size == 0


def test_sqrt_on_every_positive_number_and_the_special_cases(pkg, ctx):
    """The kernels' square root (a short expansion when a whole wavefront holds positive normal numbers from 2^-96 on, the
    compiler's full IEEE expansion otherwise) against the correctly rounded root, which is what f32::sqrt returns: all
    2^31 non-negative bit patterns in order (wavefronts of the short form, of the full one, and the mixed ones at the range
    borders), then shuffled samples that mix usual and unusual numbers inside every wavefront, negatives and NaNs included."""
    chunk = 1 << 26
    for start in range(0, 1 << 31, chunk):
        x = np.arange(start, start + chunk, dtype=np.uint32).view(np.float32)
        with np.errstate(invalid="ignore"):
            want = np.sqrt(x)
        assert_same_floats(ctx.eval_math(pkg.RT_MATH_SQRT, x), want, f"sqrt, bit patterns from {start:#x}")
    rng = np.random.default_rng(17)
    bits = rng.integers(0, 1 << 32, 1 << 22, dtype=np.uint64).astype(np.uint32)
    mixed = np.concatenate([bits.view(np.float32), np.tile(SPECIAL, 4096), rng.uniform(0, 1e-28, 1 << 20).astype(np.float32),
                            (10.0 ** rng.uniform(-30, 30, 1 << 21)).astype(np.float32)])
    rng.shuffle(mixed)
    with np.errstate(invalid="ignore"):
        want = np.sqrt(mixed)
    assert_same_floats(ctx.eval_math(pkg.RT_MATH_SQRT, mixed), want, "sqrt, mixed wavefronts")
