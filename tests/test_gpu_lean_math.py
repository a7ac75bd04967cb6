"""GPU: the library's short float64 device math (dc::lean in csrc/dc_kernels.hip.h) against numpy,
in units in the last place.  The float64 kernels (neutral / dynamic models, the z side of the
extended model) are compared with the float64 oracle at 1e-9 .. 1e-11 relative: these helpers
must stay a few ulp."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulps(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_exp(hip_ctx):
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.uniform(-700, 700, 200_000), rs.uniform(-2, 2, 200_000), rs.normal(0, 1e-3, 1000),
                        [0.0, -0.0, 1.0, -1.0, 709.0, -745.0]])
    got = hip_ctx.selftest_math(0, x)
    assert _ulps(got, np.exp(x)).max() <= 2.0
    edge = hip_ctx.selftest_math(0, np.array([800.0, -800.0, np.inf, -np.inf, np.nan, 1e300, -1e300]))
    assert edge[0] == np.inf and edge[1] == 0.0 and edge[2] == np.inf and edge[3] == 0.0
    assert np.isnan(edge[4]) and edge[5] == np.inf and edge[6] == 0.0


def test_log_and_log1p(hip_ctx):
    rs = np.random.RandomState(1)
    x = np.concatenate([np.exp(rs.uniform(-700, 700, 200_000)), rs.uniform(0.5, 2.0, 200_000),
                        1.0 + rs.normal(0, 1e-6, 1000), [1.0, 2.0, 0.5, 5e-324, 1e-310, 1.7e308]])
    got = hip_ctx.selftest_math(1, x)
    want = np.log(x)
    err = np.abs(got - want) / np.maximum(np.spacing(np.abs(want)), 1e-300)
    # (near 1 the result is tiny: absolute error against the spacing of the ARGUMENT's log scale)
    assert (np.minimum(err, np.abs(got - want) / 2.3e-16)).max() <= 2.0
    with np.errstate(all="ignore"):
        edge = hip_ctx.selftest_math(1, np.array([0.0, -1.0, np.inf, np.nan]))
    assert edge[0] == -np.inf and np.isnan(edge[1]) and edge[2] == np.inf and np.isnan(edge[3])
    y = np.concatenate([np.exp(-np.abs(rs.uniform(0, 750, 200_000))), rs.uniform(0, 1, 100_000), [0.0, 1.0, 1e-20]])
    got = hip_ctx.selftest_math(2, y)
    assert _ulps(got[y > 0], np.log1p(y[y > 0])).max() <= 2.5
    assert got[y == 0].max() == 0.0


def test_rcp(hip_ctx):
    rs = np.random.RandomState(2)
    x = np.concatenate([rs.uniform(1e-9, 1e9, 200_000), -rs.uniform(0.1, 10, 1000), np.exp(rs.uniform(-600, 600, 100_000))])
    got = hip_ctx.selftest_math(3, x)
    assert _ulps(got, 1.0 / x).max() <= 1.5
