"""AlteredMetric::fill_Jgup on the GPU vs the oracle's statement-by-statement restatement, and the
two limits that pin the oracle: no stratification and no rotation give back J g^{mu nu}; strong stratification removes
the vertical-vertical part."""
import numpy as np
import pytest


def _inputs(n, seed):
    rng = np.random.default_rng(seed)
    return dict(nsq=rng.uniform(0.0, 4.0, n), dmu=rng.uniform(-1, 1, n), dnu=rng.uniform(-1, 1, n),
                gup=rng.uniform(0.5, 2.0, n), J=rng.uniform(0.5, 2.0, n),
                hjac=[rng.uniform(-1, 1, n) for _ in range(4)])


def test_oracle_altered_metric_limits(oracle):
    so = oracle
    a = _inputs(1000, 1)
    # no stratification, no rotation: the plain metric
    out = so.altered_jgup(np.zeros(1000), a["dmu"], a["dnu"], a["gup"], a["J"], 0.3, 0.0, a["hjac"])
    np.testing.assert_allclose(out, a["gup"] * a["J"], rtol=1e-15, atol=1e-300)
    # omega -> infinity, mu = nu = z on a Cartesian map (dXi^z/dz = 1, g^zz = 1, J = 1): 1 - 1 = 0
    one = np.ones(10)
    out = so.altered_jgup(1e30 * one, one, one, one, one, 1.0, 0.0)
    np.testing.assert_allclose(out, 0.0, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("offdiag", [False, True])
def test_altered_metric_bit_exact(oracle, offdiag):
    from somar_amd import api
    so = oracle
    a = _inputs(100003, 2)
    h = a["hjac"] if offdiag else None
    want = so.altered_jgup(a["nsq"], a["dmu"], a["dnu"], a["gup"], a["J"], 0.37, 1.3e-1, h)
    got = api.altered_jgup(a["nsq"], a["dmu"], a["dnu"], a["gup"], a["J"], 0.37, 1.3e-1, h)
    np.testing.assert_array_equal(got, want)
