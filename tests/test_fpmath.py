"""include/szg/fpmath.h — the pinned GLSL built-ins — against float64 libm. These functions
are shared by the oracle and the GPU kernels, so their accuracy is checked here
independently of both: every function must stay inside the precision class Vulkan/GLSL
allows for the built-in it stands for (exp/log/sin/cos/asin/acos: a few ULP; pow: inherited
from exp(y * log(x)))."""
import numpy as np
import pytest

from oracle import binding as ob

N = 400_000
rng = np.random.default_rng(0x5A2C)


def ulp_error(got, want):
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64)
    exp = np.floor(np.log2(np.maximum(np.abs(want), 2.0 ** -126)))
    ulp = 2.0 ** (exp - 23)
    return np.abs(got - want) / ulp


def sweep(fn, lo, hi, ref, y=None):
    x = rng.uniform(lo, hi, N).astype(np.float32)
    yy = None if y is None else np.full(N, y, np.float32)
    got = ob.builtin_eval(fn, x, yy)
    want = ref(x.astype(np.float64)) if y is None else ref(x.astype(np.float64), float(np.float32(y)))
    return ulp_error(got, want).max()


@pytest.mark.parametrize("lo,hi", [(-87.0, 88.0), (-1.0, 1.0), (-20.0, 0.0)])
def test_exp(lo, hi):
    assert sweep(0, lo, hi, np.exp) <= 1.0


@pytest.mark.parametrize("fn,ref", [(2, np.sin), (3, np.cos)])
def test_sin_cos(fn, ref):
    assert sweep(fn, -10.0, 10.0, ref) <= 1.6
    assert sweep(fn, -0.02, 0.02, ref) <= 1.0


def test_asin_acos():
    assert sweep(4, -1.0, 1.0, np.arcsin) <= 2.5
    assert sweep(5, -1.0, 1.0, np.arccos) <= 1.5
    assert sweep(5, 0.99, 1.0, np.arccos) <= 1.5


@pytest.mark.parametrize("y,lo,hi,rel", [(1.2, 1e-4, 20.0, 3e-6), (1.5, 0.3, 3.3, 1e-6), (5.0, 1e-3, 1.0, 1e-5),
                                        (160.0, 0.85, 1.0, 1.2e-5), (13.7, 0.5, 1.0, 2e-6)])
def test_pow_relative_error(y, lo, hi, rel):
    """pow = exp(y * log(x)): relative error ~ |y log x| * 2^-23, as for GLSL's exp2(y * log2(x))."""
    x = rng.uniform(lo, hi, N).astype(np.float32)
    got = ob.builtin_eval(1, x, np.full(N, y, np.float32)).astype(np.float64)
    want = np.power(x.astype(np.float64), float(np.float32(y)))
    assert (np.abs(got - want) / want).max() <= rel


def test_special_values():
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    e = ob.builtin_eval(0, np.array([0.0, -np.inf, np.inf, nan, -104.5, 89.5], np.float32))
    assert e[0] == 1.0 and e[1] == 0.0 and e[2] == inf and np.isnan(e[3]) and e[4] == 0.0 and e[5] == inf
    p = ob.builtin_eval(1, np.array([0.0, 1.0, 2.0, 0.0, -1.0, 0.5], np.float32), np.array([5.0, 160.0, 0.0, 0.0, 2.0, 1.0], np.float32))
    assert p[0] == 0.0 and p[1] == 1.0 and p[2] == 1.0 and p[3] == 1.0 and np.isnan(p[4]) and abs(p[5] - 0.5) < 1e-7
    s = ob.builtin_eval(2, np.array([0.0, 1e9, nan], np.float32))
    assert s[0] == 0.0 and np.isnan(s[1]) and np.isnan(s[2])
    c = ob.builtin_eval(3, np.array([0.0], np.float32))
    assert c[0] == 1.0
    a = ob.builtin_eval(4, np.array([1.0, -1.0, 1.5, 0.0], np.float32))
    assert abs(a[0] - np.pi / 2) < 2e-7 and abs(a[1] + np.pi / 2) < 2e-7 and np.isnan(a[2]) and a[3] == 0.0
    k = ob.builtin_eval(5, np.array([1.0, -1.0, 0.0, -1.5], np.float32))
    assert k[0] == 0.0 and abs(k[1] - np.pi) < 4e-7 and abs(k[2] - np.pi / 2) < 2e-7 and np.isnan(k[3])


def test_pinned_and_libm_builds_agree_to_a_few_ulp():
    x = rng.uniform(-20, 20, N).astype(np.float32)
    for fn in (0, 2, 3):
        a, b = ob.builtin_eval(fn, x), ob.builtin_eval(fn, x, libm=True)
        assert ulp_error(a, b.astype(np.float64)).max() <= 2.5
