"""pt_pow.h restates glibc's pow (the libm function behind the reference's f64::powf: render.rs:47 gamma, material.rs:200
specular) operation by operation, as this image's libm.so.6 executes it on a CPU with FMA. No GPU here: the HOST build of the
same header (pt_test_pow_host, C ABI) against the machine's libm, bit for bit. The device build is checked against the same
libm values in tests/test_gpu_device_parity.py."""
import ctypes as C

import numpy as np
import pytest

from portrayer_amd import _hip as H


def both(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    port = np.empty_like(x); libm = np.empty_like(x)
    assert H.lib().pt_test_pow_host(x.size, x.ctypes.data_as(H._dp), y.ctypes.data_as(H._dp), port.ctypes.data_as(H._dp), libm.ctypes.data_as(H._dp)) == 0
    return port, libm


def same_bits(a, b):
    """bit-identical, any NaN equal to any NaN (the sign / payload of an invalid operation's NaN is the machine's business)"""
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | nan))


def cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as fh:
            return " fma " in fh.read()
    except OSError:
        return True


def glibc_version():
    import platform
    name, ver = platform.libc_ver()
    return ver if name == "glibc" else ""


# pt_pow.h restates ONE binary's pow: glibc 2.35's x86-64 FMA variant. On another glibc or a CPU without FMA the
# machine's libm is a different function, and a failure here would say nothing about the port.
pytestmark = [pytest.mark.skipif(not cpu_has_fma(), reason="glibc selects its non-FMA pow on this CPU; pt_pow.h restates the FMA one"),
              pytest.mark.skipif(glibc_version() != "2.35", reason=f"pt_pow.h is pinned to glibc 2.35's pow; this machine has glibc {glibc_version() or '?'}")]


def test_renderer_domain_is_bit_exact():
    rng = np.random.default_rng(11)
    n = 1_500_000
    x = np.concatenate([rng.uniform(0.0, 1.0, n), rng.uniform(0.0, 4.0, n), rng.uniform(0.0, 1.5, n)])
    y = np.concatenate([np.full(n, 1.0 / 2.2), 4.0 * rng.integers(1, 200, n).astype(np.float64), rng.uniform(0.0, 400.0, n)])
    port, libm = both(x, y)
    assert same_bits(port, libm)


def test_wide_range_and_random_bit_patterns():
    rng = np.random.default_rng(12)
    n = 1_000_000
    x = np.ldexp(rng.uniform(0.5, 1.0, n), rng.integers(-1070, 1024, n).astype(np.int32))
    y = rng.uniform(-300.0, 300.0, n)
    port, libm = both(x, y)
    assert same_bits(port, libm)
    bits = rng.integers(0, 2**64, size=(2, n), dtype=np.uint64)  # every class of double: nan, inf, subnormal, negative
    port, libm = both(bits[0].view(np.float64), bits[1].view(np.float64))
    assert same_bits(port, libm)
    x = 1.0 + (rng.uniform(-0.5, 0.5, n)) * 1e-3  # results near overflow / underflow / the subnormal range
    y = rng.uniform(-3e6, 3e6, n)
    port, libm = both(x, y)
    assert same_bits(port, libm)
    x = -rng.uniform(0.0, 10.0, n); y = rng.integers(-20, 21, n).astype(np.float64)  # negative bases, integer exponents
    port, libm = both(x, y)
    assert same_bits(port, libm)


def test_special_values():
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.5, 2.0, -2.0, 3.0, -3.0, 1e-310, -1e-310, 1e308, 1e-320, 5e-324,
                   1.7976931348623157e308, 710.0, -745.0, 1024.0, 2.0**63, 2.0**-66, -2.0**-66, 2.0**64, -2.0**64, 0.5e-300, 1.0 / 2.2, 100.0])
    x, y = np.meshgrid(sp, sp)
    port, libm = both(x.ravel(), y.ravel())
    assert same_bits(port, libm)
    assert port[(x.ravel() == 0.0) & (y.ravel() == 100.0)].tolist() == [0.0, 0.0]  # max(N.h, 0)^(4 shininess) of a grazing highlight
