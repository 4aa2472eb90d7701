"""Distance between f64 arrays in units in the last place, for the parity tests. Test infrastructure."""
import os

import numpy as np


def ulp_diff(a, b):
    return np.abs(np.ascontiguousarray(a, dtype=np.float64).view(np.int64) - np.ascontiguousarray(b, dtype=np.float64).view(np.int64))


def assert_ulp(got, want, bound, label=""):
    """max |got - want| in ulp <= bound. PT_ULP_LOG=<file> appends the measured maximum, so that the bounds written in
    the tests can be kept at what the hardware actually shows."""
    d = ulp_diff(got, want)
    m = int(d.max()) if d.size else 0
    log = os.environ.get("PT_ULP_LOG")
    if log:
        with open(log, "a") as fh:
            fh.write(f"{label or os.environ.get('PYTEST_CURRENT_TEST', '?')}: max {m} ulp (bound {bound}), {100.0 * float((d == 0).mean()) if d.size else 100.0:.4f} % bit-equal\n")
    assert m <= bound, f"{label}: max {m} ulp > {bound}"
