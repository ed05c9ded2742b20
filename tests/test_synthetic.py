"""CPU: the seeded workload generator.  Its bits must not depend on a BLAS library's threading (round 5: `dirs @ R.T` from
several threads at once produced other scans from run to run on a 256-thread host; the golden fixtures hold the sha1 of the
generated inputs), nor on whether sequences are generated here or by worker processes."""
import numpy as np


def test_rotate_rows_is_the_blas_product_bit_for_bit():
    """`rotate_rows(x, R)` == `x @ R.T` as OpenBLAS evaluates it (the fused chain fma(x2, r2, fma(x1, r1, x0 * r0))), element by
    element, on random rows of several magnitudes, rows with zeros and matrices with zeros — evaluated without BLAS through
    error-free transformations and one rounding to odd (a correctly rounded a * b + c)."""
    from pylidar_slam_amd.synthetic import rotate_rows
    rng = np.random.default_rng(11)
    for k in range(12):
        x = rng.normal(size=(20_000, 3)) * 10.0 ** float(rng.integers(-3, 3))
        x[rng.integers(0, x.shape[0], 50)] = 0.0
        m = rng.normal(size=(3, 3))
        if k % 3 == 0:
            m[rng.integers(0, 3), rng.integers(0, 3)] = 0.0
        assert np.array_equal(x @ m.T, rotate_rows(x, m)), k


def test_correctly_rounded_fma_against_exact_arithmetic():
    """The emulated fma against exact rational arithmetic (Fraction -> float is correctly rounded) on cases built to need the
    rounding to odd: a product whose low part is exactly half an ulp of the sum."""
    from fractions import Fraction
    from pylidar_slam_amd.synthetic import _fma
    rng = np.random.default_rng(5)
    a = rng.normal(size=4000)
    b = rng.normal(size=4000)
    c = rng.normal(size=4000) * 10.0 ** rng.integers(-8, 8, size=4000)
    # ties: c = -round(a * b) + a few ulps, so that a * b + c cancels almost completely
    c[:1000] = -(a[:1000] * b[:1000]) * (1.0 + rng.integers(-3, 4, size=1000) * 2.0 ** -52)
    got = _fma(a, b, c)
    for i in range(a.shape[0]):
        want = float(Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i])))
        assert got[i] == want, (i, a[i], b[i], c[i], got[i], want)


def test_worker_processes_generate_the_sequential_bits():
    """`make_c2_workloads` with worker processes (bench.py's throughput legs) returns what `make_c2_workload` returns here."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_c2_workload, make_c2_workloads
    import pylidar_slam_amd.synthetic as syn
    par = make_c2_workloads([301, 302], "pingpong_r01", 8, workers=2)
    for seq, got in zip((301, 302), par):
        want = make_c2_workload(seq, "pingpong_r01", 8)
        assert got[3] == want[3] and got[4] == want[4]
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        assert sorted(got[0]) == sorted(want[0]) and all(np.array_equal(got[0][f], want[0][f]) for f in want[0])
