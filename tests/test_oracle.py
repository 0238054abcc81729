"""Pins the CPU oracle (not-gpu).

1. against the reference's own known-answer cases (constant vectors; the only
   results the reference's tests assert: examples/basic_inverse.rs:238-253,
   examples/basic_inverse2.rs:269-284),
2. against numpy.fft float64 fixtures (tests/golden/k4_random.npz),
3. fp64 fast path vs naive O(n^2), buffer-parity rule, generator properties.
"""
import numpy as np
import pytest

from conftest import REF_ABS_TOL, REL_TOL


def _max_abs_parts(a, b):
    d = a.astype(np.complex128) - b.astype(np.complex128)
    return max(np.abs(d.real).max(), np.abs(d.imag).max())


def test_reference_known_answers(oracle, known_answers):
    batch = 5
    for case in known_answers["cases"]:
        n = case["n"]
        c = np.complex64(complex(*case["c"]))
        x = np.full(n * batch, c, dtype=np.complex64)
        expect = np.zeros(n * batch, dtype=np.complex128)
        if case["plan"] == "Forward":
            y, which = oracle.forward_ref(x, n)
            expect[::n] = complex(c) * n
        elif case["plan"] == "Inverse":
            y, which = oracle.inverse_ref(x, n)
            expect[::n] = complex(c)
        else:
            y, which = oracle.onlyinverse_ref(x, n)
            y = oracle.normalize_ref(y, n)
            expect[::n] = complex(c)
        # reference metric: max(|dre|, |dim|) < 1e-5 absolute
        assert _max_abs_parts(y, expect) < REF_ABS_TOL, case
        # with constant input every a-b is exactly 0: error must be 0
        if case["plan"] != "Forward":
            assert _max_abs_parts(y, expect.astype(np.complex64)) == 0.0
        # result-buffer parity rule processor.rs:153-157: 512 -> scratch, 16 -> src
        assert which == (int(np.log2(n)) % 2)


@pytest.mark.parametrize("lg", range(1, 11))
def test_forward_vs_numpy_fixture(oracle, k4, lg):
    n = 1 << lg
    x = k4[f"x_{n}"]
    y, _ = oracle.forward_ref(x, n)
    mx, l2 = oracle.compare(y, k4[f"fwd_{n}"])
    assert mx <= REL_TOL and l2 <= REL_TOL
    # the oracle's own fp64 path agrees with numpy to fp64 roundoff
    r = oracle.dft_f64(x, n, -1)
    assert np.abs(r - k4[f"fwd_{n}"]).max() <= 1e-12 * max(1.0, np.abs(r).max())


@pytest.mark.parametrize("lg", range(1, 11))
def test_inverse_vs_numpy_fixture(oracle, k4, lg):
    n = 1 << lg
    x = k4[f"x_{n}"]
    y, _ = oracle.onlyinverse_ref(x, n)
    mx, l2 = oracle.compare(y, k4[f"inv_unscaled_{n}"])
    assert mx <= REL_TOL and l2 <= REL_TOL
    ys, _ = oracle.inverse_ref(x, n)
    mx, l2 = oracle.compare(ys, k4[f"inv_unscaled_{n}"] / n)
    assert mx <= REL_TOL and l2 <= REL_TOL
    # Onlyinverse + Normalize == Inverse bit for bit (same operations, same order)
    yn = oracle.normalize_ref(y, n)
    assert np.array_equal(yn.view(np.uint32), ys.view(np.uint32))


def test_fp64_fast_vs_naive(oracle):
    for n in (128, 512, 2048):
        x = oracle.gen_input(n, 1, first_transform=7)
        a = oracle.dft_f64(x, n, -1)
        b = oracle.dft_f64_naive(x, -1)
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
        a = oracle.dft_f64(x, n, +1)
        b = oracle.dft_f64_naive(x, +1)
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()


def test_impulse_and_tone(oracle):
    # K2: impulse at p -> exp(-2*pi*i*p*k/n): pins sign, order and every twiddle
    n, p, q = 256, 5, 37
    x = np.zeros(n, dtype=np.complex64); x[p] = 1
    y, _ = oracle.forward_ref(x, n)
    k = np.arange(n)
    mx, _ = oracle.compare(y, np.exp(-2j * np.pi * p * k / n))
    assert mx <= REL_TOL
    # K3: tone at bin q -> n*delta(k-q)
    x = np.exp(2j * np.pi * q * k / n).astype(np.complex64)
    y, _ = oracle.forward_ref(x, n)
    e = np.zeros(n, dtype=np.complex128); e[q] = n
    mx, _ = oracle.compare(y, e)
    assert mx <= REL_TOL


def test_large_forward_vs_fp64(oracle):
    n = 1 << 16
    x = oracle.gen_input(n, 2)
    y, which = oracle.forward_ref(x, n)
    assert which == 0
    r = oracle.dft_f64(x, n, -1)
    for t in range(2):
        mx, l2 = oracle.compare(y[t * n:(t + 1) * n], r[t * n:(t + 1) * n])
        assert mx <= REL_TOL and l2 <= REL_TOL


def test_generator_properties(oracle):
    n = 1024
    a = oracle.gen_input(n, 4)
    # counter-based: transform 2 alone equals slice 2 of the batch
    b = oracle.gen_input(n, 1, first_transform=2)
    assert np.array_equal(a[2 * n:3 * n], b)
    assert a.real.min() >= -1 and a.real.max() < 1 and a.imag.min() >= -1 and a.imag.max() < 1
    assert abs(a.real.mean()) < 0.05 and abs(a.imag.mean()) < 0.05
    assert abs(a.real.std() - 1 / np.sqrt(3)) < 0.02
    # power-of-two scale is exact
    c = oracle.gen_input(n, 4, scale=2.0 ** -40)
    assert np.array_equal(c, a * np.float32(2.0 ** -40))
    # different seed -> different data
    d = oracle.gen_input(n, 1, seed=1)
    assert not np.array_equal(d, a[:n])
