"""The arithmetic claims behind the f16x2 contractions (csrc/traj_fast.hpp, csrc/gemm_f32.hpp), checked on the CPU restatement
oracle/f16x2_oracle.py: exactness of the two-term splits over their stated ranges, fp32-level accuracy of the two-MFMA contraction
and of the three-product plane form, and what happens outside the range."""
import numpy as np

from oracle import f16x2_oracle as F

EPS22 = 2.0 ** -22          # two 11-bit terms: the split keeps 22 significand bits in the worst case (fp32: 24), ~23.5 on average


def _logrand(rng, lo, hi, n):
    return (np.exp(rng.uniform(np.log(lo), np.log(hi), n)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)


def test_activation_split_is_exact_to_fp32_rounding_over_its_range():
    rng = np.random.RandomState(0)
    a = _logrand(rng, 0.25, 4.0e6, 200000)
    hd, lo = F.split_activation(a)
    r = np.abs(a.astype(np.float64) - 64.0 * hd.astype(np.float64) - lo.astype(np.float64))
    assert np.all(np.isfinite(hd)) and np.all(np.isfinite(lo))
    assert np.all(r <= EPS22 * np.abs(a))
    # below 0.25 the low term is a subnormal f16: an ABSOLUTE quantum of 2^-24, i.e. an error of at most 2^-25
    a = _logrand(rng, 1e-9, 0.25, 200000)
    hd, lo = F.split_activation(a)
    r = np.abs(a.astype(np.float64) - 64.0 * hd.astype(np.float64) - lo.astype(np.float64))
    assert np.all(r <= 2.0 ** -25 + EPS22 * np.abs(a))
    # the residual in front of the second rounding is exactly representable in fp32 (the kernel forms it in one fp32 fma)
    a = _logrand(rng, 1e-3, 4.0e6, 50000)
    hd = F.f16(a / F.SCALE)
    res64 = a.astype(np.float64) - 64.0 * hd.astype(np.float64)
    assert np.array_equal(res64.astype(np.float32).astype(np.float64), res64)


def test_weight_split_is_exact_whatever_it_multiplies():
    rng = np.random.RandomState(1)
    w = _logrand(rng, 4e-3, 1000.0, 200000)
    w_hi, W_lo = F.split_weight(w)
    r = np.abs(w.astype(np.float64) - w_hi.astype(np.float64) - W_lo.astype(np.float64) / 64.0)
    assert np.all(r <= EPS22 * np.abs(w))
    assert np.all(np.isfinite(F.f16(w_hi * F.SCALE)))                  # the [64 w_hi | .] fragment: |w| < 1023
    # the UNSCALED low term (the first build of round 6) is subnormal below |w| = 0.25: up to 2^-25 absolute, i.e. 2^-18 relative at
    # |w| = 4e-3 -- which is what failed the float64 bracket of the stiff fixtures
    w = _logrand(rng, 4e-3, 0.2, 50000)
    w_hi = F.f16(w)
    bad = np.abs(w.astype(np.float64) - w_hi.astype(np.float64) - F.f16(w - w_hi).astype(np.float64))
    assert bad.max() > 10 * EPS22 * np.abs(w).min()


def test_two_mfma_contraction_is_fp32_accurate():
    rng = np.random.RandomState(2)
    for wscale, ascale in ((1.0, 10.0), (0.1, 300.0), (0.01, 1.0), (3.0, 1e5)):
        w = (rng.randn(2000, 64) * wscale).astype(np.float32)
        a = (rng.randn(2000, 64) * ascale).astype(np.float32)
        ref = (w.astype(np.float64) * a.astype(np.float64)).sum(-1)
        mag = (np.abs(w).astype(np.float64) * np.abs(a)).sum(-1)
        err = np.abs(F.contract(w, a) - ref)
        small = (np.abs(a) < 0.25)
        bound = 2.5 * EPS22 * mag + 2.0 ** -25 * (np.abs(w) * small).sum(-1)      # DESIGN 3h's error statement (both operands + the dropped term)
        assert np.all(err <= bound), (wscale, ascale, float((err / bound).max()))
        # ... and in practice as good as fp32 itself: the median error sits within a factor two of an fp32 dot product's (products
        # rounded to fp32, then a sequential fp32 sum)
        acc = np.zeros(len(w), np.float32)
        for k in range(w.shape[1]):
            acc = (acc + (w[:, k] * a[:, k]).astype(np.float32)).astype(np.float32)
        err32 = np.abs(acc.astype(np.float64) - ref)
        assert np.median(err) <= 2.0 * np.median(err32)


def test_out_of_range_operands_are_not_finite():
    hd, lo = F.split_activation(np.float32([4.3e6, -1e7]))
    assert np.all(np.isinf(hd)) and not np.any(np.isfinite(lo))          # inf - inf in the MFMA: NaN, never a wrong finite number
    assert np.all(np.isfinite(F.split_activation(np.float32([4.19e6, 65504.0 * 64.0]))[0]))


def test_plane_products_are_fp32_accurate_on_their_range_and_degrade_below_it():
    rng = np.random.RandomState(3)
    x = _logrand(rng, 4e-3, 6.0e4, 64 * 4000).reshape(4000, 64)
    y = _logrand(rng, 4e-3, 6.0e4, 64 * 4000).reshape(4000, 64)
    ref = (x.astype(np.float64) * y.astype(np.float64)).sum(-1)
    mag = (np.abs(x).astype(np.float64) * np.abs(y)).sum(-1)
    assert np.all(np.abs(F.plane_product(x, y) - ref) <= 2.5 * EPS22 * mag)
    X1, X2 = F.planes(x)
    assert np.all(np.abs(x.astype(np.float64) - X1 - X2.astype(np.float64) / 64.0) <= EPS22 * np.abs(x))
    # entries far below the range (an adjoint scaled by 1 / chains, say 1e-7) keep an absolute quantum only: the reason the
    # TRAINER's planes stay bf16x3
    t = np.float32([1.0e-7, 3.3e-8])
    T1, T2 = F.planes(t)
    assert np.max(np.abs(t.astype(np.float64) - T1 - T2.astype(np.float64) / 64.0) / t) > 1e-4
