"""CPU restatement of the "f16x2" arithmetic of the trajectory kernels and of the GEMM engine's f16 planes -- TEST INFRASTRUCTURE, NOT
PRODUCT CODE.

The contractions of the S/T/Q nets (reference: utils/layers.py:29-37, `tf.matmul(x, W) + b` in fp32, evaluated four times per leapfrog
step by utils/dynamics.py:115-201) run on the f16 matrix pipe of gfx950 without giving up fp32 accuracy
(l2hmc_amd/csrc/traj_fast.hpp `split16` / `wsplit16`; csrc/gemm_f32.hpp `split4_f16`).  This module restates the splits in numpy
(np.float16 conversions round to nearest even, as v_cvt_pk_f16_f32 / v_fma_mix*_f16 do under the default rounding mode) so that the
claims the design rests on are checked WITHOUT a GPU (tests/test_f16x2_oracle.py):

  * activation  a  = 64 hd + lo + r,  hd = f16(a / 64), lo = f16(a - 64 hd):  |r| <= 2^-24 |a| for 0.25 <= |a| < 4.19e6, and
    |r| <= 2^-25 below (lo subnormal); the residual a - 64 hd is exact in fp32;
  * weight      w  = w_hi + W_lo / 64 + r,  w_hi = f16(w), W_lo = f16(64 (w - w_hi)):  |r| <= 2^-24 |w| for 4e-3 <= |w| < 1023;
  * the two MFMAs  [64 w_hi | w_hi] . [hd | lo]  +  [W_lo | W_lo / 64] . [hd | lo]  reproduce  sum w a  to fp32 level (products of two
    f16 numbers are exact in fp32; the only dropped term is (W_lo / 64) lo <= 2^-24 |w a|);
  * the planes of the GEMM engine:  x = X1 + X2 / 64,  X1 = f16(x), X2 = f16(64 (x - X1)),  x y = X1 Y1 + (X1 / 64) Y2 + X2 (Y1 / 64).

Only tests/ may import this module.
"""
import numpy as np

SCALE = np.float32(64.0)


def f16(x):
    """fp32 -> nearest f16 (ties to even; overflow -> inf), returned as fp32"""
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def split_activation(a):
    """(hd, lo) of traj_fast.hpp `split16`: hd = f16(a / 64), lo = f16(a - 64 hd) with the subtraction exact in fp32"""
    a = np.asarray(a, dtype=np.float32)
    hd = f16(a / SCALE)
    res = a.astype(np.float64) - 64.0 * hd.astype(np.float64)          # what the fma forms before its ONE rounding
    with np.errstate(over="ignore", invalid="ignore"):
        lo = res.astype(np.float16).astype(np.float32)
    return hd, lo


def split_weight(w):
    """(w_hi, W_lo) of `wsplit16`: w_hi = f16(w), W_lo = f16(64 (w - w_hi)); the fragments are [64 w_hi | w_hi], [W_lo | W_lo / 64]"""
    w = np.asarray(w, dtype=np.float32)
    w_hi = f16(w)
    W_lo = f16((w - w_hi) * SCALE)
    return w_hi, W_lo


def contract(w, a):
    """sum_k w[..., k] a[..., k] as the two MFMAs form it: every product exact, accumulated here in float64 (the hardware
    accumulates in fp32: its own rounding comes on top, as for the f32-input MFMA)"""
    w_hi, W_lo = split_weight(w)
    hd, lo = split_activation(a)
    d = np.float64
    first = (f16(w_hi * SCALE).astype(d) * hd.astype(d) + w_hi.astype(d) * lo.astype(d)).sum(-1)
    second = (W_lo.astype(d) * hd.astype(d) + f16(W_lo / SCALE).astype(d) * lo.astype(d)).sum(-1)
    return first + second


def planes(x):
    """(X1, X2) of gemm_f32.hpp `split4_f16`"""
    x = np.asarray(x, dtype=np.float32)
    X1 = f16(x)
    X2 = f16((x - X1) * SCALE)
    return X1, X2


def plane_product(x, y):
    """sum_k x y as the three MFMAs of gemm_xlp_kernel<..., 1> form it (float64 accumulation)"""
    X1, X2 = planes(x)
    Y1, Y2 = planes(y)
    d = np.float64
    return (X2.astype(d) * f16(Y1 / SCALE).astype(d) + f16(X1 / SCALE).astype(d) * Y2.astype(d) + X1.astype(d) * Y1.astype(d)).sum(-1)
