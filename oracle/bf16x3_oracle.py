"""CPU restatement of the "bf16x3" arithmetic of the GEMM engine -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The decoder-sized products of config 5 (reference: the dense layers of mnist_vae.py:104-111, i.e. utils/layers.py:29-37
`tf.matmul(x, W) + b` in fp32) run on the bf16 matrix pipe of gfx950 without giving up fp32 accuracy
(l2hmc_amd/csrc/gemm_f32.hpp `split3`, csrc/gemm_xl.hpp planes): every fp32 operand is split into three bf16 terms
x = h + m + l, round-to-nearest-even at each level, and of the nine cross products the six of weight >= 2^-16 are accumulated
in fp32.  This module restates exactly that in numpy so that the claims the design rests on are checked WITHOUT a GPU
(tests/test_oracle_golden.py::test_bf16x3_*):

  * the split is EXACT for every finite fp32 number whose exponent leaves room for the two residual levels
    (8 + 8 + 8 significand bits);
  * six products reproduce the fp32 product to fp32 rounding level, three do not;
  * the plane layout the kernels agree on (`to_planes`): three (rows_pad, ld) bf16 planes h | m | l, zero beyond the matrix.

Only tests/ may import this module.
"""
import numpy as np


def bf16_rne(x):
    """fp32 -> the nearest bf16 (ties to even), returned as fp32 with the low 16 bits zero (v_cvt_pk_bf16_f32 on finite input)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def split3(x):
    """(h, m, l), each bf16-valued fp32: h = rne(x), m = rne(x - h), l = rne(x - h - m) -- the residuals are exact fp32 subtractions."""
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    s = (r - m).astype(np.float32)
    return h, m, bf16_rne(s)


def to_planes(W, rows_pad=None, ld=None):
    """uint16 array (3, rows_pad, ld): the bf16 bit patterns of the h | m | l planes of W (rows, K), zero-padded -- what
    `to_planes_kernel` writes (ld = K rounded up to 32, rows_pad = rows rounded up to 128 for a weight matrix)."""
    W = np.asarray(W, dtype=np.float32)
    rows, K = W.shape
    rows_pad = rows if rows_pad is None else rows_pad
    ld = -(-K // 32) * 32 if ld is None else ld
    out = np.zeros((3, rows_pad, ld), dtype=np.uint16)
    for p, t in enumerate(split3(W)):
        out[p, :rows, :K] = (t.view(np.uint32) >> 16).astype(np.uint16)
    return out


def from_planes(P, rows, K):
    """fp32 matrix h + m + l of the planes (exact: the inverse of `to_planes`)."""
    t = (P[:, :rows, :K].astype(np.uint32) << 16).view(np.float32)
    return ((t[0] + t[1]).astype(np.float32) + t[2]).astype(np.float32)


# (weight term, activation term) of the products kept, in the order the kernels accumulate them per k-block: smallest first
SIX = (("l", "h"), ("h", "l"), ("m", "m"), ("m", "h"), ("h", "m"), ("h", "h"))
THREE = (("m", "h"), ("h", "m"), ("h", "h"))


def gemm_nt(A, B, products=SIX):
    """C[m, n] = sum_k A[m, k] B[n, k] from the given cross products of the bf16 terms.  Products of bf16 numbers are exact in
    fp32 and the MFMA accumulates in fp32; the accumulation ORDER inside the hardware is not restated (each product matrix is
    summed in float64 and rounded once, the products are then added in fp32 in the kernels' order), so this is the arithmetic
    up to fp32 summation order -- which is what the error bounds below are about."""
    ta = dict(zip("hml", split3(A)))
    tb = dict(zip("hml", split3(B)))
    C = np.zeros((np.shape(A)[0], np.shape(B)[0]), dtype=np.float32)
    for wb, wa in products:                     # B carries the weights (rows n), A the activations
        C = (C + (ta[wa].astype(np.float64) @ tb[wb].astype(np.float64).T).astype(np.float32)).astype(np.float32)
    return C
