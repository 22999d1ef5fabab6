"""The arithmetic behind vt_bf3.h, stated on the CPU (numpy; no GPU, no library call).

The block kernels' qkv / fc1 / fc2 and the F = 8 head's towers multiply fp32 operands as three bf16 pieces each,
x = h + m + l by truncation, a product = hh + hm + mh + hl + lh + mm accumulated in fp32 (v_mfma_f32_16x16x32_bf16).
These tests pin the two properties the design rests on -- the split is EXACT, and the six-term product is as accurate
as an fp32 fma chain -- with the same bit operations the kernels (vt3::split3) and the host packer (split3_host) use.
The hardware statement of the second property is tools/src/probe_bf3.hip (2.6e-7 against 3.0e-7 of sum |a b|).
"""
import numpy as np


def split3(x):
    """float32 array -> three float32 arrays whose values are bf16-representable (low 16 bits zero)."""
    x = np.asarray(x, dtype=np.float32)
    mask = np.uint32(0xFFFF0000)
    h = (x.view(np.uint32) & mask).view(np.float32)
    r1 = (x - h).astype(np.float32)
    m = (r1.view(np.uint32) & mask).view(np.float32)
    r2 = (r1 - m).astype(np.float32)
    l = (r2.view(np.uint32) & mask).view(np.float32)
    return h, m, l, r1, r2


def _operands(rs, n):
    # activations and weights as the net has them: signs, magnitudes over ten octaves, exact zeros, powers of two
    x = (rs.standard_normal(n) * np.exp2(rs.uniform(-12, 8, n))).astype(np.float32)
    x[::97] = 0.0
    x[1::101] = np.exp2(rs.randint(-10, 10, x[1::101].shape)).astype(np.float32)
    return x


def test_split_is_exact():
    rs = np.random.RandomState(0)
    x = _operands(rs, 200_000)
    h, m, l, r1, r2 = split3(x)
    for p in (h, m, l):
        assert not (p.view(np.uint32) & np.uint32(0xFFFF)).any()          # each piece is a bf16 value
    # the residuals are exact fp32 subtractions, and the last residual IS its own top half
    x64 = x.astype(np.float64)
    assert np.array_equal(r1.astype(np.float64), x64 - h.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))
    assert np.array_equal(l, r2)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x64)


def _six_term_dot(a, b):
    """sum_k a[k] b[k] the way the kernels issue it: per term one fp32-accumulated pass over K (an MFMA adds exact bf16 x bf16
    products into an fp32 accumulator), small terms first."""
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    acc = np.zeros(a.shape[:-1], dtype=np.float32)
    for pa, pb in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
        for k in range(a.shape[-1]):
            acc = (acc + (pa[..., k].astype(np.float64) * pb[..., k].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc


def _fp32_fma_dot(a, b):
    acc = np.zeros(a.shape[:-1], dtype=np.float64)
    for k in range(a.shape[-1]):            # fma: exact product, one rounding per step
        acc = (acc + a[..., k].astype(np.float64) * b[..., k].astype(np.float64)).astype(np.float32).astype(np.float64)
    return acc.astype(np.float32)


def test_six_term_product_is_as_accurate_as_an_fp32_fma_chain():
    rs = np.random.RandomState(1)
    for K in (48, 192):                       # qkv / fc1 and fc2
        a = rs.standard_normal((4000, K)).astype(np.float32)
        b = (rs.standard_normal((4000, K)) * 0.2).astype(np.float32)
        exact = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
        scale = (np.abs(a.astype(np.float64)) * np.abs(b.astype(np.float64))).sum(-1)
        e6 = np.abs(_six_term_dot(a, b).astype(np.float64) - exact) / scale
        ef = np.abs(_fp32_fma_dot(a, b).astype(np.float64) - exact) / scale
        # what the six terms drop (m l, l m, l l) is about one fp32 rounding per product (test_dropped_terms_bound); the accumulation rounds like any fp32 sum
        assert e6.max() <= 4e-7, (K, float(e6.max()))
        assert e6.max() <= 2.0 * ef.max() + 1e-8, (K, float(e6.max()), float(ef.max()))
        assert e6.mean() <= 1.5 * ef.mean() + 1e-9, (K, float(e6.mean()), float(ef.mean()))


def test_dropped_terms_bound():
    """Truncating pieces: |m| < 2^-7 |x|, |l| < 2^-15 |x|, so the dropped m l + l m + l l is below 2^-21 |a b| for ANY operands and
    ~2^-24.6 |a b| on average -- the size of ONE fp32 rounding --; in a sum the accumulator's own fp32 rounding (2^-24 per step)
    dominates it -- the statement test_six_term_product_is_as_accurate_as_an_fp32_fma_chain makes."""
    rs = np.random.RandomState(2)
    a, b = _operands(rs, 100_000), _operands(rs, 100_000)
    ah, am, al, _, _ = (p.astype(np.float64) for p in split3(a))
    bh, bm, bl, _, _ = (p.astype(np.float64) for p in split3(b))
    kept = ah * bh + ah * bm + am * bh + ah * bl + al * bh + am * bm
    ab = np.abs(a.astype(np.float64) * b.astype(np.float64))
    dropped = np.abs(a.astype(np.float64) * b.astype(np.float64) - kept)
    nz = ab > 0
    rel = dropped[nz] / ab[nz]
    assert rel.max() < 2.0 ** -21
    assert rel.mean() < 2.0 ** -24
    # the adversarial mantissa (leading piece 1.0000000, all ones below it) comes close to the bound
    worst = np.float32(np.uint32(0x3F80FFFF).view(np.float32))          # 1.0078124
    h, m, l, _, _ = (float(p[0]) for p in split3(np.array([worst])))
    w = float(worst)
    assert 2.0 ** -22 < abs(w * w - (h * h + 2 * h * m + 2 * h * l + m * m)) / (w * w) < 2.0 ** -21
