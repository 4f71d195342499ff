"""The arithmetic contract of CTL_DT_X3 (csrc/ctl_conv_x3_stage.h), checked on the CPU with torch's bfloat16 (round to nearest even, the
rounding of v_cvt_pk_bf16_f32): every fp32 number in the split's domain is EXACTLY hi + mid + lo with three bf16 numbers, and the six products
the kernels keep reproduce the fp32 product to better than the rounding of an fp32 multiply.  (The GPU side of the same claim:
tests/test_x3_gpu.py::test_split_is_exact_on_adversarial_values and the fp64 comparisons there.)"""
import torch


def split3(x):
    hi = x.to(torch.bfloat16).float()
    r1 = x - hi                                  # exact in fp32 (the kernels compute it in fp32 as well)
    mid = r1.to(torch.bfloat16).float()
    lo = r1 - mid
    return hi, mid, lo, r1


def _values():
    g = torch.Generator().manual_seed(11)
    bits = torch.randint(0, 2 ** 31 - 1, (200000,), generator=g, dtype=torch.int64).to(torch.int32).view(torch.float32)
    v = torch.cat([torch.randn(200000, generator=g), torch.randn(50000, generator=g) * 1e-12, torch.randn(50000, generator=g) * 1e12, bits, -bits,
                   torch.tensor([1.0, 1.00390625, 1.0078125, 0.99609375, 255.99998, 1.9999999, 3.3e38, -3.3e38, 1e-29, 0.0])])
    return v[torch.isfinite(v) & ((v == 0) | ((v.abs() > 1e-30) & (v.abs() < 3.38e38)))]


def test_three_bf16_numbers_hold_an_fp32_number_exactly():
    x = _values()
    hi, mid, lo, r1 = split3(x)
    assert torch.equal((x.double() - hi.double()), r1.double()), "x - hi must be exact in fp32"
    assert torch.equal(lo, lo.to(torch.bfloat16).float()), "lo must be representable in bf16 without rounding"
    assert torch.equal(hi.double() + mid.double() + lo.double(), x.double()), "hi + mid + lo == x, bit for bit"
    # each step takes 8 significant bits plus the sign of the remainder
    nz = x != 0
    assert float((r1[nz].abs() / x[nz].abs()).max()) <= 2.0 ** -8
    assert float((lo[nz].abs() / x[nz].abs()).max()) <= 2.0 ** -16


def test_six_products_are_an_fp32_product():
    g = torch.Generator().manual_seed(5)
    a = torch.randn(300000, generator=g) * torch.exp(torch.randn(300000, generator=g) * 3)
    b = torch.randn(300000, generator=g) * torch.exp(torch.randn(300000, generator=g) * 3)
    ah, am, al, _ = split3(a)
    bh, bm, bl, _ = split3(b)
    d = lambda t: t.double()
    six = d(ah) * d(bh) + d(ah) * d(bm) + d(am) * d(bh) + d(am) * d(bm) + d(ah) * d(bl) + d(al) * d(bh)      # (each product is exact in fp32: 8 x 8 bits)
    exact = d(a) * d(b)
    rel = ((six - exact).abs() / exact.abs().clamp_min(1e-300)).max()
    assert float(rel) < 2.0 ** -24, float(rel)              # the dropped mid*lo + lo*mid + lo*lo: below half an ulp of the fp32 product
    fp32_rounding = ((d((a * b)) - exact).abs() / exact.abs().clamp_min(1e-300)).max()
    assert float(rel) < float(fp32_rounding)                # ... i.e. smaller than what rounding the product to fp32 costs
