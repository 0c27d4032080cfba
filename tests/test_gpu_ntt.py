"""GPU parity: HIP NTT (through the C ABI) == oracle restatement of groth16/math/ntt.nim, bit-exact."""
import pytest

from oracle import bn254_ref as o
from tests import inputs as I

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log2n", [0, 1, 2, 3, 5, 8, 9, 10, 11, 13, 16])
def test_ntt_forward_inverse_vs_oracle(ctx, orc, log2n):
    n = 1 << log2n
    xb = I.fr_mont_bytes(I.uniform_scalars(n, seed=300 + log2n))
    f = ctx.ntt(xb, log2n, inverse=False)
    assert f == orc.ntt(xb, log2n, inverse=False)
    g = ctx.ntt(xb, log2n, inverse=True)
    assert g == orc.ntt(xb, log2n, inverse=True)
    assert ctx.ntt(f, log2n, inverse=True) == xb          # round trip


def test_ntt_matches_naive_dft_definition(ctx):
    n = 64
    xs = I.uniform_scalars(n, seed=5)
    got = I.fr_from_mont(ctx.ntt(I.fr_mont_bytes(xs), 6, inverse=False))
    assert got == o.naive_dft(xs, o.Domain(n))


def test_ntt_reference_api_mirror(ctx):
    from nim_groth16_amd import createDomain, forwardNTT, inverseNTT
    D = createDomain(256)
    xs = I.uniform_scalars(256, seed=9)
    xb = I.fr_mont_bytes(xs)
    assert I.fr_from_mont(forwardNTT(xb, D, ctx)) == o.forward_ntt(xs, o.Domain(256))
    assert inverseNTT(forwardNTT(xb, D, ctx), D, ctx) == xb
    with pytest.raises(AssertionError):
        forwardNTT(xb[:-32], D, ctx)                       # ntt.nim:57
