"""world_size-2 gloo test of the sharded-proof collective logic (no GPU): shard ranges, the single all-gather,
rank-order summation and the mask algebra must reproduce the oracle's unsharded proof bit for bit.  The
per-rank MSM partials and the final combine are oracle stand-ins here (the GPU versions are covered by
tests/test_gpu_prover.py::test_sharded_proof_single_gpu)."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from oracle import bn254_ref as o
from tests import inputs as I

ONE = o.fp_to_mont_bytes(1)


def _xyzz_g1(p):
    return bytes(128) if o.G1.is_inf(p) else o.g1_to_bytes(p) + ONE + ONE


def _xyzz_g2(p):
    return bytes(256) if o.G2.is_inf(p) else o.g2_to_bytes(p) + (ONE + bytes(32)) * 2


def _aff_g1(b):      # XYZZ (x, y, zz, zzz) -> affine; zz == 0 -> infinity
    x, y, zz, zzz = (o.fp_from_mont_bytes(b[i:i + 32]) for i in (0, 32, 64, 96))
    return o.INF_G1 if zz == 0 else (x * pow(zz, -1, o.P) % o.P, y * pow(zzz, -1, o.P) % o.P)


def _aff_g2(b):
    e = lambda k: (o.fp_from_mont_bytes(b[k:k + 32]), o.fp_from_mont_bytes(b[k + 32:k + 64]))   # noqa: E731
    x, y, zz, zzz = e(0), e(64), e(128), e(192)
    return o.INF_G2 if o.fp2_is_zero(zz) else (o.fp2_mul(x, o.fp2_inv(zz)), o.fp2_mul(y, o.fp2_inv(zzz)))


def _setup():
    rng = o.SplitMix64(77)
    tox = o.ToxicWaste(*[rng.fr() for _ in range(5)])
    return o.fake_circuit_setup(o.toy_r1cs(), tox, o.SNARKJS), rng.fr(), rng.fr()


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nim_groth16_amd.distributed import ShardedProver, shardRange
    from nim_groth16_amd.prover import Mask, Witness
    from nim_groth16_amd.zkey_types import GrothHeader, ZKey
    oz, r, s = _setup()
    zk = ZKey(header=GrothHeader("bn128", 1, oz.nvars, oz.npubs, oz.domainSize, oz.logDomainSize))
    wit = o.TOY_WITNESS

    def partials(wbytes):
        w = I.fr_from_mont(wbytes)
        Az, Bz, Cz = o.build_abc(oz.coeffs, oz.domainSize, w)
        qs = o.compute_snarkjs_scalar_coeffs(Az, Bz, Cz)
        zs = w[oz.npubs + 1:]
        wa, wb = shardRange(oz.nvars, rank, world)
        ca, cb = shardRange(oz.nvars - oz.npubs - 1, rank, world)
        ha, hb = shardRange(oz.domainSize, rank, world)
        return (_xyzz_g1(o.G1.msm_naive(w[wa:wb], oz.pointsA1[wa:wb])) +
                _xyzz_g1(o.G1.msm_naive(w[wa:wb], oz.pointsB1[wa:wb])) +
                _xyzz_g2(o.G2.msm_naive(w[wa:wb], oz.pointsB2[wa:wb])) +
                _xyzz_g1(o.G1.msm_naive(qs[ha:hb], oz.pointsH1[ha:hb])) +
                _xyzz_g1(o.G1.msm_naive(zs[ca:cb], oz.pointsC1[ca:cb])))

    def combine(gathered, count, rb, sb):
        rr = o.fr_from_mont_bytes(rb) if rb else 0
        ss = o.fr_from_mont_bytes(sb) if sb else 0
        tot = [o.INF_G1, o.INF_G1, o.INF_G2, o.INF_G1, o.INF_G1]
        for k in range(count):
            rec = gathered[768 * k:768 * (k + 1)]
            tot[0] = o.G1.add(tot[0], _aff_g1(rec[0:128]))
            tot[1] = o.G1.add(tot[1], _aff_g1(rec[128:256]))
            tot[2] = o.G2.add(tot[2], _aff_g2(rec[256:512]))
            tot[3] = o.G1.add(tot[3], _aff_g1(rec[512:640]))
            tot[4] = o.G1.add(tot[4], _aff_g1(rec[640:768]))
        it = iter(tot)
        pr = o.generate_proof_with_mask(oz, wit, rr, ss, msm_g1=lambda c, p: next(it), msm_g2=lambda c, p: next(it),
                                        quotient=lambda *a: [0] * oz.domainSize)
        # generate_proof_with_mask calls msm in the order A1, B1, B2, H1, C1 (prover.nim:282-302) == record order
        return o.g1_to_bytes(pr.pi_a), o.g2_to_bytes(pr.pi_b), o.g1_to_bytes(pr.pi_c)

    sp = ShardedProver(zk, rank, world, partials_fn=partials, combine_fn=combine)
    pr = sp.prove(Witness("bn128", oz.nvars, I.fr_mont_bytes(wit)), Mask(r, s))
    q.put((rank, pr.pi_a, pr.pi_b, pr.pi_c, pr.publicIO))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_rule():
    from nim_groth16_amd.distributed import shardRange
    for N in (0, 1, 7, 8, 1000, (1 << 20) - 3):
        for world in (1, 2, 3, 8):
            edges = [shardRange(N, k, world) for k in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == N
            assert all(edges[k][1] == edges[k + 1][0] for k in range(world - 1))     # contiguous, no overlap
            assert all(b == (N * (k + 1)) // world for k, (_, b) in enumerate(edges))  # msm.nim:109


@pytest.mark.timeout(300)
def test_sharded_proof_world2_gloo():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(rk, 2, port, q)) for rk in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oz, r, s = _setup()
    ref = o.generate_proof_with_mask(oz, o.TOY_WITNESS, r, s)
    for (_, pa, pb, pc, pub) in outs:
        assert (o.g1_from_bytes(pa), o.g2_from_bytes(pb), o.g1_from_bytes(pc)) == (ref.pi_a, ref.pi_b, ref.pi_c)
        assert I.fr_from_mont(pub) == ref.publicIO
    assert o.verify_proof(oz, ref)
