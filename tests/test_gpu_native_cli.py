"""tools/g16prove.cpp: the prove path of the reference CLI (cli/cli_main.nim -p ... -y) as a native C++ program over the
C ABI (no Python in the process).  Built with g++ here; its proof.json must equal, byte for byte, the one the Python
host mirror writes for the same files and the trivial mask, and must verify."""
import json
import os
import subprocess

import pytest

from oracle import bn254_ref as o

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    csrc = os.path.join(ROOT, "nim_groth16_amd", "csrc")
    exe = str(tmp_path / "g16prove")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "g16prove.cpp"), "-L" + csrc, "-lg16hip", "-Wl,-rpath," + csrc,
                           "-o", exe])
    return exe


@pytest.mark.parametrize("circuit", ["toy", "chain12"])
def test_native_cli_matches_python_host(ctx, tmp_path, circuit):
    from nim_groth16_amd import generateProofWithTrivialMask
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.files import exportProof, exportPublicIO, parseWitness, parseZKey, writeWitness, writeZKey
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    rng = SplitMix64(77)
    tw = ToxicWaste(*[rng.fr() for _ in range(5)])
    if circuit == "toy":
        r1cs, wit = R1CS(8, 1, 1, 3, o.toy_r1cs().constraints), o.TOY_WITNESS
    else:
        r1cs, wit = squaringChain((1 << 12) - 2, seed=4)
    zk = fakeCircuitSetup(r1cs, tw, 1, ctx)
    zpath, wpath = str(tmp_path / "c.zkey"), str(tmp_path / "c.wtns")
    writeZKey(zpath, zk)
    writeWitness(wpath, wit)
    # the Python host mirror
    pr = generateProofWithTrivialMask(0, False, parseZKey(zpath), parseWitness(wpath), ctx)
    exportProof(str(tmp_path / "py_proof.json"), pr)
    exportPublicIO(str(tmp_path / "py_public.json"), pr)
    # the native host
    exe = _build(tmp_path)
    out = subprocess.run([exe, "-z", zpath, "-w", wpath, "-o", str(tmp_path / "proof.json"), "-i",
                          str(tmp_path / "public.json"), "-n", "-y", "-t"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "verification succeeded" in out.stdout
    assert open(tmp_path / "proof.json").read() == open(tmp_path / "py_proof.json").read()
    assert open(tmp_path / "public.json").read() == open(tmp_path / "py_public.json").read()
    # the proof sharded over a device group (three members, all on this box's one GPU): the same bytes
    out = subprocess.run([exe, "-z", zpath, "-w", wpath, "-o", str(tmp_path / "proof_g.json"), "-i",
                          str(tmp_path / "public_g.json"), "-n", "-y", "--gpus", "0,0,0"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "verification succeeded" in out.stdout, out.stderr
    assert open(tmp_path / "proof_g.json").read() == open(tmp_path / "py_proof.json").read()
    # a random mask still verifies (on the GPU verifier inside the tool) and gives a different proof
    out = subprocess.run([exe, "-z", zpath, "-w", wpath, "-o", str(tmp_path / "proof2.json"), "-i",
                          str(tmp_path / "public2.json"), "-y"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "verification succeeded" in out.stdout, out.stderr
    assert json.load(open(tmp_path / "proof2.json"))["pi_a"] != json.load(open(tmp_path / "proof.json"))["pi_a"]


def test_key_from_the_unparsed_coefficient_section(ctx, orc, tmp_path):
    """g16_pkey_create_zkey: the .zkey's section 4 goes to the library as it lies on disk (44-byte entries, values c R^2:
    files/zkey.nim:169-192, io.nim:134-139).  buildABC and the proof must equal the parsed key's, for a key that runs on a
    value dictionary (Poseidon shape) and for one that does not (all-distinct values), for both witness layouts; a
    malformed section is refused."""
    from nim_groth16_amd import Mask, Witness, generateProofWithMask, loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd._lib import G16Error
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.files import parseZKey, writeZKey
    from nim_groth16_amd.synthetic import SplitMix64, poseidonMerkle
    from nim_groth16_amd.zkey_types import packCoeffs
    rng = SplitMix64(31)
    tw = ToxicWaste(*[rng.fr() for _ in range(5)])
    r, s = rng.fr(), rng.fr()
    circuits = [poseidonMerkle(11, seed=4)]
    nw, cons = 300, []
    for i in range(250):                              # > 1024 entries, every value distinct: no dictionary
        cons.append(([(1 + rng.next() % (nw - 1), rng.fr()) for _ in range(1 + i % 7)],
                     [(1 + rng.next() % (nw - 1), rng.fr()) for _ in range(1 + i % 3)], []))
    circuits.append((R1CS(nw, 1, 0, nw - 2, cons), [1] + [rng.fr() for _ in range(nw - 1)]))
    for k, (r1cs, wit) in enumerate(circuits):
        zk = fakeCircuitSetup(r1cs, tw, 1, ctx)
        path = str(tmp_path / f"c{k}.zkey")
        writeZKey(path, zk)
        parsed, raw = parseZKey(path), parseZKey(path, rawCoeffs=True)
        wb, ws = F.frSeqToMontBytes(wit), F.frSeqToStdBytes(wit)
        pk_p, pk_r = loadProvingKey(parsed, ctx), loadProvingKey(raw, ctx)
        try:
            assert pk_r.abc_info() == pk_p.abc_info() and (pk_r.abc_info()["dict_values"] > 0) == (k == 0)
            want = orc.build_abc(packCoeffs(parsed.coeffs), wb, parsed.header.logDomainSize)
            assert pk_r.build_abc(wb) == want and pk_r.build_abc(ws, mont=False) == want
            if k == 0:                                # (the random second circuit is not satisfied by its witness)
                pr = generateProofWithMask(0, False, parsed, Witness("bn128", len(wit), wb), Mask(r, s), ctx, pkey=pk_p)
                pq = generateProofWithMask(0, False, raw, Witness("bn128", len(wit), ws, std=True), Mask(r, s), ctx, pkey=pk_r)
                assert (pq.pi_a, pq.pi_b, pq.pi_c) == (pr.pi_a, pr.pi_b, pr.pi_c)
        finally:
            pk_p.destroy()
            pk_r.destroy()
    bad = raw.coeffsSection4[:-1]
    raw.coeffsSection4 = bad
    with pytest.raises(G16Error):
        loadProvingKey(raw, ctx)
    raw.coeffsSection4 = (1).to_bytes(4, "little") + (2).to_bytes(4, "little") + bytes(40)                # matrix 2 = MatrixC
    with pytest.raises(G16Error):
        loadProvingKey(raw, ctx)


def test_native_cli_rejects_bad_files(tmp_path):
    exe = _build(tmp_path)
    bad = tmp_path / "x.zkey"
    bad.write_bytes(b"nope" + bytes(20))
    out = subprocess.run([exe, "-z", str(bad), "-w", str(bad)], capture_output=True, text=True)
    assert out.returncode != 0 and "not a `zkey` file" in out.stderr


def test_snarkjs_laid_out_files_prove_and_verify_python_and_native(ctx, tmp_path):
    """config-5 stand-in: a `.zkey` / `.wtns` pair laid out the way snarkjs writes them (tests/snarkjs_layout.py:
    sections out of order, a section 10, (0,0) points on unused wires, dummy public rows; serialised independently of
    the product's writers) -> parse -> GPU prove -> verify, through the Python host AND the native C++ caller; the
    proof equals the oracle's for the trivial mask."""
    from nim_groth16_amd import extractVKey, generateProofWithTrivialMask, verifyProof
    from nim_groth16_amd.files import exportProof, exportPublicIO, parseWitness, parseZKey
    from tests.test_files_cpu import _snarkjs_like
    zpath, wpath, oz, wit, _ = _snarkjs_like(tmp_path)
    zk, wt = parseZKey(zpath, check=True, ctx=ctx), parseWitness(wpath)       # incl. the on-curve pass over every section
    pr = generateProofWithTrivialMask(0, False, zk, wt, ctx)
    ref = o.generate_proof_with_mask(oz, wit, 0, 0)
    assert (o.g1_from_bytes(pr.pi_a), o.g2_from_bytes(pr.pi_b), o.g1_from_bytes(pr.pi_c)) == (ref.pi_a, ref.pi_b, ref.pi_c)
    assert o.verify_proof(oz, ref) and verifyProof(extractVKey(zk), pr, ctx)
    exportProof(str(tmp_path / "py_proof.json"), pr)
    exportPublicIO(str(tmp_path / "py_public.json"), pr)
    exe = _build(tmp_path)
    out = subprocess.run([exe, "-z", zpath, "-w", wpath, "-o", str(tmp_path / "proof.json"), "-i",
                          str(tmp_path / "public.json"), "-n", "-y"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "verification succeeded" in out.stdout, out.stderr
    assert open(tmp_path / "proof.json").read() == open(tmp_path / "py_proof.json").read()
    assert open(tmp_path / "public.json").read() == open(tmp_path / "py_public.json").read()


def test_config5_handoff_recipe_on_external_files(ctx, tmp_path):
    """tools/prove_files.py -c -y -t -k: the checked recipe for files that arrive from outside (BASELINE config 5:
    a real circom/snarkjs pair cannot be produced in the build container).  Here the files are a snarkjs-laid-out
    fixture and a key in which ~90 % of the B points are (0,0); the tool must check every point, prove, verify on the
    GPU, write proof / public / verification_key JSON, and report the infinity fractions it found."""
    import sys
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.files import writeWitness, writeZKey
    from nim_groth16_amd.synthetic import SplitMix64, mixedCircuit
    from tests.test_files_cpu import _snarkjs_like
    tool = os.path.join(ROOT, "tools", "prove_files.py")

    def run(zpath, wpath, tag):
        out = subprocess.run([sys.executable, tool, "-z", zpath, "-w", wpath, "-c", "-y", "-t", "-o",
                              str(tmp_path / f"{tag}_proof.json"), "-i", str(tmp_path / f"{tag}_public.json"), "-k",
                              str(tmp_path / f"{tag}_vkey.json")], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "verification succeeded" in out.stdout and "points at infinity (0,0): A1 " in out.stdout
        pj, vk = json.load(open(tmp_path / f"{tag}_proof.json")), json.load(open(tmp_path / f"{tag}_vkey.json"))
        assert pj["protocol"] == "groth16" and pj["curve"] == "bn128" and len(pj["pi_b"]) == 3
        assert vk["protocol"] == "groth16" and len(vk["IC"]) == vk["nPublic"] + 1
        return out.stdout

    zpath, wpath, _, _, _ = _snarkjs_like(tmp_path)
    run(zpath, wpath, "fixture")
    m = (1 << 12) - 2
    r1cs, wit = mixedCircuit(m, seed=4, zero_pct=0, one_pct=0, lin_pct=90)
    rng = SplitMix64(9)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
    z2, w2 = str(tmp_path / "lin.zkey"), str(tmp_path / "lin.wtns")
    writeZKey(z2, zk)
    writeWitness(w2, wit)
    text = run(z2, w2, "lin")
    assert "B1+B2 compacted" in text and "A1 shared witness sort" in text
