#!/usr/bin/env python3
"""Prove from circom/snarkjs artifacts on the GPU (the `-p` path of the reference CLI, cli/cli_main.nim:162-231):
    python tools/prove_files.py --zkey circuit.zkey --wtns witness.wtns -o proof.json -i public.json [-k vkey.json] [-y] [-c]
The outputs are snarkjs-compatible (`snarkjs groth16 verify vkey.json public.json proof.json`).

BASELINE config 5 hand-off (a real circom Poseidon-Merkle .zkey cannot be produced in the build container: no circom,
snarkjs, ptau or network).  Given external files the checked recipe is
    python tools/prove_files.py -z circuit.zkey -w witness.wtns -c -y -t -k verification_key.json
      -c  every point of the key is checked to be on its curve on the GPU (the mkG1 / mkG2 asserts of the reference's
          loaders, curves.nim:95-107) and the witness is checked against the key's header
      -y  the proof is verified on the GPU against the key's own verification key (verifier.nim:31-52)
      -t  phase timings, and the fraction of points at infinity per ProverPoints array together with whether the
          library put A1 / B1+B2 on compacted entry lists (g16_pkey_inf_counts)
    snarkjs groth16 verify verification_key.json public.json proof.json        # offline, wherever snarkjs exists
(the reference's own recipe: groth16/example/prove.sh:52-59)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import secrets  # noqa: E402

from nim_groth16_amd import Context, Mask, generateProofWithMask, loadProvingKey  # noqa: E402
from nim_groth16_amd import bn128 as F  # noqa: E402
from nim_groth16_amd.files import exportProof, exportPublicIO, parseWitness, parseZKey  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-z", "--zkey", required=True)
    ap.add_argument("-w", "--wtns", required=True)
    ap.add_argument("-o", "--output", default="proof.json")
    ap.add_argument("-i", "--io", default="public.json")
    ap.add_argument("-n", "--nomask", action="store_true", help="trivial mask r = s = 0 (cli_main.nim -n)")
    ap.add_argument("-t", "--time", action="store_true")
    ap.add_argument("-k", "--vkey", default=None, help="also write the verification key as snarkjs verification_key.json")
    ap.add_argument("-c", "--check", action="store_true",
                    help="check every key point against its curve equation on the GPU while parsing")
    ap.add_argument("-y", "--verify", action="store_true",
                    help="also verify the proof against the zkey's verification key (cli_main.nim -y)")
    args = ap.parse_args()
    ctx = Context(0)
    ctx.selftest()
    t0 = time.time()
    zkey, wtns = parseZKey(args.zkey, check=args.check, ctx=ctx, rawCoeffs=True), parseWitness(args.wtns)
    assert wtns.nvars == zkey.header.nvars, "wrong witness length"        # prover.nim:236
    t1 = time.time()
    pkey = loadProvingKey(zkey, ctx)
    t2 = time.time()
    mask = Mask(0, 0) if args.nomask else Mask(secrets.randbelow(F.primeR), secrets.randbelow(F.primeR))
    proof = generateProofWithMask(0, args.time, zkey, wtns, mask, ctx, pkey=pkey)
    t3 = time.time()
    exportProof(args.output, proof)
    exportPublicIO(args.io, proof)
    if args.vkey:
        from nim_groth16_amd import extractVKey
        from nim_groth16_amd.files import exportVKey
        exportVKey(args.vkey, extractVKey(zkey))
    if args.verify:
        from nim_groth16_amd import extractVKey, verifyProof
        ok = verifyProof(extractVKey(zkey), proof, ctx)
        print("verification " + ("succeeded" if ok else "FAILED"))
        if not ok:
            raise SystemExit(1)
    if args.time:
        print(f"parsing {t1-t0:.3f}s | key upload + tables {t2-t1:.3f}s | proof {t3-t2:.3f}s")
        hdr, inf = zkey.header, pkey.inf_counts()
        sizes = {"A1": hdr.nvars, "B1": hdr.nvars, "B2": hdr.nvars, "C1": hdr.nvars - hdr.npubs - 1, "H1": hdr.domainSize}
        print("points at infinity (0,0): " + ", ".join(f"{k} {inf[k]}/{n} ({100.0 * inf[k] / max(n, 1):.1f} %)"
                                                      for k, n in sizes.items()))
        print(f"entry lists: A1 {'compacted' if inf['compact_A'] else 'shared witness sort'}, B1+B2 "
              f"{'compacted (their own sort without the ' + str(inf['B1_and_B2']) + ' dead wires)' if inf['compact_B'] else 'shared witness sort'}")


if __name__ == "__main__":
    main()
