#!/usr/bin/env python3
"""Prove from circom/snarkjs artifacts on the GPU (the `-p` path of the reference CLI, cli/cli_main.nim:162-231):
    python tools/prove_files.py --zkey circuit.zkey --wtns witness.wtns -o proof.json -i public.json [-k vkey.json] [-y]
The outputs are snarkjs-compatible (`snarkjs groth16 verify vkey.json public.json proof.json`)."""
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
    ap.add_argument("-y", "--verify", action="store_true",
                    help="also verify the proof against the zkey's verification key (cli_main.nim -y)")
    args = ap.parse_args()
    ctx = Context(0)
    ctx.selftest()
    t0 = time.time()
    zkey, wtns = parseZKey(args.zkey), parseWitness(args.wtns)
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


if __name__ == "__main__":
    main()
