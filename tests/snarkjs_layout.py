"""Hand-built `.zkey` / `.wtns` files laid out the way snarkjs really writes them (TEST INFRASTRUCTURE).

Built from the format description the reference carries (groth16/files/zkey.nim:6-91, files/container.nim:6-20,
files/witness.nim:5-15) and independent of the product's own writers -- so that the readers are not only checked
against themselves.  What a real snarkjs key has and the product's writeZKey output has not:
  * sections NOT in ascending id order (snarkjs writes 1, 2, 4, 3, 9, 8, 5, 6, 7, 10);
  * a section 10 (the ceremony's contribution log) that a prover must skip;
  * section-4 entries grouped by matrix and row in the order the constraints were compiled, with the npubs+1 dummy
    `1 * w_i` rows of matrix A behind the real constraints (snarkjs zkey_new; fake_setup.nim:59-63 restates them);
  * (0,0) = infinity points for every wire that does not occur in A (resp. B);
  * nothing after the last section, sizes as u64."""
import struct

P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
MONT = 1 << 256


def _container(magic: bytes, version: int, sections) -> bytes:
    out = bytearray(magic + struct.pack("<II", version, len(sections)))
    for sid, data in sections:
        out += struct.pack("<IQ", sid, len(data)) + data
    return bytes(out)


def snarkjs_zkey_bytes(nvars, npubs, domain_size, spec, ic, a1, b1, b2, c1, h1, coeffs, order=(1, 2, 4, 3, 9, 8, 5, 6, 7, 10)):
    """spec = (alpha1, beta1, beta2, gamma2, delta1, delta2) byte strings; point arguments: Montgomery affine bytes as
    in memory; coeffs: list of (matrix, row, col, value int in standard form)."""
    le32 = lambda x: int(x).to_bytes(32, "little")           # noqa: E731
    s2 = (struct.pack("<I", 32) + le32(P) + struct.pack("<I", 32) + le32(R) + struct.pack("<III", nvars, npubs, domain_size)
          + b"".join(spec))
    s4 = bytearray(struct.pack("<I", len(coeffs)))
    for (m, row, col, v) in coeffs:
        s4 += struct.pack("<III", m, row, col) + le32(v % R * MONT % R * MONT % R)     # DOUBLE Montgomery (zkey.nim:57)
    # section 10: csHash (64 B) + number of contributions + one fake contribution record; content is irrelevant to a
    # prover, it only has to be skipped
    s10 = bytes(range(64)) + struct.pack("<I", 1) + bytes(200)
    body = {1: struct.pack("<I", 1), 2: s2, 3: ic, 4: bytes(s4), 5: a1, 6: b1, 7: b2, 8: c1, 9: h1, 10: s10}
    return _container(b"zkey", 1, [(sid, body[sid]) for sid in order])


def snarkjs_wtns_bytes(values, order=(1, 2)):
    """.wtns v2: header (n8, r, nvars) + nvars x 32 B standard-form little-endian (witness.nim:5-15)"""
    s1 = struct.pack("<I", 32) + R.to_bytes(32, "little") + struct.pack("<I", len(values))
    s2 = b"".join((v % R).to_bytes(32, "little") for v in values)
    body = {1: s1, 2: s2}
    return _container(b"wtns", 2, [(sid, body[sid]) for sid in order])


def unused_wire_circuit():
    """A circuit shaped like circom output for
         signal input a, b;  signal output c;  signal t;   t <== a * b;   c <== t * t + a;
    wires [1, c, a, b, t]; constraints (A)(B) = (C):  a*b = t ;  t*t = c - a.
    Wire b occurs only in B and wire a only in A, so pointsA1[b], pointsB1[a], pointsB2[a] are the point at
    infinity, stored as (0,0); the public wires 1 and c occur in no product at all and get an A point only from
    their dummy rows.
    -> (nWires, nPubOut, nPubIn, nPrivIn, constraints, witness)"""
    cons = [([(2, 1)], [(3, 1)], [(4, 1)]),
            ([(4, 1)], [(4, 1)], [(1, 1), (2, R - 1)])]
    a, b = 12345, 67890
    t = a * b % R
    c = (t * t + a) % R
    return 5, 1, 0, 3, cons, [1, c, a, b, t]
