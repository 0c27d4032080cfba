"""groth16/files/container.nim: magic u32 | version u32 | nsections u32 | { id u32, size u64, data }*"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple


def magicWord(magic: str) -> int:
    """container.nim:38-44"""
    assert len(magic) == 4, "magicWord: expecting a string of 4 characters"
    return sum(ord(magic[i]) << (8 * i) for i in range(4))


def parsePrimeField(buf: memoryview, pos: int) -> Tuple[int, int, int]:
    """container.nim:48-55 -> (n8, prime, new position)"""
    (n8,) = struct.unpack_from("<I", buf, pos)
    assert n8 <= 32, "at most 256 bit primes are allowed"
    p = int.from_bytes(bytes(buf[pos + 4:pos + 4 + n8]), "little")
    return n8, p, pos + 4 + n8


def parseContainer(expectedMagic: str, expectedVersion: int, fname: str) -> Dict[int, List[memoryview]]:
    """container.nim:75-93.  Returns {section id: [memoryview of the section data, ...]} over a read-only
    memory map of the file (large point sections are never copied on the host)."""
    import mmap
    f = open(fname, "rb")
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    buf = memoryview(mm)
    magic, version, nsections = struct.unpack_from("<III", buf, 0)
    assert magic == magicWord(expectedMagic), f"not a `{expectedMagic}` file"
    assert version == expectedVersion, f"not a version {expectedVersion} `{expectedMagic}` file"
    pos = 12
    out: Dict[int, List[memoryview]] = {}
    for _ in range(nsections):
        sid, slen = struct.unpack_from("<IQ", buf, pos)
        pos += 12
        assert pos + slen <= len(buf), "truncated section"
        out.setdefault(sid, []).append(buf[pos:pos + slen])
        pos += slen
    return out


def writeContainer(magic: str, version: int, fname: str, sections: List[Tuple[int, bytes]]) -> None:
    with open(fname, "wb") as f:
        f.write(struct.pack("<III", magicWord(magic), version, len(sections)))
        for sid, data in sections:
            f.write(struct.pack("<IQ", sid, len(data)))
            f.write(data)
