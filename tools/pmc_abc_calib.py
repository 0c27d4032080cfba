#!/usr/bin/env python3
"""Calibration workload for reading FETCH_SIZE on the sparse kernel's own access pattern (MI355X_MICROARCH.md, HBM
section: "calibrate on a known byte count in your own access pattern"): the row-balanced kernel on a matrix whose gathers
all hit ONE element (col = 0), so that every fetched byte belongs to the entry STREAMS, whose size is known exactly:
  plain values:  nnz * (32 B value + 4 B column)  + 8 B per row (offsets, row ids)
  dictionary:    nnz * ( 4 B index + 4 B column)  + 8 B per row
Run under `rocprofv3 --pmc FETCH_SIZE`; tools/pmc_abc.py divides the counted bytes by these."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nim_groth16_amd import Context  # noqa: E402
from nim_groth16_amd import bn128 as F  # noqa: E402

NROWS, PER = 786432, 16


def main():
    ctx = Context(0)
    nnz = NROWS * PER
    row = np.repeat(np.arange(NROWS, dtype=np.uint32), PER)
    col = np.zeros(nnz, dtype=np.uint32)
    rng = np.random.default_rng(5)
    x = F.frSeqToMontBytes([3])
    for name in ("plain", "dict"):
        if name == "plain":
            val = rng.integers(0, 256, size=(nnz, 32), dtype=np.uint8)
            val[:, 31] &= 0x1F                      # < 2^253 < r: canonical residues
        else:
            tab = np.frombuffer(F.frSeqToMontBytes([1, F.primeR - 1, 5, 7]), dtype=np.uint8).reshape(4, 32)
            val = tab[rng.integers(0, 4, size=nnz)]
        for _ in range(3):
            y = ctx.spmv(row, col, val, x, NROWS)
        print(name, "nnz", nnz, "rows", NROWS, "stream bytes",
              nnz * (36 if name == "plain" else 8) + 8 * NROWS, "y[0][:8]", y[:8].hex(), flush=True)


if __name__ == "__main__":
    main()
