"""Host mirror of groth16/files/*.nim: the iden3 binary container and the .zkey / .wtns / .r1cs formats of the
circom / snarkjs ecosystem, plus the JSON export.  Writers (absent from the reference, which only reads) exist
so that fixtures and benchmark keys can be produced without circom/snarkjs."""
from .container import parseContainer, writeContainer  # noqa: F401
from .zkey import parseZKey, writeZKey  # noqa: F401
from .witness import parseWitness, writeWitness  # noqa: F401
from .r1cs import parseR1CS, writeR1CS  # noqa: F401
from .export_json import exportProof, exportPublicIO, exportVKey  # noqa: F401
