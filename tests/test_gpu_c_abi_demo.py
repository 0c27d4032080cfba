"""examples/c_abi_demo.c: a plain-C program (gcc, no Python / torch in the process) linked against libg16hip.so --
the C ABI as a compiled host would use it."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_caller(tmp_path):
    csrc = os.path.join(ROOT, "nim_groth16_amd", "csrc")
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + csrc, "-lg16hip",
                           "-Wl,-rpath," + csrc, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "C ABI demo OK" in out.stdout
