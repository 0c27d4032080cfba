"""Register / LDS / scratch footprint of every kernel in the built objects (llvm-readelf --notes of the gfx950 code
objects): python tools/kernel_resources.py [objects...]   (default: every .o in nim_groth16_amd/csrc)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(obj):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        subprocess.check_call(["cp", obj, local])
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not co:
            return []
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, co[0])], check=True,
                               capture_output=True, text=True).stdout
    out, cur = [], {}
    for ln in notes.split("\n"):
        m = re.match(r"\s+-?\s*\.(agpr_count|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|name|max_flat_workgroup_size):\s+(\S+)", ln)
        if not m:
            continue
        k, v = m.groups()
        if k == "agpr_count" and cur.get("name"):
            out.append(cur)
            cur = {}
        cur[k] = v
    if cur.get("name"):
        out.append(cur)
    return out


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()


if __name__ == "__main__":
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "nim_groth16_amd", "csrc", "*.o")))
    seen = set()
    for o in objs:
        for k in kernels(o):
            name = re.sub(r"g16::|\(anonymous namespace\)::", "", demangle(k["name"]))
            name = re.sub(r"Curve<Field<FpParams> ?>", "G1", re.sub(r"Curve<Fp2>", "G2", name))
            name = re.split(r"\(", name)[0].replace("void ", "")
            if (name, k.get("vgpr_count")) in seen:
                continue
            seen.add((name, k.get("vgpr_count")))
            print(f"{os.path.basename(o):20s} {name:50s} vgpr {k.get('vgpr_count'):>4s} agpr {k.get('agpr_count'):>3s} "
                  f"lds {k.get('group_segment_fixed_size'):>6s} scratch {k.get('private_segment_fixed_size'):>5s} "
                  f"wg {k.get('max_flat_workgroup_size')}")
