"""A/B builds of kernel variants: compile ONE source of libenoki-hip.so with extra -D switches and link it with the product's other
objects into build/variants/libenoki-hip-<name>.so (travels to the GPU box with the snapshot).  Inside one gpurun call the
libraries are swapped over enoki_amd/libenoki-hip.so (tools/README.md).

    python tools/build_variant.py <name> <source.hip>[,<source2.hip>] -DFOO -DBAR=2 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enoki_amd import _build as b

name, srcs, defs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]          # (several sources: "a.hip,b.hip" -- a switch in a shared header)
out_dir = os.path.join(ROOT, "build", "variants")
os.makedirs(out_dir, exist_ok=True)
objs = []
for src in srcs:
    obj = os.path.join(out_dir, f"{os.path.basename(src)}-{name}.o")
    subprocess.check_call([b.HIPCC] + b.DEVICE + b.COMMON + b.FILE_FLAGS.get(src, []) + defs + ["-c", os.path.join(b.CSRC, src), "-o", obj])
    objs.append(obj)
others = [os.path.join(b.OBJ, s + ".o") for s in b.LIB_SOURCES if s not in srcs]
lib = os.path.join(out_dir, f"libenoki-hip-{name}.so")
subprocess.check_call([b.HIPCC] + b.DEVICE + ["-shared", "-fPIC", "-o", lib] + objs + others)
print(lib)
