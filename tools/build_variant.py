"""A/B builds of kernel variants: compile ONE source of libenoki-hip.so with extra -D switches and link it with the product's other
objects into build/variants/libenoki-hip-<name>.so (travels to the GPU box with the snapshot).  Inside one gpurun call the
libraries are swapped over enoki_amd/libenoki-hip.so (tools/README.md).

    python tools/build_variant.py <name> <source.hip> -DFOO -DBAR=2 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enoki_amd import _build as b

name, src, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
out_dir = os.path.join(ROOT, "build", "variants")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"{os.path.basename(src)}-{name}.o")
subprocess.check_call([b.HIPCC] + b.DEVICE + b.COMMON + b.FILE_FLAGS.get(src, []) + defs + ["-c", os.path.join(b.CSRC, src), "-o", obj])
others = [os.path.join(b.OBJ, s + ".o") for s in b.LIB_SOURCES if s != src]
lib = os.path.join(out_dir, f"libenoki-hip-{name}.so")
subprocess.check_call([b.HIPCC] + b.DEVICE + ["-shared", "-fPIC", "-o", lib, obj] + others)
print(lib)
