import os, sys
sys.path.insert(0, "tests")
import numpy as np, tape_lib as tl
u = 2.0 ** -24
GOLDEN = os.path.join(tl.HERE, "golden")
for name in sorted(tl.ORDER_DEPENDENT_ON_GPU):
    prog = tl.suite()[name]
    z = np.load(os.path.join(GOLDEN, f"tape_{name}.npz"))
    rv = z["value"]; rg = [z[f"g{i}"] if f"g{i}" in z else None for i in range(int(z["n_grads"]))]
    gv, gg = tl.run(tl.hip_lib().hip_tape_program, prog)
    n = max(a.size for a, _ in prog.inputs)
    out = [f"value err/u {float(np.abs(gv - rv).max()) / u:10.1f} |v| {float(np.abs(rv).max()):9.3g}"]
    for i, (a, b) in enumerate(zip(rg, gg)):
        if a is not None:
            out.append(f"g{i} err/u {float(np.abs(a - b).max()) / u:8.1f} max|g| {float(np.abs(a).max()):9.3g} size {a.size}")
    print(f"{name:26s} n={n}", " | ".join(out))
