"""In-tree build of the native libraries (gfx950 only).

    python -m enoki_amd._build            # everything that is out of date
    python -m enoki_amd._build --force

Products (git-ignored, but shipped to the GPU box with the tree):
    enoki_amd/libenoki-hip.so            C ABI + HIP kernels      (csrc/*.hip, csrc/runtime.cpp)
    enoki_amd/libenoki-hip-probe.so      measurement probes       (csrc/probe.hip; tools/ only)
    enoki_amd/libenoki-hip-autodiff.so   Tape<HIPArray<float>>    (src/autodiff.cpp)          [if present]
    enoki_amd/hip*.so                    pybind11 modules         (python/*.cpp)              [if present]

hipcc cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only dev container.
"""
import concurrent.futures
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: only explicit fma() calls fuse (bit parity with the reference CPU path)
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-math-errno", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function", f"-I{os.path.join(ROOT, 'include')}"]
DEVICE = [f"--offload-arch={ARCH}"]

LIB_SOURCES = ["runtime.cpp", "dist.cpp", "elementwise.hip", "memory.hip", "reduce.hip", "scatter_binned.hip", "random.hip", "gathered.hip", "scan.hip", "bucketed.hip", "bucketed_early.hip"]
# measurement scaffolding (tools/probe_*.py): its own library on top of the public C ABI, never loaded by the product
PROBE_SOURCES = ["probe.hip", "probe_lds64.hip", "probe_valu.hip", "probe_rt.cpp"]


def kernels_sha16():
    """fingerprint of the device code: every source and header that goes into libenoki-hip.so, in a fixed order"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")) and not f.startswith("probe"))
    dev = os.path.join(ROOT, "include", "enoki", "device")
    files += sorted(os.path.join(dev, f) for f in os.listdir(dev) if f.endswith(".h"))
    files.append(os.path.join(ROOT, "include", "enoki_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(ROOT, "include")
    for base, _, files in os.walk(inc):
        hs += [os.path.join(base, f) for f in files if f.endswith(".h")]
    return hs


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + r.stdout)
    return r.stdout


# per-file code generation switches
#   bucketed_early.hip: the wave-level rewrite of uniform atomics ("atomic optimizer") turns the one-lane page-batch request of
#   walk_pages_dynamic into a ballot / mbcnt / readfirstlane sequence that WAITS for the returning ds_add on the spot
#   (s_waitcnt lgkmcnt(0): every LDS add of the previous step) -- without it the value is picked up a step later, for free
FILE_FLAGS = {"bucketed_early.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if force or _newer(obj, [src] + _headers()):
        # ENOKI_PROBE_DEFINES="-DEK_PG_TIMING" builds the measurement library (probe.hip only) with instrumentation
        extra = os.environ.get("ENOKI_PROBE_DEFINES", "").split() if os.path.basename(src).startswith("probe") else []
        extra = extra + FILE_FLAGS.get(os.path.basename(src), [])
        cmd = [HIPCC] + DEVICE + COMMON + extra + (["-x", "hip"] if src.endswith(".cpp") and "csrc" in src else []) + ["-c", src, "-o", obj]
        _run(cmd)
    return obj


def build_core(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    sources = [os.path.join(CSRC, s) for s in LIB_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(sources))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), sources))
    lib = os.path.join(HERE, "libenoki-hip.so")
    if force or _newer(lib, objs):
        _run([HIPCC] + DEVICE + ["-shared", "-fPIC", "-o", lib] + objs)
        if verbose:
            print(f"[enoki_amd] built {os.path.relpath(lib, ROOT)}")
    return lib


def build_probe(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    sources = [os.path.join(CSRC, s) for s in PROBE_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = [_compile(s, force) for s in sources]
    lib = os.path.join(HERE, "libenoki-hip-probe.so")
    if force or _newer(lib, objs + [os.path.join(HERE, "libenoki-hip.so")]):
        _run([HIPCC] + DEVICE + ["-shared", "-fPIC", "-o", lib] + objs + [f"-L{HERE}", "-lenoki-hip", "-Wl,-rpath,$ORIGIN"])
        if verbose:
            print(f"[enoki_amd] built {os.path.relpath(lib, ROOT)}")
    return lib


def build_autodiff(force=False, verbose=True):
    src = os.path.join(HERE, "src", "autodiff.cpp")
    if not os.path.exists(src):
        return None
    os.makedirs(OBJ, exist_ok=True)
    lib = os.path.join(HERE, "libenoki-hip-autodiff.so")
    if force or _newer(lib, [src, os.path.join(HERE, "src", "autodiff_impl.h")] + _headers()):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", f"-I{os.path.join(ROOT, 'include')}",
              src, "-o", lib, f"-L{HERE}", "-lenoki-hip", "-Wl,-rpath,$ORIGIN"])
        if verbose:
            print(f"[enoki_amd] built {os.path.relpath(lib, ROOT)}")
    return lib


def build_python(force=False, verbose=True):
    pydir = os.path.join(HERE, "python")
    if not os.path.isdir(pydir):
        return []
    import pybind11
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    inc = [f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}", f"-I{os.path.join(ROOT, 'include')}"]
    out = []
    jobs = []
    for f in sorted(os.listdir(pydir)):
        if not f.endswith(".cpp"):
            continue
        src = os.path.join(pydir, f)
        mod = os.path.join(HERE, f[:-4] + ext)
        out.append(mod)
        if force or _newer(mod, [src] + _headers() + [os.path.join(pydir, h) for h in os.listdir(pydir) if h.endswith(".h")]):
            jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall"] + inc +
                        [src, "-o", mod, f"-L{HERE}", "-lenoki-hip-autodiff", "-lenoki-hip", "-Wl,-rpath,$ORIGIN"])
    # a downstream extension module written against the headers only (tests/test_user_extension_gpu.py)
    user_src = os.path.join(ROOT, "tests", "cpp", "user_ext", "user_ext.cpp")
    if os.path.exists(user_src):
        user_mod = os.path.join(os.path.dirname(user_src), "user_ext" + ext)
        if force or _newer(user_mod, [user_src] + _headers()):
            jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall"] + inc +
                        [user_src, "-o", user_mod, f"-L{HERE}", "-lenoki-hip-autodiff", "-lenoki-hip",
                         "-Wl,-rpath,$ORIGIN/../../../enoki_amd"])
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(_run, jobs))
        if verbose:
            print(f"[enoki_amd] built {len(jobs)} python module(s)")
    return out


def build_checkers(force=False, verbose=True):
    """Test infrastructure: the C oracle, the reference-built oracle (only where /root/reference exists) and
    the register-machine harnesses of tests/cpp.  Building the checkers is not using them."""
    oracle = os.path.join(ROOT, "oracle")
    _run(["make", "-C", oracle, "port"])
    if os.path.isdir("/root/reference"):
        _run(["make", "-C", oracle, "ref", "ref512", "refscalar"])
    tcpp = os.path.join(ROOT, "tests", "cpp")
    inc = f"-I{os.path.join(ROOT, 'include')}"
    host = os.path.join(tcpp, "libtape_host.so")
    deps = _headers() + [os.path.join(tcpp, "tape_program.h"), os.path.join(HERE, "src", "autodiff_impl.h"),
                         os.path.join(oracle, "host_array.h")]
    if force or _newer(host, [os.path.join(tcpp, "tape_host.cpp")] + deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", inc, os.path.join(tcpp, "tape_host.cpp"), "-o", host,
              f"-L{oracle}", "-lenoki_oracle", "-Wl,-rpath,$ORIGIN/../../oracle"])
    hip = os.path.join(tcpp, "libtape_hip.so")
    if force or _newer(hip, [os.path.join(tcpp, "tape_hip.cpp"), os.path.join(HERE, "libenoki-hip-autodiff.so")] + deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", inc, os.path.join(tcpp, "tape_hip.cpp"), "-o", hip,
              f"-L{HERE}", "-lenoki-hip-autodiff", "-lenoki-hip", "-Wl,-rpath,$ORIGIN/../../enoki_amd"])
    # tests/cpp/asan_tape.cpp against the real library (runs on the GPU box), in float32 and float64
    for fuzz_name, fuzz_flags in (("fuzz_tape_hip.bin", []), ("fuzz_tape_hip_f64.bin", ["-DEK_FUZZ_DOUBLE"])):
        fuzz = os.path.join(tcpp, fuzz_name)
        if force or _newer(fuzz, [os.path.join(tcpp, "asan_tape.cpp"), os.path.join(HERE, "libenoki-hip-autodiff.so")] + deps):
            _run(["g++", "-O1", "-std=c++17", "-DEK_FUZZ_DEVICE"] + fuzz_flags + [inc, os.path.join(tcpp, "asan_tape.cpp"), "-o", fuzz,
                  f"-L{HERE}", "-lenoki-hip-autodiff", "-lenoki-hip", "-Wl,-rpath,$ORIGIN/../../enoki_amd"])
    sphere = os.path.join(tcpp, "libsphere_hip.so")
    if force or _newer(sphere, [os.path.join(tcpp, "sphere_hip.cpp"), os.path.join(HERE, "libenoki-hip.so")] + _headers()):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", inc, os.path.join(tcpp, "sphere_hip.cpp"), "-o", sphere,
              f"-L{HERE}", "-lenoki-hip", "-Wl,-rpath,$ORIGIN/../../enoki_amd"])
    compat = os.path.join(tcpp, "compat_names_hip.bin")
    if force or _newer(compat, [os.path.join(tcpp, "compat_names_hip.cpp"), os.path.join(HERE, "libenoki-hip-autodiff.so")] + _headers()):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-DENOKI_HIP_DYNAMIC_IS_DEVICE=1", inc, f"-I{os.path.join(ROOT, 'compat')}", os.path.join(tcpp, "compat_names_hip.cpp"), "-o", compat,
              f"-L{HERE}", "-lenoki-hip-autodiff", "-lenoki-hip", "-Wl,-rpath,$ORIGIN/../../enoki_amd"])
    call = os.path.join(tcpp, "libcall_hip.so")
    if force or _newer(call, [os.path.join(tcpp, "call_hip.cpp"), os.path.join(HERE, "libenoki-hip-autodiff.so")] + _headers()):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", inc, os.path.join(tcpp, "call_hip.cpp"), "-o", call,
              f"-L{HERE}", "-lenoki-hip-autodiff", "-lenoki-hip", "-Wl,-rpath,$ORIGIN/../../enoki_amd"])
    # sanitizer driver of the runtime's host side (profiles/asan_ubsan_allocator_r02.txt); run by hand on a GPU box
    asan = os.path.join(tcpp, "asan_allocator.bin")
    if force or _newer(asan, [os.path.join(tcpp, "asan_allocator.cpp"), os.path.join(CSRC, "runtime.cpp"), os.path.join(CSRC, "ek_internal.h")]):
        _run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
              "-I/opt/rocm/include", inc, os.path.join(tcpp, "asan_allocator.cpp"), "-o", asan, "-L/opt/rocm/lib", "-lamdhip64", "-ldl",
              "-Wl,-rpath,/opt/rocm/lib"])
    # fused (vectorize) == unfused, bit for bit, for every floating point function (hipcc translation unit)
    vmath = os.path.join(tcpp, "libvectorize_math_hip.so")
    if force or _newer(vmath, [os.path.join(tcpp, "vectorize_math_hip.cpp"), os.path.join(HERE, "libenoki-hip.so")] + _headers()):
        _run([HIPCC] + DEVICE + ["-x", "hip", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", inc,
                                 os.path.join(tcpp, "vectorize_math_hip.cpp"), "-o", vmath, f"-L{HERE}", "-lenoki-hip",
                                 "-Wl,-rpath,$ORIGIN/../../enoki_amd"])
    # include/enoki/ellint.h on host packets (CPU check of the elliptic integrals against the reference's golden vectors)
    ell = os.path.join(tcpp, "libellint_host.so")
    if force or _newer(ell, [os.path.join(tcpp, "ellint_host.cpp")] + _headers()):
        _run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", inc, os.path.join(tcpp, "ellint_host.cpp"), "-o", ell])
    mo = os.path.join(tcpp, "morton_host.bin")
    if force or _newer(mo, [os.path.join(tcpp, "morton_host.cpp"), os.path.join(ROOT, "include", "enoki", "morton.h")]):
        _run(["g++", "-O1", "-std=c++17", inc, os.path.join(tcpp, "morton_host.cpp"), "-o", mo])
    hf = os.path.join(tcpp, "half_host.bin")
    if force or _newer(hf, [os.path.join(tcpp, "half_host.cpp"), os.path.join(ROOT, "include", "enoki", "half.h"),
                            os.path.join(ROOT, "include", "enoki", "array.h")]):
        f16c = []
        try:
            with open("/proc/cpuinfo") as f:
                f16c = ["-mf16c"] if " f16c" in f.read() else []       # compare with the instruction the reference's build uses
        except OSError:
            pass
        _run(["g++", "-O1", "-std=c++17"] + f16c + [inc, os.path.join(tcpp, "half_host.cpp"), "-o", hf])
    rt = os.path.join(tcpp, "router_host.bin")
    if force or _newer(rt, [os.path.join(tcpp, "router_host.cpp"), os.path.join(ROOT, "include", "enoki", "array.h")]):
        _run(["g++", "-O1", "-std=c++17", inc, os.path.join(tcpp, "router_host.cpp"), "-o", rt])
    pk = os.path.join(tcpp, "packed_host.bin")
    if force or _newer(pk, [os.path.join(tcpp, "packed_host.cpp"), os.path.join(ROOT, "include", "enoki", "array.h")]):
        _run(["g++", "-O1", "-std=c++17", inc, os.path.join(tcpp, "packed_host.cpp"), "-o", pk])
    po = os.path.join(tcpp, "polar_host.bin")
    if force or _newer(po, [os.path.join(tcpp, "polar_host.cpp")] + _headers()):
        _run(["g++", "-O1", "-std=c++17", inc, os.path.join(tcpp, "polar_host.cpp"), "-o", po])
    shl = os.path.join(tcpp, "libsh_host.so")
    if force or _newer(shl, [os.path.join(tcpp, "sh_host.cpp"), os.path.join(ROOT, "include", "enoki", "sh.h")]):
        _run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", inc, os.path.join(tcpp, "sh_host.cpp"), "-o", shl])
    tr = os.path.join(tcpp, "libtransform_host.so")
    if force or _newer(tr, [os.path.join(tcpp, "transform_host.cpp")] + _headers()):
        _run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", inc, os.path.join(tcpp, "transform_host.cpp"), "-o", tr])
    # host logic of the binding's deferred nodes against a host stand-in of the C ABI, under ASan + LSan + UBSan: needs no
    # GPU, runs in the CPU suite (tests/test_host_sanitizers.py)
    for name, extra in (("asan_deferred", []), ("asan_tape", [os.path.join(HERE, "src", "autodiff_impl.h")])):
        exe = os.path.join(tcpp, name + ".bin")
        if force or _newer(exe, [os.path.join(tcpp, name + ".cpp"), os.path.join(tcpp, "host_abi_stub.h")] + extra + _headers()):
            _run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", inc,
                  os.path.join(tcpp, name + ".cpp"), "-o", exe])
    # The reference's OWN test sources compiled against this repository's headers with the device array types substituted
    # (tests/cpp/refshim): only where the reference tree exists; the binaries travel to the GPU box.
    ref_tests = "/root/reference/tests"
    if os.path.isdir(ref_tests):
        shim = os.path.join(tcpp, "refshim")
        for name, source in (("reftest_autodiff_hip", "autodiff.cpp"),):
            exe = os.path.join(tcpp, name + ".bin")
            src = os.path.join(tcpp, name + ".cpp")
            shim_files = [os.path.join(base, f) for base, _, files in os.walk(shim) for f in files]
            if force or _newer(exe, [src, os.path.join(ref_tests, source), os.path.join(HERE, "libenoki-hip-autodiff.so")] + shim_files + _headers()):
                # -I- : the reference's `#include "test.h"` must find the shim, not the file next to the test source
                _run(["g++", "-O1", "-std=c++17", "-iquote", shim, "-iquote", "/usr/include/c++/11/pstl", "-I-", f"-I{shim}", inc, f"-I{os.path.join(ROOT, 'compat')}",
                      "-DENOKI_AUTODIFF=1", f'-DREFERENCE_TEST_FILE="{os.path.join(ref_tests, source)}"', src, "-o", exe,
                      f"-L{HERE}", "-lenoki-hip-autodiff", "-lenoki-hip", "-Wl,-rpath,$ORIGIN/../../enoki_amd"])
        # the reference's own HEADERS (router, math, struct support) driving this backend through integration/enoki/hip.h, with
        # the reference's CPU arrays in the same binary as the yardstick (reference flags of oracle/Makefile)
        exe = os.path.join(tcpp, "reference_side_hip.bin")
        integ = os.path.join(ROOT, "integration")
        deps = [os.path.join(tcpp, "reference_side_hip.cpp"), os.path.join(integ, "enoki", "hip.h"), os.path.join(integ, "hip_hooks.cpp"),
                os.path.join(ROOT, "include", "enoki_hip.h"), os.path.join(HERE, "libenoki-hip.so")]
        if force or _newer(exe, deps):
            _run(["g++", "-std=c++17", "-O2", "-mavx2", "-mfma", "-mf16c", "-mbmi", "-mbmi2", "-mlzcnt", "-ffp-contract=off", "-fno-math-errno",
                  "-I/root/reference/include", f"-I{integ}", inc, deps[0], deps[2], "-o", exe, f"-L{HERE}", "-lenoki-hip",
                  "-Wl,-rpath,$ORIGIN/../../enoki_amd"])
        # the same binding on the host stand-in of the C ABI under ASan / UBSan (tests/test_host_sanitizers.py): tag bookkeeping
        exe = os.path.join(tcpp, "integration_host.bin")
        if force or _newer(exe, [os.path.join(tcpp, "integration_host.cpp"), os.path.join(tcpp, "host_abi_stub.h"),
                                 os.path.join(integ, "enoki", "hip.h"), os.path.join(integ, "hip_hooks.cpp"),
                                 os.path.join(ROOT, "include", "enoki_hip.h")]):
            _run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-mavx2", "-mfma", "-mf16c",
                  "-mbmi", "-mbmi2", "-mlzcnt", "-ffp-contract=off", "-fno-math-errno", "-I/root/reference/include", f"-I{integ}", inc,
                  os.path.join(tcpp, "integration_host.cpp"), "-o", exe])
        # ... and the reference's own TAPE (src/autodiff/autodiff.cpp) + its own autodiff test suite on top of that header: every
        # line above the C ABI in this binary is reference code.  autodiff.cpp also instantiates Tape<CUDAArray<float>> under
        # ENOKI_CUDA; that instantiation is dead code here, its references into libenoki-cuda.so stay unresolved.
        exe = os.path.join(tcpp, "reference_tape_hip.bin")
        src = os.path.join(tcpp, "reference_tape_hip.cpp")
        shim_files = [os.path.join(base, f) for base, _, files in os.walk(shim) for f in files]
        if force or _newer(exe, [src, os.path.join(integ, "enoki", "hip.h"), os.path.join(integ, "hip_hooks.cpp"), os.path.join(ref_tests, "autodiff.cpp"),
                                 "/root/reference/src/autodiff/autodiff.cpp", os.path.join(HERE, "libenoki-hip.so")] + shim_files):
            _run(["g++", "-std=c++17", "-O1", "-mavx2", "-mfma", "-mf16c", "-mbmi", "-mbmi2", "-mlzcnt", "-ffp-contract=off", "-fno-math-errno",
                  "-iquote", shim, "-iquote", "/root/reference/include/enoki", "-iquote", "/usr/include/c++/11/pstl", "-I-",
                  "-I/root/reference/include", f"-I{integ}", inc, f"-I{shim}", "-DENOKI_AUTODIFF=1", "-DENOKI_BUILD=1", "-DENOKI_AUTODIFF_BUILD=1",
                  '-DREFERENCE_TAPE_FILE="/root/reference/src/autodiff/autodiff.cpp"', f'-DREFERENCE_TEST_FILE="{os.path.join(ref_tests, "autodiff.cpp")}"',
                  src, os.path.join(integ, "hip_hooks.cpp"), "-o", exe, f"-L{HERE}", "-lenoki-hip", "-Wl,-rpath,$ORIGIN/../../enoki_amd",
                  "-Wl,--unresolved-symbols=ignore-all"])
        # tests/sphere.cpp goes through hipcc: its vectorize() calls become fused kernels (include/enoki/vectorize.h)
        exe = os.path.join(tcpp, "reftest_sphere_hip.bin")
        src = os.path.join(tcpp, "reftest_sphere_hip.cpp")
        shim_files = [os.path.join(base, f) for base, _, files in os.walk(shim) for f in files]
        if force or _newer(exe, [src, os.path.join(ref_tests, "sphere.cpp"), os.path.join(ref_tests, "ray.h"),
                                 os.path.join(HERE, "libenoki-hip.so")] + shim_files + _headers()):
            _run([HIPCC] + DEVICE + ["-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{shim}", inc, f"-I{os.path.join(ROOT, 'compat')}",
                                     f'-DREFERENCE_TEST_FILE="{os.path.join(ref_tests, "sphere.cpp")}"', src, "-o", exe,
                                     f"-L{HERE}", "-lenoki-hip", "-Wl,-rpath,$ORIGIN/../../enoki_amd"])
    if verbose:
        print("[enoki_amd] checkers up to date (oracle/, tests/cpp/)")


def build_examples(force=False, verbose=True):
    """user-level programs that go through hipcc because they use enoki::vectorize() (compile-time fusion)"""
    ex = os.path.join(ROOT, "examples")
    out = []
    for name in ("sphere_fused", "path_trace"):
        src = os.path.join(ex, name + ".cpp")
        lib = os.path.join(ex, f"lib{name}.so")
        if not os.path.exists(src):
            continue
        extra = [os.path.join(ex, "path_trace.h"), os.path.join(HERE, "libenoki-hip-autodiff.so")] if name == "path_trace" else []
        if force or _newer(lib, [src, os.path.join(HERE, "libenoki-hip.so")] + extra + _headers()):
            _run([HIPCC] + DEVICE + ["-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                                     f"-I{os.path.join(ROOT, 'include')}", src, "-o", lib, f"-L{HERE}"] +
                 (["-lenoki-hip-autodiff"] if name == "path_trace" else []) + ["-lenoki-hip", "-Wl,-rpath,$ORIGIN/../enoki_amd"])
            if verbose:
                print(f"[enoki_amd] built {os.path.relpath(lib, ROOT)}")
        out.append(lib)
    return out


def build_all(force=False, verbose=True):
    build_core(force, verbose)
    build_probe(force, verbose)
    build_autodiff(force, verbose)
    build_python(force, verbose)
    build_examples(force, verbose)
    build_checkers(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
