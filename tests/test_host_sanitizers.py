"""Host logic under AddressSanitizer + LeakSanitizer + UBSan, no GPU needed.

tests/cpp/asan_deferred.cpp compiles include/enoki/hip.h against a host stand-in of the C ABI and drives the deferred
gathers / deferred unary maps / sincos pairs of HIPArray through directed scenarios and 40 fuzzed programs that are
executed with and without deferred evaluation (same bits expected, no block left allocated).

tests/cpp/asan_tape.cpp does the same one level up: Tape<HIPArray<float>> (enoki_amd/src/autodiff_impl.h) over that stand-in,
60 fuzzed differentiable programs, values and gradients with and without deferral."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_deferred_nodes_under_sanitizers():
    exe = os.path.join(ROOT, "tests", "cpp", "asan_deferred.bin")
    if not os.path.exists(exe):
        from enoki_amd import _build
        _build.build_checkers(verbose=False)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    assert "agree with eager evaluation" in out.stdout


def test_tape_over_deferred_nodes_under_sanitizers():
    exe = os.path.join(ROOT, "tests", "cpp", "asan_tape.bin")
    if not os.path.exists(exe):
        from enoki_amd import _build
        _build.build_checkers(verbose=False)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("ENOKI_HIP_DEFER_MIN", None)
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    assert "identical values and gradients" in out.stdout



def test_reference_side_binding_under_sanitizers():
    """integration/enoki/hip.h + integration/hip_hooks.cpp against the REFERENCE's headers and the host stand-in
    (tests/cpp/integration_host.cpp): the safe_mul / safe_fmadd fragments are fused exactly when the select's operand is the
    tagged product, tags expire on writes, widened index arrays find their 32-bit origin.  Built where the reference tree
    exists (this container); the binary is what gets checked elsewhere."""
    import pytest
    exe = os.path.join(ROOT, "tests", "cpp", "integration_host.bin")
    if not os.path.exists(exe):
        pytest.skip("needs the reference's headers to build (enoki_amd/_build.py builds it where /root/reference exists)")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    assert "fused exactly when tagged" in out.stdout
