"""torch-ROCm interop: the ROCm port of the reference's tests/python/test_pytorch.py.
Arrays are built from device tensors with one device-to-device copy (no host round trip) and exported with
`.torch()` as zero-copy views through __cuda_array_interface__; torch is plumbing here, not the compute path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ek():
    import enoki_amd.hip_autodiff as m
    m.hip_init(0)
    return m


@pytest.fixture(scope="module")
def ekc():
    import enoki_amd.hip as m
    return m


def make_atan2(ek, ekc):
    class EnokiAtan2(torch.autograd.Function):
        """the PyTorch function example of the reference documentation (tests/python/test_pytorch.py:6-31)"""

        @staticmethod
        def forward(ctx, arg1, arg2):
            ctx.in1 = ek.Float32(arg1)
            ctx.in2 = ek.Float32(arg2)
            ek.set_requires_gradient(ctx.in1, arg1.requires_grad)
            ek.set_requires_gradient(ctx.in2, arg2.requires_grad)
            ctx.out = ek.atan2(ctx.in1, ctx.in2)
            out_torch = ek.detach(ctx.out).torch().clone()
            ek.hip_malloc_trim()
            return out_torch.reshape(arg1.shape)

        @staticmethod
        def backward(ctx, grad_out):
            ek.set_gradient(ctx.out, ekc.Float32(grad_out))
            ek.Float32.backward()
            result = (ek.gradient(ctx.in1).torch().clone().reshape(grad_out.shape) if ek.requires_gradient(ctx.in1) else None,
                      ek.gradient(ctx.in2).torch().clone().reshape(grad_out.shape) if ek.requires_gradient(ctx.in2) else None)
            del ctx.out, ctx.in1, ctx.in2
            ek.hip_malloc_trim()
            return result

    return EnokiAtan2.apply


def test01_set_gradient(ek, ekc):
    a = ek.Float32.full(42, 10)
    ek.set_requires_gradient(a)
    with pytest.raises(TypeError):
        ek.set_gradient(a, ek.Float32.full(-1, 10))          # gradients are plain (non-differentiable) arrays
    grad = ekc.Float32.full(-1, 10)
    ek.set_gradient(a, grad)
    assert np.allclose(grad.numpy(), ek.gradient(a).numpy())
    ek.Float32.backward()


def test02_array_to_torch(ek, ekc):
    a = ekc.Float32.full(42, 10)
    t = a.torch()
    assert isinstance(t, torch.Tensor) and t.is_cuda and t.shape == (10,)
    t += 8
    assert np.allclose(t.cpu().numpy(), 50)
    assert np.allclose(a.numpy(), 50)                        # zero-copy: the tensor aliases the array
    x = torch.linspace(0, 1, 1000, device="cuda")
    assert np.array_equal(ekc.Float32(x).numpy(), x.cpu().numpy())              # device -> device construction
    assert np.array_equal(ekc.Float32(x[::2]).numpy(), x[::2].cpu().numpy())    # non-contiguous input
    with pytest.raises(TypeError):
        ekc.Float32(x.double())


def test03_pytorch_function(ek, ekc):
    enoki_atan2 = make_atan2(ek, ekc)
    y = torch.tensor(1.0, device="cuda", requires_grad=True)
    x = torch.tensor(2.0, device="cuda", requires_grad=True)
    o = enoki_atan2(y, x)
    o.backward()
    assert np.allclose(y.grad.cpu(), 0.4) and np.allclose(x.grad.cpu(), -0.2)


def test04_pytorch_function_vector(ek, ekc):
    enoki_atan2 = make_atan2(ek, ekc)
    y = torch.linspace(-2, 2, 1001, device="cuda", requires_grad=True)
    x = torch.linspace(3, 1, 1001, device="cuda", requires_grad=True)
    o = enoki_atan2(y, x)
    o.sum().backward()
    den = (x * x + y * y).detach()
    assert torch.allclose(o, torch.atan2(y, x).detach(), atol=2e-6)
    assert torch.allclose(y.grad, x.detach() / den, rtol=1e-5) and torch.allclose(x.grad, -y.detach() / den, rtol=1e-5, atol=1e-7)


def test05_own_arrays_are_shared_not_copied(ek, ekc):
    """the __cuda_array_interface__ constructor must not intercept our own arrays: DiffArray(plain) shares the buffer"""
    a = ekc.Float32.arange(1000)
    d = ek.Float32(a)
    assert ek.detach(d).data_ptr() == a.data_ptr()
    assert np.array_equal(ekc.Float32(ekc.UInt32.arange(10)).numpy(), np.arange(10, dtype=np.float32))   # converting ctor still reachable


def test06_enoki_namespace():
    """`import enoki as ek` with the reference's spelling (FloatC / FloatD aliases, type-dispatched free functions):
    the documentation example of tests/python/test_pytorch.py runs unchanged up to the module name"""
    import enoki as ek2
    a = ek2.FloatD.full(42, 10)
    ek2.set_requires_gradient(a)
    with pytest.raises(TypeError):
        ek2.set_gradient(a, ek2.FloatD.full(-1, 10))
    ek2.set_gradient(a, ek2.FloatC.full(-1, 10))
    ek2.FloatD.backward()
    y = torch.tensor([1.0, 0.5], device="cuda"); x = torch.tensor([2.0, 3.0], device="cuda")
    yd, xd = ek2.FloatD(y), ek2.FloatD(x)
    ek2.set_requires_gradient(yd); ek2.set_requires_gradient(xd)
    out = ek2.atan2(yd, xd)                                 # differentiable overload
    ek2.backward(ek2.hsum(out))
    den = (x * x + y * y).cpu().numpy()
    assert np.allclose(ek2.gradient(yd).numpy(), x.cpu().numpy() / den, rtol=1e-5)
    assert np.allclose(ek2.atan2(ek2.FloatC(y), ek2.FloatC(x)).numpy(), np.arctan2(y.cpu().numpy(), x.cpu().numpy()), rtol=2e-6)
    import enoki.cuda_autodiff as m
    assert m is ek2.hip_autodiff
    ek2.cuda_malloc_trim()


def test07_readme_python_example():
    import enoki as ek2
    x = ek2.FloatD.linspace(0, 1, 1 << 12); ek2.set_requires_gradient(x)
    y = ek2.hsum(ek2.atan2(x, ek2.FloatD(2.0)) * ek2.exp(x))
    ek2.backward(y)
    g = ek2.gradient(x)
    xv = np.linspace(0, 1, 1 << 12)
    want = np.exp(xv) * (np.arctan2(xv, 2.0) + 2.0 / (xv * xv + 4.0))
    assert np.allclose(g.numpy(), want, rtol=1e-5) and np.allclose(g.torch().cpu().numpy(), g.numpy())
    rng = ek2.PCG32C(ek2.UInt64C(42), ek2.UInt64C.arange(1 << 12)); u = rng.next_float32().numpy()
    assert u.min() >= 0 and u.max() < 1
