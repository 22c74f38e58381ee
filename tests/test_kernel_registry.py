"""Registry behaviour (mirrors reference tests/test_kernel_registry.py:14-138)."""

import pytest

from src.kernels.registry import KernelRegistry


@pytest.fixture()
def clean_registry():
    saved = {k: list(v) for k, v in KernelRegistry._kernels.items()}
    KernelRegistry._kernels.clear()
    yield KernelRegistry
    KernelRegistry._kernels.clear()
    KernelRegistry._kernels.update(saved)


def _fn(name):
    def f(*a, **k):
        return name

    f.__name__ = name
    return f


def test_priority_ordering(clean_registry):
    lo, hi = _fn("low_ref"), _fn("high_hip")
    clean_registry.register("op", lo, priority=10, device="auto")
    clean_registry.register("op", hi, priority=100, device="cuda")
    assert clean_registry.get_best("op", "cuda") is hi
    assert clean_registry.get_best("op", "cpu") is lo


def test_device_filter_and_auto(clean_registry):
    cuda_only = _fn("cuda_only")
    clean_registry.register("op", cuda_only, priority=50, device="cuda")
    assert clean_registry.get_best("op", "cpu") is None
    assert clean_registry.get_best("op", "mps") is None
    assert clean_registry.get_best("op", "cuda") is cuda_only
    anyd = _fn("any_device")
    clean_registry.register("op", anyd, priority=5, device="auto")
    assert clean_registry.get_best("op", "mps") is anyd


def test_unknown_op(clean_registry):
    assert clean_registry.get_best("nope", "cuda") is None
    assert clean_registry.list_available("nope", "cuda") == []


def test_list_available_and_status(clean_registry):
    a, b = _fn("a_impl"), _fn("b_impl")
    clean_registry.register("x", a, priority=1, device="auto")
    clean_registry.register("x", b, priority=2, device="cuda")
    listed = clean_registry.list_available("x", "cuda")
    assert [e["name"] for e in listed] == ["b_impl", "a_impl"]
    assert listed[0] == {"name": "b_impl", "priority": 2, "device": "cuda"}
    assert clean_registry.get_status("cuda") == {"x": "b_impl"}
    assert clean_registry.get_status("cpu") == {"x": "a_impl"}


def test_equal_priority_keeps_registration_order(clean_registry):
    first, second = _fn("first"), _fn("second")
    clean_registry.register("y", first, priority=7, device="auto")
    clean_registry.register("y", second, priority=7, device="auto")
    assert clean_registry.get_best("y", "cpu") is first


def test_module_surface_and_aliases():
    import kernels
    import src.kernels as sk

    assert kernels is sk
    # `kernels.registry` (attribute) is the singleton, as in the reference; the
    # submodule is reached through from-imports under both spellings
    from kernels.registry import KernelRegistry as KR1
    from src.kernels.registry import KernelRegistry as KR2

    assert KR1 is KR2 is KernelRegistry and isinstance(sk.registry, KR1)
    info = sk.get_kernel_info()
    for key in ("verify_backend", "kv_append_backend", "verify_available", "kv_append_available", "device"):
        assert key in info
    # the reference allows cuda/triton/torch/fallback/unknown; this build adds "hip"
    assert info["verify_backend"] == "hip" and info["kv_append_backend"] == "hip"
    assert sk.get_verify_prefix("cuda") is sk.verify_prefix
    assert sk.get_kv_append("cuda") is sk.kv_append
    assert sk.get_verify_prefix("cpu") is None, "no CPU fallback may be registered"


def test_ops_refuse_cpu_tensors():
    import torch

    import src.kernels as sk

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sk.verify_prefix(torch.zeros(1, 1, 4), torch.zeros(1, 1, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sk.kv_append(*(torch.zeros(1, 1, 1, 4) for _ in range(4)))
    # shape assertions come first, as in the reference (tests/test_kv_cache.py:86-113)
    with pytest.raises(AssertionError):
        sk.kv_append(torch.zeros(2, 2, 3, 4), torch.zeros(2, 2, 3, 4), torch.zeros(1, 2, 2, 4), torch.zeros(1, 2, 2, 4))
