"""Every kernel instantiation the BASELINE configurations launch at full size has an oracle-parity case under `-m gpu`.

Runs on the CPU: the library's dispatch is asked in dry-run mode (x3d_pw_kernel_name / x3d_dw3d_kernel_name, nothing is
launched) which instantiation each launch of a full-size DRY plan runs (x3d_tf_amd/dispatch.py), and which instantiation
each registered parity case (tests/shapes.py: the lists tests/test_kernels_gpu.py and tests/test_model_gpu.py
parametrise over) runs.  The first set must be contained in the second.
"""
import pytest
import torch

from tests import shapes as S


def _kernel_case_names():
    """{kernel instantiation: first parity case that runs it} over the kernel-level cases of tests/test_kernels_gpu.py."""
    from x3d_tf_amd import hip
    got = {}

    def add(st, what):
        got.setdefault(hip.kernel_name(st), what)

    for dt in S.DTYPES:
        half = dt != S.F32
        for shp in S.PW_FWD + S.PW_FWD_XL:
            add(S.pw_fwd_struct(shp, dt, False), f"test_pw_fwd[{shp}, {dt}]")
            if half:
                add(S.pw_fwd_struct(shp, dt, True), f"test_pw_fwd[{shp}, {dt}, panel]")
        for shp in S.PW_DGRAD:
            for epi in S.PW_DGRAD_EPI:
                add(S.pw_dgrad_struct(shp, epi, dt, False), f"test_pw_dgrad[{shp}, {epi}, {dt}]")
                if half:
                    add(S.pw_dgrad_struct(shp, epi, dt, True), f"test_pw_dgrad[{shp}, {epi}, {dt}, panel]")
        for shp in S.PW_WGRAD:
            add(S.pw_wgrad_struct(shp, dt), f"test_pw_wgrad[{shp}, {dt}]")
        if half:
            for shp in S.PW_BWD:
                st = S.pw_bwd_struct(shp, dt)
                assert hip.load().x3d_pw_bwd_supported(st), f"fused backward does not cover the registered case {shp}"
                add(st, f"test_pw_bwd_oracle[{shp}, {dt}]")
        for shp in S.DW:
            add(S.dw_fwd_struct(shp, dt), f"test_dw3d_fwd[{shp}, {dt}]")
            add(S.dw_bwd_struct(shp, dt), f"test_dw3d_bwd[{shp}, {dt}]")
    return got


@pytest.fixture(scope="module")
def parity_kernels():
    # kernel-level cases only: the whole-model parity tests (tests/test_model_gpu.py) run the same instantiations again
    # inside full plans, but coverage is not allowed to lean on them
    return _kernel_case_names()


@pytest.mark.parametrize("index", [1, 2, 3, 4, 5])
def test_every_instantiation_of_a_baseline_config_has_a_parity_case(parity_kernels, index):
    from x3d_tf_amd import dispatch as D
    rows = D.baseline_kernels(index)
    assert rows, "the dry plan recorded no conv launches"
    need = D.kernel_set(rows)
    missing = {k: v for k, v in need.items() if k not in parity_kernels}
    assert not missing, (f"BASELINE config {index} {D.BASELINE_CONFIGS[index][:5]} dispatches kernel instantiations that no "
                         f"-m gpu oracle-parity case runs:\n" + "\n".join(f"  {k}   <- {v}" for k, v in sorted(missing.items())))


def test_bench_workload_is_config_3(parity_kernels):
    """bench.py's default workload (X3D-M, 64 clips of 16x224x224, bf16) is BASELINE config 3: the depthwise kernels of its
    216-channel 28 -> 14 layer (the pair VERDICT r01 found untested) are among the parity-tested instantiations by name."""
    from x3d_tf_amd import dispatch as D
    rows = [r for r in D.baseline_kernels(3) if r[0].startswith("x3d_dw3d") and r[2] == "N64 C216 T16 28x28 s2"]
    assert {r[0] for r in rows} == {"x3d_dw3d_fwd", "x3d_dw3d_bwd"}
    for _, kernel, _ in rows:
        assert kernel in parity_kernels, kernel


def test_dry_model_cannot_run():
    """A dry model records launches and nothing else: there is no CPU execution path."""
    import x3d_tf_amd as x
    from x3d_tf_amd import hip
    from x3d_tf_amd.model import X3D
    m = X3D(x.get_config("XS"), dtype=torch.float32, device="dry")
    with pytest.raises(hip.X3DHipError):
        m(torch.zeros(10, 4, 32, 32, 3), training=False)
    pl = m._plan(2, 4, 32, 32, True)
    with pytest.raises(hip.X3DHipError):
        pl.run(pl.fwd)
