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
        for shp in S.PW_FWD_TAIL:
            add(S.pw_fwd_struct(shp, dt, False), f"test_pw_fwd_tail[{shp}, {dt}]")
            if half:
                add(S.pw_fwd_struct(shp, dt, True), f"test_pw_fwd_tail[{shp}, {dt}, panel]")
        for shp in S.PW_FWD_INFER:
            add(S.pw_fwd_infer_struct(shp, dt, False), f"test_pw_fwd_infer[{shp}, {dt}]")
            if half:
                add(S.pw_fwd_infer_struct(shp, dt, True), f"test_pw_fwd_infer[{shp}, {dt}, panel]")
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
            for shp in S.PW_BWD_TAIL:
                st = S.pw_bwd_struct(shp, dt)
                assert hip.load().x3d_pw_bwd_supported(st), f"fused backward with the tail does not cover the registered case {shp}"
                add(st, f"test_pw_bwd_tail[{shp}, {dt}]")
            for shp in S.PW_BWD_RC:
                st = S.pw_bwd_rc_struct(shp, dt)
                assert hip.load().x3d_pw_bwd_supported(st), f"the recomputed-output fused backward does not cover the registered case {shp}"
                add(st, f"test_pw_bwd_rc[{shp}, {dt}]")
            for shp in S.PW_BWD_RC_STRIDED:
                st = S.pw_bwd_rc_strided_struct(shp, dt)
                assert hip.load().x3d_pw_bwd_supported(st), f"the strided recomputed-output backward does not cover the registered case {shp}"
                add(st, f"test_pw_bwd_rc_strided[{shp}, {dt}]")
                # ... which also runs the DENSE store form on the even-pixel copy (the plans' path since round 6)
                n_, ci_, co_, t_, xh_, xw_ = shp
                st = S.pw_bwd_rc_struct((n_, ci_, co_, t_, (xh_ + 1) // 2, (xw_ + 1) // 2, "store", 0), dt)
                if hip.load().x3d_pw_bwd_supported(st):
                    add(st, f"test_pw_bwd_rc_strided[{shp}, {dt}, compact]")
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


@pytest.mark.parametrize("index", [1, 5])
def test_inference_plan_is_inference_shaped(index):
    """BASELINE configs 1 and 5 (eval.py:83-89, model.py:113-127 at training=False): the recorded forward list has no
    residual-tail pass, no per-layer BatchNorm launch (one batched coefficient launch at the head) and per block exactly
    a -> b -> [SE] -> [shortcut conv] -> c, the last with the folded BN + Add + ReLU epilogue."""
    import x3d_tf_amd as x
    from x3d_tf_amd import dispatch as D
    from x3d_tf_amd.model import X3D
    variant, n, t, s, dtype, training, over = D.BASELINE_CONFIGS[index]
    flat = []
    for k, v in over.items():
        flat += [k, v]
    m = X3D(x.get_config(variant, flat or None), dtype=dtype, device="dry")
    pl = m._plan(n, t, s, s, training)
    names = [item[0] for item in pl.fwd]
    assert names[0] == "x3d_bn_eval_coef_batched" and names.count("x3d_bn_eval_coef_batched") == 1
    assert not [k for k in names if k.startswith(("x3d_tail", "x3d_bn_finalize"))]
    # the stem: one launch (conv_s -> conv_t -> BN + ReLU, x3d_stem_fwd) with 16-bit storage, the two-kernel path in fp32
    assert pl.stem_fused == (dtype != torch.float32)
    stem = ["x3d_stem_fwd"] if pl.stem_fused else ["x3d_stem_s_fwd", "x3d_dwt_fwd"]
    k0 = 1 + len(stem)
    assert names[1:k0] == stem
    blocks = m.arch.blocks
    odd_rows = lambda b: (pl.blocks[blocks.index(b)].wo % 2) == 1      # (only where the gather takes one output per load)
    want = []
    for b in blocks:
        # (a strided shortcut conv reads the even-pixel copy of the block input: one small copy launch in front of it)
        want += ["x3d_pw_fwd", "x3d_dw3d_fwd"] + (["x3d_se_fwd"] if b.has_se else []) + \
                ((["x3d_subsample2"] if b.stride == 2 and odd_rows(b) else []) + ["x3d_pw_fwd"] if b.has_shortcut_conv else []) + ["x3d_pw_fwd"]
    assert names[k0:k0 + len(want)] == want
    assert names[k0 + len(want):] == ["x3d_pw_fwd", "x3d_pool_fwd", "x3d_dense_fwd", "x3d_dense_fwd", "x3d_softmax_xent", "x3d_view_mean"]
    assert len(want) <= 6 * len(blocks) and sum(3 + b.has_se + b.has_shortcut_conv * (1 + (b.stride == 2 and odd_rows(b))) for b in blocks) == len(want)
    for B in pl.blocks:                      # the `c` conv carries the epilogue; `a` and the shortcut stay raw
        assert B.sc.out_scale_shift and B.sc.out_add and B.sc.out_act == 1 and not B.sc.stats
        assert bool(B.sc.out_add_scale_shift) == bool(B.spec.has_shortcut_conv)
        assert not B.sa.out_scale_shift
    # activation buffers are shared between blocks: two block outputs ping-pong
    ys = {B.y.data_ptr() for B in pl.blocks}
    assert len(ys) == 2 and len({B.a_raw.data_ptr() for B in pl.blocks}) == 1
    for prev, cur in zip(pl.blocks, pl.blocks[1:]):
        assert cur.x.data_ptr() == prev.y.data_ptr() and cur.y.data_ptr() != cur.x.data_ptr()
