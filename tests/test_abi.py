"""The C-ABI library loads on a CPU-only host and exports exactly what include/x3d_hip.h declares.
(No compute calls here: there is no GPU in the build container.)"""
import os
import re

from x3d_tf_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "x3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(x3d_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == hip.exported_symbols()


def test_library_exports_every_declared_symbol():
    lib = hip.load()
    for name in _declared():
        assert getattr(lib, name) is not None
    header = int(re.search(r"#define X3D_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "x3d_hip.h")).read()).group(1))
    assert lib.x3d_version() == header == hip.ABI_VERSION
    assert lib.x3d_last_error() is not None


def test_product_library_reads_no_environment():
    """the kernel-selection A/B switches exist only in -DX3D_EXPERIMENTS builds (csrc/common.h x3d_env_int): the product
    library must not even import getenv, so no X3D_* variable can change which kernel a launch takes"""
    import shutil
    import subprocess
    import pytest
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    if not os.path.exists(nm):
        pytest.skip("no nm in this image")
    from x3d_tf_amd import build
    if "-DX3D_EXPERIMENTS" in build.FLAGS:
        pytest.skip("experiments build")
    out = subprocess.run([nm, "-D", "--undefined-only", hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in out


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """a library built from another version of the header must not load (its argument lists may have shifted)"""
    import pytest
    monkeypatch.setattr(hip, "ABI_VERSION", hip.ABI_VERSION + 1)
    with pytest.raises(hip.X3DHipError, match="ABI version"):
        hip.load(hip.LIB_PATH)


def test_struct_layouts_match_header():
    """field order of the ctypes mirrors == field order of the C structs"""
    text = open(os.path.join(ROOT, "include", "x3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    pairs = {"x3d_pw_fwd_args": hip.PwFwdArgs, "x3d_pw_dgrad_args": hip.PwDgradArgs,
             "x3d_pw_wgrad_args": hip.PwWgradArgs, "x3d_pw_bwd_args": hip.PwBwdArgs,
             "x3d_pw_pack_item": hip.PwPackItem, "x3d_bn_eval_item": hip.BnEvalItem, "x3d_eval_views_args": hip.EvalViewsArgs, "x3d_dw3d_fwd_args": hip.Dw3dFwdArgs,
             "x3d_dw3d_bwd_args": hip.Dw3dBwdArgs, "x3d_se_bnb_bwd_args": hip.SeBnbBwdArgs,
             "x3d_bn_fold": hip.BnFold, "x3d_train_clip_args": hip.TrainClipArgs}
    for cname, cls in pairs.items():
        body = re.search(r"typedef struct \{([^{}]*)\} " + cname + ";", text).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            first = names[0].split()[-1].lstrip("*")
            fields.append(first)
            fields += [n.strip().lstrip("*") for n in names[1:]]
        fields = [re.sub(r"\[\d+\]$", "", f) for f in fields]   # float mean[3] -> mean
        assert fields == [f[0] for f in cls._fields_], cname


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(hip.X3DHipError):
        hip.load(str(tmp_path / "libx3d_hip.so"))


def test_dw3d_kernel_name_dry_run():
    """x3d_dw3d_kernel_name runs the depthwise dispatch without launching (no GPU needed): the names are the
    instantiations bench.py's roofline row and the rocprofv3 summaries refer to."""
    f = hip.Dw3dFwdArgs()
    f.x = f.w = f.y = 256
    f.dtype = hip.BF16
    f.N, f.C, f.T, f.H, f.W, f.stride = 64, 54, 16, 112, 112, 2
    assert hip.dw3d_kernel_name(f) == "dw3d_fwd_di_kernel<bf16, 2>"             # stride 2, strips of four: the de-interleaved LDS plane
    b = hip.Dw3dBwdArgs()
    b.dv = b.braw = b.coef_nc = b.araw = b.a_scale_shift = b.w = b.ga = b.a_sums = b.dw = 256
    b.dtype = hip.BF16
    b.N, b.C, b.T, b.H, b.W, b.stride = 64, 108, 16, 56, 56, 1
    assert hip.dw3d_kernel_name(b) == "dw3d_bwd_s1r_kernel<bf16, 8, 1, 6>"      # dw_s1.hip
    b.C, b.H, b.W, b.stride = 54, 112, 112, 2
    assert hip.dw3d_kernel_name(b) == "dw3d_bwd_s2r_kernel<bf16, 8, 2, 6>"      # dw_s2.hip: the headline's largest launches
    b.dtype = hip.F32
    assert hip.dw3d_kernel_name(b).startswith("dw3d_bwd_kernel<float, 2, 2, "), hip.dw3d_kernel_name(b)   # fp32 storage keeps the older kernels
    f.stride = 3
    import pytest
    with pytest.raises(hip.X3DHipError):
        hip.dw3d_kernel_name(f)
    assert hip.load().x3d_crc32c(b"123456789", 9, 0) == 0xE3069283


def test_cpp_client_links_against_the_c_abi(tmp_path):
    """tools/bench_dw3d.cpp is a C++ client of include/x3d_hip.h with no Python / torch in the loop: it must compile
    and link against libx3d_hip.so (run on a GPU box: profiles/r01e_bench_dw3d_cabi.txt)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("hipcc not available")
    out = tmp_path / "bench_dw3d"
    libdir = os.path.join(ROOT, "x3d-tf_amd")
    r = subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "bench_dw3d.cpp"),
                        "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lx3d_hip", "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists()
