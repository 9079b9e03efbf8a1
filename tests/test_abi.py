"""CPU-side checks of the drop-in boundary: the shared library builds/loads and exports exactly the symbols
include/conformer_hip.h declares (no compute calls: there is no GPU here), the ctypes table matches the
header, and the product refuses to run without its HIP path."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "conformer_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = re.findall(r"\b(?:int|int64_t|size_t|const char\*)\s+(cfm_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S)
    out = {}
    for name, args in decls:
        args = args.strip()
        out[name] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


@pytest.fixture(scope="module")
def lib():
    from conformer_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_library(verbose=False)
    return _lib.load()


def test_header_declares_the_path():
    fns = header_functions()
    for need in ("cfm_layernorm_fwd_f32", "cfm_gemm_bias_f32", "cfm_gemm_bias_swish_f32", "cfm_gemm_bias_glu_f32",
                 "cfm_gemm_bias_residual_f32", "cfm_relpos_table_f32", "cfm_relpos_attention_fwd_f32",
                 "cfm_dwconv_bn_swish_fwd_f32", "cfm_subsample_conv1_relu_f32", "cfm_subsample_conv2_relu_f32"):
        assert need in fns


def test_library_exports_every_declared_symbol(lib):
    from conformer_amd import _lib
    fns = header_functions()
    assert set(fns) == set(_lib.SIGNATURES), set(fns) ^ set(_lib.SIGNATURES)
    for name, nargs in fns.items():
        assert hasattr(lib, name), f"{name} missing from libconformer_hip.so"
        assert len(_lib.SIGNATURES[name][1]) == nargs, f"ctypes arity of {name} differs from the header"


def test_host_side_entry_points(lib):
    assert lib.cfm_version() == 1
    assert lib.cfm_strerror(0) == b"ok" and lib.cfm_strerror(-2) == b"unsupported configuration"
    for n in (7, 8, 31, 200, 1000, 20000):
        assert lib.cfm_subsampled_length(n) == ((n - 1) // 2 - 1) // 2


def test_argument_validation_without_gpu(lib):
    """Shape/NULL checks run before any HIP call, so they are testable on CPU."""
    assert lib.cfm_layernorm_fwd_f32(None, None, None, None, None, None, 4, 32, 1e-5, None) == -3
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p += (-p) % 16
    assert lib.cfm_layernorm_fwd_f32(p, p, p, p, None, None, 1, 30, 1e-5, None) == -1       # d % 4
    assert lib.cfm_gemm_bias_f32(p, p, p, p, 4, 4, 6, 8, 4, None) == -1                     # K % 4
    assert lib.cfm_relpos_attention_fwd_f32(p, p, p, 384, p, 128, p, p, None, p, 128, None, 1, 4, 1, 128, None) == -2


def test_no_cpu_fallback():
    from conformer_amd import ops
    from conformer_amd._lib import ConformerHipError
    with pytest.raises(ConformerHipError):
        ops.layernorm(torch.zeros(2, 32), torch.ones(32), torch.zeros(32))
    from model.modules.encoder import Encoder
    enc = Encoder(80, 1, 32, 4, 7).eval()
    with torch.no_grad(), pytest.raises(ConformerHipError):
        enc(torch.zeros(1, 80, 40), None)


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under conformer_amd/ or model/ may reference it."""
    for base in ("conformer_amd", "model"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b|oracle/|conformer_oracle", txt, flags=re.M), \
                        os.path.join(dp, f)


def test_state_dict_contract():
    """Keys/shapes of SURVEY Appendix A: a reference checkpoint loads strictly."""
    from model.conformer import Conformer
    from oracle import conformer_oracle as O
    m = Conformer(370, 80, 4, 144, 4, 31, 320, 1, 0.0)
    P = O.make_params(370, 80, 4, 144, 4, 31, 320, seed=1)
    assert set(m.state_dict().keys()) == set(P.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(P[k].shape), k
    assert not m.encoder.rel_pe.div_term.requires_grad
    assert "encoder.layers.0.conv.deepwise_conv.weight" in P
