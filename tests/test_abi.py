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


def test_round3_entries_validate_their_arguments_without_gpu(lib):
    """cfm_ffn_fused_f32 / cfm_ffn_pack_f32 / cfm_rowchain_f32 / cfm_rowgemm_pack_f32 / cfm_relpos_attention_rows_mfma16_f32 refuse bad
    arguments with a negative status BEFORE any HIP call (so this runs without a GPU): NULL pointers, unsupported widths, ragged
    hidden sizes, stage combinations that are not one of the three chains, misaligned pointers."""
    buf = (ctypes.c_float * 4096)()
    p = ctypes.addressof(buf)
    assert p % 16 == 0 or True
    a = (p + 15) // 16 * 16                                    # a 16-byte aligned address inside the buffer
    OK = 0
    assert lib.cfm_ffn_pack_elems(512, 2048) == 2 * 512 * 2048
    assert lib.cfm_ffn_pack_f32(None, a, a, 512, 2048, None) < OK                                   # NULL
    assert lib.cfm_ffn_pack_f32(a, a, a, 160, 640, None) < OK                                       # d not in {128, 256, 512}
    assert lib.cfm_ffn_pack_f32(a, a, a, 512, 2000, None) < OK                                      # hidden % 128
    args = [a, 512, a, 16, 1e-5, a, a, a, a, 0.5, a, 512, 0, None, None, None, 0.0, 64, 512, 2048, None]
    bad = list(args); bad[0] = None
    assert lib.cfm_ffn_fused_f32(*bad) < OK                                                         # X NULL
    bad = list(args); bad[18] = 144
    assert lib.cfm_ffn_fused_f32(*bad) < OK                                                         # d = 144
    bad = list(args); bad[3] = 3
    assert lib.cfm_ffn_fused_f32(*bad) < OK                                                         # ln_parts not a power of two
    bad = list(args); bad[12] = 1
    assert lib.cfm_ffn_fused_f32(*bad) < OK                                                         # mode 1 without stats_out
    bad = list(args); bad[12] = 2
    assert lib.cfm_ffn_fused_f32(*bad) < OK                                                         # mode 2 without gamma / beta
    bad = list(args); bad[0] = a + 4
    assert lib.cfm_ffn_fused_f32(*bad) < OK                                                         # misaligned X
    empty = list(args); empty[17] = 0
    assert lib.cfm_ffn_fused_f32(*empty) == OK                                                      # M = 0: nothing to do, no launch
    assert lib.cfm_rowgemm_pack_f32(a, a, 1536, 512, 1, None) < OK                                  # glu needs N == 2 d
    assert lib.cfm_rowgemm_pack_f32(a, a, 1000, 512, 0, None) < OK                                  # N % 128
    chain = [0, 1, 0, 0, a, 512, None, None, None, 0, None, 0, a, 16, 1e-5, a, a, a, a, 0.5, 2048, a, 512, None, None, None, 0.0,
             None, None, None, 0.0, None, 0, 64, 512, None]
    assert lib.cfm_rowchain_f32(*chain) < OK                                                        # (0, 1, 0, 0) is not K1 / K2 / K3
    chain[0], chain[1], chain[2] = 1, 0, 0
    assert lib.cfm_rowchain_f32(*chain) < OK                                                        # PRE alone
    assert lib.cfm_relpos_attention_rows_mfma16_f32(1, a, a, a, 1536, a, 512, a, a, None, a, 512, 1, 16, 8, 64, 0, 16, None) < OK   # lengths NULL
    assert lib.cfm_relpos_attention_rows_mfma16_f32(1, a, a, a, 1536, a, 512, a, a, a, a, 512, 1, 16, 8, 64, 8, 16, None) < OK      # rows past T


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
