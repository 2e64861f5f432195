"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, and exports exactly the symbols that
include/xfm_hip.h declares; the ctypes binding mirrors the header; the product path refuses to run without the HIP
extension or on CPU tensors (no silent fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "xfm_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xfm_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from xfm_amd import build
    lib_path = build.build()
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/xfm_hip.h but not exported"


def test_binding_covers_the_header_and_abi_version():
    from xfm_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.xfm_abi_version() == _lib.ABI_VERSION
    hdr = open(HEADER).read()
    assert f"#define XFM_ABI_VERSION {_lib.ABI_VERSION}" in hdr


def test_struct_layouts_match_header_field_order():
    from xfm_amd import _lib
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for cname, struct in (("xfm_ln_fwd_args", _lib.LnFwdArgs), ("xfm_ln_bwd_args", _lib.LnBwdArgs),
                          ("xfm_attn_args", _lib.AttnArgs), ("xfm_embed_args", _lib.EmbedArgs), ("xfm_adamw_args", _lib.AdamWArgs)):
        end = hdr.index("} " + cname + ";")
        body = hdr[hdr.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                name = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?\s*$", part.strip())
                fields.append(name[0])
        assert fields == [f[0] for f in struct._fields_], cname


def test_argument_errors_are_reported_not_crashed():
    from xfm_amd import _lib
    lib = _lib.load()
    rc = lib.xfm_gemm_nt(None, 0, None, 0, None, 0, None, None, 0, 1, 1, 64, 0, 0, None)
    assert rc == -1 and b"null operand" in lib.xfm_last_error()
    rc = lib.xfm_layernorm_fwd(None, 768, 0, None)
    assert rc == -1


def test_rccl_entry_points_report_argument_errors():
    """xfm_dp_*: no communicator -> an error code and a message, never a crash (XFM_E_ARG; XFM_E_UNSUPPORTED where librccl.so cannot be
    loaded -- the library itself must load without it: it is resolved at the first xfm_dp_* call, not at link time)."""
    from xfm_amd import _lib
    lib = _lib.load()
    rc = lib.xfm_dp_bucket_allreduce(None, None, 16, 0, 1, None)
    assert rc in (-1, -3) and lib.xfm_last_error()
    rc = lib.xfm_dp_init(None, 0, 1, None)
    assert rc in (-1, -3)
    import subprocess
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH if hasattr(_lib, "LIB_PATH") else os.path.join(ROOT, "xfm_amd", "libxfm_hip.so")],
                            capture_output=True, text=True).stdout
    assert "rccl" not in needed   # no link-time dependency on RCCL


def test_product_path_refuses_cpu_tensors():
    from xfm_amd import _lib, functional as Fx
    a = torch.zeros(4, 64, dtype=torch.bfloat16)
    with pytest.raises(_lib.XfmHipError):
        Fx.gemm_nt(a, a)


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "xfm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("tools/oracle", ""), f"{f} references the oracle"


def test_register_bound_kernels_keep_their_budgets():
    """The 256 x 256 GEMM kernels run two waves per SIMD (<= 256 VGPRs) inside counted-vmcnt pipelines, the short-sequence attention
    backward kernels three (<= 168): a compiler or source change that spills them to scratch (VMEM traffic the wait counts do not
    know about) or drops their occupancy must fail here, not show up as a slow bench.  The shipped ViT forward kernel sits at its
    256-register limit with two spilled registers outside its loops; more than that is a regression."""
    from xfm_amd import build
    use = build.resource_usage()
    seen = 0
    for name, u in use.items():
        if "gemm_nt_256_kernel" in name or "gemm_tn_256_kernel" in name:
            seen += 1
            assert u["vgprs"] + u.get("agprs", 0) <= 256, (name, u)
            assert u["spill"] == 0 and u["scratch"] == 0, (name, u)
        if "attn_bwd_dq_short_kernel" in name or "attn_bwd_dkv_short_kernel" in name:
            seen += 1
            assert u["vgprs"] <= 168 and u["spill"] == 0 and u["scratch"] == 0, (name, u)
        if "gemm_tn_group_kernelILb0E" in name:   # the grouped weight-gradient kernel's default build (the throttled build spills 5 registers)
            seen += 1
            assert u["vgprs"] <= 256 and u["spill"] == 0 and u["scratch"] == 0, (name, u)
        if "attn_fwd_vit_kernelILb1ELi13ELb1E" in name:   # <bias, 13 tiles, tiled bias>: the pre-training step's instantiation
            seen += 1
            assert u["scratch"] <= 16, (name, u)
    assert seen >= 27, sorted(use)[:20]


def test_persistent_gemm_epilogue_issues_the_stores_its_wait_counts_assume():
    """gemm_nt_256_kernel<EPI, PERSIST, D> retires the next tile's first staging units with `s_waitcnt vmcnt(6 + NS)`: the 2 D
    staging loads are OLDER than the NS output stores of an interior tile, so the wait is only sufficient if the compiler really
    issues >= NS store instructions per lane on that path (fewer -- merged or dropped stores -- and a wave would read a ring slot
    whose direct-to-LDS load has not landed; more are harmless, the wait only gets stricter).  Pinned here from the gfx950 ISA:
    the full-width path of every instantiation holds exactly NS `global_store_dwordx4` (16: bf16 / GELU-dgrad; 32: fp32 and
    GELU + gelu'), the ragged-edge path stores element-wise with narrower instructions."""
    import re
    from xfm_amd import build
    isa = build.kernel_isa()
    want = {0: 16, 1: 32, 2: 32, 3: 16}   # EPI_BF16, EPI_F32, EPI_GELU, EPI_DGELU (EPI_F32_ACC waits without the allowance)
    seen = 0
    for name, body in isa.items():
        m = re.match(r"_Z18gemm_nt_256_kernelILi(\d)ELb([01])ELi(\d)EE", name)
        if not m or int(m.group(1)) not in want:
            continue
        seen += 1
        x4 = sum(1 for l in body if l.startswith("global_store_dwordx4"))
        assert x4 == want[int(m.group(1))], (name, x4)
        if m.group(2) == "1":   # the allowance itself: vmcnt(6 + NS) is in the persistent instantiation's code
            assert any(re.search(r"s_waitcnt vmcnt\(%d\)" % (6 + want[int(m.group(1))]), l) for l in body), name
    assert seen == 16, seen

