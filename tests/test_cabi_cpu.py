"""CPU-side checks of the C-ABI library (no compute calls): it loads, exports every symbol that
include/qgemul.h declares, classifies the golden descriptors, and refuses to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import golden_io as G
from qublas_amd import capi
from qublas_amd.desc import CLASS_LINEAR, CLASS_TREE, Qu, desc_from_dict, lower, qgemul_desc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "qgemul.h")).read()
    return sorted(set(re.findall(r"\b(qgemul_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    names = declared_symbols()
    assert set(names) == set(capi.EXPORTS), set(names) ^ set(capi.EXPORTS)
    for n in names:
        assert hasattr(L, n), n
    assert L.qgemul_abi_version() == 1
    assert L.qgemul_strerror(0) == b"ok"


def test_product_library_has_no_diagnostic_switches():
    """The product library reads NO environment variable and contains no ablation kernel variant: the A/B switches of
    tools/ (QG_ABLATE selects kernels that are wrong by construction, QG_NO_PLANE_MASK drops a launch the result depends on)
    exist only in libqugemm_diag.so (-DQG_DIAG).  Checked on the built file: neither the switch names nor `getenv` nor a
    symbol of an ablation instantiation are in it."""
    import os
    import subprocess
    path = capi.LIB_PATH
    assert path.endswith("libqugemm.so")
    blob = open(path, "rb").read()
    for name in (b"QG_ABLATE", b"QG_NO_PLANE_MASK", b"QG_LIMB32", b"QG_NO_DEEP", b"QG_PP_", b"QG_NO_KARA", b"QG_GEMV_CH", b"QG_NO_FAST_PACK"):
        assert name not in blob, name
    nm = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True).stdout
    assert "getenv" not in nm
    assert b"qgemul_diag_set_stamps" not in blob
    diag = os.path.join(os.path.dirname(path), "libqugemm_diag.so")
    if os.path.exists(diag):                      # built by tools users; when present it must be the one that has them
        assert b"QG_ABLATE" in open(diag, "rb").read()


def test_struct_layout_matches_header():
    # sizeof(qgemul_desc): 4+4 + 3*8 + 6*8 + 8*8 + 4+4 + 2*2*40*8
    assert C.sizeof(qgemul_desc) == 8 + 24 + 48 + 64 + 8 + 1280


@pytest.mark.parametrize("j", G.gemm_cases("real") + G.gemm_cases("cplx"), ids=lambda j: j["name"])
def test_classify_golden_descriptors(j):
    d = desc_from_dict(j)
    info = capi.classify(d)
    assert info.supported == 1
    name = j["name"]
    if "_L_" in name or name.endswith("classL"):
        if not j["is_complex"] or "_basic_L_" in name:
            assert info.cls == CLASS_LINEAR, (name, info.reason)
        else:
            # TFComplexMul can never be proven linear: (b - a) is always formed in the default-merged
            # format (QuBLAS.h:3515), which cannot hold b - a over the operands' full range
            assert info.cls == CLASS_TREE, name
    if "classT" in name or "default" in name:
        assert info.cls == CLASS_TREE, name
    assert info.max_bits <= 62


def test_classify_limbs_and_kernel():
    by = {j["name"]: j for j in G.gemm_cases("real")}
    i = capi.classify(desc_from_dict(by["e43_L_16x16x512_full_wideC"]))
    assert (i.cls, i.limbs[0], i.limbs[1], i.kernel) == (CLASS_LINEAR, 1, 1, 1)
    i = capi.classify(desc_from_dict(by["e88z_L_4x4x4096_full"]))
    assert (i.cls, i.limbs[0], i.limbs[1], i.kernel) == (CLASS_LINEAR, 3, 3, 2)
    i = capi.classify(desc_from_dict(by["u44_L_8x8x64_full"]), capi.OPT_BALANCED_LIMBS)
    assert (i.cls, i.limbs[0], i.limbs[1], i.kernel) == (CLASS_LINEAR, 2, 2, 2)       # unsigned 8 bits: two plain balanced limbs ...
    i = capi.classify(desc_from_dict(by["u44_L_8x8x64_full"]))
    assert (i.cls, i.limbs[0], i.limbs[1], i.kernel) == (CLASS_LINEAR, 1, 1, 1)       # ... one when the operand is stored centred (x - 128)
    i = capi.classify(desc_from_dict(by["e88z_L_4x4x4096_full"]), capi.OPT_FORCE_TREE)
    assert i.kernel == 4
    # README example / configuration 1: per-product and per-node quantisation -> tree class
    i = capi.classify(desc_from_dict(by["c1_nn_classT"]))
    assert i.cls == CLASS_TREE


def test_reference_artefacts_flag_is_opt_in():
    """C of an unsigned WRP::TCPL format with exactly 32 value bits: refused where a value can leave the format (the reference's own
    result there is an artefact of ArbiInt<32>::allOnes() = -1), reproduced with QG_DESC_REFERENCE_ARTEFACTS (host-word containers,
    like WRP::TCPL_SAT); as a LEVEL type it stays refused either way; a C that cannot leave the format runs without the flag."""
    from qublas_amd.desc import Tags, WRP, TRN
    u32 = Qu(32, 0, False, TRN.TCPL, WRP.TCPL)
    e = Qu(12, 4)
    kw = dict(mul_args=Tags(25, 8), add_args=[Qu(40, 8)])
    st, info = capi.classify_status(lower(e, e, u32, 64, 64, 64, **kw))
    assert st == capi.QG_EUNSUPPORTED and b"allOnes" in info.reason
    st, info = capi.classify_status(lower(e, e, u32, 64, 64, 64, reference_artefacts=True, **kw))
    assert st == capi.QG_OK and info.host_elem_bytes[2] == 8, info.reason
    st, info = capi.classify_status(lower(e, e, Qu(40, 8), 64, 64, 64, mul_args=Tags(25, 8), add_args=[u32], reference_artefacts=True))
    assert st == capi.QG_EUNSUPPORTED
    ue = Qu(4, 3, False)
    st, info = capi.classify_status(lower(ue, ue, u32, 64, 64, 64, mul_args=Tags(9, 6, False), add_args=[Qu(20, 6, False)]))
    assert st == capi.QG_OK, info.reason        # (non-negative sums below 2^32: nothing to wrap)


def test_rejections():
    e = Qu(4, 3)
    # WRP::TCPL_SAT is a stub in the reference (QuBLAS.h:2336-2344: the value goes into the storage word unclamped).  As C's OfMode it
    # runs (general kernels, host-word containers: tests/golden/ref_gemm_real_8); where the value provably stays in its format it is
    # a no-op; as the mode of a level whose sums can leave the format it is refused
    st, info = capi.classify_status(lower(e, e, Qu(4, 3, OfMode=4), 4, 4, 4))            # default tags: the root is already inside (4,3)
    assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == "tree_i32", info.reason
    st, info = capi.classify_status(lower(e, e, Qu(4, 3, OfMode=4), 4, 4, 4, add_args=[Qu(12, 3)]))   # a wide root into C's 32-bit word
    assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == "tree_i64" and info.packed_bytes[2] == 4 * 4 * 4, info.reason
    from qublas_amd.desc import Tags
    st, info = capi.classify_status(lower(e, e, Qu(4, 3, OfMode=4), 300, 300, 64, mul_args=Tags(9, 6), add_args=[Qu(19, 6)]))
    assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == "mfma_i8" and b"combine" in info.reason, info.reason
    st, info = capi.classify_status(lower(e, e, e, 64, 64, 64, mul_args=Qu(9, 6, OfMode=4), add_args=[Qu(19, 6, OfMode=4)]))
    assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == "mfma_i8", info.reason
    st, info = capi.classify_status(lower(e, e, e, 64, 64, 64, add_args=[Qu(5, 3, OfMode=4)]))
    assert st == capi.QG_EUNSUPPORTED and b"TCPL_SAT" in info.reason
    # a 32-bit rounding shift with an RND mode is a reference width artefact (tests/test_oracle_golden.py)
    d = lower(Qu(30, 30), Qu(1, 0), Qu(6, -2, QuMode=4), 4, 4, 4, mul_args=Qu(31, 30))
    st, info = capi.classify_status(d)
    assert st == capi.QG_EUNSUPPORTED, info.reason
    # converting into an unsigned WRP::TCPL format of exactly 32 value bits: another ArbiInt<32>::allOnes artefact
    # (tests/test_oracle_golden.py::test_wrap_into_32_bits_is_a_reference_artefact_only_for_unsigned) ...
    for c in (Qu(32, 0, False, OfMode=3), Qu(16, 16, False, OfMode=3)):
        st, info = capi.classify_status(lower(e, e, c, 4, 4, 4))
        assert st == capi.QG_EUNSUPPORTED, info.reason
    # ... unless no value can leave the range: the reference's `val & allOnes` is then a no-op as well (unsigned operands, sums
    # below 2^32: uint32-style formats run; tests/golden/ref_scalar_6 holds in-range conversions into these types)
    u = Qu(4, 3, False)
    for c in (Qu(32, 0, False, OfMode=3), Qu(16, 16, False, OfMode=3)):
        st, info = capi.classify_status(lower(u, u, c, 4, 4, 4))
        assert st == capi.QG_OK, info.reason
    # ... and only that: 31 / 33 unsigned bits and the signed 32-storage-bit format wrap arithmetically in the reference too
    for c in (Qu(31, 0, False, OfMode=3), Qu(33, 0, False, OfMode=3), Qu(31, 0, True, OfMode=3), Qu(32, 0, False, OfMode=0)):
        st, info = capi.classify_status(lower(e, e, c, 4, 4, 4))
        assert st == capi.QG_OK, info.reason
    # wider than 120 bits
    d = lower(Qu(30, 30), Qu(30, 30), Qu(8, 8), 4, 4, 4, mul_args=Qu(60, 60))
    st, _ = capi.classify_status(d)
    assert st == capi.QG_EUNSUPPORTED
    # The UNROUNDED product of two operands may use all of an int64 on the 64-bit kernels: signed 32-bit words (Q15.16) run there;
    # two unsigned 32-bit words (a 64-bit magnitude) or a 33-bit pair need the 128-bit kernel (round 2 refused them)
    q = Qu(15, 16)
    st, info = capi.classify_status(lower(q, q, q, 4, 4, 64), capi.OPT_RUNTIME_MODES)
    assert st == capi.QG_OK and info.max_bits == 64 and capi.KERNEL_NAMES[info.kernel] == "tree_i64", info.reason
    st, info = capi.classify_status(lower(q, q, q, 4, 4, 64))     # (default: the 32-bit-word form on the 32-bit tree kernel's frame)
    assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == "tree_i32" and info.reason.endswith(b"saturating word adds"), info.reason
    for a, b in ((Qu(16, 16, False), Qu(16, 16, False)), (Qu(16, 16), Qu(15, 16))):
        st, info = capi.classify_status(lower(a, b, q, 4, 4, 64))
        assert st == capi.QG_OK and capi.KERNEL_NAMES[info.kernel] == "tree_i128", info.reason
    # operand ELEMENTS stay one-word values
    assert capi.classify_status(lower(Qu(40, 30), q, q, 4, 4, 4))[0] == capi.QG_EUNSUPPORTED
    d = lower(e, e, e, 4, 4, 4)
    d.n_levels = 5
    st, _ = capi.classify_status(d)
    assert st == capi.QG_EINVAL


def test_no_cpu_fallback():
    """Without a gfx950 device the engine must fail loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    e = Qu(4, 3)
    d = lower(e, e, e, 4, 4, 4)
    A = np.zeros(16, np.int32)
    out = np.full(16, 77, np.int32)
    with pytest.raises(capi.QgemulError) as ei:
        capi.run(d, out, A, A)
    assert ei.value.status == capi.QG_ENOGPU
    assert (out == 77).all()


def test_int32_kernels_need_32bit_formats():
    """Values may be small while a declared format is wide: the 32-bit tree kernels keep every format's
    bounds in 32-bit registers, so the planner must fall back to the 64-bit kernels (found on the GPU:
    golden case c5_tf_L_tn_4x4x64_full has 37-bit level formats)."""
    by = {j["name"]: j for j in G.gemm_cases("cplx")}
    i = capi.classify(desc_from_dict(by["c5_tf_L_tn_4x4x64_full"]))
    assert capi.KERNEL_NAMES[i.kernel] == "tree_cplx"
    i = capi.classify(desc_from_dict(by["c5_tf_default_8x8x64_full"]))
    assert capi.KERNEL_NAMES[i.kernel] == "tree_cplx_i32"
    e = Qu(4, 3)
    d = lower(e, e, Qu(16, 3), 64, 64, 64, add_args=[Qu(40, 3)])     # wide level type, tree class (product rounds)
    assert capi.KERNEL_NAMES[capi.classify(d).kernel] == "tree_i64"
    d = lower(e, e, Qu(16, 3), 64, 64, 64, add_args=[Qu(20, 3)])
    assert capi.KERNEL_NAMES[capi.classify(d).kernel] == "tree_i32"


def test_smoke_expectations_hold_without_a_gpu():
    """__graft_entry__.smoke() asserts which kernel each of its cases runs on; the planner decides that on the host, so the
    expectations are checked here too (a planner change once turned the smoke red only on the GPU box)."""
    import __graft_entry__ as g
    for kernel, d, *_ in g.smoke_cases():
        assert capi.KERNEL_NAMES[capi.classify(d).kernel] == kernel
