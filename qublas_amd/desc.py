"""Descriptor lowering in Python — a mirror of the lowering in include/QuBLAS_amd.h and include/qgemul_reference_binding.hpp, used by the tests and the
benchmark harness to build `qgemul_desc` structures (include/qgemul.h) without a C++ compile.

It restates the reference's compile-time result-type rules:
  * defaults                       /root/reference/include/QuBLAS.h:2355-2359
  * MulMerger / AddMerger          QuBLAS.h:3107-3120, :3125-3139
  * a full Qu type used as tag     QuBLAS.h:3097-3099 (unwrapped into its five tags)
  * BasicComplexMul / TFComplexMul QuBLAS.h:3426-3445, :3510-3534 (incl. the crossed cdbT/badT use
                                   and the never-honoured baT)
  * Reducer level types            QuBLAS.h:4906-4921, :4960-4984
The C++ header is the product-side lowering; tests/test_lowering.py checks both against the formats
the reference's own types report (tests/golden/ref_gemm_*.jsonl, written by oracle/ref_driver.hpp).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional, Sequence, Tuple, Union

QGEMUL_ABI_VERSION = 1
QG_MAX_LEVELS = 40


class RND:
    POS_INF, NEG_INF, ZERO, INF, CONV = 0, 1, 2, 3, 4


class TRN:
    TCPL, SMGN = 5, 6


class SAT:
    TCPL, ZERO, SMGN = 0, 1, 2


class WRP:
    TCPL, TCPL_SAT = 3, 4


QU_MODES = {"RND::POS_INF": 0, "RND::NEG_INF": 1, "RND::ZERO": 2, "RND::INF": 3, "RND::CONV": 4,
            "TRN::TCPL": 5, "TRN::SMGN": 6}
OF_MODES = {"SAT::TCPL": 0, "SAT::ZERO": 1, "SAT::SMGN": 2, "WRP::TCPL": 3}

CMUL_NONE, CMUL_BASIC, CMUL_TF = 0, 1, 2
DESC_LEFTOVER0_COPY = 1   # qgemul_desc.flags (include/qgemul.h)
DESC_REFERENCE_ARTEFACTS = 2   # opt-in: C of an unsigned WRP::TCPL format with exactly 32 value bits comes out unwrapped, as in the reference
CLASS_LINEAR, CLASS_TREE = 1, 2


class qfmt(C.Structure):
    _fields_ = [("I", C.c_int16), ("F", C.c_int16), ("S", C.c_uint8), ("Q", C.c_uint8),
                ("O", C.c_uint8), ("pad", C.c_uint8)]


class qgemul_desc(C.Structure):
    _fields_ = [("abi", C.c_uint32), ("transA", C.c_uint8), ("is_complex", C.c_uint8),
                ("cmul", C.c_uint8), ("flags", C.c_uint8),
                ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
                ("a", qfmt * 2), ("b", qfmt * 2), ("c", qfmt * 2),
                ("mul", qfmt * 8),
                ("n_levels", C.c_uint32), ("reserved2", C.c_uint32),
                ("level_add", (qfmt * QG_MAX_LEVELS) * 2),
                ("level", (qfmt * QG_MAX_LEVELS) * 2)]


class qgemul_opts(C.Structure):
    _fields_ = [("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
                ("device", C.c_int32), ("flags", C.c_uint32)]


QG_MAX_EW = 4
EW_ADD, EW_SUB, EW_MUL, EW_PASS = 1, 2, 3, 4


class qgemul_ew_stage(C.Structure):
    _fields_ = [("op", C.c_uint8), ("x_first", C.c_uint8), ("e_scalar", C.c_uint8), ("reserved", C.c_uint8),
                ("e", qfmt), ("r", qfmt), ("t", qfmt)]


class qgemul_epilogue(C.Structure):
    _fields_ = [("n_stages", C.c_uint32), ("reserved", C.c_uint32), ("stage", qgemul_ew_stage * QG_MAX_EW), ("d", qfmt)]


class qgemul_epilogue_cplx(C.Structure):
    _fields_ = [("part", qgemul_epilogue * 2), ("e_complex", C.c_uint8 * QG_MAX_EW), ("reserved", C.c_uint8 * 4)]


class qgemul_ep_args(C.Structure):
    _fields_ = [("e_packed", C.c_void_p * QG_MAX_EW), ("e_scalar", C.c_int64 * QG_MAX_EW), ("e_scalar_im", C.c_int64 * QG_MAX_EW)]


class qgemul_info(C.Structure):
    _fields_ = [("cls", C.c_int32), ("supported", C.c_int32), ("max_bits", C.c_int32),
                ("in_bits", C.c_int32 * 2), ("limbs", C.c_int32 * 2), ("kernel", C.c_int32),
                ("host_elem_bytes", C.c_int32 * 3), ("host_imag_off", C.c_int32 * 3),
                ("packed_bytes", C.c_int64 * 3), ("ops", C.c_double), ("reason", C.c_char * 96)]


@dataclass(frozen=True)
class Qu:
    """A scalar fixed-point format: Qu<intBits<I>, fracBits<F>, isSigned<S>, QuMode<Q>, OfMode<O>>
    with the reference's defaults (QuBLAS.h:2355-2359)."""
    intBits: int = 8
    fracBits: int = 8
    isSigned: bool = True
    QuMode: int = TRN.TCPL
    OfMode: int = SAT.TCPL

    @property
    def W(self) -> int:
        return self.intBits + self.fracBits

    @property
    def storage_bits(self) -> int:
        return 1 + self.W  # always a sign bit, QuBLAS.h:2384-2385

    @property
    def host_bytes(self) -> int:
        # ArbiInt<N>::data_t, QuBLAS.h:353; beyond 64 bits a little-endian std::array<uint64_t, ceil(N/64)> (:572-573)
        return 4 if self.storage_bits <= 32 else 8 if self.storage_bits <= 64 else 8 * ((self.storage_bits + 63) // 64)

    @property
    def raw_min(self) -> int:
        return -(1 << self.W) if self.isSigned else 0

    @property
    def raw_max(self) -> int:
        return (1 << self.W) - 1

    def as_tuple(self) -> Tuple[int, int, int, int, int]:
        return (self.intBits, self.fracBits, int(self.isSigned), self.QuMode, self.OfMode)

    def c(self) -> qfmt:
        return qfmt(self.intBits, self.fracBits, int(self.isSigned), self.QuMode, self.OfMode, 0)

    @staticmethod
    def from_tuple(t: Sequence[int]) -> "Qu":
        return Qu(int(t[0]), int(t[1]), bool(t[2]), int(t[3]), int(t[4]))


@dataclass(frozen=True)
class Qcomplex:
    real: Qu
    imag: Qu


Elem = Union[Qu, Qcomplex]


@dataclass(frozen=True)
class Tags:
    """A loose tag list `<intBits<..>, fracBits<..>, isSigned<..>, QuMode<..>, OfMode<..>, FullPrec>`;
    None = tag absent.  `Tags.of(Qu(...))` is a full type used as tag (all five present)."""
    intBits: Optional[int] = None
    fracBits: Optional[int] = None
    isSigned: Optional[bool] = None
    QuMode: Optional[int] = None
    OfMode: Optional[int] = None
    FullPrec: bool = False

    @staticmethod
    def of(q: Qu) -> "Tags":
        return Tags(q.intBits, q.fracBits, q.isSigned, q.QuMode, q.OfMode, False)


TagLike = Union[Tags, Qu, None]


def _tags(t: TagLike) -> Tags:
    if t is None:
        return Tags()
    if isinstance(t, Qu):
        return Tags.of(t)
    return t


def _common(x, y, default):
    return x if x == y else default


def mul_merge(a: Qu, b: Qu, t: TagLike = None) -> Qu:
    """MulMerger::resType, QuBLAS.h:3107-3120."""
    t = _tags(t)
    fp = t.FullPrec
    return Qu(
        t.intBits if t.intBits is not None else (a.intBits + b.intBits if fp else max(a.intBits, b.intBits)),
        t.fracBits if t.fracBits is not None else (a.fracBits + b.fracBits if fp else max(a.fracBits, b.fracBits)),
        t.isSigned if t.isSigned is not None else (a.isSigned or b.isSigned),
        t.QuMode if t.QuMode is not None else _common(a.QuMode, b.QuMode, TRN.TCPL),
        t.OfMode if t.OfMode is not None else _common(a.OfMode, b.OfMode, SAT.TCPL),
    )


def add_merge(a: Qu, b: Qu, t: TagLike = None) -> Qu:
    """AddMerger::resType, QuBLAS.h:3125-3139 (also used by Qsub, :3214)."""
    t = _tags(t)
    fp = t.FullPrec
    return Qu(
        t.intBits if t.intBits is not None else (max(a.intBits, b.intBits) + 1 if fp else max(a.intBits, b.intBits)),
        t.fracBits if t.fracBits is not None else max(a.fracBits, b.fracBits),
        t.isSigned if t.isSigned is not None else (a.isSigned or b.isSigned),
        t.QuMode if t.QuMode is not None else _common(a.QuMode, b.QuMode, TRN.TCPL),
        t.OfMode if t.OfMode is not None else _common(a.OfMode, b.OfMode, SAT.TCPL),
    )


@dataclass(frozen=True)
class BasicComplexMul:
    """BasicComplexMul<acT<>, bdT<>, adT<>, bcT<>, acbdT<>, adbcT<>, loose tags…>.
    A sub-op without its own tag sees the wrapper's loose tags (its default is xT<toArgs…>,
    whose ::list is the whole argument list, QuBLAS.h:3429-3435 + :164-170)."""
    acT: TagLike = None
    bdT: TagLike = None
    adT: TagLike = None
    bcT: TagLike = None
    acbdT: TagLike = None
    adbcT: TagLike = None
    loose: TagLike = None


@dataclass(frozen=True)
class TFComplexMul:
    """TFComplexMul<abT<>, cdT<>, baT<>, abcT<>, cdbT<>, badT<>, ABT<>, BCT<>, loose tags…>."""
    abT: TagLike = None
    cdT: TagLike = None
    baT: TagLike = None   # accepted and ignored, exactly like the reference (QuBLAS.h:3515)
    abcT: TagLike = None
    cdbT: TagLike = None
    badT: TagLike = None
    ABT: TagLike = None
    BCT: TagLike = None
    loose: TagLike = None


MulArgs = Union[Tags, Qu, BasicComplexMul, TFComplexMul, None]


def _pick(own: TagLike, loose: TagLike) -> Tags:
    return _tags(own) if own is not None else _tags(loose)


def complex_mul_slots(x: Qcomplex, y: Qcomplex, m: MulArgs) -> Tuple[int, List[Qu]]:
    """Resolved result formats of every sub-operation, in the slot order of include/qgemul.h."""
    a, b, c, d = x.real, x.imag, y.real, y.imag
    zero = Qu(0, 0, False, 0, 0)
    if m is None:
        m = BasicComplexMul()  # Qmul(c1, c2) without arguments, QuBLAS.h:3422-3424
    if isinstance(m, (Tags, Qu)):
        raise ValueError("complex Qgemul needs BasicComplexMul/TFComplexMul (or no) QgemulMulArgs")
    if isinstance(m, BasicComplexMul):
        ac = mul_merge(a, c, _pick(m.acT, m.loose))
        bd = mul_merge(b, d, _pick(m.bdT, m.loose))
        ad = mul_merge(a, d, _pick(m.adT, m.loose))
        bc = mul_merge(b, c, _pick(m.bcT, m.loose))
        re = add_merge(ac, bd, _pick(m.acbdT, m.loose))
        im = add_merge(ad, bc, _pick(m.adbcT, m.loose))
        return CMUL_BASIC, [ac, bd, ad, bc, re, im, zero, zero]
    ab = add_merge(a, b, _pick(m.abT, m.loose))
    cd = add_merge(c, d, _pick(m.cdT, m.loose))
    ba = add_merge(b, a, None)                      # quirk 2: always the default merge
    A = mul_merge(ab, c, _pick(m.abcT, m.loose))
    B = mul_merge(cd, b, _pick(m.badT, m.loose))    # quirk 1: B is quantised with badT
    Cc = mul_merge(ba, d, _pick(m.cdbT, m.loose))   # quirk 1: C is quantised with cdbT
    re = add_merge(A, B, _pick(m.ABT, m.loose))
    im = add_merge(B, Cc, _pick(m.BCT, m.loose))
    return CMUL_TF, [ab, cd, ba, A, B, Cc, re, im]


def n_levels_for(K: int) -> int:
    n = 0
    while K > 1:
        K = (K + 1) // 2
        n += 1
    return n


def lower(A: Elem, B: Elem, Cc: Elem, M: int, N: int, K: int, *,
          add_args: Optional[Sequence[Elem]] = None, mul_args: MulArgs = None,
          transposed_a: bool = False, reference_artefacts: bool = False) -> qgemul_desc:
    """Lower one Qgemul<QgemulAddArgs<add_args…>, QgemulMulArgs<mul_args>, QgemulTransposedA<t>>
    call on element types (A, B, C) and runtime sizes to the C-ABI descriptor."""
    cx = isinstance(A, Qcomplex)
    if cx != isinstance(B, Qcomplex) or cx != isinstance(Cc, Qcomplex):
        raise ValueError("mixed real/complex operands are outside the Qgemul path (SURVEY.md §2)")
    if K < 1 or M < 0 or N < 0:
        raise ValueError("bad shape")
    d = qgemul_desc()
    d.abi = QGEMUL_ABI_VERSION
    d.transA = int(bool(transposed_a))
    d.is_complex = int(cx)
    d.flags = DESC_REFERENCE_ARTEFACTS if reference_artefacts else 0   # (opt-in, include/qgemul.h)
    d.M, d.N, d.K = M, N, K
    nl = n_levels_for(K)
    if nl > QG_MAX_LEVELS:
        raise ValueError("K too large")
    d.n_levels = nl
    levels = list(add_args) if add_args else []
    if not cx:
        d.a[0] = d.a[1] = A.c()
        d.b[0] = d.b[1] = B.c()
        d.c[0] = d.c[1] = Cc.c()
        if isinstance(mul_args, (BasicComplexMul, TFComplexMul)):
            raise ValueError("complex multiplier tags on a real Qgemul")
        prod = mul_merge(A, B, mul_args)
        d.cmul = CMUL_NONE
        d.mul[0] = prod.c()
        prev = prod
        for l in range(nl):
            if levels:
                t = levels[min(l, len(levels) - 1)]
                if not isinstance(t, Qu):
                    raise ValueError("a real Qgemul needs real level types")
            else:
                t = prev  # Qadd<nullptr_t>: default merge of two equal types = that type
            d.level_add[0][l] = d.level_add[1][l] = t.c()
            d.level[0][l] = d.level[1][l] = t.c()
            prev = t
    else:
        d.a[0], d.a[1] = A.real.c(), A.imag.c()
        d.b[0], d.b[1] = B.real.c(), B.imag.c()
        d.c[0], d.c[1] = Cc.real.c(), Cc.imag.c()
        cmul, slots = complex_mul_slots(A, B, mul_args)
        d.cmul = cmul
        for i, s in enumerate(slots):
            d.mul[i] = s.c()
        prev = (slots[6], slots[7]) if cmul == CMUL_TF else (slots[4], slots[5])
        for l in range(nl):
            if levels:
                t = levels[min(l, len(levels) - 1)]
                if not isinstance(t, Qcomplex):
                    raise ValueError("a complex Qgemul needs complex level types (QuBLAS.h:4966)")
                buf = (t.real, t.imag)
            else:
                buf = prev
            for p in range(2):
                # a complex type passed as Qadd tag is ignored: the add is the default merge of the
                # incoming format, the level buffer then converts
                d.level_add[p][l] = add_merge(prev[p], prev[p], None).c()
                d.level[p][l] = buf[p].c()
            prev = buf
    return d


def desc_to_dict(d: qgemul_desc) -> Dict:
    """Same shape as the JSON records oracle/ref_driver.hpp prints."""
    f = lambda q: [q.I, q.F, q.S, q.Q, q.O]
    nl = d.n_levels
    return {
        "M": d.M, "N": d.N, "K": d.K, "transA": d.transA, "is_complex": d.is_complex, "cmul": d.cmul, "flags": d.flags,
        "a": [f(d.a[0]), f(d.a[1])], "b": [f(d.b[0]), f(d.b[1])], "c": [f(d.c[0]), f(d.c[1])],
        "mul": [f(d.mul[i]) for i in range(8)], "n_levels": nl,
        "level_add": [[f(d.level_add[0][l]), f(d.level_add[1][l])] for l in range(nl)],
        "level": [[f(d.level[0][l]), f(d.level[1][l])] for l in range(nl)],
    }


def desc_from_dict(j: Dict) -> qgemul_desc:
    """Build a descriptor straight from a golden record (formats as the reference resolved them)."""
    d = qgemul_desc()
    d.abi = QGEMUL_ABI_VERSION
    d.transA, d.is_complex, d.cmul = j["transA"], j["is_complex"], j["cmul"]
    d.flags = j.get("flags", 0)
    d.M, d.N, d.K = j["M"], j["N"], j["K"]
    mk = lambda t: qfmt(t[0], t[1], t[2], t[3], t[4], 0)
    for p in range(2):
        d.a[p], d.b[p], d.c[p] = mk(j["a"][p]), mk(j["b"][p]), mk(j["c"][p])
    for i in range(8):
        d.mul[i] = mk(j["mul"][i])
    d.n_levels = j["n_levels"]
    for l in range(d.n_levels):
        for p in range(2):
            d.level_add[p][l] = mk(j["level_add"][l][p])
            d.level[p][l] = mk(j["level"][l][p])
    return d


def elem_parts(e: Elem) -> Tuple[Qu, Qu]:
    return (e.real, e.imag) if isinstance(e, Qcomplex) else (e, e)


def host_layout(e: Elem) -> Tuple[int, int, Tuple[int, int]]:
    """(element bytes, byte offset of .imag, (bytes of real, bytes of imag)) of the reference's
    host element: int32/int64 per part, a complex element is struct {real; imag;} (QuBLAS.h:2512-2513)."""
    if not isinstance(e, Qcomplex):
        return e.host_bytes, 0, (e.host_bytes, 0)
    sr, si = e.real.host_bytes, e.imag.host_bytes
    ar, ai = min(sr, 8), min(si, 8)       # alignment: int32_t / int64_t / uint64_t[n]
    off = (sr + ai - 1) // ai * ai
    al = max(ar, ai)
    size = (off + si + al - 1) // al * al
    return size, off, (sr, si)


# ---------------------------------------------------------------------------------------------
# Qreduce (SURVEY.md §8-f "next" #1): the reference's tree reduction of a vector,
# Qreduce<L...>(v) (/root/reference/include/QuBLAS.h:4960-4990, :5014-5018), expressed on the SAME engine path:
# a batch of `rows` vectors of length `length` is the Qgemul  C[rows x 1] = A[rows x length] * ones[length x 1]
# whose product format is the element's own format (Qmul(a, 1) into a's format is the identity: no
# rounding shift, value in range; for a signed SAT::SMGN element type the leaf format is the element's
# with SAT::TCPL, see lower_reduce) and whose level list is L.  The result type is the reducer's
# result type: the last level type, or the element type without levels (len 1: the element itself).
ONE = Qu(1, 0, False)   # the constant 1 as an unsigned 1-bit integer


def reduce_result_type(e: Qu, levels, length: int) -> Qu:
    if length <= 1 or not levels:
        return e
    return levels[min(n_levels_for(length), len(levels)) - 1]


def lower_reduce(e: Qu, rows: int, length: int, levels=None) -> qgemul_desc:
    if isinstance(e, Qcomplex):
        raise ValueError("complex Qreduce is not lowered yet")
    levels = list(levels) if levels else []
    ec = reduce_result_type(e, levels, length)
    if e.isSigned and e.OfMode == SAT.SMGN:
        # the one element type for which Qmul(a, 1) into a's own format is not the identity: symmetric saturation clamps the raw
        # minimum -2^W, which the reference's Qreduce adds as it is (tests/golden/ref_scalar_7).  The leaf conversion
        # therefore targets a's format with SAT::TCPL (the identity on every raw value), and the default level type — the
        # merge of two elements, i.e. the element type itself (AddMerger, QuBLAS.h:3125-3139) — is named explicitly.
        leaf = Qu(e.intBits, e.fracBits, e.isSigned, e.QuMode, SAT.TCPL)
        d = lower(e, ONE, leaf if length <= 1 else ec, rows, 1, length, add_args=levels or [e], mul_args=leaf, transposed_a=True)
        # ... and where level 0's type IS the element type, the reference's copy of an odd leftover into the level-0 buffer is a
        # same-type copy (QuBLAS.h:4977-4980): the raw minimum survives it, so the descriptor asks for the unconverted copy
        if (levels or [e])[0] == e:
            d.flags |= DESC_LEFTOVER0_COPY
        return d
    return lower(e, ONE, ec, rows, 1, length, add_args=levels, mul_args=e, transposed_a=True)


# ---- the VARIADIC Qreduce<L...>(q1, q2, ...) (readme.md:62; QuBLAS.h:4924-4951): any number of scalars of any types.  Level l adds
# neighbours with Qadd<T_l> (T_l = L[min(l, n-1)], no L: default merge); an odd leftover is added AFTER the recursion over the
# pair sums, with the CURRENT level's type (the vector overload copies it into the next level instead).  Every Qadd<T>(x, y) of two
# scalars of different types is one two-term Qgemul on the engine: both operands written exactly in a common super-format (the
# alignment shifts Qadd performs itself, QuBLAS.h:3190), times 1, level 0 = the node's result type.
def add_node(fx: Qu, fy: Qu, tags: TagLike = None):
    """(descriptor of the K = 2 Qgemul that computes Qadd<tags>(x, y), super-format, (shift of x, shift of y), result type)"""
    fr = add_merge(fx, fy, tags)
    F = max(fx.fracBits, fy.fracBits)
    sup = Qu(max(fx.intBits, fy.intBits), F, fx.isSigned or fy.isSigned)
    d = lower(sup, ONE, fr, 1, 1, 2, add_args=[fr], mul_args=sup, transposed_a=True)
    return d, sup, (F - fx.fracBits, F - fy.fracBits), fr


def reduce_variadic(values: Sequence[int], fmts: Sequence[Qu], levels: Sequence[Qu], node, layer: int = 0):
    """The reference's variadic reduction order and types; node(x, fx, y, fy, tags) -> (value, format) performs one Qadd."""
    n = len(values)
    if n == 1:
        return values[0], fmts[0]
    tags = levels[min(layer, len(levels) - 1)] if levels else None
    pv, pf = [], []
    for i in range(n // 2):
        v, f = node(values[2 * i], fmts[2 * i], values[2 * i + 1], fmts[2 * i + 1], tags)
        pv.append(v)
        pf.append(f)
    rv, rf = reduce_variadic(pv, pf, levels, node, layer + 1)
    if n % 2:
        return node(rv, rf, values[-1], fmts[-1], tags)
    return rv, rf


# ---- element-wise epilogue: the lazy tensor operators after a Qgemul ----

@dataclass(frozen=True)
class Ew:
    """One lazy tensor operator applied to the running value x (QuBLAS.h:3780-3877, front-ends :4079-4100):
    op in {"add", "sub", "mul"}; e = the other operand's element type; tags = the operator's own toArgs;
    x_first False means Qop(e, x); scalar True means e is an isScalar operand (autoCall, :3767-3778);
    into = element type of the tensor the result is assigned to before the next operator (None: the operator's own
    result type, i.e. no conversion; ignored for the last stage, whose result goes into D)."""
    op: str
    e: "Qu"
    tags: TagLike = None
    x_first: bool = True
    scalar: bool = False
    into: Optional["Qu"] = None


def ew_result(x: "Qu", st: Ew) -> "Qu":
    first, second = (x, st.e) if st.x_first else (st.e, x)
    return mul_merge(first, second, st.tags) if st.op == "mul" else add_merge(first, second, st.tags)


def lower_epilogue(c: "Qu", stages, d: "Qu") -> qgemul_epilogue:
    """Resolve the chain D = cvt_d(Qop_n(... Qop_1(C, e_1) ..., e_n)) to the C-ABI struct."""
    stages = list(stages)
    if len(stages) > QG_MAX_EW:
        raise ValueError("too many element-wise stages")
    ep = qgemul_epilogue()
    ep.n_stages = len(stages)
    x = c
    for k, st in enumerate(stages):
        r = ew_result(x, st)
        ep.stage[k].op = {"add": EW_ADD, "sub": EW_SUB, "mul": EW_MUL}[st.op]
        ep.stage[k].x_first = 1 if st.x_first else 0
        ep.stage[k].e_scalar = 1 if st.scalar else 0
        ep.stage[k].e = st.e.c()
        ep.stage[k].r = r.c()
        x = st.into if (st.into is not None and k + 1 < len(stages)) else r
        ep.stage[k].t = x.c()
    ep.d = d.c()
    return ep


# ---- the same after a COMPLEX Qgemul: two part-wise chains (include/qgemul.h, qgemul_epilogue_cplx) ----

@dataclass(frozen=True)
class EwC:
    """One lazy tensor operator on a complex running value x.  e: Qcomplex (complex Qadd / Qsub, QuBLAS.h:3549-3589) or
    Qu (an operator between a complex and a real value: Qmul :3604-3644, Qadd :3648-3676, Qsub :3680-3707).
    real_tags / imag_tags = realT<...> / imagT<...> (or the two types of the <Qu1, Qu2> form, :3566-3568); tags = loose
    tags, which a part without its own wrapper sees (the extractor's default realT<toArgs...>, :3551-3552).  complex (+|-) real
    hands ALL its tags to the one real Qadd / Qsub (:3654, :3670, :3686, :3701): only `tags` applies there."""
    op: str
    e: "Elem"
    tags: TagLike = None
    real_tags: TagLike = None
    imag_tags: TagLike = None
    x_first: bool = True
    scalar: bool = False
    into: Optional["Qcomplex"] = None


def lower_epilogue_cplx(c: "Qcomplex", stages, d: "Qcomplex") -> qgemul_epilogue_cplx:
    stages = list(stages)
    if len(stages) > QG_MAX_EW:
        raise ValueError("too many element-wise stages")
    epc = qgemul_epilogue_cplx()
    x = [c.real, c.imag]
    opc = {"add": EW_ADD, "sub": EW_SUB, "mul": EW_MUL}
    for p in range(2):
        epc.part[p].n_stages = len(stages)
    for k, st in enumerate(stages):
        cplx = isinstance(st.e, Qcomplex)
        if cplx and st.op == "mul":
            raise ValueError("complex x complex multiplication mixes the parts: not an element-wise epilogue stage")
        epc.e_complex[k] = 1 if cplx else 0
        r = [None, None]
        for p in range(2):
            sg = epc.part[p].stage[k]
            e = (st.e.real, st.e.imag)[p] if cplx else st.e
            own = st.real_tags if p == 0 else st.imag_tags
            sg.x_first = 1 if st.x_first else 0
            sg.e_scalar = 1 if st.scalar else 0
            sg.e = e.c()
            if cplx or st.op == "mul":
                t = _pick(own, st.tags)
            else:
                t = _tags(st.tags)
                if p == 1 and (st.op == "add" or st.x_first):
                    sg.op = EW_PASS                 # the imaginary part is carried over
                    sg.e_scalar = 1
                    r[p] = x[p]
                    sg.r = r[p].c()
                    continue
                if p == 1:
                    sg.e_scalar = 1                 # real - complex: Qsub(zero of the operand's type, x.imag)
            first, second = (x[p], e) if st.x_first else (e, x[p])
            r[p] = mul_merge(first, second, t) if st.op == "mul" else add_merge(first, second, t)
            sg.op = opc[st.op]
            sg.r = r[p].c()
        for p in range(2):
            x[p] = ((st.into.real, st.into.imag)[p] if (st.into is not None and k + 1 < len(stages)) else r[p])
            epc.part[p].stage[k].t = x[p].c()
    epc.part[0].d = d.real.c()
    epc.part[1].d = d.imag.c()
    return epc

