"""The planner (qublas_amd/csrc/qg_plan.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU — the reference
builds all of its tests that way (CMakeLists.txt:17,26).  Descriptors: every reference-generated golden GEMM, the element-wise
cases, and a few thousand random ones (valid, borderline and deliberately malformed: huge shifts, negative widths, level counts
at the array bound).  The sanitized driver must agree with the product library's classification and report nothing."""
import ctypes as C
import os
import random
import subprocess

import pytest

import golden_io as G
from qublas_amd import capi
from qublas_amd.desc import Qu, Tags, desc_from_dict, lower, qgemul_desc, qgemul_epilogue

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_driver(tmp):
    exe = os.path.join(tmp, "plan_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "san", "plan_san_driver.cpp"), os.path.join(ROOT, "qublas_amd", "csrc", "qg_plan.cpp"), "-o", exe]
    subprocess.check_call(cmd)
    return exe


def random_descs(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        wa, wb = rng.randint(1, 30), rng.randint(1, 30)
        fa, fb = rng.randint(-4, 12), rng.randint(-4, 12)
        ea = Qu(max(wa - max(fa, 0), 0), fa, rng.random() < 0.7, rng.randint(0, 6), rng.randint(0, 3))
        eb = Qu(max(wb - max(fb, 0), 0), fb, rng.random() < 0.7, rng.randint(0, 6), rng.randint(0, 3))
        ec = Qu(rng.randint(0, 40), rng.randint(-6, 20), rng.random() < 0.7, rng.randint(0, 6), rng.randint(0, 4))
        kw = {}
        if rng.random() < 0.6:
            kw["mul_args"] = Tags(rng.randint(0, 40), rng.randint(-4, 24))
        if rng.random() < 0.6:
            kw["add_args"] = [Qu(rng.randint(0, 50), rng.randint(-4, 24), True, rng.randint(0, 6), rng.randint(0, 3)) for _ in range(rng.randint(1, 3))]
        M, N, K = rng.choice([1, 7, 64, 4096, 70000]), rng.choice([1, 5, 256, 16384]), rng.choice([1, 2, 3, 100, 1024, 4096, 65536, 200000])
        try:
            d = lower(ea, eb, ec, M, N, K, **kw)
        except (ValueError, OverflowError):
            continue
        if rng.random() < 0.1:      # malformed on purpose: the planner must reject, not misbehave
            which = rng.randint(0, 3)
            if which == 0:
                d.n_levels = rng.choice([0, 39, 40, 41, 1000])
            elif which == 1:
                d.c[0].F = rng.choice([-30000, 30000])
            elif which == 2:
                d.mul[0].I = rng.choice([-200, 32767])
            else:
                d.a[0].Q = 200
        out.append(d)
    return out


def compact_form_descs(n, seed):
    """descriptors whose roundings add a constant and whose overflows clamp / test / wrap: the planner's compact step records
    (QFix: fast_mode 3 / 4 / 5, cplx_fixed_ok 2 / 3 / 8 + features, gemv_fixed 3 / 5) are built for most of them"""
    from qublas_amd.desc import BasicComplexMul, Qcomplex, TFComplexMul, lower_reduce
    rng = random.Random(seed)
    QM, OM = [5, 0, 1], [0, 0, 2, 1, 3]     # TRN::TCPL, RND::POS_INF, RND::NEG_INF; SAT::TCPL, SAT::SMGN, SAT::ZERO, WRP::TCPL

    def rq(bits, om=OM):
        i = rng.randint(0, bits)
        return Qu(i + rng.choice([0, 0, 2]), bits - i - rng.choice([0, 0, 3]), rng.random() < 0.85, rng.choice(QM), rng.choice(om))
    out = []
    while len(out) < n:
        kind = rng.random()
        K = rng.choice([16, 17, 64, 100, 1024, 4096])
        try:
            if kind < 0.4:          # real, per-level formats
                ea = rq(rng.choice([4, 8, 12, 16]))
                d = lower(ea, ea, rq(12), rng.choice([1, 70]), rng.choice([1, 33]), K, mul_args=rq(rng.choice([8, 12, 20])) if rng.random() < 0.6 else None,
                          add_args=[rq(rng.choice([10, 18, 26, 30])) for _ in range(rng.randint(0, 3))] or None)
            elif kind < 0.55:       # Qreduce lowering (incl. signed SAT::SMGN element types)
                d = lower_reduce(rq(rng.choice([4, 8, 12])), rng.choice([1, 9]), rng.choice([1, 2, 16, 64, 4096]), [rq(rng.choice([10, 18])) for _ in range(rng.randint(0, 2))] or None)
            else:                   # complex: clamping modes, or (a third) any rounding / overflow kind
                allk = rng.random() < 0.34
                qm, om = (list(range(7)), [0, 1, 2, 3]) if allk else (QM, [0, 0, 2])

                def c(b):
                    parts = []
                    for _ in range(2):
                        i = rng.randint(0, b)
                        parts.append(Qu(i, b - i, rng.random() < 0.85, rng.choice(qm), rng.choice(om)))
                    return Qcomplex(*parts)
                ea = c(rng.choice([5, 8, 11]))
                mul = TFComplexMul(abcT=Tags(rng.randint(4, 12), rng.randint(-2, 8)), ABT=rq(rng.choice([8, 14, 29]), [0, 2]) if rng.random() < 0.3 else None) if rng.random() < 0.5 \
                    else BasicComplexMul(acT=Tags(rng.randint(4, 24), rng.randint(-2, 10)) if rng.random() < 0.5 else None)
                d = lower(ea, ea, c(12), 9, 7, K, mul_args=mul, add_args=[c(rng.choice([10, 16, 24])) for _ in range(rng.randint(0, 2))] or None)
        except (ValueError, OverflowError):
            continue
        out.append(d)
    return out


def test_planner_under_asan_ubsan(tmp_path):
    exe = build_driver(str(tmp_path))
    descs = [desc_from_dict(j) for j in G.gemm_cases("real") + G.gemm_cases("cplx")] + random_descs(3000, 5) + compact_form_descs(1500, 6)
    blob = bytearray()
    ep0 = qgemul_epilogue()
    for d in descs:
        blob += bytes(d) + bytes(ep0)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], input=bytes(blob), capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr.decode()[-3000:]
    lines = r.stdout.decode().splitlines()
    assert len(lines) == len(descs)
    # same verdicts as the product library (the same source compiled by hipcc)
    L = capi.lib()
    n_ok = 0
    forms = {"fast": set(), "cplx": set(), "gemv": set()}
    for ln, d in zip(lines, descs):
        _, st, cls, bits, _, _, _, fm, cf, gf = ln.split()
        forms["fast"].add(int(fm)); forms["cplx"].add(int(cf)); forms["gemv"].add(int(gf))
        info = capi.qgemul_info()
        assert L.qgemul_classify(C.byref(d), 0, C.byref(info)) == int(st)
        if int(st) == 0:
            assert info.cls == int(cls) and info.max_bits == int(bits)
            n_ok += 1
    assert n_ok > 500
    # the record-building code ran under the sanitizers for every form
    assert {1, 2, 3, 4, 5} <= forms["fast"] and {1, 2, 3} <= forms["cplx"] and any(f >= 8 for f in forms["cplx"]) and {1, 2, 3, 5} <= forms["gemv"], forms
