"""The C++ drop-in boundary end to end on the GPU (run with -m gpu):
  * oracle/_ref/ref_binding_demo — the REFERENCE header's own types + include/qgemul_reference_binding.hpp,
    built in the build container (the header cannot travel), executed here;
  * tests/binding/amd_header_run.cpp — the standalone include/QuBLAS_amd.h, compiled here with clang++ -std=c++23.
Both call Qgemul<...>(C, A, B) exactly as the README does and must print the matrices the reference's own
primitives produced (tests/golden)."""
import json
import os
import subprocess

import pytest

import golden_io as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _golden():
    return {j["name"]: j for j in G.gemm_cases("real")}


def _check(lines):
    gold = _golden()
    seen = 0
    for l in lines:
        r = json.loads(l)
        assert "error" not in r, r
        if r["name"] in gold:
            assert r["C"] == gold[r["name"]]["C"], r["name"]
            seen += 1
    return seen


def test_reference_header_binding_runs_on_gpu():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_binding_demo")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_binding_demo was not built (needs the reference header, build container only)")
    out = subprocess.check_output([exe], text=True)
    assert _check(out.strip().splitlines()) == 4


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs AMD clang (C++23)")
def test_standalone_header_runs_on_gpu(tmp_path):
    exe = tmp_path / "amd_header_run"
    lib = os.path.join(ROOT, "qublas_amd")
    subprocess.check_call([CLANG, "-std=c++23", "-O1", "-w", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "binding", "amd_header_run.cpp"), "-o", str(exe), "-L" + lib, "-lqugemm",
                           "-Wl,-rpath," + lib])
    out = subprocess.check_output([str(exe)], text=True)
    lines = out.strip().splitlines()
    assert _check(lines) == 4
    q = [json.loads(l) for l in lines if '"qreduce"' in l]
    assert q and q[0]["C"] == [80]
