"""Qreduce on the engine path (SURVEY.md §8-f "next" #1): the reference's vector tree reduction
(Qreduce<L...>(v), QuBLAS.h:4960-4990) lowered as the Qgemul  C[rows x 1] = A[rows x len] * ones.
The truth tables come from the real reference header (tests/golden/ref_scalar_4.jsonl.gz: lengths
1..1000, 0/1/2-entry level lists, result types as the reference reports them)."""
import numpy as np
import pytest

import golden_io as G
from qublas_amd.desc import ONE, Qu, lower_reduce, reduce_result_type


def _tables():
    # part 4: the reduce tables; part 7: signed SAT::SMGN (and TCPL / ZERO) element types with the raw minimum -2^W present;
    # part 8: the raw minimum as the LAST element (odd lengths: the odd leftover of level 0, copied unconverted)
    return [t for part in (4, 7, 8) for t in G.scalar_tables(part) if t["kind"] == "reduce"]


def _batch(oracle, t):
    """All seeds of one table as the rows of one batched reduce: A is len x rows column-major (transposed)."""
    fin = Qu.from_tuple(t["fin"])
    levels = [Qu.from_tuple(x) for x in t["levels"]]
    L = oracle.lib()
    n, rows = t["len"], len(t["seeds"])
    A = np.zeros(n * rows, dtype=np.int32)
    for r, seed in enumerate(t["seeds"]):
        A[r * n:(r + 1) * n] = [L.qoracle_synth(fin.c(), seed, t["dist"], i, 0) for i in range(n)]
    for r in range(rows):
        for i in t.get("min_at", []):
            A[r * n + i] = fin.raw_min
    ec = reduce_result_type(fin, levels, n)
    return fin, levels, ec, A, np.ones(n, dtype=np.int32), rows


@pytest.mark.parametrize("t", _tables(), ids=lambda t: t["name"])
def test_reduce_lowering_matches_reference(oracle, t):
    fin, levels, ec, A, ones, rows = _batch(oracle, t)
    for (y, fr) in t["y"]:
        assert list(ec.as_tuple()) == fr            # result type rule
    d = lower_reduce(fin, rows, t["len"], levels)
    out = oracle.gemm(d, A, ones, ec)
    assert [int(v) for v in out] == [y for (y, _) in t["y"]]


@pytest.mark.gpu
@pytest.mark.parametrize("t", _tables(), ids=lambda t: t["name"])
def test_reduce_on_gpu_matches_reference(oracle, t):
    from qublas_amd import capi
    fin, levels, ec, A, ones, rows = _batch(oracle, t)
    d = lower_reduce(fin, rows, t["len"], levels)
    out = capi.run(d, np.zeros(rows, dtype=oracle.host_dtype(ec)), A, ones)
    assert [int(v) for v in out] == [y for (y, _) in t["y"]]


@pytest.mark.gpu
def test_batched_reduce_large(oracle):
    """65536 vectors of length 1024, two-entry level list, against the CPU restatement."""
    from qublas_amd import capi
    e = Qu(4, 3)
    levels = [Qu(6, 3, True, 4, 2), Qu(12, 3)]
    rows, n = 4096, 1024
    d = lower_reduce(e, rows, n, levels)
    A = oracle.fill(e, rows * n, 7)
    ones = np.ones(n, np.int32)
    ec = reduce_result_type(e, levels, n)
    got = capi.run(d, np.zeros(rows, np.int32), A, ones)
    exp = oracle.gemm(d, A, ones, ec, nthreads=8)
    assert np.array_equal(got, exp)


# ---- the variadic overload Qreduce<L...>(q1, q2, ...): tests/golden/ref_scalar_10 (reference-generated: readme's 4-argument call,
# lengths 3 / 5 / 6 / 7, mixed argument types, 0 / 1 / 2-entry level lists and a TypeList)
def _variadic():
    return [t for t in G.scalar_tables(10) if t["kind"] == "reduce_variadic"]


def test_variadic_reduce_order_and_types(oracle):
    """the reduction order and result-type rules of qublas_amd.desc.reduce_variadic, every Qadd by the oracle's scalar restatement"""
    from qublas_amd.desc import add_merge, reduce_variadic
    L = oracle.lib()

    def node(x, fx, y, fy, tags):
        fr = add_merge(fx, fy, tags)
        return L.qoracle_add(x, fx.c(), y, fy.c(), fr.c(), 0), fr

    ts = _variadic()
    assert len(ts) == 48
    for t in ts:
        v, f = reduce_variadic(t["x"], [Qu.from_tuple(q) for q in t["fx"]], [Qu.from_tuple(q) for q in t["levels"]], node)
        assert (v, list(f.as_tuple())) == (t["y"], t["fr"]), t["name"]


def test_variadic_node_lowering(oracle):
    """... and every Qadd as the two-term Qgemul the engine runs (desc.add_node), on the oracle's GEMM"""
    from qublas_amd.desc import add_node, reduce_variadic

    def node(x, fx, y, fy, tags):
        d, sup, (sx, sy), fr = add_node(fx, fy, tags)
        A = np.array([x << sx, y << sy], dtype=oracle.host_dtype(sup))
        return int(oracle.gemm(d, A, np.ones(2, np.int32), fr)[0]), fr

    for t in _variadic():
        v, f = reduce_variadic(t["x"], [Qu.from_tuple(q) for q in t["fx"]], [Qu.from_tuple(q) for q in t["levels"]], node)
        assert (v, list(f.as_tuple())) == (t["y"], t["fr"]), t["name"]


@pytest.mark.gpu
def test_variadic_reduce_on_gpu(oracle):
    from qublas_amd import capi
    from qublas_amd.desc import add_node, reduce_variadic

    def node(x, fx, y, fy, tags):
        d, sup, (sx, sy), fr = add_node(fx, fy, tags)
        A = np.array([x << sx, y << sy], dtype=oracle.host_dtype(sup))
        return int(capi.run(d, np.zeros(1, dtype=oracle.host_dtype(fr)), A, np.ones(2, np.int32))[0]), fr

    for t in _variadic():
        v, f = reduce_variadic(t["x"], [Qu.from_tuple(q) for q in t["fx"]], [Qu.from_tuple(q) for q in t["levels"]], node)
        assert (v, list(f.as_tuple())) == (t["y"], t["fr"]), t["name"]
