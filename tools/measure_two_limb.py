import sys, os, json
sys.path.insert(0, os.getcwd())
from qublas_amd import capi
from qublas_amd.desc import Qcomplex, Qu, RND, SAT, Tags, BasicComplexMul, lower
R63 = Qu(6, 3, True, RND.POS_INF, SAT.TCPL); I63N = Qu(6, -3, True, RND.POS_INF, SAT.TCPL); C5 = Qcomplex(R63, I63N)
E77 = Qu(7, 7)
cases = [("c5 linear (stacked 2x2 limbs + combine)", lower(C5, C5, C5, 2048, 2048, 2048, mul_args=BasicComplexMul(acT=Qu(14, 6), bdT=Qu(14, -6), adT=Qu(14, 0), bcT=Qu(14, 0), acbdT=Qu(15, 6), adbcT=Qu(15, 0)), add_args=[Qcomplex(Qu(30, 6), Qu(30, 0))])),
         ("4096^3 int<7,7> real 2x2 limbs", lower(E77, E77, Qu(20, 8), 4096, 4096, 4096, mul_args=Tags(15, 14), add_args=[Qu(27, 14)]))]
with capi.Context(0) as ctx:
    for name, d in cases:
        arms = []
        for nm, fl in (("two-group", 0), ("lockstep", capi.OPT_LOCKSTEP_TILES)):
            p = capi.Plan(ctx, d, fl); pb = p.info.packed_bytes
            a, b, c = ctx.alloc(pb[0]), ctx.alloc(pb[1]), ctx.alloc(pb[2])
            p.fill(capi.OPERAND_A, 1, 0, a); p.fill(capi.OPERAND_B, 2, 0, b)
            p.time_execute(c, a, b, 5, 5)
            arms.append((nm, p, a, b, c, []))
        for _ in range(7):
            for nm, p, a, b, c, ts in arms:
                ts.append(p.time_execute(c, a, b, 1, 20))
        for nm, p, a, b, c, ts in arms:
            ts.sort(); print(json.dumps({"case": name, "kernel": nm, "ms_median": ts[3], "ms_min": ts[0]}), flush=True)
