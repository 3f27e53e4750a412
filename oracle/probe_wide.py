#!/usr/bin/env python3
"""Which tables of oracle/ref_cases_wide.cpp can the reference header COMPILE?  (build container only)

The reference's multi-word ArbiInt code is ill-formed for some combinations of widths and modes.  This script compiles every
table id on its own (-DPROBE=<id> -fsyntax-only, which instantiates exactly that table) and writes
oracle/ref_cases_wide_enabled.inc: N_IDS flags, 1 = the reference compiles it.  The file is committed (the reference snapshot
is fixed); re-run after editing ref_cases_wide.cpp.  A compile that fails also records the first error line in
oracle/ref_cases_wide_probe.log."""
import concurrent.futures as cf
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SRC = os.path.join(HERE, "ref_cases_wide.cpp")


def used_ids():
    txt = open(SRC).read()
    ids = set(int(m) for m in re.findall(r"CASE\((\d+),", txt))
    for base in re.findall(r"cvt_all_w<D, (\d+),", txt):
        ids.update(range(int(base), int(base) + 28))
    n = int(re.search(r"N_IDS = (\d+)", txt).group(1))
    return sorted(ids), n


def probe(i):
    r = subprocess.run([CLANG, "-std=c++23", "-w", "-fsyntax-only", "-ferror-limit=1", f"-DPROBE={i}", "-I/root/reference/include", SRC],
                       capture_output=True, text=True)
    first = next((l for l in r.stderr.splitlines() if "error:" in l), "")
    return i, r.returncode == 0, first


def main():
    if not os.path.isdir("/root/reference/include"):
        sys.exit("reference header not present")
    ids, n = used_ids()
    ok = [0] * n
    log = []
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        for i, good, first in ex.map(probe, ids):
            ok[i] = int(good)
            if not good:
                log.append(f"{i}: {first}")
    with open(os.path.join(HERE, "ref_cases_wide_enabled.inc"), "w") as f:
        for r in range(0, n, 28):
            f.write(", ".join(str(x) for x in ok[r:r + 28]) + ",\n")
    with open(os.path.join(HERE, "ref_cases_wide_probe.log"), "w") as f:
        f.write("\n".join(log) + "\n")
    print(f"{sum(ok)} of {len(ids)} tables compile; {len(log)} do not (oracle/ref_cases_wide_probe.log)")


if __name__ == "__main__":
    main()
