#!/usr/bin/env python3
"""Regenerate tests/golden/ref_*.jsonl.gz from the REAL reference header.

Runs only in the build container (needs /root/reference and AMD clang for C++23): builds the
drivers of this directory into oracle/_ref/ (git-ignored) and stores what they print.  The
fixtures are data — resolved formats, raw integer inputs (or generator seeds) and raw integer
outputs; no reference source text is stored.  tests/golden/ref_rounding_kat.json is separate: it
transcribes the 40 known answers of the reference's own rounding tests by hand.
"""
import gzip
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
JOBS = {"ref_cases_real": range(9), "ref_cases_cplx": range(4), "ref_cases_scalar": range(11), "ref_cases_eltwise": range(3),
        "ref_cases_cplx_eltwise": range(3), "ref_cases_bitstream": range(1),
        "ref_cases_cplx_bitstream": range(1), "ref_cases_wide": range(6)}
OUT = {"ref_cases_real": "ref_gemm_real", "ref_cases_cplx": "ref_gemm_cplx", "ref_cases_scalar": "ref_scalar",
       "ref_cases_eltwise": "ref_eltwise", "ref_cases_cplx_eltwise": "ref_cplx_eltwise", "ref_cases_bitstream": "ref_bitstream",
       "ref_cases_cplx_bitstream": "ref_cplx_bitstream", "ref_cases_wide": "ref_wide"}


def main():
    if not os.path.isdir("/root/reference/include"):
        sys.exit("reference header not present: fixtures can only be regenerated in the build container")
    subprocess.check_call(["make", "-C", HERE, "-j4", "ref"])
    os.makedirs(GOLD, exist_ok=True)
    for exe, parts in JOBS.items():
        for p in parts:
            txt = subprocess.check_output([os.path.join(HERE, "_ref", exe), str(p)])
            path = os.path.join(GOLD, f"{OUT[exe]}_{p}.jsonl.gz")
            with gzip.GzipFile(path, "wb", mtime=0) as f:
                f.write(txt)
            print(path, len(txt))


if __name__ == "__main__":
    main()
