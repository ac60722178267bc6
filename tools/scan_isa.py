#!/usr/bin/env python3
"""Scan the device assembly of the HIP library for a code pattern hipcc 7.2 was caught emitting for a `min` of uniform
64-bit values: a vector compare (v_cmp_*_[ui]64, result in VCC) followed by an s_cselect that reads SCC, with no scalar
compare in between (SCC then still holds the carry of an earlier s_sub / s_add).  Prints the suspicious sites.

Usage: python tools/scan_isa.py   (needs hipcc; compiles insilicoseq_amd/csrc/iss_mi355x.hip to assembly)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCC_WRITERS = re.compile(r"^\s*(s_cmp|s_cmpk|s_add|s_sub|s_addc|s_subb|s_and|s_or|s_xor|s_andn2|s_orn2|s_nand|s_nor|s_xnor|"
                         r"s_lshl|s_lshr|s_ashr|s_bfe|s_min|s_max|s_abs|s_not|s_bcnt|s_wqm|s_quadmask|s_bitcmp|s_absdiff)")


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I", os.path.join(ROOT, "include"), "-S",
                               "--cuda-device-only", "-o", asm, os.path.join(ROOT, "insilicoseq_amd", "csrc", "iss_mi355x.hip")])
        lines = open(asm).read().splitlines()
    func, hits = None, []
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            func = m.group(1)
        if not re.match(r"^\s*s_cselect", line):
            continue
        j = i - 1
        while j >= 0 and not SCC_WRITERS.match(lines[j]) and not lines[j].startswith(".LBB") and not re.match(r"^_Z", lines[j]):
            j -= 1
        prev = lines[j].strip() if j >= 0 else ""
        carry = re.match(r"^(s_sub|s_add|s_addc|s_subb|s_lshl|s_lshr|s_mul)", prev) is not None
        vcmp = any(re.search(r"v_cmp_(lt|gt|le|ge)_[ui]64", x) for x in lines[j:i])
        if not prev.startswith("s_cmp") and not prev.startswith(".LBB") and (carry or vcmp):
            hits.append((func, i + 1, prev, line.strip()))
    for h in hits:
        print("%s: line %d: `%s` ... `%s`" % h)
    print("%d suspicious site(s) in %d lines of device assembly" % (len(hits), len(lines)))
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
