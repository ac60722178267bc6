#!/usr/bin/env python3
"""Scan the device assembly of the HIP library for a code pattern hipcc 7.2 was caught emitting for a `min` of uniform
64-bit values: a vector compare (v_cmp_*_[ui]64, result in VCC) followed by an s_cselect that reads SCC, with no scalar
compare in between (SCC then still holds the carry of an earlier s_sub / s_add).  Prints the suspicious sites.

Also: flat_* memory operations in the worker-set kernels of MT mode (k_mt_*_w) and spilled vector registers in the hot kernels
(k_main, k_main_g) -- neither changes a result, both cost what the kernels were built to save.

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
    # Second scan: the worker-set kernels of MT mode (k_mt_*_w) read their jobs from a table in HBM; their pointers are rebuilt as
    # (kernel argument + distance) behind an empty asm (iss_mt_compat.hip.h: as_global) so that the accesses are global_*, not
    # flat_* -- flat operations count on the LDS counter the resolver's barriers wait for.  A compiler that folds the expression
    # brings the flat accesses back without changing a result: fail the build check instead.  The grouped hot kernel keeps its
    # rows in registers: private memory in k_main_g would be the rows in scratch.
    # (k_mt_walk_w -- the sequential walker, the rare path -- reaches its LDS-staged tables through generic pointers like the
    #  single-worker k_mt_walk does: WALK_W_FLAT_MAX is today's count with room for the compiler's mood, not a target.  A hot
    #  kernel may spill ONE loop-invariant register pair that it reloads once per group of passes: k_main_g<2, 2>.)
    WALK_W_FLAT_MAX, HOT_SPILL_MAX = 120, 2
    func, flat, scratch = None, {}, {}
    for line in lines:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            func = m.group(1)
        if func and re.search(r"k_mt_\w+_w", func) and re.match(r"^\s*flat_", line):
            flat[func] = flat.get(func, 0) + 1
    flat = {f: n for f, n in flat.items() if not ("k_mt_walk_w" in f and n <= WALK_W_FLAT_MAX)}
    name = None
    for line in lines:  # (the metadata lists .name, then .vgpr_spill_count, per kernel)
        m = re.match(r"^\s*\.name:\s*(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"^\s*\.vgpr_spill_count:\s*(\d+)", line)
        # the plain variants without --store_mutations: k_main<false, true, *> and every k_main_g
        if m and name and ("k_main_g" in name or "6k_mainILb0ELb1E" in name) and int(m.group(1)) > HOT_SPILL_MAX:
            scratch[name] = int(m.group(1))
    for f, n in sorted(flat.items()):
        print("%s: %d flat_* memory operation(s) in a worker-set kernel" % (f, n))
    for f, n in sorted(scratch.items()):
        print("%s: %d spilled vector register(s) in a hot kernel" % (f, n))
    print("%d worker-set kernel(s) with flat accesses, %d hot kernel(s) with spills" % (len(flat), len(scratch)))
    return 1 if hits or flat or scratch else 0


if __name__ == "__main__":
    sys.exit(main())
