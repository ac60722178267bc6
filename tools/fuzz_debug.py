#!/usr/bin/env python3
"""Re-run one configuration of the randomized differential test and show where the device's reads leave the oracle's.
Usage: ISS_FUZZ_OFFSET=18000 python tools/fuzz_debug.py 45 [env switches are honoured, e.g. ISS_LIGHT_INDELS=2]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import _fuzz_config  # noqa: E402
from insilicoseq_amd.engine import ReadEngine  # noqa: E402
from oracle import oracle as O  # noqa: E402

k = int(sys.argv[1])
c = _fuzz_config(k)
dense, genome, n = c["dense"], c["genome"], c["n"]
print("RL", dense.read_length, "n", n, "first", c["first"], "seq_type", c["seq_type"], "gc_bias", c["gc_bias"], "frag", c["frag"], "mut", c["mut"],
      "genome len", len(genome), "ins max", float(np.max(dense.ins)), "del max", float(np.max(dense.dele)))
orc = O.Oracle(dense)
fl, fsd = c["frag"] if c["frag"] else (None, None)
kw = dict(sequence_type=c["seq_type"], gc_bias=c["gc_bias"])
with ReadEngine(0) as eng:
    eng.load_model(dense)
    gid = eng.add_genome(genome)
    exp = orc.simulate(O.Rng().seed_philox(c["seed"]), genome, n, first_ordinal=c["first"], fragment_length=fl, fragment_sd=fsd,
                       store_mutations=c["mut"], want_coords=True, **kw)
    eng.set_fragment(fl, fsd)
    eng.mutations_reserve(max(64 * n * dense.read_length, 1 << 21) if c["mut"] else 0)
    eng.generate(gid, n, first_ordinal=c["first"], seed=c["seed"], **kw)
    eng.synchronize()
    got = eng.download(0, n)
    print("stats", eng.stats_read())
    for key in ("r1_base", "r1_qual", "r2_base", "r2_qual"):
        bad = np.argwhere(got[key] != exp[key])
        print(key, "mismatching cells:", len(bad), "rows:", sorted(set(int(b[0]) for b in bad))[:20])
        for r in sorted(set(int(b[0]) for b in bad))[:4]:
            print("  row", r, "coords", exp["coords"][r])
            print("   got", bytes(got[key][r]).decode("latin1"))
            print("   exp", bytes(exp[key][r]).decode("latin1"))
    exp2 = orc.simulate(O.Rng().seed_philox(c["seed"]), genome, n, first_ordinal=c["first"], fragment_length=fl, fragment_sd=fsd,
                        store_mutations=True, want_coords=True, **kw)
    rows = exp2["mutations"]
    badrows = set()
    for key in ("r1_base", "r2_base"):
        badrows |= set(int(b[0]) for b in np.argwhere(got[key] != exp[key]))
    for r in sorted(badrows)[:4]:
        sel = rows["pair"] == r
        print("oracle rows of pair", r, [(int(m), int(t), int(p), chr(a), chr(b)) for m, t, p, a, b in zip(rows["mate"][sel], rows["type"][sel], rows["position"][sel], rows["ref"][sel], rows["alt"][sel])])
    print("genome", bytes(genome).decode("latin1"))
