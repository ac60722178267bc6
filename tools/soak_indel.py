# Large bit-exact comparisons of the indel-heavy paths with the CPU oracle on the GPU box: python tools/soak_indel.py
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import dense_model, random_genome, mixed_genome
from insilicoseq_amd.engine import ReadEngine
import test_gpu_parity as T
eng = ReadEngine(0)
for model, indel, n, L, mixed in [("novaseq", (0.001, 0.003), 300000, 3000000, False), ("hiseq", (0.002, 0.002), 150000, 500000, True),
                                  ("miseq", (0.001, 0.003), 60000, 400000, False), ("miseq-36", (0.01, 0.02), 100000, 20000, False),
                                  ("nextseq", (0.0005, 0.001), 60000, 900000, True), ("novaseq", (0.02, 0.05), 50000, 100000, False)]:
    t = time.time()
    g = (mixed_genome if mixed else random_genome)(77, L)
    _, stats = T._compare(eng, dense_model(model, indel), g, n, 1234, 2**34 + 9, "metagenomics", False)
    print(model, indel, n, "ok", stats, "%.1f s" % (time.time() - t), flush=True)
