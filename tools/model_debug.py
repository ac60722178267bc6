import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["ISS_DEBUG_MODEL"] = "1"
from helpers import dense_model
from insilicoseq_amd.engine import ReadEngine
for m in ("novaseq", "nextseq", "hiseq", "miseq"):
    print(m, flush=True)
    e = ReadEngine(0); e.load_model(dense_model(m)); e.close()
