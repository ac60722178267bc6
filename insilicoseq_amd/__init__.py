"""insilicoseq_amd -- MI355X-native read-generation engine behind InSilicoSeq's
``worker_iterator`` / ``ErrorModel`` boundary (include/iss_mi355x.h)."""
from .model import DenseModel, KDErrorModel, ModelError  # noqa: F401

__all__ = ["DenseModel", "KDErrorModel", "ModelError"]
