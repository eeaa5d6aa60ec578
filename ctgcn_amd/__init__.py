"""ctgcn_amd — MI355X-native CTGCN embedding engine (hot path only).

The sparse aggregation of CoreDiffusion over nested k-core adjacency matrices and the k-core peel that
produces them run as hand-written HIP kernels for gfx950 behind a C ABI (include/ctgcn_hip.h,
ctgcn_amd/csrc/libctgcn_hip.so); this package is the host-side mirror of the reference's interface for that
path: layers.CoreDiffusion, models.{CDN,CGCN,CTGCN}, helper.DataLoader.get_core_adj_list,
preprocessing.StructureInfoGenerator.  There is no CPU fallback.
"""
from .core_adj import CoreAdj  # noqa: F401
from .layers import CoreDiffusion, MLP  # noqa: F401
from .models import CDN, CGCN, CTGCN  # noqa: F401
from .helper import DataLoader  # noqa: F401

__all__ = ["CoreAdj", "CoreDiffusion", "MLP", "CDN", "CGCN", "CTGCN", "DataLoader"]
