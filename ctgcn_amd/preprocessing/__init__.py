from .structure_generation import StructureInfoGenerator  # noqa: F401
from .walk_generation import WalkGenerator  # noqa: F401
