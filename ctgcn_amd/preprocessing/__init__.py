from .structure_generation import StructureInfoGenerator  # noqa: F401
