"""``runtime.validation`` of the reference (runtime/validation.py:13-35)."""
from . import validate_input_paths, validate_parameters

__all__ = ["validate_input_paths", "validate_parameters"]
