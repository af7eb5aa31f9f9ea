"""``runtime.device`` of the reference (runtime/device.py:12-40): the same two functions, defined in the package."""
from . import setup_device, setup_random_seed

__all__ = ["setup_device", "setup_random_seed"]
