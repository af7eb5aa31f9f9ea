"""``runtime.output`` of the reference (runtime/output.py:21-120)."""
from . import save_outputs, setup_output_directory, stylized_image_path_from_names, stylized_image_path_from_paths

__all__ = ["save_outputs", "setup_output_directory", "stylized_image_path_from_names", "stylized_image_path_from_paths"]
