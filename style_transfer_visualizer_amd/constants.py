"""Implementation constants (not user-configurable), as in the reference's constants.py."""
from __future__ import annotations

# ImageNet statistics used by torchvision's pretrained VGG (reference constants.py:11-12)
IMAGENET_MEAN = [0.485, 0.456, 0.406]
IMAGENET_STD = [0.229, 0.224, 0.225]

# raw F.F^T entries are clamped to this before normalisation (reference constants.py:15)
GRAM_MATRIX_CLAMP_MAX = 5e5

MIN_DIMENSION = 64
MAX_DIMENSION = 3000
COLOR_MODE_RGB = "RGB"
DENORM_VIEW_SHAPE = (1, 3, 1, 1)

VIDEO_QUALITY_MIN = 1
VIDEO_QUALITY_MAX = 10

# above this many steps the runner suggests CSV logging (reference constants.py:40)
CSV_LOGGING_RECOMMENDED_STEPS = 2000

# Presentation-side constants of the reference's constants.py (video encoding, comparison-grid colours, the
# resolution its image-loading tests use): nothing on the hot path reads them; kept so that code written against
# `style_transfer_visualizer.constants` finds the same names here.
VIDEO_CODEC = "libx264"
ENCODING_BLOCK_SIZE = 16
COLOR_BLACK = (0, 0, 0)
COLOR_WHITE = (255, 255, 255)
COLOR_BEIGE = (240, 236, 226)
COLOR_GREY = (60, 67, 74)
RESOLUTION_FULL_HD = (1920, 1080)
