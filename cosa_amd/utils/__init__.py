"""Host-side mirror of the reference's `utils` package for the hot path (seg_helper, rrm_utils, torch_helper)."""
from . import seg_helper, torch_helper  # noqa: F401
