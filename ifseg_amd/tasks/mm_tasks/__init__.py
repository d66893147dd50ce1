from .segmentation import SegmentationTask  # noqa: F401
