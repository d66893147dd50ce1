from .seg_criterion import SegCriterion  # noqa: F401
