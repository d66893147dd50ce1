from .segofa import (SegOFAModel, segofa_base_architecture, segofa_large_architecture,  # noqa: F401
                     segofa_huge_architecture, segofa_medium_architecture, segofa_tiny_architecture)
from .config import SegOFAConfig, make_config  # noqa: F401
