from .infer import EAST

__all__ = ["EAST"]
