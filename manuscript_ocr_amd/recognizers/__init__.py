from ._trba import TRBA

__all__ = ["TRBA"]
