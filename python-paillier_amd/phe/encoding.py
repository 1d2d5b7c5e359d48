"""`phe.encoding` namespace of the reference (phe/encoding.py); the codec lives in phe/codec.py."""
from .codec import EncodedNumber

__all__ = ["EncodedNumber"]
