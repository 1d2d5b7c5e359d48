"""`phe.paillier` namespace of the reference (phe/paillier.py), re-exported from this package's modules."""
from .codec import EncodedNumber
from .keys import (DEFAULT_KEYSIZE, PaillierPrivateKey, PaillierPrivateKeyring, PaillierPublicKey,
                   generate_paillier_keypair)
from .ciphertext import EncryptedNumber, EncryptedVector
from .util import getprimeover, invert, isqrt, mulmod, powmod

__all__ = ["DEFAULT_KEYSIZE", "EncodedNumber", "EncryptedNumber", "EncryptedVector", "PaillierPrivateKey",
           "PaillierPrivateKeyring", "PaillierPublicKey", "generate_paillier_keypair"]
