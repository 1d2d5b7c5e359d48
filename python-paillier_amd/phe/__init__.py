"""Drop-in `phe` package backed by hand-written HIP kernels for AMD MI355X (gfx950).

Exports the same names as the reference package (phe/__init__.py:1-8 of data61/python-paillier 1.5.0):
EncodedNumber, generate_paillier_keypair, EncryptedNumber, PaillierPrivateKey, PaillierPublicKey,
PaillierPrivateKeyring and the `phe.util` / `phe.paillier` / `phe.encoding` modules — plus the batched
siblings (EncryptedVector, *_batch methods) the GPU path is built around.  Put python-paillier_amd/ on
sys.path instead of the reference and `import phe` resolves here.
"""
__version__ = "1.5.0+mi355x.r1"

from .codec import EncodedNumber
from .keys import generate_paillier_keypair
from .ciphertext import EncryptedNumber, EncryptedVector
from .keys import PaillierPrivateKey, PaillierPublicKey
from .keys import PaillierPrivateKeyring

from . import util
from . import paillier
from . import encoding
