"""Key objects of the drop-in: same constructor signatures, attributes, equality/hash and error
behaviour as the reference (phe/paillier.py:37-68 keypair generation, :71-194 PaillierPublicKey,
:197-380 PaillierPrivateKey, :383-439 PaillierPrivateKeyring), with the arithmetic delegated to a
per-key GPU Engine and batched siblings (`*_batch`) added next to every scalar method.

Key generation and the once-per-key constants (p_inverse, hp, hq) are cold, host-side integer work
(out of the hot path); everything per-ciphertext runs on the GPU.
"""
import random
import threading

try:
    from collections.abc import Mapping
except ImportError:  # pragma: no cover
    Mapping = dict

import numpy as np

from . import util
from ._engine import Engine, random_lt_n, random_lt_n_limbs
from .codec import EncodedNumber
from .sharding import shard_bounds

DEFAULT_KEYSIZE = 3072
# scalar encrypt() / obfuscate() refill the obfuscator pool with this many r^n per launch when it runs dry (0 = one
# exponentiation per call, as the reference does): 4096 rows still fit the latency geometry of one launch
SCALAR_POOL_REFILL = 4096


def _gpu_prime_search_available():
    """the batched Miller-Rabin launch needs the real library and a device (the test emulator has neither)"""
    try:
        from . import _native
        return hasattr(_native.Context, "encrypt_dev") and _native.device_count() > 0
    except Exception:
        return False


def generate_paillier_keypair(private_keyring=None, n_length=DEFAULT_KEYSIZE):
    """New (PaillierPublicKey, PaillierPrivateKey) with an n of exactly n_length bits (phe/paillier.py:37-68)."""
    half = n_length // 2
    draw_pair = None
    if half >= 256 and _gpu_prime_search_available():
        # both primes of a try from ONE pair of launches (phe/primes.py: the smallest probable prime above a random
        # start with the top bit set — the gmpy2 branch of getprimeover, phe/util.py:113-116)
        from . import primes
        draw_pair = lambda: primes.getprimeover_batch(half, 2)
    while True:
        if draw_pair is not None:
            p, q = draw_pair()
            if p == q:
                continue
        else:
            p = util.getprimeover(half)
            q = p
            while q == p:
                q = util.getprimeover(half)
        n = p * q
        if n.bit_length() == n_length:
            break
    public_key = PaillierPublicKey(n)
    private_key = PaillierPrivateKey(public_key, p, q)
    if private_keyring is not None:
        private_keyring.add(private_key)
    return public_key, private_key


class PaillierPublicKey(object):
    def __init__(self, n):
        self.g = n + 1
        self.n = n
        self.nsquare = n * n
        self.max_int = n // 3 - 1
        self._engine = None
        self._engine_lock = threading.Lock()
        self._fleet = None
        self._fleet_factory = None               # device -> Engine for the further devices of a fleet (a key pair installs its own)

    # one GPU context per key, created on first use (a private key shares its context with its public key); the creation
    # is guarded so that two threads never build two contexts — and two obfuscator pools — for one key
    def _get_engine(self):
        eng = self._engine
        if eng is None:
            with self._engine_lock:
                if self._engine is None:
                    self._engine = Engine(self.n)
                eng = self._engine
        return eng

    def _get_fleet(self):
        """One engine per device of PHE_HIP_DEVICES (phe/fleet.py), the key's ordinary engine first — or None when no
        single-process fan-out is asked for.  A key pair shares one fleet, as it shares one engine."""
        fl = self._fleet
        if fl is not None:
            return fl
        from . import fleet
        devices = fleet.configured_devices()
        if devices is None:
            return None
        primary = self._get_engine()
        with self._engine_lock:
            if self._fleet is None or self._fleet.engine(0) is not self._engine:
                make = self._fleet_factory or (lambda device: Engine(self.n, device=device))
                self._fleet = fleet.Fleet(self._engine or primary, make, devices)
            return self._fleet

    def _engine_for(self, store):
        """the engine that owns a resident array (a vector made by encrypt_batch_sharded lives on ITS device), else the
        key's ordinary engine"""
        primary = self._get_engine()
        ctx = getattr(store, "ctx", None)
        if ctx is None or ctx is primary.ctx:
            return primary
        fl = self._fleet
        if fl is not None:
            eng = fl.engine_of(ctx)
            if eng is not None:
                return eng
        device = getattr(ctx, "device", None)
        if device is None or device == primary.device:
            return primary          # another context of the primary's GPU (e.g. the public engine a key pair's engine replaced)
        if fl is not None:
            eng = fl.engine_on(device)
            if eng is not None:
                return eng
        # never launch one device's kernels on another device's pointers (nothing here enables peer access)
        raise RuntimeError("resident rows live on device %r, which no engine of this key serves (engines on %r): move them "
                           "through the host (vector.to_host())" % (device, [primary.device] + (fl.devices[1:] if fl else [])))

    def __repr__(self):
        return "<PaillierPublicKey {}>".format(hex(hash(self))[2:][:10])

    def __eq__(self, other):
        return self.n == other.n

    def __hash__(self):
        return hash(self.n)

    def __getstate__(self):  # engines hold device handles; keys pickle as plain numbers
        return {"n": self.n}

    def __setstate__(self, state):
        self.__init__(state["n"])

    def get_random_lt_n(self):
        return random.SystemRandom().randrange(1, self.n)

    # ---- scalar API (batch of one on the GPU) ---------------------------------------------------
    def raw_encrypt(self, plaintext, r_value=None):
        if not isinstance(plaintext, int):
            raise TypeError('Expected int type plaintext but got: %s' % type(plaintext))
        r = r_value or self.get_random_lt_n()          # falsy r_value (None, 0) => random, as in the reference
        return self.raw_encrypt_batch([plaintext], [r])[0]

    def encrypt(self, value, precision=None, r_value=None):
        encoding = value if isinstance(value, EncodedNumber) else EncodedNumber.encode(self, value, precision)
        return self.encrypt_encoded(encoding, r_value)

    def encrypt_encoded(self, encoding, r_value):
        from .ciphertext import EncryptedNumber
        if r_value is None:
            # fresh obfuscator: one fused launch (1 + n*m) * r^n, flagged obfuscated like
            # encrypt_encoded + obfuscate() in the reference (phe/paillier.py:189-193)
            eng = self._get_engine()
            if hasattr(eng.ctx, "encrypt_dev"):
                # an obfuscator r^n made ahead of time (precompute_obfuscators), used once: r^n * (1 + n m), one small
                # launch.  One scalar encryption is one latency-bound exponentiation on an otherwise idle GPU: when the
                # pool is dry, draw and exponentiate a launch-full of obfuscators in the same ~time and keep the rest
                if not isinstance(encoding.encoding, int):
                    raise TypeError('Expected int type plaintext but got: %s' % type(encoding.encoding))
                plain = eng.plain_limbs([encoding.encoding % self.n])
                limbs = eng.encrypt_from_obfuscators(plain)
                if limbs is None and SCALAR_POOL_REFILL:
                    eng.fill_obfuscator_pool(SCALAR_POOL_REFILL)
                    limbs = eng.encrypt_from_obfuscators(plain)
                if limbs is not None:
                    number = EncryptedNumber(self, eng.to_ints(limbs.to_host())[0], encoding.exponent)
                    number._EncryptedNumber__is_obfuscated = True
                    return number
            c = self.raw_encrypt(encoding.encoding, self.get_random_lt_n())
            number = EncryptedNumber(self, c, encoding.exponent)
            number._EncryptedNumber__is_obfuscated = True
            return number
        c = self.raw_encrypt(encoding.encoding, r_value or 1)   # r_value == 0 => obfuscator 1, not obfuscated
        return EncryptedNumber(self, c, encoding.exponent)

    # ---- offline / online split ------------------------------------------------------------------------
    def precompute_obfuscators(self, count, sharded=False):
        """Make `count` obfuscators r^n mod n^2 (fresh r from the OS CSPRNG) ahead of time and keep them in HBM.
        encrypt_batch without r_values and EncryptedVector.obfuscate() then consume them — one product per element
        instead of a modular exponentiation — and fall back to drawing on the spot when the pool is short.  Every
        obfuscator is used once.  Returns the number available.  With a fleet (PHE_HIP_DEVICES) the pool is cut the way the
        batch that will consume it is cut: `encrypt_batch` of `count` rows by default, `encrypt_batch_sharded` of `count` rows
        (one part per device whatever the size) with sharded=True."""
        fl = self._get_fleet()
        count = int(count)
        bounds = fl.shards(count, min_rows=1 if sharded else None) if fl is not None else []
        if len(bounds) > 1:
            # the devices a batch of `count` rows fans out to fill THEIR pools with the shares that batch will take from them
            # (a pool is never shared across devices; a batch too small to fan out, and every scalar call, is served by the
            # primary engine alone — so a small `count` goes there whole instead of being spread where nothing would use it)
            futures = [fl._pool.submit(fl.engine(k).fill_obfuscator_pool, hi - lo) for k, (lo, hi) in enumerate(bounds)]
            for f in futures:
                f.result()
            return self.obfuscators_available()
        return self._get_engine().fill_obfuscator_pool(count)

    def obfuscators_available(self, per_device=False):
        """obfuscators made ahead of time and not used yet; per_device=True: the list, one entry per engine that exists (an
        obfuscator is only ever used by the device that holds it)"""
        fl = self._fleet
        counts = [eng.obfuscators_available() for eng in (fl.made() if fl is not None else [self._get_engine()])]
        return counts if per_device else sum(counts)

    def discard_obfuscators(self):
        """forget the obfuscators made ahead of time: the next encryptions draw and exponentiate on the spot again"""
        fl = self._fleet
        for eng in (fl.made() if fl is not None else [self._get_engine()]):
            eng.clear_obfuscator_pool()

    # ---- batched API ------------------------------------------------------------------------------
    def raw_encrypt_batch(self, plaintexts, r_values=None):
        """List of ints -> list of int ciphertexts; r_values=None draws fresh obfuscators."""
        plaintexts = list(plaintexts)
        for m in plaintexts:
            if not isinstance(m, int):
                raise TypeError('Expected int type plaintext but got: %s' % type(m))
        if r_values is None:
            r_values = random_lt_n(self.n, len(plaintexts))
        eng = self._get_engine()
        return eng.to_ints(eng.raw_encrypt(plaintexts, list(r_values)))

    def encrypt_batch(self, values, precision=None, r_values=None, device=False):
        """Encode + encrypt a whole sequence / numpy array -> EncryptedVector (one kernel launch).
        device=True keeps the ciphertexts resident in HBM for later homomorphic operations.
        float64 / integer numpy arrays are encoded and drawn without a Python integer per element."""
        from .ciphertext import EncryptedVector
        eng = self._get_engine()
        fresh = r_values is None
        signed = EncodedNumber.encode_signed(values, precision) if eng.n_limbs >= 4 else None
        if signed is not None:
            mag, neg, exps = signed
            m = EncodedNumber.signed_to_limbs(self, mag, neg, eng.n_limbs)
        else:
            m, exps = EncodedNumber.encode_many(self, values, precision)
        count = len(exps)
        limbs = None
        fl = self._get_fleet() if (not device and isinstance(m, np.ndarray)) else None
        if fl is not None and len(fl.shards(count)) > 1:
            # single-process fan-out (phe/fleet.py): contiguous shards, one device each, results into ONE host array
            if fresh:
                def shard(e, lo, hi):
                    if not hasattr(e.ctx, "encrypt_dev"):           # (a backend without resident rows: draw, then one call)
                        return e.raw_encrypt(m[lo:hi], random_lt_n_limbs(self.n, hi - lo, e.n_limbs))
                    got = e.encrypt_from_obfuscators(m[lo:hi])      # the device's own pool first (each obfuscator used once)
                    return got.to_host() if got is not None else e.raw_encrypt_fresh(m[lo:hi], False)
            else:
                r_all = r_values if isinstance(r_values, np.ndarray) else eng.plain_limbs(list(r_values))
                shard = lambda e, lo, hi: e.raw_encrypt(m[lo:hi], r_all[lo:hi])
            return EncryptedVector(self, np.concatenate(fl.run(count, shard)), exps, obfuscated=fresh)
        if fresh and isinstance(m, np.ndarray) and hasattr(eng.ctx, "encrypt_dev"):
            # online part only: (1 + n m) * r^n with r^n from the pool made by precompute_obfuscators (each used once)
            limbs = eng.encrypt_from_obfuscators(m)
            if limbs is None and 0 < count <= SCALAR_POOL_REFILL // 4:
                eng.fill_obfuscator_pool(SCALAR_POOL_REFILL)       # a small batch is as latency-bound as a scalar call
                limbs = eng.encrypt_from_obfuscators(m)
        if limbs is not None:
            if not device:
                limbs = limbs.to_host()
        elif fresh and isinstance(m, np.ndarray) and hasattr(eng.ctx, "encrypt_dev"):
            limbs = eng.raw_encrypt_fresh(m, device)
        else:
            r = random_lt_n_limbs(self.n, count, eng.n_limbs) if fresh else list(r_values)
            limbs = eng.raw_encrypt_dev(m, r) if device else eng.raw_encrypt(m, r)
        return EncryptedVector(self, limbs, exps, obfuscated=fresh)

    def encrypt_batch_sharded(self, values, precision=None):
        """encrypt_batch with the ciphertexts left RESIDENT, one EncryptedVector per device of the fleet (contiguous shards in
        order; a single vector on the ordinary device when no fleet is configured): the per-device list form of the fan-out.
        Each vector is operated on by the engine of its own device; `PaillierPrivateKey.decrypt_batch` takes the list."""
        from .ciphertext import EncryptedVector
        eng = self._get_engine()
        fl = self._get_fleet()
        if fl is None or not hasattr(eng.ctx, "encrypt_dev"):
            return [self.encrypt_batch(values, precision, device=hasattr(eng.ctx, "encrypt_dev"))]
        signed = EncodedNumber.encode_signed(values, precision) if eng.n_limbs >= 4 else None
        if signed is not None:
            mag, neg, exps = signed
            m = EncodedNumber.signed_to_limbs(self, mag, neg, eng.n_limbs)
        else:
            m, exps = EncodedNumber.encode_many(self, values, precision)
            m = m if isinstance(m, np.ndarray) else eng.plain_limbs([v % self.n for v in m])
        exps = np.asarray(exps, dtype=np.int64)

        def shard(e, lo, hi):
            got = e.encrypt_from_obfuscators(m[lo:hi])
            return got if got is not None else e.raw_encrypt_fresh(m[lo:hi], True)
        bounds = fl.shards(len(exps), min_rows=1)
        parts = fl.run(len(exps), shard, min_rows=1)
        return [EncryptedVector(self, part, exps[lo:hi], obfuscated=True) for part, (lo, hi) in zip(parts, bounds)]


class PaillierPrivateKey(object):
    def __init__(self, public_key, p, q):
        if not p * q == public_key.n:
            raise ValueError('given public key does not match the given p and q.')
        if p == q:
            raise ValueError('p and q have to be different')
        self.public_key = public_key
        self.p, self.q = (q, p) if q < p else (p, q)
        self.psquare = self.p * self.p
        self.qsquare = self.q * self.q
        self.p_inverse = util.invert(self.p, self.q)
        if public_key.g == public_key.n + 1:
            # h_function in closed form: g = 1 + n and n^2 = 0 (mod p^2), so g^(p-1) = 1 + (p-1) n = 1 + p (-q mod p)
            # (mod p^2), L(.) = -q mod p and hp = (-q)^-1 mod p — the value phe/paillier.py:356-360 computes with a
            # modular exponentiation (pinned to the reference's numbers by tests/test_api.py and the golden fixtures)
            self.hp = util.invert((-self.q) % self.p, self.p)
            self.hq = util.invert((-self.p) % self.q, self.q)
        else:
            self.hp = self.h_function(self.p, self.psquare)
            self.hq = self.h_function(self.q, self.qsquare)
        self._engine = None
        self._engine_lock = threading.Lock()

    @staticmethod
    def from_totient(public_key, totient):
        p_plus_q = public_key.n - totient + 1
        p_minus_q = util.isqrt(p_plus_q * p_plus_q - public_key.n * 4)
        q = (p_plus_q - p_minus_q) // 2
        p = p_plus_q - q
        if not p * q == public_key.n:
            raise ValueError('given public key and totient do not match.')
        return PaillierPrivateKey(public_key, p, q)

    def _get_engine(self):
        eng = self._engine
        if eng is not None:
            return eng
        with self._engine_lock:
            if self._engine is None:
                fresh = Engine(self.public_key.n, self.p, self.q, self.hp, self.hq, self.p_inverse)
                # The public key of a key pair works through the private key's engine from here on: one GPU context per key
                # pair, and encryptions under it take the key owner's CRT form (Engine.owner_encrypt: same ciphertext bits,
                # about half the work).  Obfuscators made ahead of time stay where they are: the new engine adopts the old
                # engine's pool OBJECT (device pointers are valid across contexts), so a thread that is still inside the old
                # engine and a thread inside the new one take rows under the same lock — nothing is handed out twice.
                pub = self.public_key
                with pub._engine_lock:
                    old = pub._engine
                    if old is not None and old is not fresh:
                        fresh._obf = old._obf
                    pub._engine = fresh
                    # the further devices of a fleet get key-pair engines too (built on first use), each with a pool of its own
                    n, p, q, hp, hq, pinv = pub.n, self.p, self.q, self.hp, self.hq, self.p_inverse
                    pub._fleet_factory = lambda device: Engine(n, p, q, hp, hq, pinv, device=device)
                    old_fleet, pub._fleet = pub._fleet, None
                    if old_fleet is not None:
                        # the public key already fanned out (encrypt_batch_sharded / precompute_obfuscators before the first
                        # private-key call): its resident parts live on devices 1..k and its pools hold obfuscators there.
                        # The key pair's fleet is built NOW, engine for engine on the same devices, each adopting the pool of
                        # the engine it replaces; the old contexts resolve to their successors (Fleet.succeed)
                        from . import fleet
                        pub._fleet = fleet.Fleet(fresh, pub._fleet_factory, old_fleet.devices).succeed(old_fleet)
                self._engine = fresh
            return self._engine

    def _get_fleet(self):
        """the key pair's fleet (PaillierPublicKey._get_fleet): every engine of it holds the private key"""
        self._get_engine()
        return self.public_key._get_fleet()

    def __repr__(self):
        return "<PaillierPrivateKey for {}>".format(repr(self.public_key))

    def __eq__(self, other):
        return self.p == other.p and self.q == other.q

    def __hash__(self):
        return hash((self.p, self.q))

    def __getstate__(self):
        return {"n": self.public_key.n, "p": self.p, "q": self.q}

    def __setstate__(self, state):
        self.__init__(PaillierPublicKey(state["n"]), state["p"], state["q"])

    # once-per-key helpers, host integers (phe/paillier.py:356-364)
    def h_function(self, x, xsquare):
        return util.invert(self.l_function(util.powmod(self.public_key.g, x - 1, xsquare), x), x)

    def l_function(self, x, p):
        return (x - 1) // p

    def crt(self, mp, mq):
        u = (mq - mp) * self.p_inverse % self.q
        return mp + u * self.p

    # ---- scalar API ---------------------------------------------------------------------------------
    def raw_decrypt(self, ciphertext):
        if not isinstance(ciphertext, int):
            raise TypeError('Expected ciphertext to be an int, not: %s' % type(ciphertext))
        return self.raw_decrypt_batch([ciphertext])[0]

    def decrypt(self, encrypted_number):
        return self.decrypt_encoded(encrypted_number).decode()

    def decrypt_encoded(self, encrypted_number, Encoding=None):
        from .ciphertext import EncryptedNumber
        if not isinstance(encrypted_number, EncryptedNumber):
            raise TypeError('Expected encrypted_number to be an EncryptedNumber not: %s' % type(encrypted_number))
        if self.public_key != encrypted_number.public_key:
            raise ValueError('encrypted_number was encrypted against a different key!')
        if Encoding is None:
            Encoding = EncodedNumber
        encoded = self.raw_decrypt(encrypted_number.ciphertext(be_secure=False))
        return Encoding(self.public_key, encoded, encrypted_number.exponent)

    # ---- batched API ----------------------------------------------------------------------------------
    def raw_decrypt_batch(self, ciphertexts):
        ciphertexts = list(ciphertexts)
        for c in ciphertexts:
            if not isinstance(c, int):
                raise TypeError('Expected ciphertext to be an int, not: %s' % type(c))
        eng = self._get_engine()
        # ciphertexts are residues mod n^2; reduce stray larger ints exactly like powmod would
        return eng.to_ints(eng.raw_decrypt([c % self.public_key.nsquare for c in ciphertexts]))

    def decrypt_batch(self, vector, Encoding=None):
        """EncryptedVector (or list of EncryptedNumber) -> list of decoded ints/floats."""
        from .ciphertext import EncryptedVector
        if isinstance(vector, (list, tuple)) and vector and all(isinstance(v, EncryptedVector) for v in vector):
            # the per-device list of encrypt_batch_sharded: every part on the engine of its own device, concurrently
            fl = self._get_fleet()
            if fl is None or len(vector) == 1:
                return [x for v in vector for x in self.decrypt_batch(v, Encoding)]
            import concurrent.futures as cf
            with cf.ThreadPoolExecutor(len(vector)) as pool:
                return [x for part in pool.map(lambda v: self.decrypt_batch(v, Encoding), vector) for x in part]
        if not isinstance(vector, EncryptedVector):
            vector = EncryptedVector.from_numbers(self.public_key, vector)
        if self.public_key != vector.public_key:
            raise ValueError('encrypted_number was encrypted against a different key!')
        eng = self._get_engine()
        plain_decode = Encoding is None or Encoding is EncodedNumber
        fl = self._get_fleet()
        if vector.on_device and fl is not None:
            eng = self.public_key._engine_for(vector._store)       # a resident vector is decrypted where it lives
        if not vector.on_device and fl is not None and len(fl.shards(len(vector))) > 1:
            # single-process fan-out: contiguous shards of the host vector, one device each, decoded in shard order
            limbs, exps = vector.limbs(be_secure=False), vector.exponent_array

            def shard(e, lo, hi):
                plain = e.raw_decrypt(limbs[lo:hi])
                if plain_decode:
                    return EncodedNumber.decode_limbs(self.public_key, plain, exps[lo:hi])
                return Encoding.decode_many(self.public_key, e.to_ints(plain), exps[lo:hi].tolist())
            return [x for part in fl.run(len(vector), shard) for x in part]
        if vector.on_device or hasattr(eng.ctx, "decrypt_dev"):
            # chunks: the download and decoding of one chunk (and, for a host vector, the upload of the next) overlap
            # the kernels; device pointers are valid across contexts of the same GPU
            out, exps = [], vector.exponent_array
            limbs = vector.limbs(be_secure=False)
            chunks = eng.raw_decrypt_dev_chunks(limbs) if vector.on_device else eng.raw_decrypt_host_chunks(limbs)
            for lo, hi, plain in chunks:
                if plain_decode:
                    out += EncodedNumber.decode_limbs(self.public_key, plain, exps[lo:hi])
                else:
                    out += Encoding.decode_many(self.public_key, eng.to_ints(plain), exps[lo:hi].tolist())
            return out
        plain = eng.raw_decrypt(vector.limbs(be_secure=False))
        if plain_decode:
            return EncodedNumber.decode_limbs(self.public_key, plain, vector.exponent_array)
        return Encoding.decode_many(self.public_key, eng.to_ints(plain), vector.exponents)


class PaillierPrivateKeyring(Mapping):
    def __init__(self, private_keys=None):
        private_keys = private_keys or []
        self.__keyring = {k.public_key: k for k in private_keys}

    def __getitem__(self, key):
        return self.__keyring[key]

    def __len__(self):
        return len(self.__keyring)

    def __iter__(self):
        return iter(self.__keyring)

    def __delitem__(self, public_key):
        del self.__keyring[public_key]

    def add(self, private_key):
        if not isinstance(private_key, PaillierPrivateKey):
            raise TypeError("private_key should be of type PaillierPrivateKey, not %s" % type(private_key))
        self.__keyring[private_key.public_key] = private_key

    def decrypt(self, encrypted_number):
        return self.__keyring[encrypted_number.public_key].decrypt(encrypted_number)
