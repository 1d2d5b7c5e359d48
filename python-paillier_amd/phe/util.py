"""Cold-path integer helpers with the names the reference's phe.util exports.

These are NOT the hot path: in this package every per-ciphertext powmod/mulmod/invert of the
reference (phe/util.py:38-103) is executed by the HIP kernels through phe._engine.  What remains here
is once-per-key work (prime generation, modular inverse of key constants, integer square root) and
the base64 codecs of the serialisation docs; they run on CPython integers and are kept only so that
code written against `phe.util` keeps importing.
"""
import math
import secrets
from base64 import urlsafe_b64decode, urlsafe_b64encode
from binascii import hexlify, unhexlify

# the reference advertises which bigint engine it found; this build uses neither gmpy2 nor pycrypto
HAVE_GMP = False
HAVE_CRYPTO = False


def powmod(a, b, c):
    """a**b % c for scalar, cold uses (key setup, primality testing)."""
    if a == 1:
        return 1
    return pow(a, b, c)


def mulmod(a, b, c):
    return a * b % c


def extended_euclidean_algorithm(a, b):
    """(g, s, t) with g = gcd(a, b) = s*a + t*b."""
    old_r, r = a, b
    old_s, s = 1, 0
    old_t, t = 0, 1
    while r:
        quot = old_r // r
        old_r, r = r, old_r - quot * r
        old_s, s = s, old_s - quot * s
        old_t, t = t, old_t - quot * t
    return old_r, old_s, old_t


def invert(a, b):
    """Multiplicative inverse of a modulo b; ZeroDivisionError if none exists."""
    g, s, _ = extended_euclidean_algorithm(a, b)
    if g != 1:
        raise ZeroDivisionError('invert() no inverse exists')
    return s % b


def isqrt(N):
    return math.isqrt(N)


def improved_i_sqrt(n):
    return math.isqrt(n)


def _small_primes(limit):
    sieve = bytearray([1]) * (limit + 1)
    sieve[0:2] = b"\x00\x00"
    for i in range(2, int(limit ** 0.5) + 1):
        if sieve[i]:
            sieve[i * i::i] = bytearray(len(sieve[i * i::i]))
    return [i for i, flag in enumerate(sieve) if flag]


first_primes = _small_primes(2000)


def miller_rabin(n, k):
    """k rounds of Miller-Rabin with random bases; False means composite."""
    if n < 4:
        return n in (2, 3)
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for _ in range(k):
        a = 2 + secrets.randbelow(n - 3)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def is_prime(n, mr_rounds=25):
    if n < 2:
        return False
    for p in first_primes:
        if n == p:
            return True
        if n % p == 0:
            return False
    return miller_rabin(n, mr_rounds)


def getprimeover(N):
    """A random N-bit prime from the OS entropy source."""
    while True:
        cand = secrets.randbits(N) | (1 << (N - 1)) | 1
        if N < 2:
            cand = 2 + secrets.randbelow(2)
        while cand.bit_length() == N or N < 2:
            if is_prime(cand):
                return cand
            cand += 2
            if N < 2:
                break


def base64url_encode(payload):
    if not isinstance(payload, bytes):
        payload = payload.encode('utf-8')
    return urlsafe_b64encode(payload).decode('utf-8').rstrip('=')


def base64url_decode(payload):
    payload += '=' * (-len(payload) % 4)
    return urlsafe_b64decode(payload.encode('utf-8'))


def base64_to_int(source):
    return int(hexlify(base64url_decode(source)), 16)


def int_to_base64(source):
    assert source != 0
    digits = hex(source)[2:].rstrip('L')
    if len(digits) % 2:
        digits = '0' + digits
    return base64url_encode(unhexlify(digits))
