#!/usr/bin/env python3
"""Key generation: phe.primes.generate_paillier_keypairs on the GPU against the reference's
gmpy2 route (random start + gmpy2.next_prime = libgmp mpz_nextprime, timed here through oracle/ on one core) for
the same number of primes.  Usage (GPU box): python tools/bench_keygen.py [n_length] [count].  Prints one JSON object."""
import json
import os
import secrets
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-paillier_amd")):
    sys.path.insert(0, p)

from phe import primes  # noqa: E402
from oracle.paillier_oracle import COracle  # noqa: E402


def main():
    n_length = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    orc = COracle()
    primes.next_primes([secrets.randbits(256) | (1 << 255)])               # library load / first launch out of the timings
    res = {"n_length": n_length, "prime_bits": n_length // 2}
    for k in (1, count):
        t0 = time.perf_counter()
        pairs = primes.generate_paillier_keypairs(k, n_length)
        dt = time.perf_counter() - t0
        ok = all(pub.n.bit_length() == n_length and orc.is_probable_prime(priv.p) and orc.is_probable_prime(priv.q)
                 for pub, priv in pairs)
        res["gpu_%d_keypairs" % k] = {"seconds": dt, "keypairs_per_s": k / dt, "primes_confirmed_by_gmp": ok}
    starts = [secrets.randbits(n_length // 2) | (1 << (n_length // 2 - 1)) for _ in range(2 * count)]
    t0 = time.perf_counter()
    want = [orc.next_prime(s) for s in starts]
    dt = time.perf_counter() - t0
    res["cpu_gmp_nextprime_one_core"] = {"primes": len(starts), "seconds": dt, "keypairs_per_s": count / dt}
    t0 = time.perf_counter()
    got = primes.next_primes(starts)
    res["gpu_next_primes_same_starts"] = {"seconds": time.perf_counter() - t0, "equal_to_gmp": got == want}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
