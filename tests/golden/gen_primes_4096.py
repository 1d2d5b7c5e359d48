#!/usr/bin/env python3
"""Two 2048-bit primes for the 4096-bit emulator test (tests/test_emu_core.py::test_wave_pair_sweeps_with_three_limbs_per_lane):
drawn from a fixed seed, Fermat tests to four bases.  Only p and q are stored; the test derives the key from them as
phe/paillier.py:224-235 does.     python tests/golden/gen_primes_4096.py > tests/golden/paillier_4096_primes.json"""
import json
import random

rng = random.Random(40961)


def prime(bits):
    while True:
        cand = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        if all(pow(a, cand - 1, cand) == 1 for a in (2, 3, 5, 7)):
            return cand


while True:
    p, q = sorted((prime(2048), prime(2048)))
    if p != q and (p * q).bit_length() == 4096:
        break
print(json.dumps({"p": "%x" % p, "q": "%x" % q, "seed": 40961}))
