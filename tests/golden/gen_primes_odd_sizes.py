#!/usr/bin/env python3
"""Prime pairs for key sizes that are not multiples of 1024 bits (tests of the wave-pair sweeps' last-trip variants and of the
ragged limb counts: tests/test_emu_core.py, tests/test_gpu_ladder.py).  Drawn from a fixed seed per size, Fermat tests to four
bases; only p and q are stored, the tests derive the key as phe/paillier.py:224-235 does.
    python tests/golden/gen_primes_odd_sizes.py > tests/golden/paillier_odd_sizes_primes.json"""
import json
import random


def pair(key_bits):
    rng = random.Random(key_bits * 7 + 1)

    def prime(bits):
        while True:
            cand = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
            if all(pow(a, cand - 1, cand) == 1 for a in (2, 3, 5, 7)):
                return cand
    while True:
        p, q = sorted((prime(key_bits // 2), prime(key_bits // 2)))
        if p != q and (p * q).bit_length() == key_bits:
            return {"p": "%x" % p, "q": "%x" % q}


print(json.dumps({str(b): pair(b) for b in (2100, 2240, 1600)}))
