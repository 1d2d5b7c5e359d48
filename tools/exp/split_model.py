"""Integer model of the split-modulus arithmetic of csrc/split_core.h (checks the algebra and the lazy bounds).

An element x of Z/n^2 is kept in half-width Montgomery form, x~ = x*R mod n^2 with R = 2^(29h) >= 16 n, written in
"n-adic digits with a minus sign":   x~ = X0 - n*X1  (mod n^2),   X0 < 2n, X1 < 2n (lazily reduced).

With MQ a Montgomery product modulo n that also returns its quotient m (X0*Y0 + m*n = u*R exactly):

    x*y :  (u, m) = MQ(X0, Y0);   Z0 = u;   Z1 = (m + X0*Y1 + X1*Y0) * R^-1 mod n
    x^2 :  (u, m) = MQ(X0, X0);   Z0 = u;   Z1 = (m + X0*(2*X1))     * R^-1 mod n

because x~ y~ = X0 Y0 - n (X0 Y1 + X1 Y0) = u R - n (m + X0 Y1 + X1 Y0) (mod n^2).  A squaring is two half-width
Montgomery passes (one quarter of the multiply-adds of a full-width pass each), a product is one pass plus one
pass with two products per digit."""
import random


class Split:
    def __init__(self, n, h):
        self.n, self.h = n, h
        self.n2 = n * n
        self.R = 1 << (29 * h)
        assert self.R >= 16 * n
        self.nprime = (-pow(n, -1, self.R)) % self.R
        self.r1 = self.R % n
        self.E = self.pair_of(self.R % self.n2)  # the pair of 1

    def pair_of(self, xt):
        """constant pair for Montgomery form xt = z0 + z1*n: (z0, -z1 mod n)"""
        z0, z1 = xt % self.n, xt // self.n
        return z0, (self.n - z1) % self.n

    def value(self, X):
        return (X[0] - self.n * X[1]) * pow(self.R, -1, self.n2) % self.n2

    def MQ(self, a, b):
        t = a * b
        m = t * self.nprime % self.R
        u, rem = divmod(t + m * self.n, self.R)
        assert rem == 0
        return u, m

    def RED(self, v):
        m2 = v * self.nprime % self.R
        return (v + m2 * self.n) // self.R

    def square(self, X):
        u, m = self.MQ(X[0], X[0])
        return u, self.RED(m + X[0] * (2 * X[1]))

    def mul(self, X, Y):
        u, m = self.MQ(X[0], Y[0])
        return u, self.RED(m + X[0] * Y[1] + X[1] * Y[0])

    def conv(self, chunks):
        """integer sum(chunks[j] * R^j), chunks < R  ->  pair"""
        X0 = X1 = 0
        for j, x in enumerate(chunks):
            D = self.pair_of(pow(self.R, j + 2, self.n2))
            u, m = self.MQ(x, D[0])
            X0 += u
            X1 += self.RED(m + x * D[1])
        X = (X0, X1)
        if len(chunks) > 1:
            X = self.mul(X, self.E)
        return X

    def exit(self, X, mp=0):
        """canonical value of x * (1 + n*mp) mod n^2"""
        n = self.n
        u, m = self.MQ(X[0], 1)
        t = self.RED(X[1] * (n - 1) + m * (n - 1))
        if mp:
            t += self.RED(mp * X[0])
        t = self.RED(t * self.r1) % n
        v = u + n * t
        assert v < 2 * self.n2
        return v - self.n2 if v >= self.n2 else v


def check(bits, h, seed):
    rnd = random.Random(seed)
    n = rnd.getrandbits(bits) | (1 << (bits - 1)) | 1
    S = Split(n, h)
    n2 = n * n
    for _ in range(5):
        x = rnd.randrange(n2)
        chunks = []
        while x or not chunks:
            chunks.append(x % S.R)
            x //= S.R
        x = sum(c * S.R ** j for j, c in enumerate(chunks))
        for pad in (0, 2):
            X = S.conv(chunks + [0] * pad)
            assert S.value(X) == x, "conv"
            assert X[0] < 2 * n and X[1] < 3 * n
            assert S.exit(X) == x, "exit"
    for _ in range(200):
        X = (rnd.randrange(2 * n), rnd.randrange(3 * n))
        Y = (rnd.randrange(2 * n), rnd.randrange(3 * n))
        if rnd.random() < 0.2:
            X = (2 * n - 1, 3 * n - 1)
        if rnd.random() < 0.2:
            Y = (2 * n - 1, 3 * n - 1)
        Z = S.mul(X, Y)
        assert S.value(Z) == S.value(X) * S.value(Y) % n2
        assert Z[0] < 2 * n and Z[1] < 2 * n, (Z[0] / n, Z[1] / n)
        Q = S.square(X)
        assert S.value(Q) == pow(S.value(X), 2, n2)
        assert Q[0] < 2 * n and Q[1] < 2 * n
    r = rnd.randrange(1, n)
    mp = rnd.randrange(n)
    X = S.conv([r])
    acc = X
    for bit in bin(n)[3:]:
        acc = S.square(acc)
        if bit == "1":
            acc = S.mul(acc, X)
    assert S.exit(acc, mp) == (1 + n * mp) * pow(r, n, n2) % n2
    assert S.exit(acc) == pow(r, n, n2)
    return True


if __name__ == "__main__":
    for bits, h in ((64, 3), (128, 5), (256, 9), (1024, 36), (2048, 72)):
        for seed in range(3):
            check(bits, h, seed)
        print("ok", bits, h)
