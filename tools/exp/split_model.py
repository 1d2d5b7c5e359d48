"""Integer model of the split-modulus ("n-adic") arithmetic used by csrc/split_core.h.

Elements of Z/n^2 are kept as pairs (X0, X1) with  x = X0*beta + X1*n (mod n^2),  beta = R^-1 mod n^2,
R = 2^(29h) >= 16 n.  Every product then costs half-width (mod n) Montgomery passes only:

    square:   (u, m) = MQ(X0, X0);  X1' = MAC2(X0, 2*X1, m, gamma)
    multiply: (u, m) = MQ(X0, Y0);  X1' = MAC2(X0, Y1, m, gamma) + MONT(X1, Y0)

MQ is a Montgomery product that also returns its quotient m (X0*Y0 + m*n = u*R exactly), gamma = -R^-1 mod n.
This file checks the algebra and the lazy-reduction bounds with Python integers (no limbs)."""
import random


def egcd_inv(a, n):
    return pow(a, -1, n)


class Split:
    def __init__(self, n, h):
        self.n, self.h = n, h
        self.n2 = n * n
        self.R = 1 << (29 * h)
        assert self.R >= 16 * n
        self.nprime = (-egcd_inv(n, self.R)) % self.R
        self.rho = egcd_inv(self.R, n)
        self.gamma = (n - self.rho) % n
        self.beta = egcd_inv(self.R, self.n2)
        self.r1 = self.R % n
        self.r2 = self.R * self.R % n
        self.E = self.rep(1, 1)

    # constants.  K*R^j = z0 + z1*n (mod n^2) gives the pair (z0, z1*rho):
    #   j = 1: rep_1(K) = (X0, X1)            (X1*R = z1 mod n)
    #   j = 2: (D0, D1') with D1' = D1*R      (D1*R^2 = z1 mod n), the multiplier-side constant of conv()
    def rep(self, K, j):
        Z = K * pow(self.R, j, self.n2) % self.n2
        return Z % self.n, (Z // self.n) * self.rho % self.n

    def value(self, X):
        return (X[0] * self.beta + X[1] * self.n) % self.n2

    def MQ(self, a, b):
        t = a * b
        m = t * self.nprime % self.R
        u, rem = divmod(t + m * self.n, self.R)
        assert rem == 0
        return u, m

    def MAC2(self, a, b, m, g):
        v = a * b + m * g
        m2 = v * self.nprime % self.R
        return (v + m2 * self.n) // self.R

    def MONT(self, a, b):
        return self.MAC2(a, b, 0, 0)

    def square(self, X):
        u, m = self.MQ(X[0], X[0])
        return u, self.MAC2(X[0], 2 * X[1], m, self.gamma)

    def mul(self, X, Y):
        u, m = self.MQ(X[0], Y[0])
        return u, self.MAC2(X[0], Y[1], m, self.gamma) + self.MONT(X[1], Y[0])

    def conv(self, chunks):
        """integer sum(chunks[j] * R^j) -> rep_1"""
        X0 = X1 = 0
        for j, x in enumerate(chunks):
            D0, D1 = self.rep(pow(self.R, j, self.n2), 2)
            u, m = self.MQ(x, D0)
            X0 += u
            X1 += self.MAC2(x, D1, m, self.gamma)
        X = (X0, X1)
        if len(chunks) > 1:
            X = self.mul(X, self.E)
        return X

    def exit(self, X, mp=0):
        """plain canonical value of x * (1 + n*mp) mod n^2"""
        n = self.n
        u, m = self.MQ(X[0], 1)
        t = X[1] + self.MONT(m, n - 1)
        if mp:
            t += self.MONT(self.MONT(mp, self.r2), u)
        t = self.MONT(t, self.r1)
        t %= n  # canonicalize
        v = u + n * t
        assert v < 2 * self.n2
        return v - self.n2 if v >= self.n2 else v


def check(bits, h, seed):
    rnd = random.Random(seed)
    while True:
        n = rnd.getrandbits(bits) | (1 << (bits - 1)) | 1
        if n % 3 and n % 5:
            break
    S = Split(n, h)
    n2 = n * n
    # rep / value round trip
    for _ in range(5):
        x = rnd.randrange(n2)
        X = S.conv([x % S.R, x // S.R])
        assert S.value(X) == x, "conv"
        assert X[0] < 2 * n and X[1] < 5 * n
        assert S.exit(X) == x, "exit"
    # products with worst-case lazy operands
    for _ in range(200):
        X = (rnd.randrange(2 * n), rnd.randrange(5 * n))
        Y = (rnd.randrange(2 * n), rnd.randrange(5 * n))
        if rnd.random() < 0.2:
            X = (2 * n - 1, 5 * n - 1)
        if rnd.random() < 0.2:
            Y = (2 * n - 1, 5 * n - 1)
        Z = S.mul(X, Y)
        assert S.value(Z) == S.value(X) * S.value(Y) % n2
        assert Z[0] < 2 * n and Z[1] < 5 * n, (Z[0] / n, Z[1] / n)
        Q = S.square(X)
        assert S.value(Q) == pow(S.value(X), 2, n2)
        assert Q[0] < 2 * n and Q[1] < 5 * n
    # a whole encryption
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
