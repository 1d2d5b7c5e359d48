/*
 * oracle/paillier_oracle.c — CPU restatement of the python-paillier hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under python-paillier_amd/ may import, link
 * or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / timed CPU baseline.
 *
 * What it restates (all paths relative to /root/reference):
 *   phe/util.py:38-50    powmod   -> orc_powmod   (gmpy2.powmod  == mpz_powm)
 *   phe/util.py:53-64    mulmod   -> orc_mulmod   (gmpy2.mul+mod == mpz_mul, mpz_mod)
 *   phe/util.py:85-103   invert   -> orc_invert   (gmpy2.invert  == mpz_invert)
 *   phe/paillier.py:102-139  PaillierPublicKey.raw_encrypt   -> orc_raw_encrypt
 *   phe/paillier.py:603-624  EncryptedNumber.obfuscate       -> orc_obfuscate
 *   phe/paillier.py:328-374  PaillierPrivateKey.raw_decrypt,
 *                            l_function, crt                 -> orc_raw_decrypt
 *   phe/paillier.py:356-360  h_function                      -> orc_h_function
 *   phe/paillier.py:705-719  EncryptedNumber._raw_add        -> orc_raw_add
 *   phe/paillier.py:721-751  EncryptedNumber._raw_mul        -> orc_raw_mul
 *   phe/util.py:106-124  getprimeover (gmpy2 branch :114-116: bit_set + next_prime) -> orc_next_prime
 *                        (gmpy2.next_prime == mpz_nextprime), orc_probab_prime (mpz_probab_prime_p)
 *
 * The arithmetic engine is libgmp (GMP 6.2.1, /usr/lib/x86_64-linux-gnu/libgmp.so.10):
 * the very library gmpy2 (requirements.txt:2, gmpy2>=2.0.4 — not vendored under
 * /root/reference, not installable here) wraps, so this is the "reference gmpy2
 * path" executed without the Python interpreter in between.
 *
 * Pinning: tests/test_oracle.py checks this file against (a) the reference's own
 * known-answer vectors (phe/tests/paillier_test.py:128-149, phe/tests/util_test.py:29-58)
 * and (b) the JSON fixtures in tests/golden/, which tests/golden/gen_golden.py produced by importing
 * the real reference from /root/reference in this container.
 *
 * Data layout at this boundary = the C-ABI's (include/phe_hip.h): numbers are
 * little-endian arrays of uint32 limbs, batches are row-major (B, limbs).
 */
#include <gmp.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_NO_INVERSE 3

/* ---- limb <-> mpz ------------------------------------------------------ */
static void limbs_to_mpz(mpz_t z, const uint32_t *limbs, size_t n) {
    mpz_import(z, n, -1, sizeof(uint32_t), 0, 0, limbs);
}
static void mpz_to_limbs(uint32_t *limbs, size_t n, const mpz_t z) {
    size_t count = 0;
    memset(limbs, 0, n * sizeof(uint32_t));
    if (mpz_sgn(z) != 0) mpz_export(limbs, &count, -1, sizeof(uint32_t), 0, 0, z);
    (void)count;
}

/* ---- phe/util.py primitives ------------------------------------------- */
/* phe/util.py:38-50.  The a==1 early return (:45-46) is value-neutral; the
 * engine-size thresholds (:35-36, :47) only pick the engine, never the value. */
static void orc_powmod_z(mpz_t out, const mpz_t a, const mpz_t b, const mpz_t c) {
    if (mpz_cmp_ui(a, 1) == 0) { mpz_set_ui(out, 1); return; }
    mpz_powm(out, a, b, c);
}
/* phe/util.py:53-64 */
static void orc_mulmod_z(mpz_t out, const mpz_t a, const mpz_t b, const mpz_t c) {
    mpz_mul(out, a, b);
    mpz_mod(out, out, c);
}
/* phe/util.py:85-103: returns 0 on success, ORC_NO_INVERSE when the reference
 * raises ZeroDivisionError('invert() no inverse exists'). */
static int orc_invert_z(mpz_t out, const mpz_t a, const mpz_t b) {
    if (mpz_invert(out, a, b) == 0) return ORC_NO_INVERSE;
    return ORC_OK;
}

/* ---- key material ------------------------------------------------------ */
typedef struct {
    mpz_t n, nsquare, max_int, n_minus_max_int;
} orc_pub;
typedef struct {
    mpz_t p, q, psquare, qsquare, p_inverse, hp, hq, pm1, qm1;
} orc_priv;

/* phe/paillier.py:86-90 */
static void pub_init(orc_pub *k, const uint32_t *n, size_t n_limbs) {
    mpz_inits(k->n, k->nsquare, k->max_int, k->n_minus_max_int, NULL);
    limbs_to_mpz(k->n, n, n_limbs);
    mpz_mul(k->nsquare, k->n, k->n);
    mpz_fdiv_q_ui(k->max_int, k->n, 3);
    mpz_sub_ui(k->max_int, k->max_int, 1);
    mpz_sub(k->n_minus_max_int, k->n, k->max_int);
}
static void pub_clear(orc_pub *k) {
    mpz_clears(k->n, k->nsquare, k->max_int, k->n_minus_max_int, NULL);
}

/* phe/paillier.py:362-364 */
static void l_function(mpz_t out, const mpz_t x, const mpz_t p) {
    mpz_sub_ui(out, x, 1);
    mpz_fdiv_q(out, out, p);
}
/* phe/paillier.py:356-360: invert(L(g^(x-1) mod x^2, x), x) with g = n+1 */
static int h_function(mpz_t out, const mpz_t n, const mpz_t x, const mpz_t xsquare) {
    mpz_t g, e, t;
    mpz_inits(g, e, t, NULL);
    mpz_add_ui(g, n, 1);
    mpz_sub_ui(e, x, 1);
    orc_powmod_z(t, g, e, xsquare);
    l_function(t, t, x);
    int rc = orc_invert_z(out, t, x);
    mpz_clears(g, e, t, NULL);
    return rc;
}
/* phe/paillier.py:217-235 (p<q ordering, psquare, qsquare, p_inverse, hp, hq) */
static int priv_init(orc_priv *k, const orc_pub *pub, const uint32_t *p, const uint32_t *q,
                     size_t pq_limbs) {
    mpz_inits(k->p, k->q, k->psquare, k->qsquare, k->p_inverse, k->hp, k->hq, k->pm1, k->qm1, NULL);
    limbs_to_mpz(k->p, p, pq_limbs);
    limbs_to_mpz(k->q, q, pq_limbs);
    if (mpz_cmp(k->q, k->p) < 0) mpz_swap(k->p, k->q);
    mpz_mul(k->psquare, k->p, k->p);
    mpz_mul(k->qsquare, k->q, k->q);
    int rc = orc_invert_z(k->p_inverse, k->p, k->q);
    if (!rc) rc = h_function(k->hp, pub->n, k->p, k->psquare);
    if (!rc) rc = h_function(k->hq, pub->n, k->q, k->qsquare);
    mpz_sub_ui(k->pm1, k->p, 1);
    mpz_sub_ui(k->qm1, k->q, 1);
    return rc;
}
static void priv_clear(orc_priv *k) {
    mpz_clears(k->p, k->q, k->psquare, k->qsquare, k->p_inverse, k->hp, k->hq, k->pm1, k->qm1, NULL);
}

/* ---- the five hot functions, scalar form ------------------------------ */
/* phe/paillier.py:102-139 with an explicit r (r_value is an input: bit-exact
 * parity needs the obfuscator fixed; the reference draws it from SystemRandom). */
static int raw_encrypt_z(mpz_t c, const orc_pub *k, const mpz_t m, const mpz_t r, mpz_t t0, mpz_t t1) {
    if (mpz_cmp(k->n_minus_max_int, m) <= 0 && mpz_cmp(m, k->n) < 0) {
        /* :125-130 "sneaky shortcut using inverses" */
        mpz_sub(t0, k->n, m);
        mpz_mul(t0, k->n, t0);
        mpz_add_ui(t0, t0, 1);
        mpz_mod(t0, t0, k->nsquare);
        int rc = orc_invert_z(t0, t0, k->nsquare);
        if (rc) return rc;
    } else {
        /* :134 */
        mpz_mul(t0, k->n, m);
        mpz_add_ui(t0, t0, 1);
        mpz_mod(t0, t0, k->nsquare);
    }
    orc_powmod_z(t1, r, k->n, k->nsquare); /* :137 */
    orc_mulmod_z(c, t0, t1, k->nsquare);   /* :139 */
    return ORC_OK;
}
/* phe/paillier.py:603-624 with explicit r */
static void obfuscate_z(mpz_t c_out, const orc_pub *k, const mpz_t c_in, const mpz_t r, mpz_t t0) {
    orc_powmod_z(t0, r, k->n, k->nsquare);
    orc_mulmod_z(c_out, c_in, t0, k->nsquare);
}
/* phe/paillier.py:328-354, :366-374 */
static void raw_decrypt_z(mpz_t m, const orc_priv *k, const mpz_t c, mpz_t mp, mpz_t mq, mpz_t u) {
    orc_powmod_z(mp, c, k->pm1, k->psquare);
    l_function(mp, mp, k->p);
    orc_mulmod_z(mp, mp, k->hp, k->p);
    orc_powmod_z(mq, c, k->qm1, k->qsquare);
    l_function(mq, mq, k->q);
    orc_mulmod_z(mq, mq, k->hq, k->q);
    /* crt :373-374; mq - mp may be negative, mpz_mod returns the non-negative residue
     * exactly like Python % and gmpy2.mod */
    mpz_sub(u, mq, mp);
    orc_mulmod_z(u, u, k->p_inverse, k->q);
    mpz_mul(u, u, k->p);
    mpz_add(m, mp, u);
}
/* phe/paillier.py:705-719 */
static void raw_add_z(mpz_t out, const orc_pub *k, const mpz_t a, const mpz_t b) {
    orc_mulmod_z(out, a, b, k->nsquare);
}
/* phe/paillier.py:721-751; returns 1 for 'Scalar out of bounds', ORC_NO_INVERSE if invert fails */
static int raw_mul_z(mpz_t out, const orc_pub *k, const mpz_t c, const mpz_t s, mpz_t t0, mpz_t t1) {
    if (mpz_sgn(s) < 0 || mpz_cmp(s, k->n) >= 0) return 1;
    if (mpz_cmp(k->n_minus_max_int, s) <= 0) {
        int rc = orc_invert_z(t0, c, k->nsquare); /* :747 */
        if (rc) return rc;
        mpz_sub(t1, k->n, s);                     /* :748 */
        orc_powmod_z(out, t0, t1, k->nsquare);    /* :749 */
    } else {
        orc_powmod_z(out, c, s, k->nsquare);      /* :751 */
    }
    return ORC_OK;
}

/* ---- exported scalar primitives on limb arrays (for tests) ------------- */
int orc_powmod(const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *out, size_t limbs) {
    mpz_t za, zb, zc, zo;
    mpz_inits(za, zb, zc, zo, NULL);
    limbs_to_mpz(za, a, limbs); limbs_to_mpz(zb, b, limbs); limbs_to_mpz(zc, c, limbs);
    orc_powmod_z(zo, za, zb, zc);
    mpz_to_limbs(out, limbs, zo);
    mpz_clears(za, zb, zc, zo, NULL);
    return ORC_OK;
}
int orc_mulmod(const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *out, size_t limbs) {
    mpz_t za, zb, zc, zo;
    mpz_inits(za, zb, zc, zo, NULL);
    limbs_to_mpz(za, a, limbs); limbs_to_mpz(zb, b, limbs); limbs_to_mpz(zc, c, limbs);
    orc_mulmod_z(zo, za, zb, zc);
    mpz_to_limbs(out, limbs, zo);
    mpz_clears(za, zb, zc, zo, NULL);
    return ORC_OK;
}
int orc_invert(const uint32_t *a, const uint32_t *b, uint32_t *out, size_t limbs) {
    mpz_t za, zb, zo;
    mpz_inits(za, zb, zo, NULL);
    limbs_to_mpz(za, a, limbs); limbs_to_mpz(zb, b, limbs);
    int rc = orc_invert_z(zo, za, zb);
    if (!rc) mpz_to_limbs(out, limbs, zo);
    mpz_clears(za, zb, zo, NULL);
    return rc;
}
/* hp, hq, p_inverse as PaillierPrivateKey.__init__ computes them (phe/paillier.py:233-235);
 * p and q are returned ordered p<q (phe/paillier.py:224-229). */
int orc_private_constants(const uint32_t *n, size_t n_limbs, const uint32_t *p, const uint32_t *q,
                          size_t pq_limbs, uint32_t *p_sorted, uint32_t *q_sorted, uint32_t *hp,
                          uint32_t *hq, uint32_t *p_inverse) {
    orc_pub pub; orc_priv priv;
    pub_init(&pub, n, n_limbs);
    int rc = priv_init(&priv, &pub, p, q, pq_limbs);
    if (!rc) {
        mpz_to_limbs(p_sorted, pq_limbs, priv.p);
        mpz_to_limbs(q_sorted, pq_limbs, priv.q);
        mpz_to_limbs(hp, pq_limbs, priv.hp);
        mpz_to_limbs(hq, pq_limbs, priv.hq);
        mpz_to_limbs(p_inverse, pq_limbs, priv.p_inverse);
    }
    priv_clear(&priv); pub_clear(&pub);
    return rc;
}

/* ---- batch entry points (same layout as the C-ABI), threaded ----------- */
typedef enum { OP_ENCRYPT, OP_OBFUSCATE, OP_DECRYPT, OP_ADD, OP_MUL } orc_op;
typedef struct {
    orc_op op;
    const orc_pub *pub;
    const orc_priv *priv;
    size_t n_limbs;            /* limbs of n (s1); ciphertexts have 2*n_limbs */
    size_t scalar_limbs;       /* limbs per scalar row for OP_MUL */
    const uint32_t *in0, *in1; /* op-dependent */
    uint32_t *out;
    size_t begin, end;
    int status;
    size_t bad_index;
} orc_job;

static void *orc_worker(void *arg) {
    orc_job *j = (orc_job *)arg;
    const size_t s1 = j->n_limbs, s2 = 2 * j->n_limbs;
    mpz_t a, b, o, t0, t1, t2;
    mpz_inits(a, b, o, t0, t1, t2, NULL);
    for (size_t i = j->begin; i < j->end && j->status == ORC_OK; ++i) {
        int rc = ORC_OK;
        switch (j->op) {
        case OP_ENCRYPT: /* in0 = m (B,s1), in1 = r (B,s1), out = c (B,s2) */
            limbs_to_mpz(a, j->in0 + i * s1, s1);
            limbs_to_mpz(b, j->in1 + i * s1, s1);
            rc = raw_encrypt_z(o, j->pub, a, b, t0, t1);
            if (!rc) mpz_to_limbs(j->out + i * s2, s2, o);
            break;
        case OP_OBFUSCATE: /* in0 = c (B,s2), in1 = r (B,s1), out = c' (B,s2) */
            limbs_to_mpz(a, j->in0 + i * s2, s2);
            limbs_to_mpz(b, j->in1 + i * s1, s1);
            obfuscate_z(o, j->pub, a, b, t0);
            mpz_to_limbs(j->out + i * s2, s2, o);
            break;
        case OP_DECRYPT: /* in0 = c (B,s2), out = m (B,s1) */
            limbs_to_mpz(a, j->in0 + i * s2, s2);
            raw_decrypt_z(o, j->priv, a, t0, t1, t2);
            mpz_to_limbs(j->out + i * s1, s1, o);
            break;
        case OP_ADD: /* in0, in1 = c (B,s2) */
            limbs_to_mpz(a, j->in0 + i * s2, s2);
            limbs_to_mpz(b, j->in1 + i * s2, s2);
            raw_add_z(o, j->pub, a, b);
            mpz_to_limbs(j->out + i * s2, s2, o);
            break;
        case OP_MUL: /* in0 = c (B,s2), in1 = scalars (B,scalar_limbs) */
            limbs_to_mpz(a, j->in0 + i * s2, s2);
            limbs_to_mpz(b, j->in1 + i * j->scalar_limbs, j->scalar_limbs);
            rc = raw_mul_z(o, j->pub, a, b, t0, t1);
            if (!rc) mpz_to_limbs(j->out + i * s2, s2, o);
            break;
        }
        if (rc) { j->status = rc; j->bad_index = i; }
    }
    mpz_clears(a, b, o, t0, t1, t2, NULL);
    return NULL;
}

static int run_jobs(orc_job proto, size_t B, int nthreads, size_t *bad_index) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > B && B > 0) nthreads = (int)B;
    orc_job *jobs = (orc_job *)calloc((size_t)nthreads, sizeof(orc_job));
    pthread_t *tids = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = proto;
        jobs[t].begin = B * (size_t)t / (size_t)nthreads;
        jobs[t].end = B * (size_t)(t + 1) / (size_t)nthreads;
        jobs[t].status = ORC_OK;
        if (nthreads == 1) orc_worker(&jobs[t]);
        else pthread_create(&tids[t], NULL, orc_worker, &jobs[t]);
    }
    int status = ORC_OK;
    for (int t = 0; t < nthreads; ++t) {
        if (nthreads > 1) pthread_join(tids[t], NULL);
        if (jobs[t].status != ORC_OK && status == ORC_OK) {
            status = jobs[t].status;
            if (bad_index) *bad_index = jobs[t].bad_index;
        }
    }
    free(jobs); free(tids);
    return status;
}

int orc_encrypt_batch(const uint32_t *n, size_t n_limbs, const uint32_t *m, const uint32_t *r,
                      uint32_t *c, size_t B, int nthreads) {
    orc_pub pub; pub_init(&pub, n, n_limbs);
    orc_job j; memset(&j, 0, sizeof j);
    j.op = OP_ENCRYPT; j.pub = &pub; j.n_limbs = n_limbs; j.in0 = m; j.in1 = r; j.out = c;
    int rc = run_jobs(j, B, nthreads, NULL);
    pub_clear(&pub);
    return rc;
}
int orc_obfuscate_batch(const uint32_t *n, size_t n_limbs, const uint32_t *c_in, const uint32_t *r,
                        uint32_t *c_out, size_t B, int nthreads) {
    orc_pub pub; pub_init(&pub, n, n_limbs);
    orc_job j; memset(&j, 0, sizeof j);
    j.op = OP_OBFUSCATE; j.pub = &pub; j.n_limbs = n_limbs; j.in0 = c_in; j.in1 = r; j.out = c_out;
    int rc = run_jobs(j, B, nthreads, NULL);
    pub_clear(&pub);
    return rc;
}
int orc_decrypt_batch(const uint32_t *n, size_t n_limbs, const uint32_t *p, const uint32_t *q,
                      size_t pq_limbs, const uint32_t *c, uint32_t *m, size_t B, int nthreads) {
    orc_pub pub; orc_priv priv;
    pub_init(&pub, n, n_limbs);
    int rc = priv_init(&priv, &pub, p, q, pq_limbs);
    if (!rc) {
        orc_job j; memset(&j, 0, sizeof j);
        j.op = OP_DECRYPT; j.pub = &pub; j.priv = &priv; j.n_limbs = n_limbs; j.in0 = c; j.out = m;
        rc = run_jobs(j, B, nthreads, NULL);
    }
    priv_clear(&priv); pub_clear(&pub);
    return rc;
}
int orc_add_batch(const uint32_t *n, size_t n_limbs, const uint32_t *a, const uint32_t *b,
                  uint32_t *out, size_t B, int nthreads) {
    orc_pub pub; pub_init(&pub, n, n_limbs);
    orc_job j; memset(&j, 0, sizeof j);
    j.op = OP_ADD; j.pub = &pub; j.n_limbs = n_limbs; j.in0 = a; j.in1 = b; j.out = out;
    int rc = run_jobs(j, B, nthreads, NULL);
    pub_clear(&pub);
    return rc;
}
int orc_mul_batch(const uint32_t *n, size_t n_limbs, const uint32_t *c, const uint32_t *scalars,
                  size_t scalar_limbs, uint32_t *out, size_t B, int nthreads, size_t *bad_index) {
    orc_pub pub; pub_init(&pub, n, n_limbs);
    orc_job j; memset(&j, 0, sizeof j);
    j.op = OP_MUL; j.pub = &pub; j.n_limbs = n_limbs; j.scalar_limbs = scalar_limbs;
    j.in0 = c; j.in1 = scalars; j.out = out;
    int rc = run_jobs(j, B, nthreads, bad_index);
    pub_clear(&pub);
    return rc;
}
/* phe/util.py:114-116: the smallest (probable) prime above `start`; 1 if it does not fit `limbs` words */
int orc_next_prime(const uint32_t *start, uint32_t *out, size_t limbs) {
    mpz_t a, p;
    mpz_inits(a, p, NULL);
    limbs_to_mpz(a, start, limbs);
    mpz_nextprime(p, a);
    const int too_big = mpz_sizeinbase(p, 2) > 32 * limbs;
    if (!too_big) mpz_to_limbs(out, limbs, p);
    mpz_clears(a, p, NULL);
    return too_big;
}
/* 2 = certainly prime, 1 = probably prime, 0 = composite (mpz_probab_prime_p with `reps` rounds) */
int orc_probab_prime(const uint32_t *n, size_t limbs, int reps) {
    mpz_t a;
    mpz_init(a);
    limbs_to_mpz(a, n, limbs);
    const int r = mpz_probab_prime_p(a, reps);
    mpz_clear(a);
    return r;
}

const char *orc_gmp_version(void) { return gmp_version; }
