/*
 * phe_hip.h — C-ABI of the MI355X (gfx950) batched Paillier engine.
 *
 * This is the drop-in boundary for the one data-parallel hot path of data61/python-paillier
 * (`phe` 1.5.0).  The reference has no FFI of its own: its backend seam is the three module
 * functions phe/util.py:38-50 (powmod), :53-64 (mulmod), :85-103 (invert), called one Python int
 * at a time from the five raw_* functions of phe/paillier.py.  A scalar seam cannot batch, so the
 * boundary sits one level up and every entry point below replaces a whole loop of calls to one of
 * those raw_* functions (file:line cited per function).  INTEGRATION.md shows the ctypes stub a
 * python-paillier maintainer would add.
 *
 * Conventions
 *   - Numbers are little-endian arrays of uint32 limbs ("int.to_bytes(4*limbs, 'little')").
 *     Batches are row-major (batch, limbs), caller-owned.
 *   - n_limbs  = limbs of n (e.g. 64 for a 2048-bit key);  ciphertexts have ct_limbs = 2*n_limbs.
 *   - Every function returns a status: PHE_HIP_OK, or an error with a message retrievable through
 *     phe_hip_last_error() (thread-local).  Nothing throws across the boundary.  Argument/range
 *     validation that the reference does in Python (TypeError/ValueError) stays in the host
 *     language; the kernels compute on whatever limbs they are given, modulo the stated ranges.
 *   - Plain functions take HOST pointers (ordinary pageable memory) and are synchronous.  From 131072 rows on,
 *     phe_hip_encrypt / _encrypt_owner / _obfuscate / _decrypt move the batch in chunks through pinned staging buffers
 *     on three internal streams (uploads and downloads under the kernels); smaller batches upload, compute, download.
 *     A handful of rows (every operand within 32 KiB: phe_hip_encrypt / _encrypt_owner / _obfuscate / _decrypt / _mulmod /
 *     _add_plain / _powmod) is copied by the CPU into a pinned, device-mapped buffer that the kernels read and write across
 *     PCIe themselves — one stream synchronisation instead of three blocking copies (a one-row phe_hip_mulmod at 2048 bits:
 *     67 -> 38 us); PHE_HIP_NO_MAPPED_STAGING=1 keeps the copies.  Same results either way.
 *     *_dev functions take DEVICE pointers (hipMalloc'd / torch CUDA tensors) and enqueue on
 *     `stream` (a hipStream_t passed as void*, NULL = default stream) without synchronising.
 *   - A context is bound to one device and is not thread-safe: calls on one context must not overlap in HOST time (one
 *     per host thread / rank, or a lock around the calls).  On the DEVICE there is one stream order per context: the
 *     window tables and intermediates of a call belong to the context, so a *_dev call issued on another stream than
 *     the previous one first makes its stream wait for that call's work (an event, no host synchronisation).  Work on
 *     different contexts runs concurrently.
 *   - All results are canonical residues, bit-identical to the reference's gmpy2/CPython values.
 */
#ifndef PHE_HIP_H
#define PHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHE_HIP_OK 0
#define PHE_HIP_EINVAL 1      /* bad argument (NULL, size mismatch, unsupported key width, even modulus) */
#define PHE_HIP_EHIP 2        /* HIP runtime error (no device, out of memory, launch failure) */
#define PHE_HIP_ENOINVERSE 3  /* an element has no inverse: reference raises ZeroDivisionError (phe/util.py:96-102) */

typedef struct phe_hip_ctx phe_hip_ctx;

/* Message for the last non-OK status returned on this thread. */
const char* phe_hip_last_error(void);

/* Version of this interface as the LIBRARY was built with it.  A binding compares it with the PHE_HIP_ABI_VERSION of the header
 * it mirrors before the first call (phe/_native.py lib()): argument lists changed under unchanged symbol names twice
 * (round 5: phe_hip_gather_rows_dev / phe_hip_scatter_rows_dev gained a row-count bound), and a stale mirror would still link.
 * Bumped on every change to an existing signature or to the meaning of an argument; new entry points alone do not bump it. */
#define PHE_HIP_ABI_VERSION 6
int phe_hip_abi_version(void);

/* Number of visible HIP devices. */
int phe_hip_device_count(int* count);

/* ---- contexts ---------------------------------------------------------------------------- */

/* Public-key context: what PaillierPublicKey.__init__ holds (phe/paillier.py:86-90) plus the
 * Montgomery constants of n^2 and the window schedule of the exponent n. */
int phe_hip_ctx_create_public(const uint32_t* n, int n_limbs, int device, phe_hip_ctx** out);

/* Private-key context: additionally what PaillierPrivateKey.__init__ holds (phe/paillier.py:217-235).
 * p < q required (the reference orders them at :224-229); hp, hq, p_inverse are the values the
 * reference computes at :233-235 (the host language passes them in; they are once-per-key). */
int phe_hip_ctx_create_private(const uint32_t* n, int n_limbs, const uint32_t* p, const uint32_t* q,
                               const uint32_t* hp, const uint32_t* hq, const uint32_t* p_inverse,
                               int pq_limbs, int device, phe_hip_ctx** out);

void phe_hip_ctx_destroy(phe_hip_ctx* ctx);

/* Geometry chosen for this key, encoded G*100 + L (lanes per limb group, 29-bit limbs per lane) for the n^2
 * kernels and for the p^2/q^2 kernels, and the number of limb groups one launch keeps in flight.
 * Any pointer may be NULL. */
int phe_hip_ctx_info(const phe_hip_ctx* ctx, int* n_limbs, int* ct_limbs, int* lanes_limbs_pub,
                     int* lane_limbs_priv, int* rows_in_flight, int* has_private);

/* Which kernels run the batch-uniform exponentiations of this context: *split_pub / *split_priv = 1 when the
 * split-modulus kernels (k_modexp_split, half-width pair arithmetic modulo n / p / q, csrc/split_core.h) are in
 * use for encrypt+obfuscate / decrypt, 0 for the full-width kernels (k_modexp_uniform; PHE_HIP_ENGINE=full or a
 * key width no split kernel is compiled for).  lane_limbs_* of phe_hip_ctx_info describe the kernels in use.
 * Results are identical either way.  Any pointer may be NULL. */
int phe_hip_ctx_engine(const phe_hip_ctx* ctx, int* split_pub, int* split_priv);

/* Tuning: workgroups (256 threads) resident per CU for the modexp kernels; 0 = ask the HIP occupancy API
 * per kernel (the default). */
int phe_hip_ctx_set_blocks_per_cu(phe_hip_ctx* ctx, int blocks_per_cu);

/* The geometry ladder.  A number is owned by a limb group of G lanes; rung 0 is the narrowest group the key width offers
 * (most limbs per lane: the best throughput once every SIMD of the GPU has a wave).  A batch too small to fill the GPU
 * that way runs on a wider rung — fewer limbs per lane, a shorter dependent chain per product, more lanes per number —
 * chosen per call from the batch size.  Results are identical on every rung.
 * phe_hip_ctx_ladder: the rungs as G*100 + L, narrowest first, for the n-side and the p/q-side kernels (up to `capacity`
 * entries each are written; *n_pub / *n_priv = how many exist; the widest rung is G = 64, one number per wavefront, for a
 * handful of numbers).  phe_hip_ctx_set_group: 0 = choose by batch size (default), G in {2, 4, 8, 16, 64} = always the rung of
 * G-lane groups (the next wider one if there is none) — for tests and measurements.
 * phe_hip_ctx_last_launch: what the last encrypt / obfuscate / decrypt / powmod / pair call took: *path = bit set of
 * 1 (r^n modulo the scaled modulus k*n), 2 (key owner's CRT form), 4 (decrypt halves side by side in one grid),
 * 8 (host batch pipelined through pinned chunks), 16 (fused obfuscate kernel), 32 (every number on a PAIR of wavefronts: the
 * lowest-latency form, taken for a handful of numbers on the G = 64 rung), 64 (the L-function / CRT tail of a decrypt on one
 * wavefront per ciphertext instead of one thread: every batch from 1024-bit keys up, small batches below), 128 (the rungs of
 * 16 lanes / of the whole wave on the "late" sweeps: scaled modulus, quotient product after the shift, rows = the limbs the
 * modulus needs instead of the G*L the lanes hold; PHE_HIP_NO_LATE=1 keeps the round-3 kernels), 256 (phe_hip_mulmod[_dev]: the
 * product as ONE plain product + ONE fold against the key's table in LDS instead of two Montgomery products — batches from 2048
 * rows on, keys whose table fits a CU's LDS (up to ~2700 bits); PHE_HIP_NO_TABLE_MUL=1 keeps the Montgomery kernels), 512 (with 256:
 * that product by tiles of 64 per workgroup with ONE ELEMENT PER LANE — batches from 16384 rows on; the fold's table words come
 * through the scalar cache instead of LDS; PHE_HIP_NO_TILE_MUL=1 keeps the kernel with the table in LDS; keys of ~810 ... 1024 bits take
 * the tiles on an 8-wave workgroup, 8 x 9 columns, with no table-in-LDS form below 16384 rows: PHE_HIP_NO_TILE8=1 keeps the Montgomery
 * kernels there); *geom_pub / *geom_priv = G*100 + L of the
 * exponentiation kernels used.  Any pointer may be NULL. */
int phe_hip_ctx_ladder(const phe_hip_ctx* ctx, int* pub_geoms, int* priv_geoms, int capacity, int* n_pub, int* n_priv);
/* The MEASURED ladder.  Without it a rung's time is an estimate from its shape; with it the rung of a call is the one whose measured
 * launch time at that batch size is least.  `table`: text, one line per measurement "key_bits family G rows ns" — key width in bits
 * (lines of other widths are ignored), kernel family (1: fixed-exponent r^n of encrypt / obfuscate, 2: the CRT halves of decrypt,
 * 3: per-element exponents of _raw_mul), lanes per number of the rung, batch size, launch time in nanoseconds; '#' starts a comment.
 * Made by `tools/bench_sweep.py --calibrate` (every rung pinned in turn; python-paillier_amd/phe/ladder_gfx950.txt is the committed
 * one, loaded by the host mirror when a context is made).  Between two measured sizes the time is interpolated, below the smallest it
 * stays (latency-bound), above the largest it scales with the rows; a family is chosen by measurement only when EVERY rung it could
 * take has a table.  table = NULL forgets the table.  *accepted = lines kept (may be NULL). */
int phe_hip_ctx_load_ladder(phe_hip_ctx* ctx, const char* table, int* accepted);
int phe_hip_ctx_set_group(phe_hip_ctx* ctx, int group);
int phe_hip_ctx_last_launch(const phe_hip_ctx* ctx, int* path, int* geom_pub, int* geom_priv);

/* Frees the grow-only device and pinned buffers of the context (window tables, intermediates, staging of the host-pointer
 * entry points and of their chunk pipeline); the next call allocates what it needs.  Synchronises the device. */
int phe_hip_ctx_release_scratch(phe_hip_ctx* ctx);

/* ---- the hot path, host buffers ------------------------------------------------------------ */

/* c[i] = (1 + n*m[i]) * r[i]^n mod n^2      — PaillierPublicKey.raw_encrypt(m, r_value=r),
 * phe/paillier.py:102-139.  m: (batch, n_limbs) any value < 2^(32*n_limbs) (m >= n wraps like the
 * reference's `% nsquare`, :134);  r: (batch, n_limbs), 0 < r < n;  c: (batch, ct_limbs). */
int phe_hip_encrypt(phe_hip_ctx* ctx, const uint32_t* m, const uint32_t* r, uint32_t* c, size_t batch);

/* The same value, bit for bit, for a caller that holds the private key (private context): r^n from its CRT halves modulo
 * p^2 and q^2 — see phe_hip_encrypt_owner_dev below.  EINVAL on public contexts and for key widths without the needed
 * geometries (phe_hip_ctx_owner_encrypt tells).  Pipelined from 131072 rows on like phe_hip_encrypt. */
int phe_hip_encrypt_owner(phe_hip_ctx* ctx, const uint32_t* m, const uint32_t* r, uint32_t* c, size_t batch);

/* c_out[i] = c_in[i] * r[i]^n mod n^2       — EncryptedNumber.obfuscate(), phe/paillier.py:603-624
 * (with the obfuscator r an explicit input; the reference draws it at :621). */
int phe_hip_obfuscate(phe_hip_ctx* ctx, const uint32_t* c_in, const uint32_t* r, uint32_t* c_out, size_t batch);

/* m[i] = PaillierPrivateKey.raw_decrypt(c[i]) — phe/paillier.py:328-354 incl. l_function :362-364
 * and crt :366-374.  c: (batch, ct_limbs) < n^2;  m: (batch, n_limbs).  Needs a private context. */
int phe_hip_decrypt(phe_hip_ctx* ctx, const uint32_t* c, uint32_t* m, size_t batch);

/* out[i] = a[i]*b[i] mod n^2                — EncryptedNumber._raw_add, phe/paillier.py:705-719
 * (= phe.util.mulmod, phe/util.py:53-64).  a, b, out: (batch, ct_limbs); b < n^2. */
int phe_hip_mulmod(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t batch);

/* out[i] = c[i] * (1 + n*m[i]) mod n^2      — adding a plaintext to a ciphertext: EncryptedNumber._add_encoded,
 * phe/paillier.py:673-675 (raw_encrypt(m, r_value=1) followed by _raw_add).  c, out: (batch, ct_limbs);
 * m: (batch, n_limbs), any value < 2^(32*n_limbs) (reduced mod n like raw_encrypt does). */
int phe_hip_add_plain(phe_hip_ctx* ctx, const uint32_t* c, const uint32_t* m, uint32_t* out, size_t batch);

/* out[i] = base[i]^e[i] mod n^2             — the powmod of EncryptedNumber._raw_mul,
 * phe/paillier.py:749/:751 (= phe.util.powmod, phe/util.py:38-50).  base: (batch, ct_limbs) < n^2;
 * e: (batch, exp_limbs).  The negative-scalar branch (:745-749) is composed by the host from
 * phe_hip_invert + this function, partitioned on the same threshold n - max_int.  A handful of numbers (the scalar
 * EncryptedNumber.__mul__ is a batch of one) run each on a pair of wavefronts with a sliding-window schedule of its own
 * exponent, which this entry point — it sees the exponents — makes on the host; phe_hip_powmod_dev takes fixed windows. */
int phe_hip_powmod(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* e, int exp_limbs, uint32_t* out,
                   size_t batch);

/* out = prod_i base[i]^e[i] mod n^2         — the encrypted dot product sum_i k_i * E(x_i): what the reference
 * computes as a chain of EncryptedNumber.__mul__ (phe/paillier.py:721-751 _raw_mul -> :751 powmod) and __add__
 * (:705-719 _raw_add -> mulmod), e.g. np.dot over ciphertexts (phe/tests/math_test.py:44-58) and
 * examples/logistic_regression_encrypted_model.py:170-177.  The product is one canonical residue, independent of
 * the order of the factors, so chunks of the batch share one square-and-multiply ladder (Straus' interleaving) and
 * the per-chunk products are joined by a pairwise mulmod tree.  base: (batch, ct_limbs) < n^2;
 * e: (batch, exp_limbs); out: ONE row of ct_limbs words (1 for an empty batch).  Rows whose scalar is on the
 * negative branch (:745-749) are handled by the host as a second product that is inverted once. */
int phe_hip_multiexp(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* e, int exp_limbs, uint32_t* out,
                     size_t batch);

/* out[i] = a[i]^-1 mod n^2                  — phe.util.invert (phe/util.py:85-103) as used at
 * phe/paillier.py:747.  Montgomery's simultaneous inversion: a product tree of mulmod launches, ONE
 * scalar inversion of the root, and the tree walked back down.  On PHE_HIP_ENOINVERSE *bad_index
 * is the first row with gcd(a, n^2) != 1 (the reference raises ZeroDivisionError for it). */
int phe_hip_invert(phe_hip_ctx* ctx, const uint32_t* a, uint32_t* out, size_t batch, size_t* bad_index);

/* ---- decimal wire format ---------------------------------------------------------------------
 * The reference's interchange formats carry every ciphertext as str(int): docs/serialisation.rst:24-43
 * ("values": [[str(ciphertext), exponent], ...]) and phe/command_line.py:120-131, :267-276 ({"v": str(c), "e": ...}).
 * These entry points are the batch form of that str() / int(): rows of little-endian 32-bit words <-> rows of
 * `width` ASCII digits, most significant first, left-padded with '0' (strip / add the padding on the host; the
 * digits are exactly str(int)'s).  phe_hip_decimal_width(words) digits hold any number of `words` words.
 * EINVAL (+ *bad_index = first offending row) for a character that is not a digit (int() raises ValueError) or a
 * value that does not fit `words` words; EINVAL if a number needs more than `width` digits. */
int phe_hip_decimal_width(int words);
int phe_hip_to_decimal(phe_hip_ctx* ctx, const uint32_t* limbs, int words, char* digits, int width, size_t batch);
int phe_hip_from_decimal(phe_hip_ctx* ctx, const char* digits, int width, uint32_t* limbs, int words, size_t batch,
                         size_t* bad_index);

/* ---- batched primality (key generation) ---------------------------------------------------------
 * pass[i] = 1 iff n[i] is a strong probable prime to base[i] (Miller-Rabin), one modulus PER ROW — the test that
 * getprimeover (phe/util.py:106-124: gmpy2.next_prime, or is_prime -> miller_rabin at :381-443) runs candidate after
 * candidate, here over all sieved candidates of a search window (and, to confirm a hit, over many bases) in one
 * launch.  n, base: (batch, limbs) little-endian words; n odd and > 3, 2 <= base <= n - 2 (EINVAL otherwise).
 * Needs no context (no key); runs on `device`.  Host pointers. */
int phe_hip_miller_rabin(int device, const uint32_t* n, const uint32_t* base, int limbs, uint8_t* pass, size_t batch);

/* ---- the hot path, device buffers (resident operands; asynchronous on `stream`) -------------- */
int phe_hip_encrypt_dev(phe_hip_ctx* ctx, const uint32_t* m, const uint32_t* r, uint32_t* c, size_t batch, void* stream);
/* raw_encrypt for the HOLDER of the private key (private context): the same value (1 + n*m) * r^n mod n^2, bit for bit, but
 * r^n is taken modulo p^2 and modulo q^2 (half-width numbers: about half the multiply-adds in all) and lifted to n^2 by
 * the Chinese remainder theorem.  The reference has no such function (its raw_encrypt, phe/paillier.py:102-139, only knows
 * the public key); the drop-in uses it when the key object it is asked to encrypt under shares an engine with its private
 * key.  EINVAL on public contexts and for key widths without the needed geometries (use phe_hip_encrypt_dev then). */
int phe_hip_ctx_owner_encrypt(const phe_hip_ctx* ctx, int* offered); /* 1 if the two entry points below will work */
int phe_hip_encrypt_owner_dev(phe_hip_ctx* ctx, const uint32_t* m, const uint32_t* r, uint32_t* c, size_t batch, void* stream);
int phe_hip_obfuscate_dev(phe_hip_ctx* ctx, const uint32_t* c_in, const uint32_t* r, uint32_t* c_out, size_t batch, void* stream);
int phe_hip_decrypt_dev(phe_hip_ctx* ctx, const uint32_t* c, uint32_t* m, size_t batch, void* stream);
int phe_hip_mulmod_dev(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t batch, void* stream);
int phe_hip_add_plain_dev(phe_hip_ctx* ctx, const uint32_t* c, const uint32_t* m, uint32_t* out, size_t batch, void* stream);
/* Chains of homomorphic additions on resident vectors (EncryptedNumber._raw_add, phe/paillier.py:705-719, is one
 * mulmod(a, b, n^2) = two Montgomery products here): out[i] = a[i] * b[i] * R^-1 mod n^2, canonical — ONE product.
 * R = 2^bits with bits from phe_hip_mont_radix_bits (fixed per context).  b_is_row != 0: b is a single row used for
 * every i (e.g. R^(d+1) mod n^2, which turns a value carrying d missing factors of R back into the plain residue).
 * The caller keeps count of the missing powers of R (the host mirror does, per vector: phe/ciphertext.py). */
int phe_hip_montmul_dev(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* b, int b_is_row, uint32_t* out, size_t batch,
                        void* stream);
int phe_hip_mont_radix_bits(phe_hip_ctx* ctx, int* bits);

/* Resident ciphertext rows in the engine's own form ("pair form", csrc/split_core.h).  A vector that is only multiplied —
 * chains of EncryptedNumber.__add__ (= _raw_add, phe/paillier.py:705-719), sum() trees, obfuscators r^n made ahead of time —
 * can stay in the representation the exponentiation kernels compute in: x*R mod n^2 = X0 - n*X1 with X0, X1 half-width
 * numbers, stored as rows of phe_hip_pair_words() words (2 * ceil((bits(n)+4)/29) rounded to the limb-group size: 144
 * for a 2048-bit key's 128-word ciphertext) of 29-bit limbs.  A homomorphic addition on such rows is ONE pair product
 * (5 half-width limb products against 16 for mulmod, 8 for phe_hip_montmul_dev), nothing is left to settle, and the rows
 * never pass through 32-bit words in between.
 *   phe_hip_to_pair_dev   : c (batch, ct_limbs), any value < 2^(32*ct_limbs)  ->  pair rows
 *   phe_hip_pair_mul_dev  : out[i] = a[i] * b[i] (pair rows; b_is_row != 0: the single row b for every i)
 *   phe_hip_from_pair_dev : pair rows -> the canonical residue mod n^2, bit-identical to what the chain of _raw_add returns;
 *                           with m != NULL (plaintexts, (batch, n_limbs)) the residue of x * (1 + n*m): raw_encrypt(m, r)
 *                           from r^n kept in the pair form (phe/paillier.py:134-139), or adding a plaintext (:673-675).
 *   phe_hip_pair_powmod_dev : out[i] = a[i]^e[i] (pair rows in and out; e: (batch, exp_limbs) words, max_exp_bits as for
 *                           phe_hip_powmod_dev): EncryptedNumber.__mul__ / _raw_mul by a non-negative scalar
 *                           (phe/paillier.py:721-751) on resident rows, without the conversion in and the exit that
 *                           phe_hip_powmod_dev pays per element (6 % of a 56-bit scalar multiplication).
 * The contents of a pair row are NOT canonical (lazy reduction): compare ciphertexts only after from_pair.
 * EINVAL without the split-modulus engine (PHE_HIP_ENGINE=full or a key width it has no kernel for). */
int phe_hip_pair_words(const phe_hip_ctx* ctx, int* words);
int phe_hip_to_pair_dev(phe_hip_ctx* ctx, const uint32_t* c, uint32_t* pair, size_t batch, void* stream);
int phe_hip_pair_mul_dev(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* b, int b_is_row, uint32_t* out, size_t batch,
                         void* stream);
int phe_hip_from_pair_dev(phe_hip_ctx* ctx, const uint32_t* pair, const uint32_t* m, uint32_t* c, size_t batch, void* stream);
int phe_hip_pair_powmod_dev(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* e, int exp_limbs, int max_exp_bits, uint32_t* out,
                            size_t batch, void* stream);
/* rows dot products  out[r] = prod_i a_i ^ e[r][i]  over a resident vector in the pair form (phe_hip_multiexp_rows_dev without
 * the conversion of every ciphertext into the form; non-negative scalars only: the negative branch of _raw_mul inverts
 * residues).  e: (rows, batch, exp_limbs); out: (rows, ct_limbs) canonical residues. */
int phe_hip_pair_multiexp_rows_dev(phe_hip_ctx* ctx, const uint32_t* pair_base, const uint32_t* e, int exp_limbs, int max_exp_bits,
                                   uint32_t* out, size_t batch, size_t rows, void* stream);
/* out (ONE pair row) = the product of all `batch` pair rows: the homomorphic sum of a resident vector — sum(enc_list) in
 * the reference is a chain of _raw_add (phe/paillier.py:705-719); the product of residues is independent of the order —
 * as a pairwise tree of log2(batch) launches queued back to back on `stream`, no host round trip between the levels. */
int phe_hip_pair_reduce_dev(phe_hip_ctx* ctx, const uint32_t* pair, size_t batch, uint32_t* out, void* stream);
/* max_exp_bits: upper bound on the bit length of every e[i] (0 = 32*exp_limbs) */
int phe_hip_powmod_dev(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* e, int exp_limbs, int max_exp_bits,
                       uint32_t* out, size_t batch, void* stream);
/* out: one row of ct_limbs words on the device, complete when `stream` has drained */
int phe_hip_multiexp_dev(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* e, int exp_limbs, int max_exp_bits,
                         uint32_t* out, size_t batch, void* stream);

/* the decimal conversions on device buffers; both synchronise `stream` (they report per-row errors) */
int phe_hip_to_decimal_dev(phe_hip_ctx* ctx, const uint32_t* limbs, int words, char* digits, int width, size_t batch,
                           void* stream);
int phe_hip_from_decimal_dev(phe_hip_ctx* ctx, const char* digits, int width, uint32_t* limbs, int words, size_t batch,
                             size_t* bad_index, void* stream);

/* The matrix form: out[r] = prod_i b_i^e[r][i] mod n^2 for r < rows — a plaintext matrix times an encrypted vector,
 * i.e. `rows` of the dot products above over the SAME ciphertexts (examples/logistic_regression_encrypted_model.py:
 * 170-177 scores every sample against one encrypted weight vector).  e: (rows, batch, exp_limbs); out: (rows, ct_limbs).
 * b_i = base[i], or base_inv[i] (= base[i]^-1 mod n^2, e.g. from phe_hip_invert_dev) where neg[r*batch + i] != 0 —
 * the reference's negative-scalar branch (phe/paillier.py:745-749) differs from row to row; base_inv and neg may
 * both be NULL.  Needs the split-modulus engine (EINVAL otherwise: call phe_hip_multiexp_dev row by row). */
int phe_hip_multiexp_rows_dev(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* base_inv, const uint32_t* e,
                              const uint8_t* neg, int exp_limbs, int max_exp_bits, uint32_t* out, size_t batch, size_t rows,
                              void* stream);

/* The same `rows` products when the matrix is SPARSE or has many rows: out[r] = prod over the entries of row r of
 * b[col]^exp — entries in CSR order (row_ptr: rows + 1 offsets, cols / e / neg: one item per entry), or dense rows of
 * `batch` entries when row_ptr and cols are NULL.  This is the loop of Bob.encrypted_score,
 * examples/logistic_regression_encrypted_model.py:170-177 (`_, idx = x.nonzero(); for i in idx: score += x[0, i] * w[i]`),
 * for all samples at once: the 2^w-ary tables of every ciphertext (and of base_inv, if given) are built once, then one
 * limb group runs ONE ladder per row over that row's entries only.  order (optional): the rows sorted by entry count,
 * so that the groups of a wavefront get ladders of similar length.  cols must be < batch (not checked: the arrays
 * live on the device).  Needs the split-modulus engine. */
int phe_hip_multiexp_csr_dev(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* base_inv, size_t batch,
                             const uint64_t* row_ptr, const uint32_t* cols, const uint32_t* e, const uint8_t* neg,
                             int exp_limbs, int max_exp_bits, const uint32_t* order, uint32_t* out, size_t rows,
                             void* stream);

/* phe_hip_invert on device buffers.  Synchronises `stream` internally (the root of the product tree makes one
 * round trip to the host); results are complete on return. */
int phe_hip_invert_dev(phe_hip_ctx* ctx, const uint32_t* a, uint32_t* out, size_t batch, size_t* bad_index, void* stream);
/* out[i] = mask[i] ? b[i] : a[i] for rows of `limbs` words; mask is one byte per row (device).  Used to keep the
 * branch select of _raw_mul (phe/paillier.py:745-751: inverted base for scalars >= n - max_int) on the device. */
int phe_hip_select_rows_dev(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* b, const uint8_t* mask, uint32_t* out,
                            int limbs, size_t batch, void* stream);
/* Rows of `limbs` words moved by an index list (device, `count` 32-bit row indices): gather dst[j] = src[idx[j]], scatter
 * dst[idx[j]] = src[j].  With them the negative-scalar branch of _raw_mul (phe/paillier.py:745-749: powmod(invert(c), n - s))
 * inverts only the rows that take it — gather, phe_hip_invert_dev on the subset, scatter into a copy of the vector — instead of
 * inverting the whole vector and selecting.
 * src_rows / dst_rows: the rows the INDEXED buffer holds.  An index at or beyond it never leaves that buffer: the scatter skips the
 * row, the gather writes a row of zeros.  Ordered like every *_dev entry point (one stream order per context). */
int phe_hip_gather_rows_dev(phe_hip_ctx* ctx, const uint32_t* src, size_t src_rows, const uint32_t* idx, uint32_t* dst, int limbs,
                            size_t count, void* stream);
int phe_hip_scatter_rows_dev(phe_hip_ctx* ctx, const uint32_t* src, const uint32_t* idx, uint32_t* dst, size_t dst_rows, int limbs,
                             size_t count, void* stream);

/* ---- device memory helpers for hosts without a tensor library ------------------------------- */
int phe_hip_malloc(phe_hip_ctx* ctx, size_t bytes, void** dptr);
int phe_hip_free(phe_hip_ctx* ctx, void* dptr);
int phe_hip_memcpy_h2d(phe_hip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int phe_hip_memcpy_d2h(phe_hip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int phe_hip_memcpy_d2d(phe_hip_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes, void* stream);
int phe_hip_stream_sync(phe_hip_ctx* ctx, void* stream);
/* a stream that does not synchronise with the NULL stream: *_dev launches queued on it overlap the blocking copies
 * above (hosts without a tensor library; with torch pass torch.cuda.current_stream().cuda_stream instead) */
int phe_hip_stream_create(phe_hip_ctx* ctx, void** stream);
int phe_hip_stream_destroy(phe_hip_ctx* ctx, void* stream);

/* ---- multi-GPU: the one exchange step of the path -------------------------------------------------
 * Batches shard by contiguous row ranges, one process per GPU, and nothing crosses GPUs while encrypting / decrypting /
 * adding (the reference has no multi-device code to cite).  Only when every device must hold the whole
 * ciphertext vector are the shards concatenated: ONE all-gather over RCCL (xGMI inside a node).  A host framework that
 * already owns a process group can do it itself (python-paillier_amd/phe/sharding.py uses torch.distributed); hosts
 * without one use these: rank 0 makes an id, the host hands its 128 bytes to the other ranks by whatever channel it has,
 * every rank creates its communicator on its context's device, then gathers device rows.  RCCL (librccl.so.1) is
 * loaded on first use; PHE_HIP_EHIP with a message if it is missing.
 * phe_hip_allgather_dev: local = (rows, limbs) words on this rank's device, all = (world * rows, limbs): rank r's rows at
 * [r * rows, (r + 1) * rows) on every rank; equal `rows` on all ranks (pad the last shard).  Asynchronous on `stream`. */
typedef struct phe_hip_comm phe_hip_comm;
int phe_hip_comm_unique_id(uint8_t id[128]);
int phe_hip_comm_create(phe_hip_ctx* ctx, const uint8_t id[128], int rank, int world, phe_hip_comm** out);
int phe_hip_allgather_dev(phe_hip_comm* comm, const uint32_t* local, uint32_t* all, size_t rows, int limbs, void* stream);
void phe_hip_comm_destroy(phe_hip_comm* comm);

/* ---- diagnostics ----------------------------------------------------------------------------- */
/* Runs the cross-lane primitives of csrc/wave_gfx950.h on lane ids: out is 1026 uint32 — for 16-lane groups down1 / up1 /
 * bcast0 (3 x 64), the ballot of (lane % 3 == 0) folded per lane (64) and as two words, then down1 / up1 / bcast0 for groups of
 * 8, 4, 2 and 64 lanes (3 x 64 each).  Used by tests/test_gpu_parity.py::test_wave_primitives to pin the emulator's semantics. */
int phe_hip_selftest_prims(int device, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* PHE_HIP_H */
