// kernels_s1a.hip — split-modulus kernels with ONE lane per number, L in {18} (see split_kernels.inc): the CRT halves of
// 1024-bit keys (p, q of 512 bits = 18 limbs) without any cross-lane step
#define PHE_PART s1a
#define PHE_PART_G 1
#define PHE_FOR_EACH_L(X) X(18)
#include "split_kernels.inc"
