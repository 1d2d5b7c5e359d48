// kernels_s8d.hip — split-modulus kernels for groups of 8 lanes, L in {3} (see split_kernels.inc): the 8-lane rung of
// 1024-bit keys' CRT halves (p, q of 512 bits = 18 limbs)
#define PHE_PART s8d
#define PHE_PART_G 8
#define PHE_FOR_EACH_L(X) X(3)
#include "split_kernels.inc"
