// kernels_s4d.hip — split-modulus kernels for groups of 4 lanes, L in {5} (see split_kernels.inc): the 4-lane rung of
// 1024-bit keys' CRT halves (p, q of 512 bits = 18 limbs)
#define PHE_PART s4d
#define PHE_PART_G 4
#define PHE_FOR_EACH_L(X) X(5)
#include "split_kernels.inc"
