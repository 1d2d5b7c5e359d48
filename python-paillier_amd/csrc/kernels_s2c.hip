// kernels_s2c.hip — split-modulus kernels for groups of 2 lanes, L in {27} (see split_kernels.inc)
#define PHE_PART s2c
#define PHE_PART_G 2
#define PHE_FOR_EACH_L(X) X(27)
#include "split_kernels.inc"
