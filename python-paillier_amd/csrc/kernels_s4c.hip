// kernels_s4c.hip — split-modulus kernels for groups of 4 lanes, L in {27} (see split_kernels.inc)
#define PHE_PART s4c
#define PHE_PART_G 4
#define PHE_FOR_EACH_L(X) X(27)
#include "split_kernels.inc"
