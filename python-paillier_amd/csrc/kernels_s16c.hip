// kernels_s16c.hip — split-modulus kernels for groups of 16 lanes, L in {14, 18} (see split_kernels.inc)
#define PHE_PART s16c
#define PHE_PART_G 16
#define PHE_FOR_EACH_L(X) X(14) X(18)
#include "split_kernels.inc"
