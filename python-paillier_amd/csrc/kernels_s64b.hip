// kernels_s64b.hip — split-modulus kernels for the whole wavefront as one limb group, L in {3, 5} (see split_kernels.inc)
#define PHE_PART s64b
#define PHE_PART_G 64
#define PHE_FOR_EACH_L(X) X(3) X(5)
#include "split_kernels.inc"
