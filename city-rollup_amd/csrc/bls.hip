// libcityprover_hip.so, second translation unit: the BLS12-381 side of the Groth16 wrap -
// G1 / G2 multi-scalar multiplication, the scalar-field NTT, the quotient polynomial and the proof assembly. Compiled in
// parallel with cityprover.hip (the Goldilocks / Plonky2 side); both share core.h.
#include "core.h"
#include "msm.h"
#include "msm.inc"
#include "fr_ntt.h"
#include "fr_ntt.inc"
#include "groth16.inc"
#include "groth16_pack.inc"
