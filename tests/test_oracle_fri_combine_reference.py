"""P7(vi): `fri_combine_initial` pinned against a REFERENCE proof by recovering alpha and zeta.

For query q with point x_q the value opened in the first FRI layer must equal
    alpha^2 * (R0_q(alpha) - O0(alpha)) / (x_q - zeta)  +  (R1_q(alpha) - O1(alpha)) / (x_q - g*zeta)
where R0_q / O0 are the alpha-reductions of the 256 opened leaf values / 256 openings in the order
[constants, sigmas, wires, zs, partial products, quotient chunks] and R1_q / O1 those of the 2 Z
polynomials / their `zs_next` openings. Clearing denominators and treating (zeta, zeta^2) as two
independent unknowns gives one linear equation per query whose coefficients are polynomials in
alpha; any three queries force a 3x3 determinant D(alpha) = 0, and gcd(D_123, D_124) isolates alpha.
Zeta then follows from a 2x2 linear solve, must satisfy w == zeta^2, and all 28 queries must agree.
This pins on real plonky2 output: the batch order of the opened polynomials, the alpha-power and
shift-by-count conventions of ReducingFactor, and g = omega_n for the `next` opening point."""
import json
import os

import pytest

import oracle_lib as O
from proof_format import parse_proof
from reference_challenges import LOG_DEG, ONE, emul, recover_alpha_zeta


@pytest.mark.parametrize("which", range(10))
def test_reference_proof_fri_combine_initial(golden_dir, which):
    from proof_format import reference_proofs
    pf = parse_proof(reference_proofs(golden_dir)[which][1])
    alpha, zeta = recover_alpha_zeta(pf)     # asserts: a single common root alpha, zeta^2 consistency, all 28 queries
    z = zeta
    for _ in range(LOG_DEG):
        z = emul(z, z)
    assert z != ONE                          # zeta is not in the trace subgroup (the prover asserts this)
    assert alpha != (0, 0)
