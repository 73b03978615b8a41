/* ORACLE — TEST INFRASTRUCTURE ONLY. Internal link between plonky2_quotient.c and plonky2_gates.c. */
#ifndef CITY_ORACLE_GATES_INTERNAL_H
#define CITY_ORACLE_GATES_INTERNAL_H
#include "cityoracle.h"
#include "goldilocks.h"
/* constraints of the gates implemented in plonky2_gates.c; -1 = not one of them */
int or_extra_gate_num_constraints(const or_gate *g);
/* unfiltered constraints in push order; consts = the gate's local constants (after the selectors). Returns the count. */
int or_extra_gate_eval(const or_gate *g, const gl2_t *consts, const gl2_t *wires, gl2_t *out);
#endif
