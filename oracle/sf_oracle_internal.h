/*
 * sf_oracle_internal.h -- helpers shared by the oracle's translation units (CPU ORACLE, test
 * infrastructure, NOT product code; see sf_oracle.h).
 */
#ifndef SF_ORACLE_INTERNAL_H
#define SF_ORACLE_INTERNAL_H
#include <stdint.h>

#define SFO_LANES 256 /* virtual lanes of the canonical block reduction */

uint64_t sfo_mix(uint64_t z);
double   sfo_block_sum(const double* x, int n);
float    sfo_residual2(const float c[12], const float* p, const float* q);
int      sfo_finite3(const float* p);
void     sfo_quat_to_R(const double q[4], double R[9]);
void     sfo_R_to_quat(const double R[9], double q[4]);
int      sfo_pnp_solve6(const double ne[28], double lambda, double d[6]);

#endif
