/* euler_host.h — internal declarations shared by the host C file and the HIP driver. */
#ifndef EULER_HOST_H
#define EULER_HOST_H
#include "euler.h"
#ifdef __cplusplus
extern "C" {
#endif
uint32_t euler_rng_next_u32(uint64_t* state);
float    euler_rng_next_float(uint64_t* state);
int      euler_half_tank_grids(int32_t X, int32_t Y, uint8_t* solid, uint8_t* source, uint8_t* sink, uint8_t* fluid);
int      euler_half_tanks_grids(int32_t X, int32_t Y, int32_t tanks, uint8_t* solid, uint8_t* source, uint8_t* sink, uint8_t* fluid);
uint64_t euler_rng_jump(uint64_t state, uint64_t steps);   /* xorshift64* state after `steps` more steps */
int      euler_seed_markers_rows(const uint8_t* fluid, int32_t X, int32_t Y, int32_t row_lo, int32_t row_hi, uint64_t* rng_state,
                                 float* markers_xy, uint32_t* keys, uint64_t cap, uint64_t* n_total, uint64_t* n_kept);
#define EULER_RNG_SEED 0x9bd185c449534b91ull /* main.c:204 */
#ifdef __cplusplus
}
#endif
#endif
