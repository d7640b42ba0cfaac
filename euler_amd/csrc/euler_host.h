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
#define EULER_RNG_SEED 0x9bd185c449534b91ull /* main.c:204 */
#ifdef __cplusplus
}
#endif
#endif
