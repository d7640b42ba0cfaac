"""Round 6 timing experiment: what the multilevel cycle costs when it runs on a second stream BESIDE k_precond_tile instead of behind it.
The variant build (EULER_HIP_LIB = a library whose k_pcg.hip carries tools/micro/ablations/overlap_cycle.patch, compiled with -DEU_EXP_OVERLAP) launches the cycle on the sums the PREVIOUS iteration left - wrong
results, the right amount of work: an upper bound for what splitting the tile pass (r update + sums | triangular solves) could hide.  Only the projection is run, again and
again, from the same advected velocities (nothing the solve leaves feeds the next one).

    python tools/r06/overlap_experiment.py [N] [workload]     -> microseconds per PCG iteration (100 iterations per solve, tol 0)

Measured (one box, alternating): 8192^2 half tank 620.2 / 643.9 us per iteration as shipped, 630.1 / 620.5 with the cycle beside the tile pass - no difference: a chain of
five small dependent launches does not run faster in the gaps of a pass that fills the chip than behind it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
from euler_amd import scenarios

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
workload = sys.argv[2] if len(sys.argv) > 2 else "half_tank"
sim = ea.Simulation(N, N, precond=ea.PRECOND_IC0_TILE_MG, dot_mode=ea.DOT_TREE, tol=0.0, max_iterations=100)
if workload == "half_tank":
    sim.load_half_tank()
else:
    sim.load_text(getattr(scenarios, workload)(), upscale=True)
dt = sim.timestep(0.1)
for st in (ea.STAGE_ADVECT_MARKERS, ea.STAGE_REFRESH_COUNTS, ea.STAGE_SOURCES, ea.STAGE_EXTRAPOLATE, ea.STAGE_ADVECT_VELOCITY):
    sim.stage(st, dt)
sim.stage(ea.STAGE_PROJECT, dt)      # warm-up
sim.get(ea.F_COUNT)
REPS = 5
t0 = time.perf_counter()
for _ in range(REPS):
    sim.stage(ea.STAGE_PROJECT, dt)
sim.get(ea.F_COUNT)
t = time.perf_counter() - t0
print("%s %d %s: %.1f us per iteration (%d solves of 100 iterations, set-up and the stage's other kernels included)" % (os.environ.get("EULER_HIP_LIB", "product"), N, workload, 1e6 * t / (REPS * 100), REPS))
