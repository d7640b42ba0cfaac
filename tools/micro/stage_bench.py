"""Development aid: the marker stages of one 8192^2 half-tank substep, repeated, for `rocprofv3 --kernel-trace --stats`
(EULER_HIP_LIB selects an experimental build of the library)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
workload = sys.argv[2] if len(sys.argv) > 2 else "half_tank"
sim = ea.Simulation(N, N, precond=ea.PRECOND_IC0_TILE, tol=0.0 if workload == "half_tank" else None, max_iterations=20)
if workload == "half_tank":
    sim.load_half_tank()
else:
    from euler_amd import scenarios
    sim.load_text(getattr(scenarios, workload)(), upscale=True)
if os.environ.get("EU_TWO_PASS"):      # (rounds 1-5's separate advection and binning passes)
    sim.set_option(ea.OPT_MARKERS_TWO_PASS, 1)
if os.environ.get("EU_NO_TILE_MAP"):   # (the grid passes visit every tile)
    sim.set_option(ea.OPT_NO_TILE_MAP, 1)
for _ in range(3):
    sim.step()
for _ in range(8):      # whole substeps, stage by stage (what one stage leaves for the next - the lazy flags of euler_sim - stands as in a frame)
    dt = sim.timestep(0.1)
    for st in (ea.STAGE_ADVECT_MARKERS, ea.STAGE_REFRESH_COUNTS, ea.STAGE_SOURCES, ea.STAGE_EXTRAPOLATE, ea.STAGE_ADVECT_VELOCITY, ea.STAGE_PROJECT):
        sim.stage(st, dt)
sim.get(ea.F_COUNT)      # (waits for the stream)
print("markers", sim.stats().n_markers)
