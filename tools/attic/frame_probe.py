"""Development aid: frames of the 16384^2 dam break (BASELINE configs[3] on one GPU) with every solve converged (multilevel mode, tol 1e-6), for
`rocprofv3 --kernel-trace --stats`: where a frame's time goes once the solves take ~25 iterations instead of 100."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
from euler_amd import scenarios

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE_MG, tol=1e-6, max_iterations=2000).load_text(scenarios.dam_break(), upscale=True)
t0 = time.time()
while time.time() - t0 < 120:
    sim.step()
    st = sim.stats()
    if st.last_pcg_iterations >= 15 * st.last_substeps:      # (the impact, not the trickle of iterations in free fall)
        break
for _ in range(2):
    sim.step()

t0 = time.time()
sub = its = 0
for _ in range(frames):
    sim.step()
    st = sim.stats()
    sub += st.last_substeps; its += st.last_pcg_iterations

dt = time.time() - t0
print("frames %d: %.1f ms per frame, %d substeps, %d iterations, %.3e cells*steps/s" % (frames, 1e3 * dt / frames, sub, its, N * N * frames / dt))
