"""Development aid: frames of BASELINE configs[1] (1024^2 dam break, tile-local mode, resident solver) in the expensive phase, for `rocprofv3 --kernel-trace`:
where the time between the kernels goes (tools/r04/runs/gap_probe.sh reads the trace)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
from euler_amd import scenarios

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE).load_text(scenarios.dam_break(), upscale=True)
n = 0
while n < 200:
    sim.step(); n += 1
    st = sim.stats()
    if st.last_pcg_iterations >= 100 * st.last_substeps and st.last_substeps >= 3:
        break
print("preroll frames", n)
t0 = time.time()
sub = its = 0
for _ in range(frames):
    sim.step()
    st = sim.stats()
    sub += st.last_substeps; its += st.last_pcg_iterations
dt = time.time() - t0
print("MARK frames %d: %.3f ms per frame, %d substeps, %d iterations, %.3f ms per substep" % (frames, 1e3 * dt / frames, sub, its, 1e3 * dt / max(sub, 1)))
