"""Development aid: iterations of the multilevel mode to the reference's tolerance on a fixed state, for builds with other cycle parameters (EULER_HIP_LIB)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
from euler_amd import scenarios

res = []
for name, N, scn in (("tank8192", 8192, None), ("dam2048", 2048, "dam_break"), ("fall2048", 2048, "waterfall")):
    s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE_MG, tol=1e-6, max_iterations=3000, resident=ea.RESIDENT_OFF)
    if scn is None:
        s.load_half_tank()
    else:
        s.load_text(getattr(scenarios, scn)(), upscale=True)
    frames = 3 if scn is None else (45 if scn == "dam_break" else 30)
    its = sub = 0
    t0 = time.time()
    for f in range(frames):
        s.step()
        st = s.stats()
        if st.last_pcg_iterations:
            its += st.last_pcg_iterations; sub += st.last_substeps
    res.append("%s %d/%d=%.1f (%.1fs)" % (name, its, sub, its / max(sub, 1), time.time() - t0))
    s.close()
print(os.environ.get("EULER_HIP_LIB", "production").split("libeuler_hip_")[-1], " | ".join(res))
