"""Development aid: multilevel iterations per solve on the 16384^2 dam break around impact and on the 4096^2 waterfall, for builds with other cycle parameters."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
from euler_amd import scenarios

res = []
for name, N, scn, pre, frames in (("dam16384", 16384, "dam_break", None, 8), ("dam4096", 4096, "dam_break", None, 30), ("fall4096", 4096, "waterfall", 60, 20)):
    s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE_MG, tol=1e-6, max_iterations=3000, resident=ea.RESIDENT_OFF).load_text(getattr(scenarios, scn)(), upscale=True)
    n = 0
    if pre is None:
        while n < 80:
            s.step(); n += 1
            if s.stats().last_pcg_iterations >= 15 * s.stats().last_substeps:      # (the impact, not the first trickle of iterations in free fall)
                break
    else:
        for _ in range(pre):
            s.step()
    its = sub = 0
    per = []
    for f in range(frames):
        s.step()
        st = s.stats()
        its += st.last_pcg_iterations; sub += st.last_substeps
        per.append(st.last_pcg_iterations)
    res.append("%s %d/%d=%.1f %s" % (name, its, sub, its / max(sub, 1), per[:8]))
    s.close()
print(os.environ.get("EULER_HIP_LIB", "production").split("libeuler_hip_")[-1], " | ".join(res))
