"""Development aid / record: a long run in the round-4 form of the PCG iteration (A s' formed twice, p eight iterations at a time) against the round-3 form
(EULER_TILE_STORE_AS=1 EULER_P_STEPS=2), digests of every field per frame.  usage: forms_soak.py out.json   (run once per environment, compare the files)"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import euler_amd as ea
from euler_amd import scenarios

out = {}
for name, pc, size, scn, frames, kw in (("tile_1024_dam_break", ea.PRECOND_IC0_TILE, 1024, "dam_break", 150, dict(resident=ea.RESIDENT_OFF)),
                                        ("mg_2048_waterfall_converged", ea.PRECOND_IC0_TILE_MG, 2048, "waterfall", 60, dict(max_iterations=2000)),
                                        ("mg_1024_dam_break_converged", ea.PRECOND_IC0_TILE_MG, 1024, "dam_break", 100, dict(max_iterations=2000)),
                                        ("parity_1024_waterfall", ea.PRECOND_IC0, 1024, "waterfall", 40, {}),
                                        ("parity_1024_dam_break", ea.PRECOND_IC0, 1024, "dam_break", 60, {})):
    s = ea.Simulation(size, size, dot_mode=ea.DOT_TREE, precond=pc, **kw).load_text(getattr(scenarios, scn)(), upscale=True)
    h = hashlib.sha1()
    its = 0
    t0 = time.time()
    for f in range(frames):
        s.step()
        its += s.stats().last_pcg_iterations
        for fld in (ea.F_U, ea.F_V, ea.F_PRESSURE, ea.F_COUNT):
            h.update(np.ascontiguousarray(s.get(fld)).tobytes())
    h.update(np.ascontiguousarray(s.get(ea.F_MARKERS)).tobytes())
    out[name] = {"digest": h.hexdigest(), "pcg_iterations": its, "frames": frames, "markers": int(s.stats().n_markers), "seconds": round(time.time() - t0, 1),
                 "max_p": float(np.abs(s.get(ea.F_PRESSURE)).max()), "max_u": float(np.abs(s.get(ea.F_U)).max())}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out))
