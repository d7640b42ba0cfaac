#!/usr/bin/env python3
"""round 4: the resident solver at BASELINE configs[1] (1024^2 dam break in its solving phase): ms per frame and us per iteration, resident f64 / f32 against the multi-kernel tile mode"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import euler_amd as ea
from euler_amd import scenarios

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name, kw in (("multi-kernel f64", dict(resident=ea.RESIDENT_OFF)), ("resident f64", {}), ("resident f32", dict(pcg_precision=ea.PCG_F32))):
    s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, **kw).load_text(scenarios.dam_break(), upscale=True)
    for _ in range(60):
        s.step()
        if s.stats().last_pcg_iterations >= 100:
            break
    s.step()
    st0 = s.stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        s.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st1 = s.stats()
    its = st1.total_pcg_iterations - st0.total_pcg_iterations
    sub = st1.total_substeps - st0.total_substeps
    s.profile_reset(); s.profile_enable(["resident_pcg", "apply_a", "precond_tile"])
    for _ in range(4):
        s.step()
    prof = s.profile(); s.profile_enable([])
    its2 = s.stats().total_pcg_iterations - st1.total_pcg_iterations
    pcg_ms = sum(v[0] for v in prof.values())
    print("%-18s %dx%d: %.2f ms/frame, %d substeps, %d iterations in %d frames; PCG kernels %.1f us/iteration (%s); resident info %s; last residual %.3g"
          % (name, N, N, 1e3 * dt / K, sub, its, K, 1e3 * pcg_ms / max(its2, 1), {k: (round(v[0], 2), v[1]) for k, v in prof.items()}, s.resident_info(), st1.last_residual), flush=True)
    s.close()
