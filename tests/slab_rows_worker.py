"""Worker for tests/test_slab_rows.py: run with torch.distributed.run, N processes (gloo), all sharing cuda:0.
Every rank holds ONE row slab of the job (euler_config.slab_nranks: rows of its bands only, markers inside them with their
global keys) and, as the thing to match, a plain single-GPU run of the same scenario in the same process.  Rank 0 prints one
JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

import euler_amd as ea
from euler_amd import scenarios
from euler_amd.slab import SLAB_LOCAL, RcclComm, TorchComm, attach_p2p


def load(sim, workload):
    if workload == "half_tank":
        sim.load_half_tank()
    elif workload == "closed_box":            # a closed box full of water: no contact with the air, A singular along the region's indicator (coarse modes: k_null_sums / k_null_apply)
        W, H = 40, 30
        rows = ["X" * W] + ["X" + "0" * (W - 2) + "X" for _ in range(H - 2)] + ["X" * W]
        sim.load_text("\n".join(rows) + "\n", upscale=True)
    elif workload.startswith("golden:"):      # one of the reference's five scenario texts, as stored with the golden fixtures
        from golden_util import load as gload, scenario_text
        sim.load_text(scenario_text(gload(workload[7:] + "_frames.npz")), upscale=True)
    else:
        sim.load_text(getattr(scenarios, workload)(), upscale=True)
    return sim


def main():
    X, Y, workload, frames, precond = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    p2p = "p2p" in sys.argv[6:]
    rccl = "rccl" in sys.argv[6:]       # one rank per GPU, the library's own RCCL communicator (1 rank on a 1-GPU box)
    if rccl:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        local = 0
        dist.init_process_group("gloo")
        torch.cuda.set_device(0)
    rank, world = dist.get_rank(), dist.get_world_size()
    rainbow = "rainbow" in sys.argv[6:]       # --rainbow: the dye fields on row slabs (ghost rows of g_r / g_g / g_b, coloured frames)
    maxit = 100
    nu = 0.0
    for a in sys.argv[6:]:
        if a.startswith("nu="):               # the velocity-diffusion extension (euler_config.viscosity) on row slabs
            nu = float(a[3:])
    for a in sys.argv[6:]:
        if a.startswith("maxit="):            # lift the reference's iteration cap: solves run to tolerance (the coarse-correction modes: tolerance parity between 1 and N ranks)
            maxit = int(a[6:])
    ref = ea.Simulation(X, Y, device=local, dot_mode=ea.DOT_TREE, precond=precond, rainbow=rainbow, max_iterations=maxit, viscosity=nu)       # single GPU, the whole grid
    if not any(a.startswith("load=") for a in sys.argv[6:]):
        load(ref, workload)
    slab = (rank, world)
    for a in sys.argv[6:]:
        if a.startswith("bands="):            # an explicit partition, "0-1,1-3,3-8": band ranges per rank (euler_config.slab_band_lo / hi)
            lo, hi = a[6:].split(",")[rank].split("-")
            slab = (rank, world, int(lo), int(hi))
    if "noexchange" in sys.argv[6:]:
        os.environ["EULER_TORCH_NO_EXCHANGE"] = "1"      # the communicator without the fused operation: the library's halo + all-gather fallback
    for a in sys.argv[6:]:
        if a.startswith("caps="):             # tiny exchange capacities (dt-chain candidates, deletions) to drive the overflow path
            os.environ["EULER_SLAB_CAPS"] = a[5:]
    sim = ea.Simulation(X, Y, device=local, dot_mode=ea.DOT_TREE, precond=precond, slab=slab, rainbow=rainbow, max_iterations=maxit, viscosity=nu)   # one slab
    for a in sys.argv[6:]:
        if a.startswith("split="):            # multilevel mode: the level the ranks all-gather (EULER_OPT_MG_SPLIT_LEVEL: the split cycle on grids this small)
            sim.set_option(ea.OPT_MG_SPLIT_LEVEL, int(a[6:]))
    comm = RcclComm(sim, SLAB_LOCAL) if rccl else TorchComm(sim, SLAB_LOCAL)
    out = {"world": world, "frames": []}
    if p2p:
        out["p2p_ok"] = attach_p2p(sim)
    snap_save = snap_load = None
    for a in sys.argv[6:]:
        if a.startswith("save="):
            snap_save = a[5:]
        if a.startswith("load="):
            snap_load = a[5:]
    if snap_load and "loadfail" in sys.argv[6:]:
        # a part file is missing / corrupt for ONE rank: euler_load_state (collective) must come back with an error on EVERY rank - nobody waits
        # in the restore's all-reduce for a rank that has already left
        failed, msg = False, ""
        try:
            sim.load_state(snap_load)
        except ea.EulerError as e:
            failed, msg = True, str(e)
        agg = [None] * world
        dist.all_gather_object(agg, [failed, msg])
        if rank == 0:
            print(json.dumps({"world": world, "loadfail": {"failed": [a[0] for a in agg], "msg": [a[1] for a in agg]}}))
        dist.barrier()
        dist.destroy_process_group()
        return
    if snap_load:                             # resume a job that another set of ranks (another partition) saved; the single-GPU run resumes its own file
        sim.load_state(snap_load)
        ref.load_state(snap_load + ".ref")
    else:
        load(sim, workload)
    lo, hi = sim.slab_rows()
    out["rows"] = [lo, hi]
    if "events" in sys.argv[6:]:
        # Drive the dt chain (main.c:497-501) across the ranks: a strong diagonal flow towards the walls inside two patches - one in the
        # lowest slab, one straddling the first slab boundary - makes markers cross a cell edge and THEN hit a solid, which shortens dt
        # for every LATER marker of the reference's array, wherever it lives.  One substep: the markers move before any dot product.
        u = np.zeros((Y, X), np.float32); v = np.zeros((Y, X), np.float32)
        from euler_amd.slab import slab_bands
        edge = 64 * slab_bands((Y + 63) // 64, 0, world)[1]            # the first slab boundary
        for (y0, y1) in ((2, 40), (edge - 20, edge + 20)):
            u[y0:y1, 1:X // 3] = -7.0       # (from the face between the wall and the first water column on: the flow runs INTO the wall)
            v[y0:y1, 1:X // 3] = -5.0
        ref.set(ea.F_U, u); ref.set(ea.F_V, v)
        sim.set(ea.F_U, u[lo:hi]); sim.set(ea.F_V, v[lo:hi])          # collective: ghost rows follow
        if "deletions" in sys.argv[6:]:
            # ... and a block of sink cells in the water, across the slab boundary: every marker inside is deleted by the next
            # refresh_marker_counts (main.c:109-112), the swap-with-last order re-keys the survivors from the back of the array
            sink = ref.get(ea.F_SINK).copy()
            sink[edge - 6:edge + 6, X // 2:X // 2 + 12] = 1
            sink[8:14, X - 30:X - 20] = 1
            ref.set(ea.F_SINK, sink)
            sim.set(ea.F_SINK, sink[lo:hi])
        if "overflow" in sys.argv[6:]:
            # more deletions than the (shrunken) exchange capacity: EVERY rank must come back with the error - nobody hangs in the next
            # exchange, nobody carries on with a truncated list
            failed, msg = False, ""
            try:
                sim.substep(sim.timestep(0.1))
                sim.substep(sim.timestep(0.1))
            except ea.EulerError as e:
                failed, msg = True, str(e)
            agg = [None] * world
            dist.all_gather_object(agg, [failed, msg])
            if rank == 0:
                print(json.dumps({"world": world, "overflow": {"failed": [a[0] for a in agg], "msg": [a[1] for a in agg]}}))
            dist.barrier()
            dist.destroy_process_group()
            return
        n0 = int(ref.stats().n_markers)
        dt_r, dt_s = ref.timestep(0.1), sim.timestep(0.1)
        ref.substep(dt_r); sim.substep(dt_s)
        m, k, rm = sim.get(ea.F_MARKERS), sim.get(ea.F_MARKER_KEYS), ref.get(ea.F_MARKERS)
        ev = {"dt": [dt_r, dt_s], "dt_events": [int(ref.stats().marker_dt_events), int(sim.stats().marker_dt_events)],
              "n_markers": [int(ref.stats().n_markers), int(sim.stats().n_markers), n0],
              "markers_at_keys": bool(np.array_equal(m.view(np.uint32), rm[k].view(np.uint32))),
              "mismatch": [int((m.view(np.uint32) != rm[k].view(np.uint32)).any(axis=1).sum()), len(k),
                           int(k[(m.view(np.uint32) != rm[k].view(np.uint32)).any(axis=1)].min()) if (m.view(np.uint32) != rm[k].view(np.uint32)).any() else -1,
                           float(np.abs(m - rm[k]).max()) if len(k) else 0.0],
              "count_differ": int((sim.get(ea.F_COUNT) != ref.get(ea.F_COUNT)[lo:hi]).sum())}
        bad = np.nonzero((m.view(np.uint32) != rm[k].view(np.uint32)).any(axis=1))[0]
        ev["bad_detail"] = [[int(i), int(k[i]), m[i].tolist(), rm[k[i]].tolist()] for i in bad[:6]] + [len(m), int(ref.stats().n_markers)]
        agg = [None] * world
        dist.all_gather_object(agg, ev)
        ev["bad_detail_per_rank"] = [a["bad_detail"] for a in agg]
        ev["markers_at_keys"] = all(a["markers_at_keys"] for a in agg)
        keys_all = [None] * world
        dist.all_gather_object(keys_all, k.tolist())
        flat = np.sort(np.concatenate([np.asarray(x, np.int64) for x in keys_all]))
        ev["keys_are_a_permutation"] = bool(len(flat) == len(rm) and np.array_equal(flat, np.arange(len(rm))))
        ev["mismatch_per_rank"] = [a["mismatch"] for a in agg]
        ev["count_differ"] = max(a["count_differ"] for a in agg)
        out["events"] = ev
        frames = 0
    free0, total = torch.cuda.mem_get_info()
    stages = "stages" in sys.argv[6:]         # euler_stage on the slab handle (collective): the six stages of a substep one by one against the single GPU's substep
    for f in range(frames):
        if stages:
            dt_r, dt_s = ref.timestep(0.1), sim.timestep(0.1)
            assert dt_r == dt_s, (dt_r, dt_s)
            ref.substep(dt_r)
            for st in range(6):
                sim.stage(st, dt_s)
        else:
            ref.step()
            sim.step()
        if comm.error:
            raise RuntimeError(comm.error)
        sr, ss = ref.stats(), sim.stats()
        d = {}
        for name, fld in (("u", ea.F_U), ("v", ea.F_V), ("p", ea.F_PRESSURE)):
            a, b = sim.get(fld), ref.get(fld)[lo:hi]
            d["d" + name] = float(np.abs(a - b).max())
        d["pmax"] = float(np.abs(ref.get(ea.F_PRESSURE)).max())
        d["count_differ"] = int((sim.get(ea.F_COUNT) != ref.get(ea.F_COUNT)[lo:hi]).sum())
        d["prev_count_differ"] = int((sim.get(ea.F_PREV_COUNT) != ref.get(ea.F_PREV_COUNT)[lo:hi]).sum())
        # markers: the local ones sit, bit for bit, where their keys say in the single-GPU array, inside the own rows
        m, k, rm = sim.get(ea.F_MARKERS), sim.get(ea.F_MARKER_KEYS), ref.get(ea.F_MARKERS)
        ok = len(m) == len(k) and (len(k) == 0 or int(k.max()) < len(rm))
        d["markers_at_keys"] = bool(ok and np.array_equal(m.view(np.uint32), rm[k].view(np.uint32)))
        d["markers_in_rows"] = bool(len(m) == 0 or (np.floor(m[:, 1]).min() >= lo and np.floor(m[:, 1]).max() < hi))
        # ... and all ranks together hold every key exactly once
        keys_all = [None] * world
        dist.all_gather_object(keys_all, k.tolist())
        flat = np.sort(np.concatenate([np.asarray(x, np.int64) for x in keys_all])) if sum(len(x) for x in keys_all) else np.zeros(0, np.int64)
        d["keys_are_a_permutation"] = bool(len(flat) == len(rm) and np.array_equal(flat, np.arange(len(rm))))
        d["keys_cover_own_count"] = bool(len(flat) == int(ss.n_markers) and np.array_equal(flat, np.arange(len(flat))))
        d["n_markers"] = [int(sr.n_markers), int(ss.n_markers), len(m)]
        d["iters"] = [sr.last_pcg_iterations, ss.last_pcg_iterations]
        d["residual"] = [float(sr.last_residual), float(ss.last_residual)]
        d["substeps"] = [1, 1] if stages else [sr.last_substeps, ss.last_substeps]
        d["rng"] = [int(sr.rng_state) == int(ss.rng_state), sr.source_exhausted == ss.source_exhausted]
        d["dt_events"] = [int(sr.marker_dt_events), int(ss.marker_dt_events)]
        if rainbow:                            # the dye: bit-exact while u, v are (no reduction of its own), else within their tolerance
            dd = 0.0
            for fld in (ea.F_DYE_R, ea.F_DYE_G, ea.F_DYE_B):
                a, b = sim.get(fld), ref.get(fld)[lo:hi]
                fl = ref.get(ea.F_COUNT)[lo:hi] > 0
                dd = max(dd, float(np.abs(a - b)[fl].max()) if fl.any() else 0.0)
            d["ddye"] = dd
        if "render" in sys.argv[6:]:          # euler_render on a slab handle is collective and returns the single-GPU frame on every rank
            d["render_equal"] = bool(sim.draw(120, 50) == ref.draw(120, 50) and sim.draw(X, Y) == ref.draw(X, Y))
        # worst over the ranks
        agg = [None] * world
        dist.all_gather_object(agg, d)
        w = dict(agg[0])
        for o in agg[1:]:
            for key in ("du", "dv", "dp", "count_differ", "prev_count_differ") + (("ddye",) if "ddye" in w else ()):
                w[key] = max(w[key], o[key])
            for key in ("markers_at_keys", "markers_in_rows", "keys_are_a_permutation", "keys_cover_own_count") + (("render_equal",) if "render_equal" in w else ()):
                w[key] = w[key] and o[key]
            w["rng"] = [w["rng"][0] and o["rng"][0], w["rng"][1] and o["rng"][1]]
            w.setdefault("local_markers", [agg[0]["n_markers"][2]]).append(o["n_markers"][2])
        out["frames"].append(w)
    if snap_save:
        sim.save_state(snap_save)             # every rank: its part file; rank 0: the manifest too
        if rank == 0:
            ref.save_state(snap_save + ".ref")
    free1, _ = torch.cuda.mem_get_info()
    out["calls"] = comm.counts
    out["split_active"] = sim.get_option(ea.OPT_MG_SPLIT_ACTIVE)
    if rank == 0:
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
