"""Multi-rank pressure solve (DESIGN.md "Multi-GPU").
CPU: the slab partition logic.  GPU: two / three gloo ranks sharing the test box's single MI355X
run the real kernels through the real communicator callbacks and are compared with a 1-GPU run."""
import json
import os
import subprocess
import sys

import pytest

import ranks

from euler_amd.slab import slab_bands

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_partition_covers_all_bands():
    for nb in (1, 2, 3, 16, 17, 128, 256):
        for n in (1, 2, 3, 4, 8):
            if n > nb:
                continue
            spans = [slab_bands(nb, r, n) for r in range(n)]
            assert spans[0][0] == 0 and spans[-1][1] == nb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(n - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1


def check_same_iterates(f, workload):
    """One frame of a multi-rank run with the single-GPU run's preconditioner against that run: only the dot products are summed
    in another order.  Tolerance: |dp| <= 1e-8 max|p|, velocities 1e-6, identical cell grid, substeps, iteration counts +-1 and
    bit-identical marker arrays - except on the half tank at rest, whose right-hand side lives on two rows and whose 100
    unconverged iterations amplify a dot product's rounding a billionfold: there |dp| <= 1e-6 max|p| and the
    markers within 1e-5 of a cell."""
    sensitive = workload == "half_tank"
    assert f["cells_differing"] == 0 and f["substeps"][0] == f["substeps"][1], f
    assert f["marker_diff"] <= 1e-5 if sensitive else f["markers_equal"], f
    assert f["dp"] <= (1e-6 if sensitive else 1e-8) * max(f["pmax"], 1.0), f
    assert f["du"] < 1e-6 and f["dv"] < 1e-6, f
    assert abs(f["iters"][0] - f["iters"][1]) <= 1, f


def run_workers(nproc, X, Y, workload, frames, coupling, port, extra=(), fusion=False):
    env = dict(os.environ)
    if fusion:      # opt-in: k_search_apply reading the neighbouring slabs' z / s through IPC mappings - a per-handle option the worker sets on every rank
        extra = list(extra) + ["fusion"]
    # (ranks started directly, tests/ranks.py: the environment torch.distributed.run gives them, without the launcher's two seconds per case)
    rc, out, err = ranks.launch(nproc, os.path.join(ROOT, "tests", "slab_worker.py"), [X, Y, workload, frames, coupling] + list(extra), port, env=env)
    assert rc == 0, (out[-1500:], err[-3000:])
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("fusion", [False, True])
@pytest.mark.parametrize("nproc,X,Y,workload,frames,coupling", [(2, 192, 256, "half_tank", 3, 1), (3, 200, 330, "waterfall", 12, 1), (2, 256, 256, "dam_break", 30, 1),
                                                               (4, 256, 512, "dam_break", 30, 0)])
def test_p2p_mailboxes_carry_the_iteration_exchanges(nproc, X, Y, workload, frames, coupling, fusion):
    """csrc/comm_p2p.hip: the three scalar all-reduces and the ghost-row exchange of every PCG iteration as
    direct writes into the peers' IPC-mapped mailboxes (here: 2-4 processes sharing the box's one GPU; on a node
    the same mappings cross xGMI).  The sums are formed in rank order, so every rank holds bit-identical scalars;
    against the 1-GPU run the same tolerances as the host-staged transport apply (exact coupling), and with
    slab-local coupling the replicated state must still agree on every rank."""
    d = run_workers(nproc, X, Y, workload, frames, coupling, 29551, extra=("p2p",), fusion=fusion)
    assert d["p2p_ok"], d["p2p_error"]
    assert d["ranks_agree"]
    assert d["p2p_calls"]["allreduce"] > 0 and d["p2p_calls"]["halo"] > 0
    assert d["calls"]["allreduce"] <= 1 and d["calls"]["halo"] == 0         # nothing per-iteration went through the host (1 = euler_p2p_connect's agreement)
    assert d["calls"]["chain"] == 0       # exact coupling: the band pipeline runs on across the ranks through the mailboxes
    solved = 0
    for f in d["frames"]:
        assert f["finite"], f
        if coupling == 1:
            check_same_iterates(f, workload)
        solved += f["iters"][1] > 0
    assert solved > 0


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,X,Y,workload,frames", [(2, 192, 256, "half_tank", 3), (3, 200, 330, "waterfall", 12), (2, 256, 256, "dam_break", 30)])
def test_exact_coupling_matches_single_gpu(nproc, X, Y, workload, frames):
    """EULER_SLAB_EXACT: same preconditioner, same iterates; only the dot products are summed in a
    different order (per-rank partials + all-reduce).  Tolerance: |dp| <= 1e-8 * max|p|, velocities
    1e-6 absolute, identical cell grid, identical marker arrays, identical iteration counts."""
    d = run_workers(nproc, X, Y, workload, frames, 1, 29531)
    assert d["ranks_agree"] and d["calls"]["chain"] > 0 and d["calls"]["halo"] > 0
    solved = 0
    for f in d["frames"]:
        assert f["finite"], f
        check_same_iterates(f, workload)
        solved += f["iters"][1] > 0
    assert solved > 0


@pytest.mark.gpu
def test_slab_local_preconditioner_converges_to_the_same_answer():
    """EULER_SLAB_LOCAL changes the preconditioner (no coupling across slabs), hence the iterates;
    where PCG converges inside the iteration budget the answers agree to solver tolerance."""
    d = run_workers(2, 128, 128, "half_tank", 3, 0, 29532)
    assert d["ranks_agree"] and d["calls"]["chain"] == 0
    for f in d["frames"]:
        assert f["finite"] and f["cells_differing"] == 0
        if f["residual"][0] <= 1e-6 and f["residual"][1] <= 1e-6:
            assert f["dp"] <= 1e-4 * max(f["pmax"], 1.0), f


def run_rccl_worker(nproc, X, Y, workload, frames, coupling, port, extra=()):
    rc, out, err = ranks.launch(nproc, os.path.join(ROOT, "tests", "rccl_worker.py"), [X, Y, workload, frames, coupling] + list(extra), port)
    assert rc == 0, (out[-1500:], err[-3000:])
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def gpu_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
@pytest.mark.parametrize("workload,X,Y,frames,p2p", [("half_tank", 192, 256, 3, False), ("dam_break", 256, 256, 30, True)])
def test_builtin_rccl_communicator(workload, X, Y, frames, p2p):
    """The library's own RCCL communicator (csrc/comm_rccl.hip) over backend "nccl": as many ranks as the
    box has GPUs (1 on the test box: RCCL refuses two ranks on one device - the communicator code path of
    the PCG driver still runs, every exchange through RCCL on the handle's stream), compared with a plain
    single-GPU run.  The same worker also drives TorchComm's four operations on raw device pointers over RCCL."""
    n = max(1, min(gpu_count(), 4))
    d = run_rccl_worker(n, X, Y, workload, frames, 1, 29541, extra=("p2p",) if p2p else ())
    assert d["world"] == n and d["rccl_version"] > 0
    assert d["torch_transport_ok"], d["torch_transport_error"]
    assert d["ranks_agree"] and d["calls"]["allgather"] > 0
    if p2p:      # bench.py's configuration: mailboxes for the per-iteration exchanges, RCCL underneath for the bulk transfers
        assert d["p2p_ok"], d["p2p_error"]
        assert d["p2p_calls"]["allreduce"] > 0 and d["p2p_calls"]["halo"] > 0 and d["calls"]["allreduce"] <= 1
    else:
        assert d["calls"]["allreduce"] > 0 and d["calls"]["halo"] > 0
    solved = 0
    for f in d["frames"]:
        check_same_iterates(f, workload)
        solved += f["iters"][1] > 0
    assert solved > 0


@pytest.mark.gpu
def test_weak_scaling_workload_local_coupling_is_the_reference_preconditioner():
    """bench.py --gpus N stacks one closed dam-break tank per row slab; the slab cuts run through the solid walls
    between the tanks, where IC(0) has no coupling to lose.  So EULER_SLAB_LOCAL (slabs sweep concurrently) produces
    the 1-GPU iterates there: same tolerances as exact coupling, although no hand-off row is ever forwarded."""
    d = run_workers(2, 256, 512, "stacked_dam_break_2", 30, 0, 29561, extra=("p2p",))
    assert d["p2p_ok"] and d["ranks_agree"] and d["calls"]["chain"] == 0
    solved = 0
    for f in d["frames"]:
        assert f["finite"], f
        check_same_iterates(f, "stacked_dam_break_2")
        solved += f["iters"][1] > 0
    assert solved > 0


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_multi_rank_contract(scaling):
    """bench.py's N > 1 path end to end as the driver launches it (torch.distributed.run, one process per rank) - here two
    gloo ranks sharing the box's GPU, exchanges over the mailboxes + torch callbacks: ONE JSON line on stdout (rank 0),
    the contract's keys, whole-job value, the transports named, the workload of the chosen scaling."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", EULER_DIST_BACKEND="gloo", EULER_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "256",
           "--comm", "torch", "--p2p", "--scaling", scaling, "--max-preroll", "60", "--strong-size", "512", "--cpu-sample-size", "512"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # nothing but the JSON line reaches stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == scaling and d["value"] > 0 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0          # rank 0 times the CPU path beside it at every N
    assert len(lines[0]) < 8192                                                          # the driver keeps a bounded tail of stdout (round 3: 25.7 KB came back unparsed)
    with open(os.path.join(ROOT, d["full"])) as f:                                       # everything else: the full object beside the line
        full = json.load(f)
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and full["n_gpus"] == 2
    if scaling == "weak":          # the default N > 1 run also measures the strong-scaling unit (BASELINE configs[3]'s scenario; here at 512^2)
        st = d["summary"]["strong_512_dam_break"]
        assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["value"] > 0 and st["pcg_iterations"] > 0, st
        assert st["balance_max_over_mean"] >= 1.0
        st = full["strong_512_dam_break"]
        assert st["balance"]["partition"].startswith("fluid-balanced") and len(st["balance"]["rows_per_rank"]) == 2, st["balance"]
        assert st["converged_frames_multilevel"] is None      # (--p2p here: the multilevel mode's converged frames need the default transport, test_slab_rows.py)
    assert d["config"]["grid"] == ([256, 512] if scaling == "weak" else [256, 256])
    assert "peer-to-peer mailboxes" in d["config"]["parallelism"] and "2 row slabs" in d["config"]["parallelism"]
    assert "EVERY stage decomposed" in d["config"]["parallelism"]        # true row slabs are the N > 1 default
    assert abs(d["value"] - d["config"]["grid"][0] * d["config"]["grid"][1] * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0
