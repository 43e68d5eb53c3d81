"""`bench.py --gpus 2` under torch.distributed.run, rehearsed on one GPU (DBENCH_BACKEND=gloo: the ranks share the
card): ONE JSON line with the contract fields from rank 0, the partitioned-join section measured in child processes —
and still one line, with the torch.distributed leg's numbers and an error note, when the child of rank 0 dies the way
a process with a GPU memory fault does."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _launch(port, extra_env):
    env = {**os.environ, "DBENCH_BACKEND": "gloo", "DBENCH_PJOIN_LOG2": "20", "DBENCH_PJOIN_DEADLINE_S": "60", **extra_env}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


@pytest.mark.parametrize("fault", [False, True])
def test_two_ranks_one_line(fault):
    r, lines = _launch(29611 if fault else 29601, {"DBENCH_TEST_CHILD_FAULT": "1"} if fault else {})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["unit"] == "Mrows/s"
    pj = d["pjoin"]
    host = pj["torch_distributed_host"]
    assert host["matches_equal_single_gpu"] and host["ms_per_step"] > 0
    if fault:
        assert "error" in pj and "exit code" in pj["error"], pj
    else:
        assert "error" not in pj, pj


def test_a_hung_leg_costs_an_error_entry_and_a_nonzero_child_exit():
    """A multi-GPU leg that never returns (an RCCL collective that hangs cannot be interrupted from Python): every
    child's watchdog prints what has been measured so far and leaves with exit code 3 — a hung process that touched the
    GPU is a failure, not a success — and rank 0 still prints ONE contract line, with the torch.distributed leg's
    numbers and `pjoin.error` naming the leg."""
    r, lines = _launch(29621, {"DBENCH_TEST_CHILD_HANG": "1", "DBENCH_PJOIN_DEADLINE_S": "25"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    pj = d["pjoin"]
    assert "watchdog" in pj.get("error", "") and "never returns" in pj["error"] and pj.get("child_exit_code") == 3, pj
    assert pj["torch_distributed_host"]["ms_per_step"] > 0


def test_gpus_2_without_any_launcher_starts_its_own_ranks():
    """the driver's command form, `python bench.py --gpus N`, with nothing around it: bench.py starts its N ranks as
    child processes itself (rehearsed with two ranks sharing this GPU over gloo) and rank 0's ONE line says n_gpus = 2
    and carries the partitioned-join section; a world that does not match --gpus is refused"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update({"DBENCH_BACKEND": "gloo", "DBENCH_PJOIN_LOG2": "20", "DBENCH_PJOIN_DEADLINE_S": "60"})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    host = d["pjoin"]["torch_distributed_host"]
    assert host["matches_equal_single_gpu"] and host["ms_per_step"] > 0 and "error" not in d["pjoin"], d["pjoin"]
    assert "started ranks 0..1 as child processes" in r.stderr


def test_gpus_2_on_one_gpu_with_rccl_fails_loudly():
    """RCCL needs a device per rank: on this one-GPU box `--gpus 2` must end non-zero with no line at all (never an
    n_gpus = 1 line under a --gpus 2 command)"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box can really run two RCCL ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "DBENCH_BACKEND")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "has no GPU of its own" in r.stderr


def test_the_rccl_legs_of_the_partitioned_join_section_with_one_rank():
    """every line of the section's RCCL branch (the C++ engine's legs and the sub-join sweep behind them) runs on this
    one GPU as a process group of ONE nccl rank — the engine then joins directly, with nothing to exchange, but the
    section's own code (collectives over the group, the sweep's agreement, the fields it fills) is the code N ranks run"""
    script = f"""
import json, sys, types
sys.path.insert(0, {str(ROOT)!r})
import torch, torch.distributed as dist
import bench
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
def barrier():
    dist.barrier()
    torch.cuda.synchronize()
out = {{}}
bench.pjoin_section(types.SimpleNamespace(steps=2, warmup=1, gpus=1, pjoin_child=False), dist, 0, 1, 0, barrier, out)
print(json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
"""
    env = {**os.environ, "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29651",
           "DBENCH_PJOIN_LOG2": "20", "DBENCH_PJOIN_DEADLINE_S": "120", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
           "DBENCH_PJOIN_RERUN_MARGIN": "2.0"}  # (the re-measurement with the sweep's best count always runs)
    env.pop("DBENCH_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    pj = json.loads(lines[0])["pjoin"]
    assert "error" not in pj and "error" not in pj["torch_distributed_host"], pj
    assert pj["ms_per_step"] > 0 and pj["matches_equal_single_gpu"] and pj["checks"]["all_rows_delivered"], pj
    sweep = pj["sub_joins_sweep"]
    assert "error" not in sweep and set(sweep) == {"1", "2", "4"}, sweep
    assert len({v["matches"] for v in sweep.values()}) == 1 and all(v["ms_per_step"] > 0 for v in sweep.values()), sweep
    assert "sub_joins_rerun_error" not in pj, pj
    if "sub_joins_chosen_by" in pj:  # the re-measurement was faster than the default's leg: it is the section's figure now
        assert pj["ms_per_step"] < sweep["2"]["ms_per_step"] and pj["matches"] == sweep["2"]["matches"]
        assert abs(pj["speedup_vs_1gpu"] - pj["single_gpu_ms_per_step"] / pj["ms_per_step"]) < 1e-9
