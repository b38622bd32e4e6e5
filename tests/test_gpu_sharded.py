"""Multi-process tests of the PRODUCT path: two processes, one GPU, gloo - what an 8-GPU node runs
with one device per rank over RCCL.  The work is done by tests/workers/sharded_worker.py and by bench.py itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, world=2, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("factor", [0, 4])
def test_sharded_batched_render_and_optimisation_two_processes(gpu, factor):
    """(factor 4: a majorant supergrid - two processes run the supergrid tracer's CU-wide workgroups side by side on one device)"""
    r = _torchrun([os.path.join(ROOT, "tests", "workers", "sharded_worker.py")], env_extra={"DRT_TEST_FACTOR": str(factor)})
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-6000:]
    assert "SHARDED_WORKER_OK" in r.stdout


def test_bench_two_ranks_on_one_device(gpu):
    """bench.py's N > 1 path (image tiles dealt to the ranks, one all-reduce per step), smoke-sized."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--res", "32",
                   "--film", "64", "--spp", "4", "--no-cpu-baseline", "--no-extra-configs"],
                  env_extra={"DRT_BENCH_SAME_DEVICE": "1", "DRT_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-6000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["roofline"]["frac"] > 0
    ar = out["allreduce"]                                   # the moved size of the (compacted) gradient all-reduce
    assert ar["mode"] in ("dense", "compact") and 0 < ar["MiB"] <= ar["of_MiB"] and 0 < ar["active_fraction"] <= 1


def test_rccl_backend_runs_the_gradient_collectives_on_one_rank(gpu):
    """The `nccl` (= RCCL) code path of bench.py / distributed.py, executed: communicator init on the device and the uint8
    MAX + float SUM collectives of the compacted / dense gradient all-reduce, in a world of one rank (a single-GPU box
    cannot do more; the multi-rank logic is covered on gloo by the world-2 and world-8 tests)."""
    r = _torchrun([os.path.join(ROOT, "tests", "workers", "nccl_smoke_worker.py")], world=1, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-6000:]
    assert "NCCL_SMOKE_OK" in r.stdout


def test_config4_gradient_exchange_at_its_size_on_rccl_one_rank(gpu):
    """Config 4's 2 GiB gradient buffer (512^3 x 4 floats) through gradient_support, the pack / unpack kernels and ONE RCCL collective in a
    world of one rank: the buffer comes back bit for bit, one collective, and the step's timings are printed."""
    r = _torchrun([os.path.join(ROOT, "tests", "workers", "nccl_config4_worker.py")], world=1, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-6000:]
    assert "CONFIG4_ALLREDUCE_OK" in r.stdout
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["flat_MiB"] == 2048.0 and out["collectives"] == 1 and out["mode"] in ("compact", "dense")
