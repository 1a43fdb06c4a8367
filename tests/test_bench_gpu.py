"""bench.py's data-parallel code paths on the MI355X (one GPU is all a test box has):
* `--rehearse-dp`: process group over RCCL (backend "nccl") with ONE rank, the three-graph step with the bucketed all-reduce issued between
  the replays under capture_error_mode="thread_local" — and the eager form with the overlapped exchange (`--no-graph`);
* `--gpus 2 --backend gloo`: the launcher starts two ranks that share the card, rank 0's line says n_gpus = 2.
Small model (2 layers, 4 videos × 3 clips) so each run takes seconds; the JSON contract of the line is checked as the driver reads it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--layers", "2", "--batch", "4", "--clips", "3", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"]


def _bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, *extra], capture_output=True, text=True, timeout=280,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


@pytest.mark.timeout(290)
@pytest.mark.parametrize("mode", [[], ["--no-graph"]])
def test_rehearse_dp_over_rccl_one_rank(mode):
    rec, err = _bench("--rehearse-dp", *mode)
    assert rec["n_gpus"] == 1 and rec["rccl_ranks"] == 1 and rec["exchange"]["backend"] == "nccl"
    assert rec["exchange"]["buckets"] >= 1 and rec["exchange"]["allreduce_bytes_per_step"] > 1e7
    assert rec["config"]["degraded"] is None
    assert ("hipGraph replay (fwd + text-side bwd" in rec["config"]["launch"]) == (not mode)
    assert rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["scaling"] == "weak" and rec["higher_is_better"] is True
    assert "AccumulateGrad node's stream does not match" not in err


@pytest.mark.timeout(290)
def test_gpus_flag_starts_two_ranks_on_the_card():
    rec, _ = _bench("--gpus", "2", "--backend", "gloo")
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["config"]["parallelism"] == "dp2"
    assert rec["config"]["global_batch"] == 8 and rec["config"]["gpus_requested"] == 2
    assert rec["exchange"]["ms_per_step_without_exchange"] is not None


@pytest.mark.timeout(290)
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_single_gpu_line_has_the_contract_fields(precision):
    rec, err = _bench("--precision", precision)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in rec, k
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["warmup"] == 2 and rec["dtype"] == precision and rec["vs_baseline"] is None
    assert rec["config"]["launch"] == "hipGraph replay" and rec["config"]["degraded"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "attention")) <= set(rec["roofline"])
    assert rec["config"]["mode"] == precision
    assert "AccumulateGrad node's stream does not match" not in err
