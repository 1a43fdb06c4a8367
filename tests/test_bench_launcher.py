"""``bench.py --gpus N`` without a launcher must really start N ranks (the flag used to be parsed and ignored).
CPU check of the entry the driver uses: the parent starts N fresh processes before touching any GPU API, every rank joins the
process group (gloo here, RCCL on the node), the product's GradReducer SUM-all-reduces a small arena, rank 0's single JSON line is
relayed and carries the number of ranks that really joined; a failing rank makes the whole command fail."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", *extra],
                          capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)


def test_gpus_flag_starts_that_many_ranks():
    r = _run()
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["config"]["gpus_requested"] == 2
    assert rec["config"]["sum_ok"] is True and rec["config"]["buckets"] >= 2
    assert rec["config"]["allreduce_bytes_per_step"] >= 5 * 64 * 33 * 4


def test_a_failing_rank_fails_the_command():
    r = _run("--dry-run-fail-rank", "1")
    assert r.returncode != 0


def test_parent_makes_no_gpu_call():
    """the launcher branch must not import torch.cuda state or the kernel library: it only spawns (checked structurally)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def launch_ranks")]
    assert "import torch" not in head.replace("import torch.distributed", "")
    body = src[src.index("def launch_ranks"):src.index("# ------------------------------------------------------------------------------------------------ worker")]
    assert "torch" not in body and "svpc_amd" not in body


def test_committed_parity_records_are_readable_whether_fresh_or_stale(monkeypatch):
    """bench.py quotes the committed parity records in its JSON line.  A record is either "stale" (recorded on other sources) or read —
    and a FRESH record must not crash the reader: metadata keys ("_sources_sha16", "_attention_ab_vs_fp64") and the 64-video rows of the
    config-5 record have no mode field (round 5: the first fresh record of the round crashed the bench line with an IndexError)."""
    sys.path.insert(0, ROOT)
    import bench
    for fresh in (False, True):
        if fresh:      # make every record count as recorded on the current sources
            def load(path, _orig=bench._load_record):
                with open(path) as f:
                    rec = json.load(f)
                rec.pop("_sources_sha16", None)
                return rec, None
            monkeypatch.setattr(bench, "_load_record", load)
        for mode in ("bf16x3", "bf16", "fp32"):
            a, b = bench.recorded_parity(mode), bench.recorded_config5(mode)
            assert a is None or isinstance(a, dict)
            assert b is None or isinstance(b, dict)
            if fresh:
                assert a and "loss_rel_vs_oracle_worst" in a and b and "token_agreement_vs_oracle" in b
