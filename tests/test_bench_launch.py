"""`python bench.py --gpus N` must be startable from a bare shell (no WORLD_SIZE in the environment):
it re-launches itself as N ranks under torch.distributed.run.  Exercised here on CPU with the gloo
backend and a stub timed body (`--stub-cpu`): same spawn path, barrier, max-over-ranks, one JSON line
from rank 0 with n_gpus taken from the process group."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-cpu", "--steps", "4",
                          "--warmup", "1"] + extra, capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout + res.stderr  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    out = _run(["--gpus", "2"])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1
    # max over ranks: rank 1 sleeps 20 ms, rank 0 10 ms
    assert out["ms_per_step"] * 4 >= 19.0


def test_bench_single_rank_needs_no_launcher():
    out = _run(["--gpus", "1"])
    assert out["n_gpus"] == 1


def test_train_workload_two_ranks_gloo():
    """`--workload train_1080p_b4 --gpus 2` (BASELINE config #4 as stated): the harness -- per-rank train step with the
    flat gradient bucket, ONE all-reduce per step, max-over-ranks timing, the all-reduce's share -- on CPU / gloo
    with a stand-in torch-only model (the HIP kernels have no CPU path)."""
    out = _run(["--gpus", "2", "--workload", "train_1080p_b4"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["unit"] == "MP/s"
    ar = out["allreduce"]
    assert ar["collectives_per_step"] == 1 and ar["backend"] == "gloo" and ar["bucket_elements"] == 12 * 32 + 32 + 32 * 3 + 3
    assert 0 < ar["share_of_step"] < 1.5 and out["config"]["global_batch"] == 8
    assert abs(out["value"] - 2 * out["per_gpu_MPps"]) <= 0.2
    one = _run(["--gpus", "1", "--workload", "train_1080p_b4"])
    assert one["n_gpus"] == 1 and one["allreduce"]["collectives_per_step"] == 0


def test_hdrp_u16_workload_launch_path():
    """`--workload hdrp_u16 --gpus 2`: same self-launch / barrier / max-over-ranks skeleton (the timed body needs the GPU)."""
    out = _run(["--gpus", "2", "--workload", "hdrp_u16"])
    assert out["n_gpus"] == 2 and out["workload"] == "hdrp_u16"


def test_traffic_record_is_refused_when_sources_differ(tmp_path, monkeypatch):
    import bench
    rec = [{"workload": "4k", "kernel": "k", "bytes_per_launch": 1, "source_digest": "deadbeef"}]
    d = tmp_path / "profiles" / "r99"
    d.mkdir(parents=True)
    (d / "traffic.json").write_text(json.dumps(rec))
    os.makedirs(tmp_path / "hdrnet_amd" / "csrc")
    unit = tmp_path / "hdrnet_amd" / "csrc" / bench.DIGEST_UNITS[0]
    unit.write_text('#include "common.h"\nx')
    (tmp_path / "hdrnet_amd" / "csrc" / "common.h").write_text("y")
    (tmp_path / "hdrnet_amd" / "csrc" / "unrelated.hip").write_text("z")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got, why = bench.measured_traffic("4k", "k")
    assert got is None and "other kernel sources" in why
    rec[0]["source_digest"] = bench.source_digest()
    (d / "traffic.json").write_text(json.dumps(rec))
    got, why = bench.measured_traffic("4k", "k")
    assert got["bytes_per_launch"] == 1 and why is None
    # a file outside the forward's include closure does not invalidate the record; one inside it does
    (tmp_path / "hdrnet_amd" / "csrc" / "unrelated.hip").write_text("changed")
    assert bench.measured_traffic("4k", "k")[1] is None
    (tmp_path / "hdrnet_amd" / "csrc" / "common.h").write_text("changed")
    got, why = bench.measured_traffic("4k", "k")
    assert got is None and "other kernel sources" in why
