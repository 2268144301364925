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


def test_bench_eight_rank_launch_line(monkeypatch):
    """`--gpus 8` from a bare shell becomes the driver's own command -- python -m torch.distributed.run --nnodes=1
    --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ... -- with dmabuf IPC kept in the
    environment; checked on the command itself (nothing is spawned)."""
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    import subprocess as sp
    monkeypatch.setattr(sp, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    import argparse
    assert bench.self_launch(argparse.Namespace(gpus=8)) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) <= 65535
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # one process per GPU; ranks wrap around on a box with fewer devices
    assert [bench.device_index(r, 8) for r in range(8)] == list(range(8))
    assert [bench.device_index(r, 1) for r in range(8)] == [0] * 8
    assert bench.device_index(3, 0) == 0


def test_bench_eight_ranks_end_to_end_on_gloo():
    """The 8-rank job itself, on CPU: `bench.py --gpus 8` spawns 8 ranks under torch.distributed.run; every rank reads
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher, binds device local_rank % device_count (8 faked
    devices: one each), joins the barrier and the max-over-ranks; rank 0 alone prints the ONE JSON line."""
    os.environ["HDRNET_BENCH_FAKE_DEVICE_COUNT"] = "8"
    try:
        out = _run(["--gpus", "8"])
    finally:
        del os.environ["HDRNET_BENCH_FAKE_DEVICE_COUNT"]
    assert out["n_gpus"] == 8 and out["scaling"] == "weak"
    ranks = sorted(out["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == list(range(8))
    assert [r["local_rank"] for r in ranks] == list(range(8))
    assert [r["device_index"] for r in ranks] == list(range(8))
    assert all(r["world"] == 8 and r["master"].startswith("127.0.0.1:") and r["ipc_legacy"] == "0" for r in ranks)
    assert len({r["master"] for r in ranks}) == 1
    # max over ranks: rank 7 sleeps 80 ms
    assert out["ms_per_step"] * 4 >= 79.0


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
