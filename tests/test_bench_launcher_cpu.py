"""bench.py's launcher on CPU: `--gpus N` without a rendezvous in the environment must start N fresh ranks itself (before
anything could touch a GPU), relay rank 0's line, and fail when a rank fails or the rank count is not the one asked for.
The ranks run `--rehearse-launch`: process group (gloo), row-block shard plan, one gather, scanline order -- no render."""
import json
import os
import subprocess
import sys

from tests.conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTMI_DIST_BACKEND="gloo", **env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_2_spawns_two_ranks_and_relays_rank0_line():
    p = _run(["--gpus", "2", "--rehearse-launch", "--steps", "1", "--warmup", "0"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["ranks"] == 2 and doc["backend"] == "gloo" and doc["rehearsal"] is True


def test_world_8_ragged_shards_and_ranks_without_a_block():
    """The driver's first multi-GPU run is world 8: 1080 rows = 135 blocks of 8 (ranks 0-6 render 17 blocks, rank 7 renders 16),
    and an image of 7 rows leaves ranks 1-7 without a block; rank-major packed gather, scanline order checked on rank 0."""
    for height, want_blocks in ((1080, [17] * 7 + [16]), (7, [1] + [0] * 7)):
        p = _run(["--gpus", "8", "--rehearse-launch", "--rehearse-height", str(height), "--steps", "1", "--warmup", "0"])
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout
        doc = json.loads(lines[0])
        assert doc["n_gpus"] == 8 and doc["ranks"] == 8 and doc["rows"] == height and doc["blocks_per_rank"] == want_blocks


def test_world_8_a_failing_rank_or_a_wrong_rank_count_fails_the_launcher():
    p = _run(["--gpus", "8", "--rehearse-launch", "--rehearse-fail-rank", "5"])
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    p = _run(["--gpus", "8", "--rehearse-launch"], WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr and not p.stdout.strip()


def test_a_failing_rank_fails_the_launcher():
    p = _run(["--gpus", "2", "--rehearse-launch", "--rehearse-fail-rank", "1"])
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_rank_count_mismatch_is_an_error_not_a_one_gpu_line():
    # a rendezvous of 1 rank in the environment but --gpus 2 asked for: refuse instead of measuring one GPU
    p = _run(["--gpus", "2", "--rehearse-launch"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
    assert not p.stdout.strip()


def test_launcher_process_never_imports_torch_or_the_library():
    """The parent of `--gpus N` must not initialise anything: checked on its source, the launch path returns before the
    imports of torch / rtmi_loader."""
    src = open(BENCH).read()
    head = src[:src.index("def workload(")]
    assert "import torch" not in head and "rtmi_loader" not in head
    main_src = src[src.index("def main("):]
    assert main_src.index("launch_ranks(args, argv)") < main_src.index("import torch")
