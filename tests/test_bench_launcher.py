"""bench.py's own launcher path without a GPU: `python bench.py --gpus N` with no torchrun environment must
become N ranks (the driver's documented invocation), bind one rank per LOCAL_RANK, and report n_gpus = N.
`--dry-host` runs the original-form model on the host over gloo -- a launcher check, not a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-host", "--steps", "2", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout      # rank 0 prints ONE JSON line and NOTHING else reaches stdout
    _run.last_stderr = r.stderr
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks():
    line = _run(["--gpus", "2"])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 2 * line["config"]["per_gpu_batch"]
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"


def test_single_process_default():
    line = _run([])
    assert line["n_gpus"] == 1


def test_inside_a_torchrun_environment_the_world_size_wins():
    # the driver's N>1 invocation passes --gpus N AND sets WORLD_SIZE: no second spawn
    line = _run(["--gpus", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert line["n_gpus"] == 1


def test_two_ranks_gather_through_the_c_communicator(tmp_path):
    # the head collective of `bench.py --gpus N`: pv_comm_* with a host-memory double of librccl behind the dlopen
    so = os.path.join(str(tmp_path), "librccl_stub.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "helpers", "rccl_stub.c"), "-lrt"])
    line = _run(["--gpus", "2"], {"PV_RCCL_LIB": so})
    assert line["n_gpus"] == 2 and line["config"]["head_collective"] == "pv_comm over librccl_stub.so"
    # the library's banner (RCCL prints one to the C stdout at communicator creation) went to stderr, not behind the JSON line
    assert "RCCL version : stub" in _run.last_stderr


def test_eight_ranks_are_ready_for_config_5(tmp_path):
    """BASELINE.json configs[4] (X3D-L, global batch 256 over the 8 GPUs of a node) cannot be measured by the builder (one GPU
    per box), so everything short of the GPUs is exercised here with EIGHT ranks: spawn, LOCAL_RANK -> device / CPU binding, the
    head collective through the C communicator on a ragged global batch, and the host half of eight X3D-L conversions at 32
    clips per rank started at the same moment (plan statistics identical on every rank)."""
    so = os.path.join(str(tmp_path), "librccl_stub.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "helpers", "rccl_stub.c"), "-lrt"])
    line = _run(["--gpus", "8", "--workload", "x3d_l", "--dry-ragged", "--dry-convert"], {"PV_RCCL_LIB": so})
    assert line["n_gpus"] == 8 and line["config"]["head_collective"] == "pv_comm over librccl_stub.so"
    r = line["readiness"]
    assert r["north_star_config"]["per_gpu_batch"] == 32 and r["north_star_config"]["global_batch"] == 256
    bind = r["binding"]
    assert sorted(b["local_rank"] for b in bind) == list(range(8))
    assert sorted(b["device"] for b in bind) == ["cuda:%d" % i for i in range(8)]
    seen = set()
    ncpu = len(os.sched_getaffinity(0))
    for b in bind:
        assert b["cpus"], b
        if ncpu >= 8:                       # disjoint CPU slices whenever the job has at least one CPU per rank
            assert not (seen & set(b["cpus"])), bind
            seen |= set(b["cpus"])
    rg = r["ragged"]
    assert rg["rows"] == rg["global_batch"] == sum(rg["sizes"]) and max(rg["sizes"]) - min(rg["sizes"]) == 1
    assert rg["equal_to_torch_distributed"] is True
    cv = r["convert"]
    assert cv["fused"] is True and cv["concurrent_ranks"] == 8 and cv["ops"] > 100 and cv["arena_bytes"] > 0
    assert cv["seconds_max"] < 120
