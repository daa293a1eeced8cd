"""Multi-rank path with the REAL kernels: two fresh processes share cuda:0, collectives over gloo
(the boxes of the pool have one GPU; on an 8-GPU node the same code runs over RCCL).  Checks that
ShardedPSF.from_lens(...).psf_volume over 2 ranks equals ONE psf_lr call over the whole grid --
same verified batch-global Newton trip tables, PSFs equal up to LDS-atomic summation order -- and
that `python bench.py --gpus 2` starts its own ranks and exits 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**kw):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    return env


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_ranks_equal_one_call_over_the_whole_grid(world):
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py")],
                              env=_env(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_PORT=port),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=420))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, f"rank failed:\n{so}\n{se[-3000:]}"
    rec = [json.loads(so.strip().splitlines()[-1]) for so, _ in outs]
    r0 = next(r for r in rec if r["rank"] == 0)
    assert r0["max_abs_diff_L"] <= 3e-6 and r0["max_abs_diff_R"] <= 3e-6
    assert r0["tables"] == r0["solo_tables"]
    assert all(r["empty_shard_ok"] for r in rec)
    print(f"{world}-rank vs solo:", r0["max_abs_diff_L"], r0["max_abs_diff_R"], "own tables differ:",
          r0["own_tables_differ"], r0["tables"])


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: spawns 2 ranks, prints ONE JSON
    line carrying both rates, exits 0.  (gloo dry-run backend: both ranks share cuda:0.)"""
    env = _env(SDIRT_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--workload", "c3", "--sustain-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["gather"] is True
    assert res["value"] > 0 and res["value_no_gather"] > 0


@pytest.mark.gpu
def test_bench_config3_with_eight_ranks_on_one_gpu():
    """BASELINE config 3 as the driver would launch it on a node -- `bench.py --gpus 8 --workload c3`:
    8 ranks x 8192 points, 8192 spp, ks 21, pupil broadcast, mask all-reduce, all-gather of the
    65536-point volume to every rank -- on the ONE GPU of this pool (gloo dry-run backend: the
    control path is the node's, the numbers mean nothing).  One JSON line, n_gpus 8, rc 0."""
    env = _env(SDIRT_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2",
                        "--warmup", "1", "--workload", "c3", "--sustain-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["config"]["gather"] is True and res["config"]["name"] == "c3"
    assert res["config"]["points_per_gpu"] == 8192 and res["value"] > 0 and res["value_no_gather"] > 0
    log = os.environ.get("SDIRT_TEST_LOG_DIR")
    if log:
        with open(os.path.join(log, "bench_gpus8_dryrun_c3.log"), "w") as f:
            f.write(lines[0] + "\n")


@pytest.mark.gpu
def test_rccl_backend_accepts_the_collectives_of_the_multi_gpu_path():
    """tests/rccl_single_rank_worker.py: backend "nccl" (= RCCL) with one rank on the one GPU."""
    env = _env(MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_worker.py")], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "rccl single-rank collectives ok" in p.stdout
