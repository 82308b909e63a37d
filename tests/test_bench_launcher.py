"""CPU: `bench.py --gpus N` starts its own ranks (one child process per GPU through
torch.distributed.run) when called from a plain shell, refuses a launcher that started a different
number of ranks, and keeps the all-gather of the rendered maps inside the timed region.  `--dry`
swaps the HIP render for stand-in maps and RCCL for gloo so the launcher, the rendezvous, the
sharding arithmetic and the collective run here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True,
                          text=True, timeout=timeout)


def _json_line(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("gather,steps,gathers", [("step", 4, 5), ("image", 144, 3)])
def test_bench_starts_its_own_ranks(gather, steps, gathers):
    r = _run(["--dry", "--gpus", "2", "--steps", str(steps), "--warmup", "1", "--gather", gather])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r)
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["steps"] == steps
    assert out["rows_in_order"] and out["gathers"] == gathers          # warm-up + timed region
    # image mode: 72 calls of 2 x 1024 rays fill one 288 x 512 image; the warm-up call is flushed by the fence
    assert out["gathered_rows"][-1] == (2048 if gather == "step" else 2 * 72 * 1024)


def test_strong_scaling_splits_the_rays():
    r = _run(["--dry", "--gpus", "2", "--steps", "2", "--warmup", "0", "--workload", "zest_val_4096x192",
              "--scaling", "strong", "--gather", "step"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r)
    assert out["scaling"] == "strong" and out["rays_per_gpu"] == 2048 and out["gathered_rows"][0] == 4096


def test_rank_count_mismatch_is_an_error():
    r = _run(["--dry", "--gpus", "2", "--steps", "1", "--warmup", "0"],
             env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16", "f16x3"])
def test_bench_json_line_on_the_gpu(mode):
    """The real (non --dry) bench on one GPU, short: ONE JSON line with the contract's keys, the roofline object and
    the other operand types in `modes` - also when the headline mode is f16x3 (the line used to die on a KeyError
    after all the timing was done)."""
    r = _run(["--steps", "5", "--warmup", "2", "--mode", mode, "--no-cpu-baseline"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["dtype"] == mode and out["n_gpus"] == 1 and out["steps"] == 5 and out["value"] > 1e5
    assert out["roofline"]["bound"] == "mfma" and 0.05 < out["roofline"]["frac"] < 1.0
    assert sorted(out["modes"]) == sorted(m for m in ("bf16", "f16", "f16x3") if m != mode)
    assert out["config"]["workload"] == "nsff_static_1024x128"
