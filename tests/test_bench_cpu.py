"""bench.py's launch contract, checked without a GPU: `--gpus N` must yield N ranks or fail loudly - never run one rank and
print `n_gpus: 1` (round 2's bug); and the committed bench line of this round carries the keys the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_gpus_2_yields_a_two_rank_launch_command():
    out = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--print-launch"])
    assert out.returncode == 0, out.stderr
    cmd = out.stdout.split()
    assert "torch.distributed.run" in cmd and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "2", "--steps", "5", "--warmup", "1"]          # the ranks get the same arguments


def test_gpus_2_without_two_gpus_fails_loudly():
    sys.path.insert(0, ROOT)
    import bench
    have = bench.visible_gpu_count()
    if have is None or have >= 2:
        pytest.skip("two GPUs present, or no KFD topology to count them from")
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert out.returncode != 0
    assert "--gpus 2" in out.stderr and "visible" in out.stderr
    assert "n_gpus" not in out.stdout                                        # and certainly no line claiming a result


def test_rank_count_that_disagrees_with_gpus_is_refused():
    out = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and "n_gpus" not in out.stdout


def test_launch_command_helper_is_importable_without_side_effects():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.rank_launch_command(8, ["--gpus", "8"], port=29511)
    assert cmd[0] == sys.executable and "--nproc-per-node=8" in cmd and cmd[-2:] == ["--gpus", "8"] and "29511" in cmd


def test_committed_bench_line_of_this_round():
    """profiles/r04_bench_line.json is what `python bench.py` printed on the MI355X this round."""
    path = os.path.join(ROOT, "profiles", "r04_bench_line.json")
    if not os.path.exists(path):
        pytest.skip("no r04 bench line committed yet")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_cold", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["config"]["distinct_scans"] >= 8 and "model" not in d["config"]
    for name in ("roofline", "roofline_cold"):
        r = d[name]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / r["kernel_us"] / 1e3) / r["achieved"] < 2e-3
    assert d["roofline"]["traffic"] is None or d["roofline"]["traffic"] >= d["roofline"]["algorithmic_bytes_per_launch"]
    # the kNN rate is the searching launch's, not the loop mean's
    n_q = d["config"]["n_q"]
    assert abs(d["knn_mpts_per_s"] - n_q / d["roofline_cold"]["kernel_us"]) / d["knn_mpts_per_s"] < 2e-3
    assert d["points_per_s_loop_mean"] > 1e6 * d["knn_mpts_per_s"]
    for k in ("ms_per_scan_early_exit", "iters_run_early_exit", "lm_iterations_per_s_early_exit", "ms_per_scan_host_buffers_early_exit",
              "ms_per_step_windows", "kernel_us_steady_back_to_back"):
        assert k in d, k
    assert abs(d["lm_iterations_per_s_early_exit"] - 1e3 * d["iters_run_early_exit"] / d["ms_per_scan_early_exit"]) / d["lm_iterations_per_s_early_exit"] < 0.1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 4 and c["unit"] == d["unit"] and "early_exit" in c
    assert abs(d["value"] - d["config"]["lm_iters_per_step"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    # the figure with the map index build charged to every scan, beside `value`
    want = d["config"]["lm_iters_per_step"] * 1e3 / (d["ms_per_step"] + d["map_index_build_ms"])
    assert abs(d["value_with_map_index_build"] - want) / want < 2e-3 and d["value_with_map_index_build"] < d["value"]
    assert "traffic_ratio" in d["roofline"]
    if d["roofline"]["traffic"] is not None:
        assert abs(d["roofline"]["traffic_ratio"] - d["roofline"]["traffic"] / d["roofline"]["algorithmic_bytes_per_launch"]) < 2e-3
    assert "chain" in d and set(("extract_cloud_ms", "downsample_scan_ms", "optimize_ms")) <= set(d["chain"])
