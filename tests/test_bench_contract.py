"""CPU: the shape of bench.py's one JSON line (the driver's contract) checked on the line committed
under profiles/ by the last GPU run, and the parts of bench.py that need no GPU."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_config2_bench.json")))
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "GCUPS" and d["data"] == "synthetic" and d["dtype"] in ("int16", "int32")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    # achieved = algorithmic bytes of one launch / measured kernel time
    assert abs(r["achieved"] - r["bytes_alg_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= 0.5 * r["bytes_alg_per_launch"]
    c = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("reference", "port")
    # value = cells of all steps / elapsed: consistent with ms_per_step and the workload
    cells = d["config"]["lq"] * d["config"]["residues_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - cells / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]


def test_bench_help_and_launch_rule_need_no_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout
    # N > 1 without a launcher is refused before anything touches a device
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "torch.distributed.run" in r.stdout
