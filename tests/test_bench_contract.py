"""CPU: the shape of bench.py's one JSON line (the driver's contract) checked on the line committed
under profiles/ by the last GPU run, and the parts of bench.py that need no GPU."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _check_block(b, n_gpus=1):
    r = b["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    # achieved = algorithmic bytes of one launch / measured kernel time
    assert abs(r["achieved"] - r["bytes_alg_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= 0.5 * r["bytes_alg_per_launch"]
    assert "workload" in b["config"] and "model" not in b["config"]
    # value = cells of the WHOLE database * steps / elapsed: consistent with ms_per_step and the workload
    cells = b["config"]["lq"] * b["config"]["residues_total"]
    assert abs(b["value"] - cells / (b["ms_per_step"] * 1e-3) / 1e9) < 0.01 * b["value"]
    assert b["unit"] == "GCUPS" and b["dtype"] in ("int16", "int32", "f16", "f16+int16")
    assert r["kernel_ms"] * r.get("launches_per_step", 1) <= b["ms_per_step"] * 1.02
    assert b.get("verify", {"ok": True})["ok"] is True


def test_committed_bench_line_has_the_contract_fields():
    """The default one-GPU line: headline = config 4's ONE 10M-sequence database whole on the GPU -- the workload
    `--gpus N` deals over N ranks -- and one block per configuration 3, 2, 4 (share), 4 with relatives, 5, 5 stress."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict), ("configs", dict)):
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["scaling"] == "strong" and d["higher_is_better"] is True
    assert d["unit"] == "GCUPS" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert d["config"]["workload"].startswith("config 4:") and "ONE" in d["config"]["workload"]
    assert d["config"]["n_seqs"] == 10000000 and d["config"]["lq"] == 3000
    assert d["roofline"]["launches_per_step"] == 48 and d["roofline"]["traffic_source"].endswith("#config4_whole_f16")
    _check_block(d)
    assert set(d["configs"]) == {"2", "3", "4", "4_relatives", "4_whole", "5", "5_stress", "peptides"}
    names = {"5_stress": "5 (stress variant)", "4_relatives": "4 (relatives)", "4_whole": "4"}
    for k, b in d["configs"].items():
        assert b["config"]["workload"].startswith("config %s:" % names.get(k, k))
        _check_block(b)
    assert d["configs"]["4_whole"]["value"] == d["value"]
    # the stress variant of config 5: every one of its 100 000 sequences re-scored in the `rescore` leg
    cs = d["configs"]["5_stress"]
    assert cs["config"]["n_seqs"] == 100000 and cs["rescore"]["n_rescored"] == 100000 and cs["rescore"]["top_k_equals_main_leg"]
    # both 16-bit forms in config 5's search, each part with its own share of the cells and of the issue peak
    r5 = d["configs"]["5"]["roofline"]["binding_roof"]
    assert d["configs"]["5"]["dtype"] == "f16+int16" and abs(r5["cells_share"] + r5["other_kernel"]["cells_share"] - 1.0) < 1e-3
    assert 0.5 < r5["frac_of_issue_peak"] <= 1.0 and 0.5 < r5["other_kernel"]["frac_of_issue_peak"] <= 1.0
    # config 5 as BASELINE names it: one GPU's share of the 10M-sequence shape, and a leg that really re-scores
    c5 = d["configs"]["5"]
    assert c5["config"]["n_seqs"] == 1250000 and c5["rescore"]["n_rescored"] > 0 and c5["rescore"]["top_k_equals_main_leg"]
    assert c5["rescore"]["kernel_ms"]["rescore"] > 0 and c5["rescore"]["options"] == {"wide16": 0}
    # config 4's share with a family of relatives: the f16 flag-and-re-run route is really taken, first search and after
    cr = d["configs"]["4_relatives"]
    assert cr["config"]["n_seqs"] == 1250000 and cr["first_search"]["n_rescored"] > 0 and cr["first_search"]["cell_form"] == 2
    assert cr["steady_state"]["n_rescored"] == cr["first_search"]["n_rescored"] and cr["steady_state"]["rescore_ms"] > 0
    assert cr["kernel_ms"]["rescore"] > 0 and cr["verify"]["ok"] is True
    # short sequences: the cost model's engine choice, timed by the driver like everything else
    pe = d["configs"]["peptides"]
    assert pe["config"]["engine"] == "systolic" and pe["dtype"] == "f16" and pe["roofline"]["binding_roof"]["instr_per_cell"] == 4.25
    assert pe["value"] > 6000
    for b in (d, d["configs"]["2"], d["configs"]["3"], d["configs"]["4"]):
        assert b["dtype"] == "f16" and b["roofline"]["binding_roof"]["instr_per_cell"] == 4.25
        assert b["roofline"]["traffic"] is None or b["roofline"]["traffic_source"].startswith("profiles/traffic.json@sha256:")
    c = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample", "cpu_model", "one_thread_gcups")) <= set(c)
    assert c["kind"] in ("reference", "port")
    h = d["host_inclusive"]
    assert h["upload_bytes"] < 1.1 * d["configs"]["2"]["config"]["residues_total"]     # about one byte per residue


def test_every_point_of_the_scaling_curve_is_the_same_workload():
    """value(N) / (N * value(1)) is only a scaling efficiency if N = 1 and N > 1 run the same
    workload.  `bench.py --gpus 1` and the one-rank run of the N > 1 path (SWG_BENCH_FORCE_DIST=1: launcher, RCCL
    communicator, all-reduce merge) on the same box name the same database and agree within 1 %."""
    one = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    dist = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_dist_rehearsal_1gpu.json")))
    for k in ("workload", "lq", "n_seqs", "residues_total", "matrix"):
        assert one["config"][k] == dist["config"][k], k
    assert one["metric"] == dist["metric"] and one["scaling"] == dist["scaling"] == "strong"
    assert one["steps"] == dist["steps"] and one["warmup"] == dist["warmup"]
    assert abs(one["value"] - dist["value"]) < 0.01 * one["value"]
    assert one["roofline"]["launches_per_step"] == dist["roofline"]["launches_per_step"] == 48
    # and the code path: both go through run_config(..., sharded=True) on config 4's full size
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count('run_config(env, SHARDED, K, W, sharded=True, n_override=CONFIGS[SHARDED]["n_full"], cpu_leg=True)') == 1
    assert "run_config(env, cnum, K, W, sharded=True, n_override=n_full, cpu_leg=True)" in src


def test_bench_help_and_launch_rule_need_no_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout
    # N > 1 without a launcher: bench.py starts its ranks itself, as child processes of a parent that has not
    # touched torch or HIP -- the command it would run (nothing is launched here)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--spawn-dry-run"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout)["spawn"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    # under a launcher (RANK / WORLD_SIZE set) it does not spawn again
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"RANK" not in os.environ and "WORLD_SIZE" not in os.environ' in src
