"""bench.py's N > 1 code path run for real with 2 and 3 ranks on ONE GPU: every rank
shares cuda:0, the C distributed store (glx_dist_*) runs over its host-staged transport with
torch.distributed gloo behind the callbacks (dist.comm_for_group), all device work is the
production HIP path.  --verify recomputes a step on an
unpartitioned copy of the graph and requires bit-identical sampling + aggregation from
the partitioned, pipelined path on every rank.  (RCCL itself is exercised with
world-size 1 in test_gpu_sharded.py; N-GPU RCCL runs are the driver's scaling bench.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,features,pipeline", [(2, "replicated", "on"), (2, "sharded", "on"),
                                                     (3, "replicated", "off")])
def test_bench_multi_rank_path_on_one_gpu(world, features, pipeline, tmp_path):
    detail = str(tmp_path / "detail.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--backend", "gloo", "--share-device", "--workload", "tiny", "--batch", "2048",
           "--steps", "3", "--warmup", "2", "--features", features, "--pipeline", pipeline, "--verify",
           "--cpu-baseline", "off", "--detail-out", detail]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints exactly one JSON line
    head = json.loads(lines[0])
    # the stdout line is the compact one (<= 4 KB: the driver parses it from a ~9 KB tail); the full record is beside it
    assert len(lines[0]) <= 4096 and head["detail"] == detail
    res = json.load(open(detail))
    for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup"):
        assert head[k] == pytest.approx(res[k], rel=1e-6), k
    assert head["verified_sharded_equals_unpartitioned"] is True and set(head["placements"]) == set(res["placements"])
    assert head["rccl_ranks"] == 0 and "host-staged" in head["transport"]  # gloo rig: no RCCL communicator in this run
    assert res["n_gpus"] == world and res["verified_sharded_equals_unpartitioned"] is True
    assert res["value"] > 0 and res["scaling"] == "weak"
    # both placements are timed side by side; `value` is the one --features names
    assert res["value_features_sharded"] > 0
    halo = res["halo_exchange_hop2"]
    assert halo["from_replica"] + halo["from_own_shard"] + halo["remote"] == halo["ids"]
    assert halo["remote_distinct"] <= halo["remote"] and halo["hot_rows"] > 0
    # the third placement: no replica of anything -- every remote request row and feature row crosses the transport --
    # timed, verified and described like the others
    pure = res["placements"]["edge_cut_pure"]
    assert pure["value"] > 0 and res["value_edge_cut_pure"] == pure["value"]
    assert res["verified_legs"] == {"features_sharded": True, "features_sharded_speculated": True, "edge_cut_pure": True,
                                    "edge_cut_pure_speculated": True, "edge_cut_pure_design_r": True}
    # the reference's own distributed aggregation (owners reduce, requester folds) as an ablation of the halo exchange
    assert res["placements"]["edge_cut_pure_design_r"]["value"] == res["value_edge_cut_pure_design_r"] > 0
    # the same placements with a speculation ledger: after the first step only the aggregation exchanges counts
    for name in ("features_sharded_speculated", "edge_cut_pure_speculated"):
        leg = res["placements"][name]
        assert leg["value"] > 0 and leg["ledger"]["holding"] == 0
        # the ledger is keyed by a call's position between two confirmations: the pipelined legs issue the sampling of
        # step i + 1 before the confirmation of step i, so their prologue visits positions 2 and 3 once
        assert leg["ledger"]["speculated"] >= 2 * 3, leg["ledger"]
        assert leg["ledger"]["learned"] == (4 if pipeline == "on" else 2), leg["ledger"]
        if pipeline == "on":  # (the flat leg does not count its exchanges)
            assert leg["count_exchanges_per_step"] == 1.0, leg
            assert res["placements"][name[:-11]]["count_exchanges_per_step"] == 3.0
    ph, ps = pure["halo_exchange_hop2"], pure["sampling_exchange_hop2"]
    assert ph["from_replica"] == 0 and ph["remote"] > 0 and ph["bytes_sent"] > 0
    assert ps["from_graph_replica"] == 0 and ps["remote"] > 0
    rep = res["config"]["replicated_per_gpu"]
    assert rep["feature_rows"] > 0 and 0 < rep["feature_row_fraction"] <= 1.0
    if features == "replicated":
        assert res["value_features_replicated"] == res["value"]
        assert "features_replicated placement" in res["config"]["workload"]
        assert "features_replicated placement" in head["config"]["workload"]
    else:
        # `value` = the faster of the placement's two exchange modes; both are reported
        assert res["value"] == max(res["value_features_sharded"], res["value_features_sharded_speculated"])
        assert "features_sharded placement" in res["config"]["workload"]


def test_bench_reports_what_it_measured_when_a_rank_fails_in_a_leg(tmp_path):
    """A rank that fails inside the halo leg leaves its peer in a collective: rank 0's watchdog must still print ONE
    result line -- the placement that did finish, named as such, with the error -- and every process must leave."""
    env = dict(os.environ, GLX_BENCH_FAULT="1:features_sharded")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--backend", "gloo", "--share-device", "--workload", "tiny", "--batch", "2048",
           "--steps", "3", "--warmup", "1", "--cpu-baseline", "off", "--watchdog", "15", "--detail-out", str(tmp_path / "d.json")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    res = json.loads(lines[0])
    assert "error" in res and "features_sharded" in res["error"]
    assert res["value"] == pytest.approx(res["placements"]["features_replicated"]["value"], rel=1e-5) and res["value"] > 0
    assert "features_replicated placement (the only leg that finished)" in res["config"]["workload"]


def test_bench_launches_its_own_ranks_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` with no torchrun around it (how the driver starts the N = 1 run): bench.py re-executes
    itself under torch.distributed.run -- two ranks really run (n_gpus from the communicator, not from the flag)."""
    detail = str(tmp_path / "detail.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--workload", "tiny",
           "--batch", "2048", "--steps", "3", "--warmup", "1", "--cpu-baseline", "off", "--detail-out", detail]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    head = json.loads(lines[0])
    assert head["n_gpus"] == 2 and head["ranks"] == 2 and head["value"] > 0 and len(lines[0]) <= 4096
    assert head["verified_sharded_equals_unpartitioned"] is True  # --verify-sharded auto: the tiny graph fits beside the shards
    assert json.load(open(detail))["n_gpus"] == 2


def test_bench_refuses_more_gpus_than_the_box_has():
    """--gpus 64 on this box: ONE JSON error line, non-zero exit -- never a 1-GPU number under an N-GPU label."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--workload", "tiny", "--steps", "2"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["value"] is None and "--gpus 64" in res["error"] and res["n_gpus"] < 64


def test_bench_c5_hetero_two_ranks_on_one_gpu():
    """BASELINE configs[4] (3 edge types, per-type Topk + Sum) through the N > 1 path: one ShardedStore
    per edge type, verified against unpartitioned copies of the three graphs on every rank."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--backend", "gloo", "--share-device", "--workload", "c5", "--c5-scale", "500",
           "--batch", "1024", "--steps", "2", "--warmup", "1", "--verify", "--cpu-baseline", "off"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["verified_sharded_equals_unpartitioned"] is True and res["value"] > 0


def test_python_api_spmd_mode_two_ranks_on_one_gpu(tmp_path):
    """Graph.init(task_index, task_count) + Graph.sharded_store(): two ranks load their shards of the same TSV
    files and serve each other's requests; every rank's answers equal an unsharded load's (scripts/spmd_pyapi_check.py)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "scripts", "spmd_pyapi_check.py"),
           str(tmp_path), "--share-device"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "spmd ok rank 0" in r.stdout and "spmd ok rank 1" in r.stdout, r.stdout
