"""bench.py --gpus N starts its own N ranks.  CPU: the ranks rendezvous over gloo,
take their window blocks and rank 0 gathers; no compute (--launch-check), so no GPU is needed."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr
    line = _json_line(r.stdout)
    assert line == {"launch_check": True, "n_gpus": 2, "ranks_seen": 2, "backend": "gloo"}


def test_single_rank_needs_no_launcher():
    r = _run(["--launch-check"])
    assert r.returncode == 0, r.stderr
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_gpus_must_match_world_size():
    # under an external launcher --gpus has to agree with WORLD_SIZE: a silent 1-GPU run is refused
    r = _run(["--gpus", "4", "--launch-check"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_parent_does_not_import_torch_before_spawning():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def launch_ranks")]
    assert "import torch" not in head and "import numpy" not in head
    body = src[src.index("def main()"):]
    assert body.index("launch_ranks(args.gpus") < body.index("import torch")


def test_recorded_traffic_is_refused_when_taken_on_other_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from a file recorded by tools/profile.sh; bench.py reports it only when the file carries the
    digest of the kernel sources as they are now."""
    import importlib
    import bench
    importlib.reload(bench)
    d = bench.source_digest()
    assert len(d) == 16 and d == bench.source_digest()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_digest", lambda: d)
    (prof / "traffic_latest.json").write_text(json.dumps({"hbm_bytes_per_launch": 8.4e9, "tag": "rXX", "source_digest": d}))
    (prof / "traffic_cfg2.json").write_text(json.dumps({"hbm_bytes_per_launch": 1.5e10, "tag": "old", "source_digest": "0" * 16}))
    tr, src = bench.recorded_traffic("cfg3")
    assert tr == 8.4e9 and "rXX" in src
    tr, src = bench.recorded_traffic("cfg2")
    assert tr is None and "other sources" in src
    assert bench.recorded_traffic("cfg4") == (None, None)


def test_job_shard_weak_and_strong():
    """--scaling weak: every rank a full per-GPU batch; --scaling strong: BASELINE's job block-sharded over the ranks
    (cfg3 4096 windows, cfg4 4096 x 10 channels, cfg5 64 windows), blocks contiguous and complete."""
    import bench
    assert bench.job_shard("cfg3", "weak", None, 3, 8) == (3 * 4096, 4096, 8 * 4096)
    assert bench.job_shard("cfg4", "weak", None, 0, 2) == (0, 5120, 10240)
    for cfg, total in (("cfg3", 4096), ("cfg4", 40960), ("cfg5", 64)):
        for n in (1, 2, 3, 8):
            blocks = [bench.job_shard(cfg, "strong", None, r, n) for r in range(n)]
            assert all(b[2] == total for b in blocks)
            assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == total
            assert all(blocks[r + 1][0] == blocks[r][0] + blocks[r][1] for r in range(n - 1))
            assert max(b[1] for b in blocks) - min(b[1] for b in blocks) <= 1
    assert bench.job_shard("cfg5", "strong", 16, 1, 2) == (8, 8, 16)


def test_projection_table_arithmetic():
    """strong_scaling_projection: G GPUs on a job without a collective = one GPU on ceil(W / G) windows;
    efficiency = time(1) / (G * time(G))."""
    import bench
    t = bench.projection_table({1: 1.6, 2: 0.82, 4: 0.44, 8: 0.25}, 4096)
    assert list(t) == ["1", "2", "4", "8"]
    assert [t[g]["windows_per_gpu"] for g in t] == [4096, 2048, 1024, 512]
    assert t["1"]["efficiency_vs_G1"] == 1.0
    assert abs(t["2"]["efficiency_vs_G1"] - 1.6 / (2 * 0.82)) < 1e-12
    assert abs(t["8"]["efficiency_vs_G1"] - 0.8) < 1e-12
    assert bench.projection_table({1: 1.0, 3: 0.4}, 100)["3"]["windows_per_gpu"] == 34


def test_binary_carries_the_digest_of_its_sources(tmp_path):
    """VERDICT r04 #4b: the library names the sources it was built from (rmx_build_info, compiled in by
    __graft_entry__.build), staleness is decided by that digest and not by mtimes, and bench.py refuses a binary whose
    digest is not the tree's."""
    import __graft_entry__ as ge
    ge.build()                            # (a no-op when the binary's digest is the tree's; a minute of hipcc otherwise)
    from radio_mapper_amd import xcorr
    d = ge.source_digest()
    assert len(d) == 16 and int(d, 16) >= 0
    assert ge.binary_digest() == d
    assert xcorr.build_info()["source_digest"] == d and xcorr.build_info()["arch"] == "gfx950"
    assert not ge._stale()
    # a file without the marker, a missing file, a file with another digest: all stale
    (tmp_path / "empty.so").write_bytes(b"\x7fELF nothing here")
    assert ge.binary_digest(str(tmp_path / "empty.so")) is None and ge._stale(str(tmp_path / "empty.so"))
    assert ge._stale(str(tmp_path / "missing.so"))
    (tmp_path / "other.so").write_bytes(b"xx RMX_BUILD_INFO source_digest=0123456789abcdef arch=gfx950\0")
    assert ge.binary_digest(str(tmp_path / "other.so")) == "0123456789abcdef" and ge._stale(str(tmp_path / "other.so"))
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "sys.exit(3)" in src[src.index("build_info = xcorr.build_info()"):src.index("dev = torch.device(\"cuda\", dev_index)")]


def test_recorded_valu_fraction(tmp_path, monkeypatch):
    """roofline.valu (VERDICT r04 #4c): wave-instructions x 2 cycles / (1024 SIMDs x 2.4 GHz) over the launch time."""
    import importlib
    import bench
    importlib.reload(bench)
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_digest", lambda: "a" * 16)
    assert bench.recorded_valu(1.7) is None
    (prof / "pmc_latest.json").write_text(json.dumps({"kernel": "k_win", "tag": "rXX", "source_digest": "a" * 16,
                                                      "counters": {"SQ_INSTS_VALU": 9.03e8}}))
    v = bench.recorded_valu(1.685)
    assert abs(v["issue_ms_at_2cyc"] - 9.03e8 * 2 / (1024 * 2.4e9) * 1e3) < 1e-12
    assert abs(v["frac_of_launch"] - v["issue_ms_at_2cyc"] / 1.685) < 1e-12 and 0.4 < v["frac_of_launch"] < 0.5
    (prof / "pmc_latest.json").write_text(json.dumps({"kernel": "k_win", "tag": "old", "source_digest": "b" * 16,
                                                      "counters": {"SQ_INSTS_VALU": 9.03e8}}))
    assert "stale" in bench.recorded_valu(1.685)


def test_parse_pmc_csv_keeps_full_size_launches(tmp_path):
    """measure_pmc_in_run's parser: counters per launch of k_win, full-size launches only (the parity leg of the child
    run launches the same kernel on a smaller grid); other kernels ignored."""
    import bench
    p = tmp_path / "pmc_counter_collection.csv"
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    def row(d, grid, name, cn, cv):
        return f"{d},{d},1,1,1,1,{grid},7,\"{name}\",512,159008,0,232,0,112,{cn},{cv},0,0"
    for d, v in ((1, 1000.0), (2, 1100.0), (3, 900.0)):
        rows.append(row(d, 131072, "void rmx::k_win<false>(void const*, HIP_vector_type<float, 4u>*)", "FETCH_SIZE", v))
    rows.append(row(4, 16384, "void rmx::k_win<false>(void const*, HIP_vector_type<float, 4u>*)", "FETCH_SIZE", 5.0))   # parity leg
    rows.append(row(5, 131072, "at::native::vectorized_elementwise_kernel<4>", "FETCH_SIZE", 1e9))
    p.write_text("\n".join(rows) + "\n")
    got = bench.parse_pmc_csv(str(p))
    assert got == {"FETCH_SIZE": 1000.0, "launches_sampled": 3}
    assert bench.parse_pmc_csv(str(p), kernel_sub="no_such_kernel") == {}


def test_reference_capture_shapes_are_complete_bench_shapes():
    """`reference_capture_lengths` of the bench line: the reference's two capture lengths (iq_stream_client.py:459,
    buoy_node.py:364) at 3 and 8 buoys -- every entry a full shape (selectable with --config, with its own step counts) whose
    resident input stays far below one GPU's memory."""
    sys.path.insert(0, ROOT)
    import bench
    assert set(bench.CAPTURE_SHAPES) == {"cap8192_b3", "cap8192_b8", "cap16384_b3", "cap16384_b8"}
    for name, c in bench.CAPTURE_SHAPES.items():
        assert bench.CONFIGS[name] is c and name in bench.DEFAULT_STEPS and name in bench.OTHER_STEPS
        assert c["N"] in (8192, 16384) and c["B"] in (3, 8) and c["C"] == 1
        assert c["W"] * c["B"] * c["N"] * 8 < 2 ** 30
    # the BASELINE shapes are untouched by the addition
    assert [k for k in bench.CONFIGS if k.startswith("cfg")] == ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"]
