"""CPU: bench.py's bookkeeping -- the algorithmic bytes of SURVEY.md section 8d, and that the PMC traffic records
(profiles/traffic.json, keyed by workload) name the kernels bench.py looks for, so `roofline.traffic` of the contract line is a
measurement of THAT kernel on THAT workload or null, never a figure borrowed from another configuration."""
import importlib.util
import json
import os
import re

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_match_survey_8d():
    b = _bench()
    cfg2 = b.algorithmic_bytes_per_shaded_px(K=12, J=128, q=4)
    assert cfg2 == dict(fwd_env=2008, fwd_noenv=472, bwd_sg=2344)
    assert cfg2["fwd_env"] + cfg2["bwd_sg"] == 4352                      # fwd+bwd, trainLight mode
    cfg5 = b.algorithmic_bytes_per_shaded_px(K=24, J=512, q=4)
    assert cfg5["fwd_env"] == 6952 and cfg5["fwd_noenv"] == 808
    assert b.HBM_PEAK_GBPS == 8000.0


def test_traffic_records_are_keyed_by_workload_and_name_the_dispatched_kernels():
    recs = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key, fwd, bwd, alg_fwd, alg_bwd in (("config2_batch16_env", r"::fwd_pk_half_kernel<2, true, true, 3, 6, 16, 2[,>]", r"::sg_bwd_pk_kernel<2, true, true", 616857600, 720076800),
                                            ("config5_batch4_env", r"::fwd_pk_half_kernel<2, true, true, 2, 12, 32, 1[,>]", r"::sg_bwd_pk_kernel<2, true, true, 32[,>]", 2135654400, 2342092800)):
        assert key in recs, key
        names = list(recs[key])
        f = [n for n in names if re.search(fwd, n)]
        g = [n for n in names if re.search(bwd, n)]
        assert f and g, (key, names)
        # measured HBM bytes: never below the algorithmic bytes (minus counter noise), and no gross re-reads
        assert 0.98 * alg_fwd <= recs[key][f[0]]["hbm_bytes"] <= 1.10 * alg_fwd, (key, recs[key][f[0]])
        assert 0.98 * alg_bwd <= recs[key][g[0]]["hbm_bytes"] <= 1.25 * alg_bwd, (key, recs[key][g[0]])


def test_sq_records_feed_the_valu_roofline():
    """profiles/sq.json (tools/pmc_sq.sh): VALU-issue figures of the dispatched kernels per workload; bench.valu_roofline turns the
    record of the dominant kernel into the `roofline_valu` object of the contract line."""
    b = _bench()
    recs = json.load(open(os.path.join(ROOT, "profiles", "sq.json")))
    for key in ("config2_batch16_env", "config5_batch4_env"):
        assert key in recs, key
        v = b.valu_roofline(key, [r"sg_bwd_pk_kernel<"], 0.25)
        assert v is not None and v["bound"] == "valu_issue" and 0.3 < v["frac"] < 1.0, v
        assert abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-3
        f = b.valu_roofline(key, [r"fwd_pk_half_kernel<"], 0.15)
        assert f is not None and 0.3 < f["frac"] < 1.0, f
    assert b.valu_roofline("no_such_workload", [r"sg_bwd_pk_kernel<"], 0.25) is None


def test_sq_records_imply_physical_clocks():
    """Every record of profiles/sq.json: the kernel's cycle count over its traced duration must not exceed the part's maximum clock
    (2.4 GHz, MI355X_MICROARCH.md) -- rounds 3-4 divided by GRBM_GUI_ACTIVE / XCDs, which implied 2.65-10 GHz and under-stated every
    VALU-busy fraction (tools/parse_sq.py) -- and the busy fraction stays inside its two bounds."""
    recs = json.load(open(os.path.join(ROOT, "profiles", "sq.json")))
    n = 0
    for tag, per_kernel in recs.items():
        if tag.startswith("_"):
            continue
        for k, r in per_kernel.items():
            if k.startswith("_"):
                continue
            # (no lower bound: a kernel of one workgroup keeps one shader engine of 32 busy, and the per-engine average says so)
            assert r["effective_clock_GHz"] is None or 0.0 < r["effective_clock_GHz"] <= 2.45, (tag, k, r["effective_clock_GHz"])
            assert 0.0 < r["frac"] <= 1.0, (tag, k, r["frac"])
            if r.get("frac_at_max_clock") is not None:
                assert r["frac_at_max_clock"] <= r["frac"] + 1e-3, (tag, k)
            n += 1
    assert n >= 10


def test_bench_stdout_is_the_json_line_alone():
    """The contract: rank 0 prints ONE JSON line.  Native libraries write to file descriptor 1 behind Python's back -- RCCL prints a five-line
    banner when its first communicator is created, buffered, so it lands AFTER the line at exit (seen in the first round-5 run of the
    RCCL world-of-one leg) -- so bench.py moves descriptor 1 to stderr at start and writes its line to the saved descriptor."""
    import subprocess
    import sys
    code = ("import ctypes, importlib.util, os, sys\n"
            f"spec = importlib.util.spec_from_file_location('b', os.path.join({ROOT!r}, 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "emit = b._claim_stdout()\n"
            "libc = ctypes.CDLL(None)\n"
            "libc.printf(b'native banner, buffered until exit\\n')\n"      # what RCCL does
            "print('python print after the claim')\n"
            "os.write(1, b'raw write to fd 1\\n')\n"
            "emit('{\"metric\": 1}')\n")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-1000:]
    assert p.stdout == '{"metric": 1}\n', repr(p.stdout)
    for needle in ("native banner", "python print after the claim", "raw write to fd 1"):
        assert needle in p.stderr, (needle, p.stderr[-500:])


# --------------------------------------------------------------------------- #
# round 6: the line's size limits, the self-launch, the anchors                 #
# --------------------------------------------------------------------------- #
def check_line_limits(line: str, b=None):
    """What the driver's record keeps of the contract line (Missing 3): an 8.7 KB tail, `parsed.config` without nested
    objects, keys cut at 40 characters, strings at 120.  bench.py promises <= 6 KB, flat scalar `config`, keys <= 40, strings <= 120."""
    b = b or _bench()
    assert "\n" not in line
    assert len(line.encode()) <= b.LINE_LIMIT, len(line.encode())
    out = json.loads(line)
    cfg = out["config"]
    for k, v in cfg.items():
        assert len(k) <= b.KEY_LIMIT, k
        assert v is None or isinstance(v, (bool, int, float, str)), (k, type(v))
        if isinstance(v, str):
            assert len(v) <= b.STR_LIMIT, (k, len(v))
    for obj in ("roofline", "cpu_baseline", "roofline_valu", "eager_gpu_baseline"):
        for k, v in (out.get(obj) or {}).items():
            assert len(k) <= b.KEY_LIMIT, (obj, k)
            assert not isinstance(v, (dict, list)), (obj, k)
            if isinstance(v, str):
                assert len(v) <= b.STR_LIMIT, (obj, k, len(v))
    return out


def test_fit_line_trims_to_the_limit_and_clip_marks_the_cut():
    b = _bench()
    out = {"metric": "m", "value": 1.0, "config": {"workload": "w", **{f"none_{i}": None for i in range(400)}, "kept": 1.5},
           "roofline": {"frac": 0.4}, "roofline_valu": {"x": "y" * 100}, "eager_gpu_baseline": {"value": 35.0}}
    assert len(json.dumps(out)) > b.LINE_LIMIT
    line = b.fit_line(out)
    got = check_line_limits(line, b)
    assert got["config"]["kept"] == 1.5 and got["roofline"]["frac"] == 0.4 and "none_3" not in got["config"]
    assert b.clip("a" * 200) == "a" * 119 + "~" and b.clip("short") == "short" and b.clip(None) is None and b.clip(3.5) == 3.5


def test_flat_config_keys_in_the_source_respect_the_key_limit():
    """Every key bench.py puts into `flat` / `config` literally: <= 40 characters (the GPU test checks the emitted line itself)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    keys = set(re.findall(r'flat\[f?"([A-Za-z0-9_{}]+)"\]', src)) | set(re.findall(r'config\["([A-Za-z0-9_]+)"\]', src))
    i = src.index("        config = {")
    keys |= set(re.findall(r'"([A-Za-z0-9_]+)":', src[i:src.index("        }\n", i)]))
    keys |= set(re.findall(r'\(\("((?:rccl1|cfg\d|obj)_[A-Za-z0-9_]+)", "', src)) | set(re.findall(r'\("((?:rccl1|cfg\d|obj)_[A-Za-z0-9_]+)", "ms_per_step', src))
    assert len(keys) > 40, sorted(keys)
    for k in keys:
        assert len(k.replace("{tag}", "cfg3").replace("{side}", "bwd")) <= 40, k
    for want in ("cfg3_ms", "cfg3_ms_graph", "cfg4_ms", "cfg5_Mpix_per_s", "cfg5_ms", "rccl1_loss_ms", "rccl1_obj_ms", "obj_ms", "strong16_ms", "n_ranks_seen",
                 "scaling_anchor_Mpix_per_s", "Mpix_with_loss", "Mpix_layer_only"):
        assert want in keys, want
    for want in ("ms_per_step_cold", "obj_ms_cold", "cfg5_{side}_frac", "{tag}_{side}_frac", "{tag}_{side}_traffic_ratio"):
        assert want in src, want


def test_self_launch_command_is_the_drivers_form():
    """`python bench.py --gpus 8 ...` outside torchrun re-executes as `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...` (round 5 raised SystemExit instead)."""
    import sys
    b = _bench()
    cmd = b.self_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[:4] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) <= 65535
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    a = b.parse_args(["--gpus", "4", "--per-gpu-batch", "8", "--strong"])
    assert a.gpus == 4 and a.batch == 8 and a.strong and b.parse_args([]).batch == 16 and b.parse_args([]).gpus == 1


def test_help_goes_to_stdout():
    """_claim_stdout() ran before argparse, so --help went to stderr."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "--per-gpu-batch" in p.stdout and "--strong" in p.stdout, (p.stdout[-300:], p.stderr[-300:])


def test_anchor_file_round_trip(tmp_path, monkeypatch):
    b = _bench()
    monkeypatch.setattr(b, "ANCHOR_FILE", str(tmp_path / "anchor.json"))
    monkeypatch.delenv("SGR_BENCH_NO_ANCHOR", raising=False)
    assert b._read_anchor("cfg2_b16_env") == {}
    b._write_anchor("cfg2_b16_env", dict(Mpix_with_loss=3050.0, Mpix_layer_only=3200.0))
    a = b._read_anchor("cfg2_b16_env")
    assert a["Mpix_with_loss"] == 3050.0 and "this host" in a["source"]
    assert b._read_anchor("cfg2_b8_env") == {}
    rec = json.load(open(b.ANCHOR_FILE))
    rec["cfg2_b16_env"]["host"] = "another-box"
    json.dump(rec, open(b.ANCHOR_FILE, "w"))
    assert b._read_anchor("cfg2_b16_env") == {}       # never a figure from another box
    monkeypatch.setenv("SGR_BENCH_NO_ANCHOR", "1")
    b._write_anchor("cfg2_b8_env", dict(Mpix_with_loss=1.0))
    assert "cfg2_b8_env" not in json.load(open(b.ANCHOR_FILE))
