"""CPU-side checks: the C-ABI library loads and exports every symbol that
include/cdml.h declares (no compute call without a GPU), the ctypes table
matches the header, and the host-side logic (layout, schedule, plug-in lookup)."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "cdml.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cdml_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from cdml_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built):
    lib = built.load_library()
    syms = _header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(built.SIGNATURES) == syms          # binding table == header
    assert lib.cdml_version() == 3000
    assert lib.cdml_last_error() is not None


def test_build_id_ties_the_library_to_the_tree(built, tmp_path):
    """the loaded library says which sources it was built from, and a library built from other
    sources is refused by the loader (file times cannot tell after a copy)."""
    import shutil
    import subprocess
    import sys
    lib = built.load_library()
    sid = built.source_id()
    assert sid and lib.cdml_build_id().decode() == "CDML_BUILD_ID=" + sid == "CDML_BUILD_ID=" + built.embedded_id(built.lib_path())
    # a tree whose kernel sources differ from the ones the prebuilt library came from: same .so, one byte more in a source
    pkg = tmp_path / "collaborative-deep-metric-learning_amd"
    shutil.copytree(os.path.join(ROOT, "collaborative-deep-metric-learning_amd"), pkg,
                    ignore=shutil.ignore_patterns("__pycache__"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    shutil.copytree(os.path.join(ROOT, "cdml_amd"), tmp_path / "cdml_amd", ignore=shutil.ignore_patterns("__pycache__"))
    with open(pkg / "csrc" / "common.h", "a") as f:
        f.write("\n// edited after the library was built\n")
    code = ("import sys; sys.path.insert(0, %r); from cdml_amd import _lib\n"
            "try:\n    _lib.load_library()\nexcept _lib.CdmlError as e:\n    print('REFUSED', e)\n" % str(tmp_path))
    env = {k: v for k, v in os.environ.items() if k not in ("CDML_LIB_PATH", "CDML_ALLOW_STALE_LIB")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert "REFUSED" in r.stdout and "built from other sources" in r.stdout, r.stdout + r.stderr


def test_no_gemm_kernel_has_a_scratch_frame(built):
    """The GEMMs count their LDS-DMA loads by hand (s_waitcnt vmcnt(N)); a register spill is a vector-memory operation on
    the same counter, so a spill inside such a loop reads LDS images before they have landed.  build() keeps the compiler's
    resource remarks beside every object and refuses a build with a spilling GEMM kernel; this test reads them again."""
    import glob
    import re
    import __graft_entry__ as g
    rem = sorted(glob.glob(os.path.join(ROOT, "build", "obj", "gemm_*.remarks")))
    assert len(rem) >= 4, "build() writes <object>.remarks for every source"
    seen = 0
    for f in rem:
        name = None
        for ln in open(f):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                name = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", ln)
            if m and name and "k_gemm" in name:
                seen += 1
                assert int(m.group(1)) == 0, (name, ln)
    assert seen >= 40                                  # every instantiation of the tile kernels reports
    g._check_no_scratch([f[:-8] + ".o" for f in rem])


def test_argument_errors_need_no_gpu(built):
    """Validation happens before any HIP call, so the status/message contract is
    testable on CPU: null pointers -> CDML_E_BADARG with a message."""
    import ctypes as C
    lib = built.load_library()
    rc = lib.cdml_l2norm_fwd(None, 4, 4, 4, None, 4, None, None)
    assert rc in (-1, -3)
    assert lib.cdml_last_error()
    rc = lib.cdml_fc_lrelu_fwd(C.c_void_p(16), 48, C.c_void_p(16), 64, C.c_void_p(16), 0.2, 8, 48, 64,
                               C.c_void_p(16), 64, None)
    assert rc == -4 and b"multiple of 32" in lib.cdml_last_error()
    assert lib.cdml_fc_bwd_weight_workspace(8192, 1536, 5120) >= 5120 * 4       # one split: bias partials only
    assert lib.cdml_fc_bwd_weight_workspace(8192, 5120, 256) >= 2 * 5120 * 256 * 4   # split-K slabs
    assert lib.cdml_fc_bwd_weight_workspace(8, 48, 64) == 0
    with pytest.raises(built.CdmlError):
        built.call("cdml_step_advance", None, None)


def test_slab_length_pin_is_thread_local_and_needs_no_gpu(built):
    """cdml_x3_slab_steps (round 6): the pin of the narrow layer's K-slab length -- returns the previous pin, rounds to whole
    six-step periods, 0 / too short = back to the rule by row-tile class; one pin per thread."""
    import threading
    from cdml_amd import _lib
    lib = _lib.load_library()
    assert lib.cdml_x3_slab_steps(120) == 0
    assert lib.cdml_x3_slab_steps(64) == 120                   # 64 -> 60 (whole periods of six K-tile steps)
    assert lib.cdml_x3_slab_steps(5) == 60                     # too short: no pin
    assert lib.cdml_x3_slab_steps(120) == 0
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.cdml_x3_slab_steps(60)))
    t.start(); t.join()
    assert seen == [0]                                         # another thread starts without a pin
    assert lib.cdml_x3_slab_steps(0) == 120                    # ... and did not touch this one


def test_no_cpu_fallback_and_oracle_not_imported():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import cdml_amd; from cdml_amd import engine, inputs, "
            "losses, models, ops, train, utils, dist; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)
    pkg = os.path.join(ROOT, "collaborative-deep-metric-learning_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "oracle" not in re.sub(r'""".*?"""', "", open(os.path.join(pkg, f)).read(), flags=re.S) \
                .replace("oracle/sampler.py", ""), f


def test_layout_and_schedule():
    from cdml_amd import engine, train, utils, models, losses, inputs
    L = engine.TowerLayout(1500, 5000, 256)
    assert (L.Fp, L.Hp, L.Dp) == (1536, 5120, 256)
    assert L.numel_unpadded == 8785256                     # parameter count
    assert all(o % 4 == 0 for o in L.offsets)
    assert engine.FeatureTable.padded_stride(1500) * 4 % 128 == 0
    assert train.exponential_decay(1.0, 999999, 1000000, 0.96) == 1.0
    assert abs(train.exponential_decay(0.01, 2000001, 1000000, 0.96) - 0.01 * 0.96 ** 2) < 1e-15
    assert utils.find_class_by_name("VNet", [models]) is models.VNet
    assert utils.find_class_by_name("HingeLoss", [losses]) is losses.HingeLoss
    assert issubclass(models.VNet, models.BaseModel) and issubclass(losses.HingeLoss, losses.BaseLoss)
    assert issubclass(inputs.MPTripletPipe, inputs.BasePipe)
    with pytest.raises(StopIteration):
        utils.find_class_by_name("VedeNet", [models])


def test_read_cowatch_files(tmp_path):
    from cdml_amd import inputs
    (tmp_path / "a.train").write_text("1,2\n3,4\n")
    (tmp_path / "b.train").write_text("5,6\n")
    got = inputs.read_cowatch_files(sorted(str(p) for p in tmp_path.glob("*.train")))
    np.testing.assert_array_equal(got, [[1, 2], [3, 4], [5, 6]])
    assert got.dtype == np.int32


def test_no_kernel_uses_scratch():
    """A kernel whose register arrays fall into scratch memory runs 1.5x slower and
    looks fine otherwise (it happened: HIP's float4 struct copied through a
    conditional pointer became a memcpy to a stack slot).  hipcc reports it."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "collaborative-deep-metric-learning_amd", "csrc")
    for src in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c",
                              "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                              "-o", "/dev/null", src], capture_output=True, text=True, cwd=csrc)
        assert out.returncode == 0, out.stderr[-2000:]
        scratch = re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)
        spills = re.findall(r"VGPRs Spill: (\d+)", out.stderr)
        assert scratch and all(int(x) == 0 for x in scratch), (src, scratch)
        assert all(int(x) == 0 for x in spills), (src, spills)


def test_dataset_directory_formats(tmp_path):
    """Next-row N3: the reference's on-disk formats round-trip."""
    from cdml_amd import inputs, online_data
    rng = np.random.RandomState(0)
    feats = rng.random_sample((50, 8)).astype(np.float32)
    pairs = [[int(a), int(b)] for a, b in rng.randint(0, 50, size=(200, 2))]
    d = str(tmp_path / "ds")
    online_data.write_features(feats, {"g%d" % i: i for i in range(50)}, {i: "g%d" % i for i in range(50)}, d)
    online_data.write_cowatches(pairs, d, split_num=4, eval_num=20, test_num=10)
    ds = online_data.load_dataset(d)
    np.testing.assert_array_equal(ds["features"], feats)
    assert ds["eval_cowatches"] == pairs[:20] and ds["test_cowatches"] == pairs[20:30]
    assert [os.path.basename(f) for f in ds["train_files"]] == ["xaa.train", "xab.train", "xac.train", "xad.train"]
    got = inputs.read_cowatch_files(ds["train_files"])
    np.testing.assert_array_equal(got, np.asarray(pairs[30:], dtype=np.int32))
    assert ds["decode_map"]["7"] == "g7"
    (tmp_path / "bad.eval").write_text("1,2\nx,y\n\n3,4\n")
    assert online_data.load_cowatches(str(tmp_path / "bad.eval")) == [[1, 2], [3, 4]]
    # 15 % rule of online_data.py:262-264
    online_data.write_cowatches(pairs, str(tmp_path / "ds2"), split_num=1)
    assert len(online_data.load_cowatches(str(tmp_path / "ds2" / "cowatches.eval"))) == 30


def test_imitation_data_matches_reference_fixture(golden_dir):
    """cdml_amd.imitation_data.gen_features under np.random.seed == the array the reference's
    generator produced (fixture G2), and the other generators keep its shapes and rules."""
    import random
    from cdml_amd import imitation_data as im
    g = np.load(os.path.join(golden_dir, "imitation_features_seed0.npz"))
    key = [k for k in g.files if g[k].ndim == 2][0]
    ref = g[key]
    np.random.seed(0)
    np.testing.assert_array_equal(im.gen_features(*ref.shape), ref)
    np.random.seed(1)
    t = im.gen_triplets(5, 12)
    assert t.shape == (5, 3, 12) and t.dtype == np.float64 and 0 <= t.min() and t.max() < 1
    random.seed(2)
    np.random.seed(2)
    w = im.gen_all_watched_guids(np.arange(50), 20, low=2, high=6)
    assert len(w) == 20 and all(2 <= len(x) <= 6 and set(x) <= set(range(50)) for x in w)
    ids = im.gen_unique_id_array(5, 30, 10)
    assert len(set(ids.tolist())) == 10 and ids.min() >= 5 and ids.max() <= 30
    with pytest.raises(ValueError):
        im.gen_unique_id_array(0, 3, 9)
    d = im.arrays_to_dict(["a", "b"], np.eye(2))
    assert list(d) == ["a", "b"] and d["b"][1] == 1
    p = im.cowatch_pairs(1000, 100, seed=3)
    assert p.dtype == np.int32 and p.shape[1] == 2 and (p[:, 0] != p[:, 1]).all() and p.max() < 1000


def test_bench_self_launch_reports_failed_rank():
    """`python bench.py --gpus 2` with no launcher in the environment starts its own ranks; when a rank
    fails (here: no GPU in this container) the parent exits non-zero and names the rank and its phase --
    never a usage message, never a hang."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU: there the ranks fail at start")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1
    assert "2-rank run failed: rank" in r.stderr and "exited with code" in r.stderr and "rank 1:" in r.stderr
    assert "needs an MI355X" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_train_config_carries_the_reference_defaults_and_round_trips(tmp_path):
    """SURVEY section 5 'Config / flags': one object with the reference's names and its hard-coded values
    (train.py:358-373, 212-222)."""
    from cdml_amd.config import TrainConfig
    c = TrainConfig()
    assert (c.num_epochs, c.batch_size, c.learning_rate, c.margin, c.optimizer) == (8, 1024, 1.0, 0.8, "lars")
    assert (c.check_stop_epoch, c.best_eval_dist, c.eval_per_epoch, c.require_improve_num) == (3, 1.0, 100, 40)
    assert (c.output_size, c.learning_rate_decay_examples, c.learning_rate_decay) == (256, 1000000, 0.96)
    assert (c.clip_gradient_norm, c.regularization_penalty, c.model, c.hidden_size) == (0.0, 0.0, "VNet", 5000)
    p = tmp_path / "cfg.json"
    c2 = TrainConfig(batch_size=4096, mode="inbatch", optimizer="adam", learning_rate=0.01)
    c2.to_json(str(p))
    assert TrainConfig.from_json(str(p)) == c2 and TrainConfig.from_json(c2.to_json()) == c2
    import pytest
    with pytest.raises(ValueError):
        TrainConfig.from_json('{"batchsize": 3}')
