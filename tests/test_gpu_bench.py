"""bench.py contract on the GPU box: exactly one JSON line on stdout with the contract's keys, and the N-rank row-tiled
frame loop rehearsed with two real processes (collectives staged over gloo because RCCL refuses two ranks on one GPU):
the composed 8K image has the checksum of the single-GPU frame."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline"]


def _run(cmd, env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(env or {}))
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [line for line in r.stdout.splitlines() if line.strip()]
    assert len(lines) == 1, f"stdout must be ONE JSON line, got {len(lines)}: {r.stdout[:500]}"
    return json.loads(lines[0])


def test_bench_contract_and_two_rank_rehearsal():
    one = _run([sys.executable, "bench.py", "--workload", "c4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    for key in CONTRACT_KEYS:
        assert key in one, key
    assert one["n_gpus"] == 1 and one["unit"] == "Mpixels/s" and one["dtype"] == "f32" and one["vs_baseline"] is None
    assert one["roofline"]["bound"] == "hbm" and 0 < one["roofline"]["frac"] < 1
    # `python bench.py --gpus 2` with no launcher around it: bench.py starts its own ranks (self_launch) and relays the line
    two = _run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "2", "--warmup", "1",
                "--no-cpu-baseline"])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["parallelism"] == "rowtile2+gather"
    assert two["image_checksum"] == one["image_checksum"]
    assert abs(two["config"]["geometry_fraction"] - one["config"]["geometry_fraction"]) < 1e-12


def test_bench_under_an_external_launcher_and_forced_tiling_on_rccl():
    """The driver's N > 1 form (torch.distributed.run around bench.py) with the two-rank rehearsal, and the row-tiled loop
    with both collectives on RCCL in a world of one."""
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29655", "bench.py", "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "1", "--warmup", "1",
                "--no-cpu-baseline"])
    assert two["n_gpus"] == 2 and two["config"]["collectives"] == "gloo"
    one = _run([sys.executable, "bench.py", "--gpus", "1", "--workload", "c4", "--force-tiled", "--steps", "2", "--warmup", "1",
                "--no-cpu-baseline"])
    assert one["config"]["collectives"] == "nccl" and one["config"]["parallelism"] == "rowtile1+gather"
    assert one["image_checksum"] == two["image_checksum"]


_C4 = {}


def _single_gpu_c4_checksum():
    """The untiled 8K frame rendered by this build in this run (a committed checksum goes stale with every change of the
    arithmetic, e.g. round 3's contraction rule)."""
    if "checksum" not in _C4:
        _C4["checksum"] = _run([sys.executable, "bench.py", "--workload", "c4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                                "--no-extras"])["image_checksum"]
    return _C4["checksum"]


@pytest.mark.parametrize("ranks,no_gather", [(2, False), (4, False), (2, True)])
def test_n_rank_frame_through_the_c_abi_collectives_on_one_gpu(ranks, no_gather):
    """The multi-rank frame loop through the C-ABI's OWN collective entry points (szg_rowtile_comm_*, szg_rowtile_gather,
    szg_skyview_allgather_lut_rows with the status-word exchange) with N real processes on one GPU. RCCL itself cannot run
    there (it refuses two ranks on a device), so tests/cpp/mock_rccl.cpp stands in for librccl.so behind SZG_RCCL_LIBRARY:
    same entry points, same in-place / root-only contracts, bytes staged through a mapped file. Everything above that
    boundary is the product's code: the communicator pair, byte offsets of tiles and LUT slices, the cyclic row blocks, the
    two frames in flight, the compose. The composed 8K image must have the checksum of the single-GPU frame."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "libmock_rccl.so"], check=True)
    env = {"SZG_RCCL_LIBRARY": os.path.join(ROOT, "tests", "cpp", "libmock_rccl.so")}
    if no_gather:  # szg_rowtile_gather's grouped ncclSend / ncclRecv form (what a library without ncclGather gets)
        env["SZG_RCCL_NO_GATHER"] = "1"
    many = _run([sys.executable, "bench.py", "--gpus", str(ranks), "--backend", "nccl", "--same-device", "--steps", "2", "--warmup", "1",
                 "--no-cpu-baseline"], env=env)
    cfg = many["config"]
    assert many["n_gpus"] == ranks and cfg["parallelism"] == f"rowtile{ranks}+gather" and cfg["collectives"] == "nccl"
    assert cfg["collective_api"].startswith("szg_rowtile_comm") and cfg["rccl_ranks"] == ranks
    assert many["image_checksum"] == _single_gpu_c4_checksum()


def test_replicas_line_of_the_batch_workload_through_the_self_launcher():
    """`bench.py --gpus N --workload c5` (BASELINE's config 5: independent 4K views, one per GPU, no collective in the data
    path): bench.py starts its own ranks, every rank renders the same view, the line reports N x the pixels over the slowest
    rank's time with "replicas N" / weak scaling. Two ranks share the one GPU of the box here (--same-device), so the value
    says nothing; what is checked is the launch path, the aggregation and that no collective API is involved."""
    one = _run([sys.executable, "bench.py", "--workload", "c5", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    two = _run([sys.executable, "bench.py", "--gpus", "2", "--workload", "c5", "--backend", "gloo", "--same-device", "--steps", "2",
                "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["parallelism"] == "replicas2"
    assert two["config"]["collectives"] is None and two["config"]["rccl_ranks"] is None
    assert two["image_checksum"] == one["image_checksum"]
    # value = pixels of BOTH views / time of the slowest rank
    assert two["value"] == pytest.approx(2 * 3840 * 2160 / (two["ms_per_step"] * 1e-3) / 1e6, rel=1e-6)


def test_bench_roofline_names_the_dominant_kernel_of_each_workload():
    """`roofline` describes the longest pass of the run it belongs to, and takes `traffic` only from the committed counter
    profile of the same workload and kernel (the latest profiles/rNN_pmc_<workload>.json), and says which tree that profile
    was collected from."""
    c5 = _run([sys.executable, "bench.py", "--workload", "c5", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert c5["roofline"]["kernel"] == "k_lights" and c5["roofline"]["avg_launch_ms"] == pytest.approx(c5["pass_ms_rank0"]["lights"])
    assert c5["roofline"]["algorithmic_bytes_per_launch"] == 56 * 3840 * 2160
    import glob
    import re

    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_c5.json")))[-1]
    assert c5["roofline"]["traffic"] is not None and c5["roofline"]["traffic_source"] == os.path.relpath(latest, ROOT)
    assert re.fullmatch(r"[0-9a-f]{16}", c5["roofline"]["traffic_profile"]["source_hash"] or "") or "r02_" in latest
    assert re.fullmatch(r"[0-9a-f]{16} contract=0x[0-9a-f]{4}", c5["library_build_id"])
    c3 = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert c3["roofline"]["kernel"] == "k_composite" and c3["roofline_valu"]["kernel"] == "k_composite"
    assert 0.5 < c3["roofline_valu"]["frac_of_issue_ceiling_at_2.4GHz"] < 1.0
    assert len(c3["source_hash"]) == 16


def test_example_frame_loop_runs():
    """examples/frame_loop.py: scene tick -> shadow bounds -> baked atmosphere -> mesh raster -> lights -> atmosphere ->
    OETF for a few animated frames; the image must be finite, opaque and not black."""
    r = subprocess.run([sys.executable, "examples/frame_loop.py", "--frames", "4", "--width", "320", "--height", "180",
                        "--shadow-map", "512"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ms per frame" in r.stdout
    value = float(r.stdout.split("mean display value")[1].split()[0])
    assert 0.02 < value < 0.98


def test_example_render_gltf_runs(tmp_path):
    """examples/render_gltf.py: a GLB written to disk -> include/szg/assets.h loader -> the whole path, one frame."""
    out = tmp_path / "gltf.ppm"
    r = subprocess.run([sys.executable, "examples/render_gltf.py", "--width", "320", "--height", "180", "--out", str(out)],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "1 meshes, 1 materials" in r.stdout and "[warning]" in r.stdout  # the sphere has no occlusion texture
    covered = float(r.stdout.split("geometry covers")[1].split("%")[0])
    assert 20.0 < covered < 95.0
    value = float(r.stdout.split("mean display value")[1].split()[0])
    assert 0.02 < value < 0.98
    assert out.stat().st_size > 320 * 180 * 3


@pytest.mark.parametrize("tool,first,last", [("random_sweep_gbuffer_fuzz.py", 1280, 1400), ("random_sweep_mesh_frames.py", 1440, 1500),
                                             ("random_sweep_shadow.py", 0, 80), ("random_sweep_raster.py", 21590, 21620),
                                             ("random_sweep_frames.py", 230, 250), ("random_sweep_params_fuzz.py", 1000, 1060),
                                             # round 2: NaN cosines of zero-length march segments rely on the clamp of the LUT distance
                                             ("random_sweep_frames.py", 940, 946), ("random_sweep_frames.py", 1272, 1276), ("random_sweep_frames.py", 2098, 2101),
                                             ("random_sweep_params_fuzz.py", 1986, 1989), ("random_sweep_params_fuzz.py", 4599, 4602),
                                             ("random_sweep_params_fuzz.py", 1500, 1530), ("random_sweep_params_fuzz.py", 26695, 26710), ("random_sweep_params_fuzz.py", 77025, 77032),
                                             ("random_sweep_params_fuzz.py", 87730, 87736), ("random_sweep_params_fuzz.py", 116586, 116593),
                                             ("random_sweep_raster_fuzz.py", 100, 130),
                                             ("random_sweep_raster_fuzz.py", 1375, 1390), ("random_sweep_raster_fuzz.py", 2920, 2935)])
def test_random_sweep_tools_find_nothing(tool, first, last, extra=()):
    """tests/sweeps/random_sweep_*.py are how the round's rare parity bugs were found; a slice of each — the seed
    ranges that once held mismatches — runs here so that the tools keep working and those cases stay fixed."""
    r = subprocess.run([sys.executable, os.path.join("tests", "sweeps", tool), str(first), str(last), *extra], cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mismatching seeds: 0" in r.stdout, r.stdout[-2000:]


def test_random_sweep_of_lut_reuse():
    """tests/sweeps/random_sweep_lut_reuse.py: a long-lived pipeline with LUT reuse on a random walk of parameter changes, repeats,
    scribbled texels and invalidations; after every frame its LUTs equal a fresh pipeline's."""
    r = subprocess.run([sys.executable, "tests/sweeps/random_sweep_lut_reuse.py", "0", "40"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "done, mismatching seeds: 0" in r.stdout, r.stdout[-2000:]


def test_random_sweep_with_degenerate_lut_extents():
    """LUTs of 2 ... 7 texels a side: their marches produce NaN texels in perfectly sane atmospheres, which only the sky-view
    LUT's status word (not any inference from the parameters) can tell the composite (seeds 1721, 1884 mismatched before it)."""
    test_random_sweep_tools_find_nothing("random_sweep_frames.py", 1715, 1725, ("tiny",))
    test_random_sweep_tools_find_nothing("random_sweep_frames.py", 1880, 1890, ("tiny",))
    test_random_sweep_tools_find_nothing("random_sweep_frames.py", 320, 323, ("tiny",))  # round 2: zero-length march segments
    test_random_sweep_tools_find_nothing("random_sweep_frames.py", 748, 751, ("tiny",))


def test_random_sweep_of_the_extension_luts():
    """The opt-in multi-scattering and aerial-perspective LUTs against their scalar oracles under hostile parameter blocks."""
    test_random_sweep_tools_find_nothing("random_sweep_params_fuzz.py", 0, 40, ("extensions",))
