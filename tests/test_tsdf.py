"""TSDF fusion.  CPU: the numpy oracle against the golden volumes the reference's own CPU path produced
(tests/golden/make_golden_tsdf.py).  GPU: the HIP kernel through the C ABI against the oracle, both update rules."""
import os

import numpy as np
import pytest
import torch

import tsdf_oracle as orc
import tsdf_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tsdf_cpu_semantics.npz")
# volumes produced by the reference's own pycuda kernel text, compiled verbatim for gfx950 and run on an MI355X
# (oracle/build_ref.py + tests/golden/make_golden_tsdf_gpu.py): ``*_v08`` / ``*_v07`` = every operation rounded on its own (the pin),
# ``*_contract`` = the same text under the compiler's default fma contraction (a statistic)
GOLD_GPU = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tsdf_gpu_semantics.npz")


def _run_oracle(sc, semantics, n_frames=None):
    dim, origin, tsdf, weight, color = orc.new_volume(sc["vol_bnds"], sc["voxel_size"])
    fn = orc.integrate_gpu_semantics if semantics == "gpu" else orc.integrate_cpu_semantics
    for fr in sc["frames"][:n_frames]:
        fn(tsdf, weight, color, origin, sc["voxel_size"], orc.fold_color(fr["color"]), fr["depth"], sc["cam_intr"], fr["pose"],
           sc["trunc_margin"], 1.0)
    return tsdf, weight, color


def test_oracle_cpu_semantics_matches_the_reference():
    g = np.load(GOLD)
    sc = tsdf_scene.make(seed=int(g["seed"]))
    tsdf, weight, color = _run_oracle(sc, "cpu")
    assert tuple(g["vol_dim"]) == tsdf.shape
    assert np.array_equal(weight, g["weight"])
    assert np.array_equal(tsdf, g["tsdf"])
    assert np.array_equal(color, g["color"])
    assert int((weight > 0).sum()) > 10000 and int((tsdf == 255).sum()) > 1000     # the scene covers part of the volume only


@pytest.mark.parametrize("tag,voxel_size", [("v08", 0.08), ("v07", 0.07)])
def test_oracle_gpu_semantics_matches_the_reference_kernel(tag, voxel_size):
    """The numpy restatement of the pycuda rule against the reference kernel itself: bit for bit."""
    g = np.load(GOLD_GPU)
    sc = tsdf_scene.make(seed=int(g["seed"]))
    sc["voxel_size"] = voxel_size
    tsdf, weight, color = _run_oracle(sc, "gpu")
    assert g["tsdf_" + tag].shape == tsdf.shape
    assert np.array_equal(weight, g["weight_" + tag])
    assert np.array_equal(tsdf, g["tsdf_" + tag])
    assert np.array_equal(color, g["color_" + tag])
    # what a compiler's choice of fused multiply-adds moves (same kernel text, default contraction): observation counts of < 0.1 % of
    # the voxels, the last bit of ~1-2 % of the stored distances, and a handful of voxels whose projection lands on the neighbouring pixel
    same_w = weight == g["weight_" + tag + "_contract"]
    assert same_w.mean() > 0.999
    d = np.abs(tsdf - g["tsdf_" + tag + "_contract"])[same_w]
    assert (d > 0).mean() < 0.03 and (d > 2e-6).mean() < 5e-3      # measured: 0.7-1.6 % differ at all, 0.2 % by more than 2e-6


def test_oracle_gpu_semantics_basic_properties():
    sc = tsdf_scene.make(seed=7)
    tsdf, weight, color = _run_oracle(sc, "gpu")
    seen = weight > 0
    assert np.all(tsdf[~seen] == 255) and np.all(color[~seen] == 0)
    assert np.all(tsdf[seen] <= 1.0 + 1e-6) and np.all(tsdf[seen] >= -1.0 - 1e-6)   # truncated, normalised distances
    assert np.all(weight[seen] <= len(sc["frames"])) and np.all(weight == np.round(weight))
    b = np.floor(color / 65536); gch = np.floor((color - b * 65536) / 256); r = color - b * 65536 - gch * 256
    assert b.max() <= 255 and gch.max() <= 255 and r.max() <= 255
    # the zero crossing sits on the wall z = 3 + 0.2 x: for a voxel column through the middle of the volume the sign changes there
    dim, origin = tsdf.shape, sc["vol_bnds"][:, 0]
    ix, iy = dim[0] // 8, dim[1] // 2        # (a column away from the sphere)
    col = tsdf[ix, iy]
    zc = origin[2] + sc["voxel_size"] * np.arange(dim[2])
    wall = 3.0 + 0.2 * (origin[0] + sc["voxel_size"] * ix)
    obs = weight[ix, iy] > 0
    assert np.all(col[obs & (zc < wall - 0.45)] > 0) and np.any(col[obs & (zc > wall + 0.1)] < 0)


@pytest.mark.gpu
@pytest.mark.parametrize("voxel_size", [0.08, 0.07])   # z extent 44 (four voxels per thread, 16-byte accesses) / 50 (scalar path)
@pytest.mark.parametrize("semantics", ["gpu", "cpu"])
def test_hip_kernel_matches_oracle(semantics, voxel_size):
    from scenerf_amd.fusion import TSDFVolume
    sc = tsdf_scene.make(seed=7)
    sc["voxel_size"] = voxel_size
    vol = TSDFVolume(sc["vol_bnds"].copy(), voxel_size=sc["voxel_size"], trunc_margin=sc["trunc_margin"], semantics=semantics)
    for fr in sc["frames"]:
        vol.integrate(fr["color"], fr["depth"], sc["cam_intr"], fr["pose"], obs_weight=1.0)
    tsdf, color = vol.get_volume()
    weight = vol.get_weight()
    rt, rw, rc = _run_oracle(sc, semantics)
    assert tsdf.shape == rt.shape
    # both rules: identical decisions, identical values ("cpu": float64 projection on both sides; "gpu": fp32, every operation
    # rounded on its own on both sides)
    assert np.array_equal(weight, rw) and np.array_equal(tsdf, rt) and np.array_equal(color, rc)
    if semantics == "gpu":   # ... and the reference's own kernel (compiled from its text, run on an MI355X): bit for bit
        g = np.load(GOLD_GPU)
        tag = "v08" if voxel_size == 0.08 else "v07"
        assert np.array_equal(weight, g["weight_" + tag]) and np.array_equal(tsdf, g["tsdf_" + tag]) and np.array_equal(color, g["color_" + tag])
    assert np.all(tsdf[weight == 0] == 255)


@pytest.mark.gpu
def test_hip_tsdf_rejects_bad_arguments():
    from scenerf_amd.fusion import TSDFVolume
    with pytest.raises(RuntimeError):
        TSDFVolume(np.array([[0, 1], [0, 1], [0, 1]], dtype=float), 0.1, use_gpu=False)
    with pytest.raises(ValueError):
        TSDFVolume(np.array([[0, 1], [0, 1], [0, 1]], dtype=float), 0.1, semantics="both")
