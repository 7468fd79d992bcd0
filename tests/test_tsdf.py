"""TSDF fusion.  CPU: the numpy oracle against the golden volumes the reference's own CPU path produced
(tests/golden/make_golden_tsdf.py).  GPU: the HIP kernel through the C ABI against the oracle, both update rules."""
import os

import numpy as np
import pytest
import torch

import tsdf_oracle as orc
import tsdf_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tsdf_cpu_semantics.npz")


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
    if semantics == "cpu":
        # float64 projection on both sides: identical decisions, identical values
        assert np.array_equal(weight, rw) and np.array_equal(tsdf, rt) and np.array_equal(color, rc)
    else:
        # fp32 projection: x/z and the products may differ in the last bit between numpy and the GPU (fma contraction is off, the
        # division is not correctly rounded on every path), which can move a voxel across a pixel or truncation boundary
        same = (weight == rw)
        assert same.mean() > 0.999, "fraction of voxels with the same observation count: %.5f" % same.mean()
        both = same & (rw > 0)
        assert np.allclose(tsdf[both], rt[both], rtol=0, atol=2e-5) or (np.abs(tsdf[both] - rt[both]) > 2e-5).mean() < 1e-3
        assert (color[both] != rc[both]).mean() < 1e-3
    assert np.all(tsdf[weight == 0] == 255)


@pytest.mark.gpu
def test_hip_tsdf_rejects_bad_arguments():
    from scenerf_amd.fusion import TSDFVolume
    with pytest.raises(RuntimeError):
        TSDFVolume(np.array([[0, 1], [0, 1], [0, 1]], dtype=float), 0.1, use_gpu=False)
    with pytest.raises(ValueError):
        TSDFVolume(np.array([[0, 1], [0, 1], [0, 1]], dtype=float), 0.1, semantics="both")
