"""Host logic of the fused ResnetFC kernels: the chunk-descriptor tables they walk (csrc/fused.hip, csrc/wide.hip), read back
through the host-only C entry scenerf_hip_test_chunk_table.  No GPU needed.  For every scale mask the descriptors must cover each
16-wide K chunk of each layer exactly once, in order, with the right operand source / column / weight block, and satisfy the
structural rules the kernels rely on (layer ends on group boundaries, stages, padding)."""
import ctypes as C

import numpy as np
import pytest

from scenerf_amd import _capi
from scenerf_amd.config import RenderConfig

STRIDE = 704
D_X3, D_H, D_L = 144, 512, 2480


def _tables(kind, variant="kitti"):
    lib = _capi.load()
    rcfg = RenderConfig.kitti(precision="bf16") if variant == "kitti" else RenderConfig.bundlefusion(precision="bf16")
    cc = rcfg.to_c()
    out = (C.c_int32 * (33 * STRIDE))()
    n = lib.scenerf_hip_test_chunk_table(C.byref(cc), kind, out, len(out))
    assert n == (32 if kind == 1 else 33) * STRIDE, _capi.load().scenerf_hip_last_error()
    return np.frombuffer(out, dtype=np.int32, count=n).reshape(-1, STRIDE).copy(), [c for c, _, _ in rcfg.map_shapes()]


def _expected_chunks(mask, chans):
    """(layer, src, a_col, w_block) of every 16-wide K chunk of the forward trunk, in order.  w_stream block order: the K chunks of
    w_h[0] (144 + 2480 columns), w_fc0[0], w_h[1] (512 + 2480), w_fc0[1], w_h[2], w_fc0[2], w_h[3] (512)."""
    layer_k = [D_X3 + D_L, D_H, D_H + D_L, D_H, D_H + D_L, D_H, D_H]
    block0 = np.concatenate([[0], np.cumsum([k // 16 for k in layer_k])])
    seg_off = np.concatenate([[0], np.cumsum(chans)])
    out = []

    def zsegs(layer, wbase):
        for i, c in enumerate(chans):
            if (mask >> i) & 1:
                out.extend((layer, 2, (seg_off[i] + k) // 16, block0[layer] + (wbase + k) // 16) for k in range(0, c, 16))
            wbase += c
    out.extend((0, 1, k // 16, block0[0] + k // 16) for k in range(0, D_X3, 16))
    zsegs(0, D_X3)
    for b in range(3):
        out.extend((1 + 2 * b, 0, k // 16, block0[1 + 2 * b] + k // 16) for k in range(0, D_H, 16))
        out.extend((2 + 2 * b, 0, k // 16, block0[2 + 2 * b] + k // 16) for k in range(0, D_H, 16))
        if b < 2:
            zsegs(2 + 2 * b, D_H)
    return out, int(block0[-1])


def _fields(d):
    return dict(block=d & 1023, acol=(d >> 10) & 255, src=(d >> 18) & 3, layer=(d >> 20) & 7, end=(d >> 23) & 1, begin=(d >> 24) & 1)


@pytest.mark.parametrize("variant", ["kitti", "bundlefusion"])
def test_ring_kernel_tables(variant):
    tabs, chans = _tables(0, variant)
    assert sum(chans) == D_L
    for mask in range(32):
        n, d = int(tabs[mask, 0]), tabs[mask, 1:]
        want, nfwd = _expected_chunks(mask, chans)
        assert n == len(want)
        for i, (layer, src, acol, block) in enumerate(want):
            f = _fields(int(d[i]))
            assert (f["layer"], f["src"], f["acol"], f["block"]) == (layer, src, acol, block), (mask, i)
            assert (int(d[i]) >> 25) & 7 == i % 5                                  # ring stage
            assert f["end"] == (i + 1 == n or want[i + 1][0] != layer) and f["begin"] == (i == 0 or want[i - 1][0] != layer)
        assert np.all((d[n:n + 8] & 0x01FFFFFF) == 0)                              # the pipeline reads a few no-op entries past the end
    # backward chain: six layers of 32 resident-operand chunks, weight blocks after the forward ones
    n, d = int(tabs[32, 0]), tabs[32, 1:]
    assert n == 6 * 32 and nfwd + n == _capi.W_STREAM_BLOCKS
    for i in range(n):
        f = _fields(int(d[i]))
        assert (f["layer"], f["src"], f["acol"], f["block"]) == (i // 32, 0, i % 32, nfwd + i)
        assert f["end"] == (i % 32 == 31) and f["begin"] == (i % 32 == 0) and (int(d[i]) >> 25) & 7 == i % 5


def test_removed_kernel_kind_is_refused():
    """kind 1 was stream.hip's table (a register-streamed 64-row forward with fused.hip's results and speed, removed in round 3)."""
    import ctypes as C
    from scenerf_amd.config import RenderConfig
    lib = _capi.load()
    buf = (C.c_int32 * 8)()
    assert lib.scenerf_hip_test_chunk_table(C.byref(RenderConfig.kitti().to_c()), 1, buf, 8) < 0
    with pytest.raises(ValueError):
        RenderConfig.kitti(fwd_kernel="stream").to_c()


@pytest.mark.parametrize("variant", ["kitti", "bundlefusion"])
def test_wide_kernel_tables(variant):
    """wide.hip: header {chunks in total, chunks of layer 0, chunks of a lin_z tail (both padded to multiples of four), real chunks of a
    lin_z tail}, then one descriptor per chunk in execution order.  Layer 0 stages its Z chunks first (stage slot = list position, so
    layers 2 and 4 find them again), then the split encoding; every layer is padded with no-op chunks (bit 25: weights are loaded, MFMAs
    skipped) to a multiple of four so that a layer begins in ring slot 0; the table is zero beyond the end (the weight prefetch
    runs up to three groups ahead).  Set 32 = the backward chain: six layers of 32 resident chunks behind the forward blocks."""
    tabs, chans = _tables(2, variant)
    assert tabs.shape[0] == 33
    for mask in range(32):
        n, n0, nz, nzr = (int(x) for x in tabs[mask, :4])
        d = [int(x) for x in tabs[mask, 4:]]
        want, _ = _expected_chunks(mask, chans)
        assert n % 4 == 0 and n0 % 4 == 0 and nz % 4 == 0 and 4 + n + 12 <= STRIDE
        real_z = sum(c // 16 for i, c in enumerate(chans) if (mask >> i) & 1)
        assert nzr == real_z and nz == (real_z + 3) // 4 * 4 and n0 == (real_z + 9 + 3) // 4 * 4
        got = []
        for i in range(n):
            f = _fields(d[i])
            if (d[i] >> 25) & 1:
                assert f["src"] == 0 and f["block"] == 0 and f["acol"] == 0 and not f["begin"]      # padding: harmless loads
            else:
                got.append((f["layer"], f["src"], f["acol"], f["block"]))
            first = i == 0 or _fields(d[i - 1])["layer"] != f["layer"]
            last = i + 1 == n or _fields(d[i + 1])["layer"] != f["layer"]
            assert f["begin"] == first and f["end"] == last
            if first:
                assert i % 4 == 0 and not (d[i] >> 25) & 1        # the zero-accumulator MFMA form sits in slot 0 and is a real chunk
            if last:
                assert i % 4 == 3
        # layer 0: Z chunks first, then the encoding; every other layer in the reference order
        l0 = [w for w in want if w[0] == 0]
        l0 = [w for w in l0 if w[1] == 2] + [w for w in l0 if w[1] == 1]
        exp = l0 + [w for w in want if w[0] != 0]
        assert got == [tuple(int(v) for v in w) for w in exp], mask                # every chunk exactly once
        # the lin_z tails of layers 2 and 4 are the last nz entries of their layers
        for layer in (2, 4):
            idx = [i for i in range(n) if _fields(d[i])["layer"] == layer]
            assert len(idx) == 32 + nz
            assert all(_fields(d[i])["src"] == 0 and not (d[i] >> 25) & 1 for i in idx[:32])
            assert all(_fields(d[i])["src"] == 2 for i in idx[32:32 + nzr]) and all((d[i] >> 25) & 1 for i in idx[32 + nzr:])
        assert all(x == 0 for x in d[n:n + 12])
    n, d = int(tabs[32, 0]), tabs[32, 4:]
    _, nfwd = _expected_chunks(0, chans)
    assert n == 6 * 32 and nfwd + n == _capi.W_STREAM_BLOCKS
    for i in range(n):
        f = _fields(int(d[i]))
        assert (f["layer"], f["src"], f["acol"], f["block"]) == (i // 32, 0, i % 32, nfwd + i)
        assert f["end"] == (i % 32 == 31) and f["begin"] == (i % 32 == 0)


def test_chunk_table_argument_checks():
    lib = _capi.load()
    cc = RenderConfig.kitti(precision="bf16").to_c()
    small = (C.c_int32 * 16)()
    assert lib.scenerf_hip_test_chunk_table(C.byref(cc), 0, small, 16) == -2      # too small
    assert b"too small" in lib.scenerf_hip_last_error()
    assert lib.scenerf_hip_test_chunk_table(C.byref(cc), 7, small, 16) == -1      # unknown kind
    assert lib.scenerf_hip_test_chunk_table(None, 0, small, 16) == -1
