"""CPU: the host side of `main.py` (main.lua's flag tables, image normalisation, seeded nets) -- no GPU needed."""
import numpy as np
import pytest

from mc_cnn_amd import main as mcmain
from mc_cnn_amd import params


def test_default_tables_cover_main_lua():
    """12 (dataset, arch) tables (main.lua:68-295); spot values from the ones BASELINE.json's configs use."""
    assert len(params.TABLES) == 12
    t = params.TABLES[("kitti", "fast")]
    assert (t["pi1"], t["pi2"], t["sgm_q1"], t["sgm_q2"], t["alpha1"], t["tau_so"], t["blur_sigma"], t["blur_t"]) == \
        (4.0, 55.72, 3.0, 2.5, 1.5, 0.02, 7.74, 5.0)
    t = params.TABLES[("mb", "slow")]
    assert (t["L1"], t["tau1"], t["cbca_i1"], t["cbca_i2"], t["lr_check"], t["border_n"]) == (14, 0.02, 2, 16, 0, 5)
    t = params.TABLES[("kitti2015", "slow")]
    assert (t["cbca_i2"], t["tau1"], t["alpha1"]) == (4, 0.03, 1.75)
    assert params.NET_SHAPES[("mb", "fast")] == (5, 64)


def test_flags_override_defaults_and_stage_names():
    _, _, opt, prm = mcmain.parse(["kitti", "fast", "-a", "predict", "-left", "l.png", "-right", "r.png", "-disp_max", "70",
                                   "-pi1", "3.5", "-sgm_i", "2"])
    assert opt.disp_max == 70 and prm["pi1"] == 3.5 and prm["sgm_i"] == 2 and prm["pi2"] == 55.72
    p = params.make_params(dict(prm, sm_terminate="sgm", sm_skip="median"))
    assert (p.sm_terminate, p.sm_skip) == (3, 5)


def test_stage_switch_flags_reach_mc_params():
    """-sm_terminate / -sm_skip (main.lua:25-26) are CLI flags of the Python host too and land in mc_params"""
    _, _, opt, prm = mcmain.parse(["kitti", "slow", "-a", "predict", "-sm_terminate", "cbca1", "-sm_skip", "sgm"])
    p = params.make_params(prm)
    assert (p.sm_terminate, p.sm_skip, p.left_only) == (params.SM_TERMINATE["cbca1"], params.SM_SKIP["sgm"], 0)
    _, _, _, prm = mcmain.parse(["mb", "fast", "-a", "predict"])
    assert params.make_params(prm).sm_terminate == 0 and params.make_params(dict(prm, left_only=1)).left_only == 1


def test_normalize_follows_the_reference_order():
    """main.lua:1095: x:add(-x:mean()):div(x:std()) -- the std is taken AFTER the float32 mean subtraction"""
    rng = np.random.default_rng(3)
    x = (rng.random((1, 37, 41)) * 255).astype(np.float32)
    y = (x - np.float32(x.astype(np.float64).mean())).astype(np.float32)
    want = y / np.float32(y.astype(np.float64).std(ddof=1))
    assert np.array_equal(mcmain.normalize(x), want)


def test_normalize_is_torch_unbiased_std():
    rng = np.random.default_rng(0)
    x = (rng.random((1, 7, 9)) * 255).astype(np.float32)
    y = mcmain.normalize(x)
    assert abs(float(y.mean())) < 1e-5
    assert abs(float(y.astype(np.float64).std(ddof=1)) - 1.0) < 1e-5


def test_rgb2y_and_random_net_are_deterministic():
    img = np.stack([np.full((2, 2), 10.0), np.full((2, 2), 20.0), np.full((2, 2), 30.0)]).astype(np.float32)
    assert np.allclose(mcmain.rgb2y(img), 0.299 * 10 + 0.587 * 20 + 0.114 * 30)
    a = mcmain.load_net("random:5", "kitti", "fast")
    b = mcmain.load_net("random:5", "kitti", "fast")
    assert len(a) == 4 and a[0][0].shape == (64, 1, 3, 3) and a[1][0].shape == (64, 64, 3, 3)
    assert all(np.array_equal(x[0], y[0]) for x, y in zip(a, b))


# ---- on-disk formats ---------------------------------------------------------------------------------------
# A net file as torch.save(fname, {net_te, opt}, 'ascii') lays it out (torch7 File.lua / generic/Tensor.c), written by
# hand: a table {1: nn.Sequential{modules = {1: cudnn.SpatialConvolution{weight 2x1x1x2 (a view into a 6-element storage
# at offset 2), bias, nOutputPlane, kH, kW}, 2: cudnn.ReLU{inplace}}}, 2: {fm = 2, name = "kitti"}}.
T7_ASCII_SAMPLE = b"""3
1
2
1
1
4
2
3
V 1
13
nn.Sequential
3
3
1
2
7
modules
3
4
2
1
1
4
5
3
V 1
24
cudnn.SpatialConvolution
3
6
5
2
6
weight
4
7
3
V 1
16
torch.CudaTensor
4
2 1 1 2
2 2 2 1
2
4
8
3
V 1
17
torch.CudaStorage
6
9 0.5 -1.25 3 4e-1 7
2
4
bias
4
9
3
V 1
16
torch.CudaTensor
1
2
1
1
4
10
3
V 1
17
torch.CudaStorage
2
0.125 -8
2
12
nOutputPlane
1
2
2
2
kH
1
1
2
2
kW
1
2
1
2
4
11
3
V 1
10
cudnn.ReLU
3
12
1
2
7
inplace
5
1
1
2
3
13
2
2
2
fm
1
2
2
4
name
2
5
kitti
"""


def test_t7_reader_on_a_hand_written_ascii_net():
    from mc_cnn_amd import t7
    obj = t7.load(T7_ASCII_SAMPLE)
    assert sorted(obj) == [1, 2] and obj[2] == {"fm": 2, "name": "kitti"}
    seq = obj[1]
    assert seq.cls == "nn.Sequential" and seq["modules"][2].cls == "cudnn.ReLU" and seq["modules"][2]["inplace"] is True
    (w, b), = t7.conv_layers(seq)
    assert w.shape == (2, 1, 1, 2) and w.dtype == np.float32
    np.testing.assert_array_equal(w.reshape(2, 2), np.array([[0.5, -1.25], [3.0, 0.4]], np.float32))  # offset 2, strides 2,2,2,1
    np.testing.assert_array_equal(b, np.array([0.125, -8.0], np.float32))


def test_t7_round_trip_of_a_slow_net_and_main_loaders(tmp_path):
    from mc_cnn_amd import t7
    rng = np.random.default_rng(5)
    convs = [(rng.standard_normal((4, 1 if i == 0 else 4, 3, 3)).astype(np.float32), rng.standard_normal(4).astype(np.float32))
             for i in range(2)]
    fcs = [(rng.standard_normal((6, 8)).astype(np.float32), rng.standard_normal(6).astype(np.float32)),
           (rng.standard_normal((1, 6)).astype(np.float32), rng.standard_normal(1).astype(np.float32))]

    def seq(mods):
        return t7.T7Object("nn.Sequential", {"modules": mods, "train": False})

    relu = lambda: t7.T7Object("cudnn.ReLU", {"inplace": True})
    net_te = seq([m for w, b in convs for m in (t7.T7Object("cudnn.SpatialConvolution", {
        "weight": w, "bias": b, "nInputPlane": w.shape[1], "nOutputPlane": w.shape[0], "kH": 3, "kW": 3, "padH": 1, "padW": 1}), relu())])
    shared = relu()  # the same object twice: written once, referenced by index the second time
    net_te2 = seq([t7.T7Object("nn.SpatialConvolution1_fw", {"weight": fcs[0][0], "bias": fcs[0][1].reshape(1, -1, 1, 1)}), shared,
                   t7.T7Object("nn.SpatialConvolution1_fw", {"weight": fcs[1][0], "bias": fcs[1][1].reshape(1, -1, 1, 1)}), shared,
                   t7.T7Object("cudnn.Sigmoid", {"inplace": True})])
    path = str(tmp_path / "net_kitti_slow.t7")
    t7.save(path, [net_te, net_te2, {"a": "train_tr", "l1": 2, "fm": 4, "at": 0.5, "debug": False}])
    obj = t7.load(path)
    assert obj[3] == {"a": "train_tr", "l1": 2, "fm": 4, "at": 0.5, "debug": False}
    assert obj[2]["modules"][2] is obj[2]["modules"][4]
    got_c, got_f = t7.load_reference_net(path, "slow")
    for (w, b), (w2, b2) in zip(convs + fcs, got_c + got_f):
        np.testing.assert_array_equal(w, w2)
        np.testing.assert_array_equal(b, b2)
    # main.py picks the .t7 up by extension
    lc = mcmain.load_net(path, "kitti", "slow")
    lf = mcmain.load_fc(path, "kitti")
    np.testing.assert_array_equal(lc[1][0], convs[1][0])
    np.testing.assert_array_equal(lf[0][0], fcs[0][0])
    # binary mode carries the same objects
    import struct
    blob = struct.pack("<i", 1) + struct.pack("<d", 2.5)
    assert t7.load(blob, mode="binary") == 2.5


def test_dataset_bin_with_sidecars_round_trip(tmp_path):
    from mc_cnn_amd import binio
    x = (np.arange(2 * 3 * 4, dtype=np.float32) * 0.5).reshape(2, 3, 4)
    nnz = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.int32)
    binio.tofile(str(tmp_path / "x0.bin"), x)
    binio.tofile(str(tmp_path / "nnz.bin"), nnz)
    assert (tmp_path / "x0.bin.dim").read_text().split() == ["2", "3", "4"] and (tmp_path / "x0.bin.type").read_text() == "float32"
    np.testing.assert_array_equal(binio.fromfile(str(tmp_path / "x0.bin")), x)
    got = binio.fromfile(str(tmp_path / "nnz.bin"))
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, nnz)
    (tmp_path / "empty.bin").write_bytes(b"")
    (tmp_path / "empty.bin.dim").write_text("0\n")
    (tmp_path / "empty.bin.type").write_text("float32")
    assert binio.fromfile(str(tmp_path / "empty.bin")).size == 0


def test_png16_and_pfm_follow_adcensus(tmp_path):
    from mc_cnn_amd import binio
    d = np.array([[0.0, 1e-6, 1.0, 12.5], [3.00390625, 255.99, 100.0, 0.5]], np.float32)
    p = str(tmp_path / "d.png")
    binio.write_png16(d, p)
    back = binio.read_png16(p)
    want = np.array([[0, 0, 256, 3200], [769, 65533, 25600, 128]], np.float32) / 256.0
    np.testing.assert_array_equal(back, want.astype(np.float32))
    q = str(tmp_path / "d.pfm")
    binio.write_pfm(d, q)
    raw = open(q, "rb").read()
    head = b"Pf\n4 2\n-0.003922\n"
    assert raw.startswith(head) and np.array_equal(np.frombuffer(raw[len(head):], "<f4").reshape(2, 4), d)


def test_predict_kitti_host_logic(tmp_path):
    """predict_kitti.lua:40-83 on the host side: file layout, PNG16 ground truth, 3-pixel error, sharding, submit files."""
    from PIL import Image
    from mc_cnn_amd import binio, predict_kitti as pk
    root = tmp_path / "unzip"
    for d in ("training/image_0", "training/image_1", "training/disp_noc", "testing/image_0", "testing/image_1"):
        (root / d).mkdir(parents=True)
    H, W = 6, 9
    rng = np.random.default_rng(0)
    gts = []
    for i in range(3):
        for cam in (0, 1):
            Image.fromarray(rng.integers(0, 255, (H, W), dtype=np.uint8)).save(root / ("training/image_%d/%06d_10.png" % (cam, i)))
            Image.fromarray(rng.integers(0, 255, (H, W), dtype=np.uint8)).save(root / ("testing/image_%d/%06d_10.png" % (cam, i)))
        gt = rng.integers(0, 40, (H, W)).astype(np.float32)
        gt[rng.random((H, W)) < 0.3] = 0            # no ground truth
        binio.write_png16(gt, str(root / ("training/disp_noc/%06d_10.png" % i)))
        gts.append(gt)
    assert pk.pair_paths("p", "submit", 7) == ("p/testing/image_0/000007_10.png", "p/testing/image_1/000007_10.png")
    seen = []

    def fake_predict(im0, im1):                      # ground truth shifted by 5 where the left image is bright
        i = int(os.path.basename(im0)[:6])
        seen.append(i)
        left = np.asarray(Image.open(im0)).astype(np.float32)
        return gts[i] + np.where(left > 128, 5.0, 1.0).astype(np.float32)

    import os
    logs = []
    total, done = pk.run("test", str(root), fake_predict, n_pairs=3, log=lambda *a: logs.append(a))
    assert done == 3 and seen == [0, 1, 2] and [a[0] for a in logs] == [0, 1, 2]
    want = 0.0
    for i in range(3):
        left = np.asarray(Image.open(root / ("training/image_0/%06d_10.png" % i)))
        m = gts[i] != 0
        want += ((left > 128) & m).sum() / m.sum()
    assert abs(total - want) < 1e-12
    # a staged predictor (submit / result) with pairs in flight: same order, same total
    class Staged:
        def __init__(self):
            self.log = []

        def submit(self, im0, im1):
            self.log.append(("submit", int(os.path.basename(im0)[:6])))
            return (im0, im1)

        def result(self, t):
            self.log.append(("result", int(os.path.basename(t[0])[:6])))
            return fake_predict(*t)

    st = Staged()
    seen.clear()
    total2, done2 = pk.run("test", str(root), st, n_pairs=3, log=lambda *a: None, in_flight=2)
    assert done2 == 3 and abs(total2 - want) < 1e-12
    assert st.log == [("submit", 0), ("submit", 1), ("result", 0), ("submit", 2), ("result", 1), ("result", 2)]
    # two ranks split the pairs; each writes its own submission files
    seen.clear()
    for r in (0, 1):
        pk.run("submit", str(root), fake_predict, world=2, rank=r, out_dir=str(tmp_path / "out"), n_pairs=3, log=lambda *a: None)
    assert sorted(seen) == [0, 1, 2] and sorted(p.name for p in (tmp_path / "out").iterdir()) == ["%06d_10.png" % i for i in range(3)]
    back = binio.read_png16(str(tmp_path / "out" / "000001_10.png"))
    assert back.shape == (H, W)


@pytest.mark.parametrize("name,mode", [("net_tiny_fast_ascii.t7", "ascii"), ("net_tiny_fast_binary.t7", "binary")])
def test_t7_reader_on_fixtures_from_an_independent_writer(name, mode):
    """tests/golden/net_tiny_fast_*.t7 were written by tests/golden/make_t7_fixture.c, a C restatement of torch7's WRITER side
    (File.lua writeObject, Tensor / Storage write, THDiskFile's ASCII and binary conventions) that shares no code with the
    reader: `{net_te, opt}` of a fast-architecture net whose four parameter tensors are views into ONE torch.CudaStorage
    (written once, referenced by index afterwards), an nn.Sequential of cudnn.SpatialConvolution / cudnn.ReLU / nn.Normalize2 /
    nn.StereoJoin1.  Still not a file written by Torch itself (none exists in the image): README says so."""
    import os
    from mc_cnn_amd import t7
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
    obj = t7.load(path, mode=mode)
    assert sorted(obj) == [1, 2]
    net, opt = obj[1], obj[2]
    assert net.cls == "nn.Sequential" and net["train"] is False
    mods = [net["modules"][k] for k in sorted(net["modules"])]
    assert [m.cls for m in mods] == ["cudnn.SpatialConvolution", "cudnn.ReLU", "cudnn.SpatialConvolution", "nn.Normalize2", "nn.StereoJoin1"]
    assert mods[1]["inplace"] is True and mods[0]["nOutputPlane"] == 4 and mods[0]["padW"] == 1
    assert opt == {"a": "train_all", "l1": 2, "fm": 4, "debug": False}
    (w1, b1), (w2, b2) = t7.conv_layers(net)
    exp = (0.5 * np.sin(0.37 * np.arange(188))).astype(np.float32)   # parameter k of the shared storage
    assert w1.shape == (4, 1, 3, 3) and w2.shape == (4, 4, 3, 3)
    assert np.array_equal(w1.ravel(), exp[:36]) and np.array_equal(b1, exp[36:40])
    assert np.array_equal(w2.ravel(), exp[40:184]) and np.array_equal(b2, exp[184:188])
    # main.py's loader takes the same file (main.lua:894-898)
    if mode == "ascii":
        convs, fcs = t7.load_reference_net(path, "fast")
        assert fcs is None and len(convs) == 2 and np.array_equal(convs[0][0], w1) and np.array_equal(convs[1][1], b2)


def test_png16_and_pfm_behind_the_c_abi(tmp_path):
    """adcensus.readPNG16 / writePNG16 / writePFM (adcensus.cu:1670-1721) as host entries of libmcadcensus.so (mc_read_png16,
    mc_write_png16, mc_write_pfm: what the Lua shim binds), against the Python host's restatement (binio, through PIL / libpng)
    in both directions, on filtered and unfiltered files, 8-bit files, and on the error paths"""
    import ctypes as C
    import torch
    from PIL import Image
    import mc_cnn_amd as mc
    from mc_cnn_amd import binio
    A = mc.adcensus
    rng = np.random.default_rng(5)
    H, W = 37, 53
    d = (rng.random((H, W)) * 200).astype(np.float32)
    d[0, :5] = [0.0, 1e-6, 9.9e-6, 1.1e-5, 255.99609375]      # below / around the 1e-5 cut, the largest 16-bit value
    d[1, :3] = [0.00390625, 0.0039, 128.5]                    # exactly one count, just below it
    # C writer -> libpng reader (PIL) and the C reader; identical pixels to the Python writer's file
    pc, pp = str(tmp_path / "c.png"), str(tmp_path / "py.png")
    A.writePNG16(torch.from_numpy(d), H, W, pc)
    binio.write_png16(d, pp)
    a, b = np.asarray(Image.open(pc)), np.asarray(Image.open(pp))
    assert a.dtype == np.uint16 and a.shape == (H, W) and np.array_equal(a, b)
    assert A.png16_size(pp) == (H, W)
    for path in (pc, pp):     # pp was written by libpng with its own filter heuristics: the reader undoes every filter type
        got = torch.empty((H, W), dtype=torch.float32)
        assert A.readPNG16(got, path) == (H, W)
        want = binio.read_png16(path)
        assert got.numpy().tobytes() == want.tobytes()
    # a smooth 16-bit image written by libpng with adaptive filtering (Sub / Up / Average / Paeth rows all occur)
    ys, xs = np.mgrid[0:120, 0:200]
    smooth = (3000 + 40 * ys + 25 * xs + (7 * np.sin(xs / 9.0) * ys)).astype(np.uint16)
    ps = str(tmp_path / "smooth.png")
    Image.fromarray(smooth).save(ps, format="PNG", optimize=True)
    got = torch.empty(smooth.shape, dtype=torch.float32)
    A.readPNG16(got, ps)
    assert got.numpy().tobytes() == binio.read_png16(ps).tobytes()
    # 8-bit greyscale: libpng's expansion to 16 bits (v * 257), then / 256
    p8 = str(tmp_path / "g8.png")
    g8 = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    Image.fromarray(g8).save(p8)
    got = torch.empty((9, 11), dtype=torch.float32)
    A.readPNG16(got, p8)
    want8 = np.where(g8 == 0, np.float32(0), (g8.astype(np.uint16) * 257).astype(np.float32) / np.float32(256))
    assert np.array_equal(got.numpy(), want8)
    # PFM: byte-identical to the Python writer (header "Pf\n<W> <H>\n-0.003922\n", rows as stored)
    qc, qp = str(tmp_path / "c.pfm"), str(tmp_path / "py.pfm")
    A.writePFM(torch.from_numpy(d), qc)
    binio.write_pfm(d, qp)
    assert open(qc, "rb").read() == open(qp, "rb").read()
    # errors: reported like every other entry (rc != 0 + mc_last_error), nothing written through a too small buffer
    small = torch.full((4,), -1.0)
    with pytest.raises(mc._lib.McError, match="buffer holds"):
        A.readPNG16(small, pc)
    assert (small == -1).all()
    with pytest.raises(mc._lib.McError, match="cannot open"):
        A.png16_size(str(tmp_path / "missing.png"))
    open(str(tmp_path / "junk.png"), "wb").write(b"not a png at all")
    with pytest.raises(mc._lib.McError, match="not a PNG"):
        A.png16_size(str(tmp_path / "junk.png"))
    rgb = str(tmp_path / "rgb.png")
    Image.fromarray(rng.integers(0, 255, (5, 5, 3), dtype=np.uint8)).save(rgb)
    with pytest.raises(mc._lib.McError, match="greyscale"):
        A.png16_size(rgb)
    bad = bytearray(open(pc, "rb").read()); bad[40] ^= 0xff
    open(str(tmp_path / "crc.png"), "wb").write(bytes(bad))
    with pytest.raises(mc._lib.McError, match="CRC|zlib"):
        A.readPNG16(torch.empty((H, W)), str(tmp_path / "crc.png"))
