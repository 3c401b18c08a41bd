"""On-disk formats and the test-time loader: the TIFF reader/writer against
Pillow's libtiff and hand-built files, the infer4rec dataset against the reference's directory layout
(data/infer4rec_dataset.py), and -- on the GPU -- `validate.main` end to end on a dataset on disk
(frames in, *_denoised.tif out) against the CPU oracles fed from the same files."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

import rvdd_oracle as O
import tvl1_oracle as T
from conftest import GOLDEN, WEIGHTS, load_weights

from rvdd_release_amd import tiffio
from rvdd_release_amd.library import (iio_read, iio_write, list_video_files_at_dir, load_image, pathdiff,
                                      warpedimagefile)

RNG = np.random.default_rng(11)


# ------------------------------------------------------------------------------------------- TIFF

@pytest.mark.parametrize("dtype,C", [(np.float32, 4), (np.float32, 2), (np.float32, 3), (np.uint16, 3), (np.uint8, 3),
                                     (np.float32, 1), (np.float64, 1), (np.int16, 2), (np.uint32, 1), (np.uint8, 4)])
def test_tiff_roundtrip(tmp_path, dtype, C):
    a = (RNG.standard_normal((33, 47, C)) * 1000).astype(dtype)
    p = str(tmp_path / "a.tif")
    tiffio.write(p, a)
    b = tiffio.read(p)
    assert b.dtype == a.dtype and b.shape == a.shape and np.array_equal(a, b)


def test_tiff_2d_and_odd_sizes(tmp_path):
    a = RNG.integers(0, 255, (5, 7)).astype(np.uint8)          # odd byte count: IFD must stay word-aligned
    p = str(tmp_path / "a.tif")
    tiffio.write(p, a)
    assert np.array_equal(tiffio.read(p)[..., 0], a)


def test_pillow_reads_our_files(tmp_path):
    from PIL import Image
    p = str(tmp_path / "a.tif")
    for a in ((RNG.random((33, 47)) * 100).astype(np.float32), RNG.integers(0, 65535, (33, 47)).astype(np.uint16),
              RNG.integers(0, 255, (33, 47, 3)).astype(np.uint8)):
        tiffio.write(p, a)
        assert np.array_equal(np.array(Image.open(p)), a)


@pytest.mark.parametrize("compression", [None, "tiff_lzw", "tiff_adobe_deflate", "packbits"])
@pytest.mark.parametrize("predictor", [None, 2])
def test_we_read_pillow_rgb8_and_u16(tmp_path, compression, predictor):
    from PIL import Image
    smooth = (np.add.outer(np.arange(64), np.arange(80)) % 256).astype(np.uint8)
    rgb = np.stack([smooth, smooth[::-1], RNG.integers(0, 255, smooth.shape).astype(np.uint8)], -1)
    u16 = RNG.integers(0, 65535, (64, 80)).astype(np.uint16)
    kw = {}
    if compression:
        kw["compression"] = compression
    if predictor:
        kw["tiffinfo"] = {317: predictor}
    p = str(tmp_path / "a.tif")
    Image.fromarray(rgb).save(p, **kw)
    assert np.array_equal(tiffio.read(p), rgb)
    Image.fromarray(u16).save(p, **kw)
    assert np.array_equal(tiffio.read(p)[..., 0], u16)


@pytest.mark.parametrize("compression,predictor", [(None, None), ("tiff_lzw", None), ("tiff_lzw", 3),
                                                   ("tiff_adobe_deflate", 3)])
def test_we_read_pillow_float(tmp_path, compression, predictor):
    from PIL import Image
    f = (RNG.random((64, 80)) * 100).astype(np.float32)
    kw = {}
    if compression:
        kw["compression"] = compression
    if predictor:
        kw["tiffinfo"] = {317: predictor}
    p = str(tmp_path / "a.tif")
    Image.fromarray(f).save(p, **kw)
    assert np.array_equal(tiffio.read(p)[..., 0], f)


def test_native_lzw_decoder_equals_python(tmp_path):
    """LZW strips go through `rvdd_tiff_lzw_decode` (host code in librvdd_hip.so) when the library loads; the
    pure-Python decoder is the independent restatement.  Same bytes out, including the error behaviour."""
    import ctypes
    from PIL import Image
    from rvdd_release_amd import _lib
    dec = _lib.load().rvdd_tiff_lzw_decode
    smooth = (np.add.outer(np.arange(97), np.arange(131)) % 251).astype(np.uint8)
    for img in (smooth, RNG.integers(0, 255, (97, 131)).astype(np.uint8), np.zeros((300, 400), np.uint8)):
        p = str(tmp_path / "l.tif")
        Image.fromarray(img).save(p, compression="tiff_lzw")
        raw = open(p, "rb").read()
        # single-strip or multi-strip: decode every strip both ways
        import struct as S
        (ifd,) = S.unpack("<I", raw[4:8])
        (n,) = S.unpack("<H", raw[ifd:ifd + 2])
        tags = {}
        for i in range(n):
            tag, typ, cnt, val = S.unpack("<HHII", raw[ifd + 2 + 12 * i: ifd + 14 + 12 * i])
            tags[tag] = (typ, cnt, val)

        def values(tag):
            typ, cnt, val = tags[tag]
            if cnt == 1:
                return [val]
            f = {3: "H", 4: "I"}[typ]
            return list(S.unpack("<" + f * cnt, raw[val:val + cnt * S.calcsize(f)]))
        total = b""
        for off, cnt in zip(values(273), values(279)):
            chunk = raw[off:off + cnt]
            want = tiffio._lzw_decode(chunk)
            out = ctypes.create_string_buffer(len(want) + 16)
            got = dec(chunk, len(chunk), out, len(out))
            assert got == len(want) and out.raw[:got] == want
            assert dec(chunk, len(chunk), out, len(want) - 1) == -1 if len(want) > 1 else True       # output too small
            total += want
        assert np.array_equal(np.frombuffer(total, np.uint8)[:img.size].reshape(img.shape), img)
    bad = bytes([0xFF, 0xFF, 0xFF, 0xFF])                                     # code 511 first: not in the table
    assert dec(bad, len(bad), ctypes.create_string_buffer(64), 64) == -1
    with pytest.raises(tiffio.TiffError):
        tiffio._lzw_decode(bad)


def _handmade(path, arr, bo="<", big=False, planar=1, tile=None, rows_per_strip=None, deflate=False):
    """A TIFF writer independent of tiffio.write: endianness, BigTIFF, planar/tiled/multi-strip layouts."""
    H, W, C = arr.shape
    dt = arr.dtype.newbyteorder(bo)
    fmtcode = {"u": 1, "i": 2, "f": 3}[arr.dtype.kind]
    planes = [arr[..., c:c + 1] for c in range(C)] if planar == 2 else [arr]
    chunks = []
    for pl in planes:
        if tile:
            th, tw = tile
            for y in range(0, H, th):
                for x in range(0, W, tw):
                    blk = np.zeros((th, tw, pl.shape[2]), arr.dtype)
                    sub = pl[y:y + th, x:x + tw]
                    blk[:sub.shape[0], :sub.shape[1]] = sub
                    chunks.append(blk.astype(dt).tobytes())
        else:
            rps = rows_per_strip or H
            for y in range(0, H, rps):
                chunks.append(np.ascontiguousarray(pl[y:y + rps]).astype(dt).tobytes())
    if deflate:
        chunks = [zlib.compress(c) for c in chunks]
    osz, ofmt = (8, "Q") if big else (4, "I")
    head = (bo == "<" and b"II" or b"MM") + (struct.pack(bo + "HHHQ", 43, 8, 0, 0) if big else struct.pack(bo + "HI", 42, 0))
    body = b""
    offsets = []
    pos = len(head)
    for c in chunks:
        offsets.append(pos + len(body))
        body += c
    ent = []

    def add(tag, typ, vals):
        ent.append((tag, typ, vals))
    S, L = 3, (16 if big else 4)
    add(256, S, (W,)); add(257, S, (H,)); add(258, S, (arr.dtype.itemsize * 8,) * C)
    add(259, S, (8 if deflate else 1,)); add(262, S, (1,)); add(277, S, (C,)); add(284, S, (planar,)); add(339, S, (fmtcode,) * C)
    if tile:
        add(322, S, (tile[1],)); add(323, S, (tile[0],)); add(324, L, tuple(offsets)); add(325, L, tuple(len(c) for c in chunks))
    else:
        add(273, L, tuple(offsets)); add(278, S, (rows_per_strip or H,)); add(279, L, tuple(len(c) for c in chunks))
    ent.sort()
    ifd_off = len(head) + len(body)
    n = len(ent)
    ifd_len = (8 + 20 * n + 8) if big else (2 + 12 * n + 4)
    tail = b""
    ifd = struct.pack(bo + ("Q" if big else "H"), n)
    for tag, typ, vals in ent:
        f = {3: "H", 4: "I", 16: "Q"}[typ]
        packed = struct.pack(bo + f * len(vals), *vals)
        if len(packed) <= osz:
            field = packed.ljust(osz, b"\0")
        else:
            field = struct.pack(bo + ofmt, ifd_off + ifd_len + len(tail))
            tail += packed
        ifd += struct.pack(bo + "HH" + ofmt, tag, typ, len(vals)) + field
    ifd += struct.pack(bo + ofmt, 0)
    head = head[:4] + struct.pack(bo + "I", ifd_off) if not big else head[:8] + struct.pack(bo + "Q", ifd_off)
    with open(path, "wb") as fh:
        fh.write(head + body + ifd + tail)


@pytest.mark.parametrize("kw", [dict(bo=">"), dict(big=True), dict(planar=2), dict(tile=(16, 32)), dict(rows_per_strip=5),
                                dict(planar=2, tile=(16, 16), bo=">"), dict(rows_per_strip=7, deflate=True),
                                dict(big=True, bo=">", rows_per_strip=1)])
def test_tiff_layout_variants(tmp_path, kw):
    for dtype, C in ((np.float32, 4), (np.uint16, 3)):
        a = (RNG.random((37, 53, C)) * 4000).astype(dtype)
        p = str(tmp_path / "a.tif")
        _handmade(p, a, **kw)
        b = tiffio.read(p)
        assert b.dtype == a.dtype and np.array_equal(a, b)


def test_tiff_errors(tmp_path):
    p = str(tmp_path / "bad.tif")
    with open(p, "wb") as f:
        f.write(b"not a tiff at all")
    with pytest.raises(tiffio.TiffError):
        tiffio.read(p)
    a = RNG.random((8, 8, 2)).astype(np.float32)
    tiffio.write(p, a)
    raw = open(p, "rb").read()
    # cut the pixel data short but keep the directory: rebuild with a lying StripByteCounts
    _handmade(p, a)
    with open(p, "r+b") as f:
        f.seek(8)
        f.truncate(8 + 40)
    with pytest.raises((tiffio.TiffError, struct.error)):
        tiffio.read(p)
    with pytest.raises(tiffio.TiffError):
        tiffio.write(p, np.zeros((4, 4), np.complex64))
    assert len(raw) == 8 + a.nbytes + 2 + 12 * 12 + 4              # header + data + 12-entry IFD (2-channel tags fit inline)


# ------------------------------------------------------------------------------- library helpers

def test_library_file_helpers(tmp_path):
    d = tmp_path / "noisy" / "000"
    d.mkdir(parents=True)
    for i in (3, 0, 6):
        iio_write((np.full((4, 6, 4), i * 100.0, np.float32)), str(d / ("%08d.tiff" % i)))
    (d / "notes.txt").write_text("x")
    files = list_video_files_at_dir(str(d))
    assert [os.path.basename(f) for f in files] == ["00000000.tiff", "00000003.tiff", "00000006.tiff"]
    assert pathdiff(files[1], str(tmp_path / "noisy")) == "000"
    assert warpedimagefile("/w", "00000000", "00000003") == "/w/00000000_00000003.tif"
    img = load_image(files[2], 12)
    assert img.dtype == np.float32 and img.shape == (4, 6, 4) and np.all(img == np.float32(600.0) / np.float32(4095.0))
    with pytest.raises(AssertionError):
        list_video_files_at_dir(str(tmp_path))
    png = str(tmp_path / "a.png")
    rgb = RNG.integers(0, 255, (5, 6, 3)).astype(np.uint8)
    iio_write(rgb, png)
    assert np.array_equal(iio_read(png), rgb)


def test_tensor2im_matches_reference():
    from rvdd_release_amd.util.util import tensor2im
    g = np.load(os.path.join(GOLDEN, "ppipe_den3200_seq0.npz"))     # 'tif' came from the reference's tensor2im
    assert np.array_equal(tensor2im(torch.from_numpy(g["x"])), g["tif"])


# --------------------------------------------------------------------------------------- dataset

def write_dataset(root, seqs, iso=3200, with_flows=True, future=0):
    """The reference's validation layout (scripts/test-recurrent-feat-convunet.sh, data/infer4rec_dataset.py:64-80)."""
    for v, s in enumerate(seqs):
        nd = os.path.join(root, f"noisy_iso{iso}", "%03d" % v)
        gd = os.path.join(root, f"gt_raw_linear_RGB_iso{iso}", "%03d" % v)
        fd = os.path.join(root, "flow", f"noisy_iso{iso}", "tvl1", "noisyinputs", "%03d" % v)
        for d in (nd, gd, fd):
            os.makedirs(d, exist_ok=True)
        Tn = s.raw.shape[0]
        for t in range(Tn):
            code = "%08d" % (3 * t)
            tiffio.write(os.path.join(nd, code + ".tiff"), ((s.raw[t].permute(1, 2, 0).numpy() + 1) / 2 * 4095).astype(np.float32))
            tiffio.write(os.path.join(gd, code + ".tiff"),
                         np.round((s.gt[t].permute(1, 2, 0).numpy() + 1) / 2 * 4095).clip(0, 4095).astype(np.uint16))
            if with_flows and t > 0:
                tiffio.write(os.path.join(fd, "%08d_%s.tif" % (3 * (t - 1), code)), s.flow_prev[t].permute(1, 2, 0).numpy())
            if with_flows and future and t < Tn - 1:
                tiffio.write(os.path.join(fd, "%08d_%s.tif" % (3 * (t + 1), code)), s.flow_next[t].permute(1, 2, 0).numpy())


def _opt(root, iso=3200, **kw):
    from rvdd_release_amd.options import make_opt
    return make_opt(val_dataroot=str(root), dataroot=str(root), nFolder=f"noisy_iso{iso}", gtFolder=f"gt_iso{iso}",
                    gt_linear_RGB_Folder=f"gt_raw_linear_RGB_iso{iso}", dataset_mode="infer4rec", serial_batches=True,
                    patch_depth=2, max_dataset_size=float("inf"), **kw)


@pytest.mark.parametrize("future", [0, 1])
def test_infer4rec_dataset_layout(tmp_path, future):
    from rvdd_release_amd import synth
    from rvdd_release_amd.data import create_dataset
    seqs = [synth.make_sequence(4, 32, 48, iso=3200, seed=70 + v) for v in range(3)]
    write_dataset(str(tmp_path), seqs, future=future)
    ds = create_dataset(_opt(tmp_path, videos="000,002", future_patch_depth=future))
    per_video = 4 - 2 - future + 1
    assert len(ds) == 2 * per_video
    items = list(ds)
    assert len(items) == len(ds)
    for k, it in enumerate(items):
        v, j = (0, 2)[k // per_video], k % per_video
        s = seqs[v]
        assert it["n"].shape == (1, 4 * (2 + future), 16, 24) and it["gt"].shape == (1, 6, 32, 48)
        assert it["flow"].shape == (1, 1 + future, 2, 16, 24)
        assert it["n_path"] == [str(tmp_path / "noisy_iso3200" / ("%03d" % v) / ("%08d.tiff" % (3 * (j + 1))))]
        assert os.path.dirname(it["gt_path"][0]).endswith("gt_raw_linear_RGB_iso3200/%03d" % v)
        # values: file = (raw+1)/2*4095 in fp32; loader = 2*(file/4095) - 1
        for f in range(2 + future):
            stored = ((s.raw[j + f].permute(1, 2, 0).numpy() + 1) / 2 * 4095).astype(np.float32)
            want = 2. * torch.from_numpy((stored / np.float32(4095.0)).transpose(2, 0, 1)) - 1.
            assert torch.equal(it["n"][0, 4 * f:4 * f + 4], want)
            assert (it["n"][0, 4 * f:4 * f + 4] - s.raw[j + f]).abs().max() < 1e-6
        assert (it["gt"][0, 3:6] - s.gt[j + 1]).abs().max() < 2.5e-4            # uint16 quantisation of the ground truth
        assert torch.equal(it["flow"][0, 0], s.flow_prev[j + 1])
        if future:
            assert torch.equal(it["flow"][0, 1], s.flow_next[j + 1])


def test_infer4rec_crop_and_missing_video(tmp_path):
    from rvdd_release_amd import synth
    from rvdd_release_amd.data import create_dataset
    seqs = [synth.make_sequence(3, 32, 48, iso=3200, seed=5)]
    write_dataset(str(tmp_path), seqs)
    it = next(iter(create_dataset(_opt(tmp_path, videos="000", crop_data="8,12"))))
    assert it["n"].shape == (1, 8, 8, 12) and it["gt"].shape == (1, 6, 16, 24)
    os.makedirs(tmp_path / "noisy_iso3200" / "001")
    with pytest.raises(AssertionError):
        create_dataset(_opt(tmp_path))                                          # gt / noisy folder counts differ


# ------------------------------------------------------------------------------------------- GPU

def _oracle_from_items(sd, items, future):
    outs = []
    rec = None
    last = ''
    for it in items:
        first = os.path.dirname(it["gt_path"][0]) != last
        last = os.path.dirname(it["gt_path"][0])
        if first:
            rec = O.RecurrentOracle(sd, future=future)
        n, fl = it["n"], it["flow"]
        outs.append(rec.step(n[:, 0:4], n[:, 4:8], n[:, 8:12] if future else None, fl[:, 0],
                             fl[:, 1] if future else None, first=first))
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("name,net,feat,future", [("recurrent-convunet+feat-iso3200", "convunet-mode=fixedfeatures+feat", True, 0),
                                                  ("recurrent-convunet+feat-future-iso12800", "convunet-mode=fixedfeatures+feat", True, 1)])
def test_validate_main_on_disk(tmp_path, name, net, feat, future):
    from rvdd_release_amd import synth, validate
    from rvdd_release_amd.data import create_dataset
    iso = 12800 if "12800" in name else 3200
    seqs = [synth.make_sequence(4 + future, 64, 96, iso=iso, seed=80 + v) for v in range(2)]
    root = tmp_path / "validation"
    write_dataset(str(root), seqs, iso=iso, future=future)
    ck = tmp_path / "checkpoints"
    argv = ["--netDenoiser", net, "--path2epoch", os.path.join(WEIGHTS, name), "--val_dataroot", str(root),
            "--gtFolder", f"gt_iso{iso}", "--nFolder", f"noisy_iso{iso}", "--gt_linear_RGB_Folder", f"gt_raw_linear_RGB_iso{iso}",
            "--suffix", "t", "--checkpoints_dir", str(ck), "--val_videos", "000,001", "--future_patch_depth", str(future)]
    if feat:
        argv.append("--feature_rec")
    res = validate.main(argv)
    out_dir = ck / f"recurrent-{net}-warp-i3o3-t" / "val_visuals"
    items = list(create_dataset(_opt(root, iso=iso, videos="000,001", future_patch_depth=future)))
    want = _oracle_from_items(load_weights(name), items, future)
    assert len(items) == 2 * 3
    psnrs = []
    for it, w in zip(items, want):
        v = os.path.basename(os.path.dirname(it["n_path"][0]))
        code = os.path.splitext(os.path.basename(it["n_path"][0]))[0]
        got = tiffio.read(str(out_dir / v / f"{code}_denoised.tif"))
        assert got.dtype == np.float32 and got.shape == (64, 96, 3)
        ref = ((w[0].permute(1, 2, 0).numpy() + 1) / 2.0 * 255.0).astype(np.float32)
        assert np.abs(got - ref).max() < 1e-4 * 127.5
        psnrs.append(O.psnr(w, it["gt"][:, 3:6]))
    assert abs(res["PSNR_valLoss"] - np.mean(psnrs)) < 0.01
    log = (out_dir / "output.log").read_text().strip().splitlines()
    assert len(log) == 6 and log[0].startswith("[L1: ") and "PSNR: " in log[0]


@pytest.mark.gpu
def test_dataset_creates_missing_flows_with_tvl1(tmp_path):
    """--check_data: flow files that do not exist are computed (TV-L1 on the device) and written where the
    reference expects them; content against the NumPy TV-L1 oracle."""
    from rvdd_release_amd import synth
    from rvdd_release_amd.data import create_dataset
    seqs = [synth.make_sequence(3, 64, 96, iso=3200, seed=91)]
    write_dataset(str(tmp_path), seqs, with_flows=False)
    ds = create_dataset(_opt(tmp_path, videos="000", future_patch_depth=1))
    fdir = tmp_path / "flow" / "noisy_iso3200" / "tvl1" / "noisyinputs" / "000"
    assert sorted(os.listdir(fdir)) == ["00000000_00000003.tif", "00000003_00000000.tif", "00000003_00000006.tif",
                                        "00000006_00000003.tif"]
    item = next(iter(ds))
    nz = [tiffio.read(str(tmp_path / "noisy_iso3200" / "000" / ("%08d.tiff" % (3 * t)))) for t in range(3)]
    want_prev = T.TVL1_flow(nz[1], nz[0]).transpose(2, 0, 1)       # flow from frame 1 to frame 0
    want_next = T.TVL1_flow(nz[1], nz[2]).transpose(2, 0, 1)
    for got, want in ((item["flow"][0, 0].numpy(), want_prev), (item["flow"][0, 1].numpy(), want_next)):
        d = np.abs(got - want)
        assert d.max() < 2e-3 and d.mean() < 1e-4, (float(d.max()), float(d.mean()))


@pytest.mark.gpu
def test_fwd_ppipe_script(tmp_path):
    """dataset/fwd_ppipe.py's script on a results folder: PNGs, PSNR.txt / SSIM.txt, averages against the oracle."""
    import ppipe_oracle as P
    from rvdd_release_amd import ppipe as PP
    from rvdd_release_amd.library import iio_read
    res, val = tmp_path / "val_visuals", tmp_path / "validation"
    want_p, want_s = [], []
    for seq in (0, 4):
        os.makedirs(res / ("%03d" % seq))
        os.makedirs(val / "gt_RGB_iso3200" / ("%03d" % seq))
        n, red, blue = PP.find_gains(seq, 3200)
        for i in (3, 6):
            x = torch.from_numpy(RNG.random((1, 3, 24, 40)).astype(np.float32) * 1.6 - 0.9)
            img = P.tensor2im(x)
            tiffio.write(str(res / ("%03d" % seq) / ("%08d_denoised.tif" % i)), img)
            gt = P.to_uint8(P.ppipe(P.normalise_bit_depth(P.tensor2im((x + 0.03).clamp(-1, 1)), 8), 1 / n, red, blue, 3200))
            iio_write(gt, str(val / "gt_RGB_iso3200" / ("%03d" % seq) / ("%08d.png" % i)))
            mine = P.to_uint8(P.ppipe(P.normalise_bit_depth(img, 8), 1 / n, red, blue, 3200))
            want_p.append(P.psnr_u8(mine, gt))
            want_s.append(P.ssim(mine, gt))
    ap, as_ = PP.main(["--validation_path", str(val), "--result_folder", str(res), "--videos", "0,4", "--first", "3",
                       "--last", "6", "--step", "3", "--bit_depth", "8", "--ISO", "3200"])
    assert abs(ap - np.mean(want_p)) < 0.05 and abs(as_ - np.mean(want_s)) < 1e-3     # +-1 LSB on a few pixels at most
    png = iio_read(str(res / "000" / "00000003_processed_pipeline.png"))
    assert png.dtype == np.uint8 and png.shape == (24, 40, 3)
    assert (res / "PSNR.txt").read_text().rstrip().endswith("dB  ###")


@pytest.mark.gpu
def test_validate_main_no_warp(tmp_path):
    """scripts/test-non_recurrent-no_warp-convunet.sh: `--no_warp`: no flow folder is needed, none is created."""
    from rvdd_release_amd import synth, validate
    from rvdd_release_amd.data import create_dataset
    name = "non_recurrent-convunet-no_warp-iso3200"
    seqs = [synth.make_sequence(4, 64, 96, iso=3200, seed=95)]
    root = tmp_path / "validation"
    write_dataset(str(root), seqs, with_flows=False)
    ck = tmp_path / "checkpoints"
    validate.main(["--netDenoiser", "convunet-mode=fixedfeatures", "--path2epoch", os.path.join(WEIGHTS, name), "--val_dataroot",
                   str(root), "--gtFolder", "gt_iso3200", "--nFolder", "noisy_iso3200", "--gt_linear_RGB_Folder",
                   "gt_raw_linear_RGB_iso3200", "--no_warp", "--suffix", "t", "--checkpoints_dir", str(ck), "--val_videos", "000"])
    assert os.listdir(root / "flow" / "noisy_iso3200" / "tvl1" / "noisyinputs" / "000") == []     # write_dataset made the folder only
    items = list(create_dataset(_opt(root, videos="000", no_warp=True)))
    assert items[0]["flow"] == [] and len(items) == 3
    rec = O.RecurrentOracle(load_weights(name), future=0, no_warp=True)
    out_dir = ck / "recurrent-convunet-mode=fixedfeatures-i3o3-t" / "val_visuals" / "000"
    for k, it in enumerate(items):
        w = rec.step(it["n"][:, 0:4], it["n"][:, 4:8], None, None, None, first=(k == 0))
        got = tiffio.read(str(out_dir / (os.path.splitext(os.path.basename(it["n_path"][0]))[0] + "_denoised.tif")))
        ref = ((w[0].permute(1, 2, 0).numpy() + 1) / 2.0 * 255.0).astype(np.float32)
        assert np.abs(got - ref).max() < 1e-4 * 127.5
