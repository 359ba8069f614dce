"""Parity goldens on REALISTIC inputs -- authoring container only (needs /root/reference).

Every other golden feeds the model the quick-start recipe (quick_start1.py:12-29): images U[0,1) BEFORE the /255 of set_test_input, so
the three image channels are ~0.002 next to labels / coordinates of O(1) and the image path is numerically almost silent.  The real
caller (demo/demo_face.py:150-192, demo/demo_pose.py:160-200) feeds BGR - IMG_MEAN in [-112, 154], edge-map / skeleton labels and
bounding-box masks.  This script builds such inputs from the reference's own demo assets and runs the REAL reference on them:

  images   demo/face_examples/images/<clip>/*.png, demo/dance_example/images/<vid>/*.jpg, cropped with the crop the reference's loader
           computes (stored with the raster goldens g7 / g9, which the imported loaders produced), PIL `resize` to 256 x 256 (face) or to
           128 x 256 + `resize_square` padding (pose), RGB -> BGR, - IMG_MEAN: the loaders' statements (dataset_video_face.py:316-321,
           dataset_video_pose.py:412-417,450-457) without cv2 (absent here; cvtColor(RGB2BGR) is a channel reversal);
  labels   the reference's own edge maps / bounding boxes / skeleton class maps of those frames as stored in g7 / g9; the face maps go
           through oracle/skimage_resize.py to 256 x 256 (a restatement, parity unpinned -- here it only synthesises an input, and the
           result is stored), `vl2ch` one-hot as the demo scripts do.
  weights  the counter PRNG with bias_std = 0.02 (no checkpoint is reachable), n_blocks = 4, K = 3 as in the demo scripts.

Stored per pair (tests/golden/g10_*.npz): the inputs as uint8 (image bytes before the mean subtraction, class maps, masks) and, as in
capture_seed_sweep.py, crops / lattice / row checksums of rec_tar_img from the reference in fp32 AND fp64, flows, and the pin of
oracle/tsnet_oracle.py on the same inputs.  tests/helpers.golden_case rebuilds the tensors from the bytes.

    python oracle/capture_demo_input_goldens.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch
from PIL import Image, ImageOps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import capture_goldens as CG  # noqa: E402
import capture_seed_sweep as CS  # noqa: E402

IMG_MEAN = np.array((101.84807705937696, 112.10832843463207, 111.65973036298041), dtype=np.float32)      # demo_face.py:27, demo_pose.py:615 (BGR)
FACE = os.path.join(CG.REF, "demo", "face_examples", "images")
DANCE = os.path.join(CG.REF, "demo", "dance_example", "images")


def unpack(bits, width):
    return np.unpackbits(bits, axis=-1)[..., :width]


def face_frame(z, meta, clip, f):
    """(bgr uint8 (256,256,3), edge map 0/1 (256,256), bbox 0/1 (256,256)) of frame f of a face clip: dataset_video_face.py:312-321"""
    import skimage_resize as SR
    c = meta["clips"][clip]
    ys, ye, xs, xe = c["crop"]                                              # get_crop_coords order: [min_y, max_y, min_x, max_x]
    img = Image.open(os.path.join(FACE, clip, c["files"][f].replace(".txt", ".png"))).convert("RGB")
    img = img.crop((xs, ys, xe, ye)).resize((256, 256))                     # self.crop (PIL box order), then PIL's default filter (bicubic)
    bgr = np.asarray(img)[:, :, ::-1].copy()
    w = c["size"][0]
    edges = unpack(z[f"{clip}_edges"][f], w) * 255                          # the loader resizes the uint8 0 / 255 map
    bbox = unpack(z[f"{clip}_bbox"][f], w) * 255
    return bgr, SR.resize_bool(edges.astype(np.uint8)).astype(np.uint8), SR.resize_bool(bbox.astype(np.uint8)).astype(np.uint8)


def pose_frame(z, meta, vid, f):
    """(bgr uint8 (256,256,3), class map (256,256) uint8, bbox 0/1 (256,256)) of frame f of a dance clip: dataset_video_pose.py:346,412-417"""
    c = meta["clips"][vid]
    img = Image.open(os.path.join(DANCE, vid, c["files"][f].replace("_keypoints.json", ".jpg"))).convert("RGB")
    img = img.crop(tuple(c["crop"])).resize((128, 256))
    img = ImageOps.expand(img, (64, 0, 64, 0))                              # resize_square: 128 x 256 -> 256 x 256, black bars
    bgr = np.asarray(img)[:, :, ::-1].copy()
    return bgr, z[f"{vid}_cls_256"][f], unpack(z[f"{vid}_bbox_256"][f], 256)


def tensors(frames_src, frames_tar, label_nc):
    """the demo scripts' call: K sources shared by every driving frame (ref_img_list = ref_imgs.unsqueeze(dim=1), demo_face.py:177-183);
    a batch of B driving frames repeats them"""
    B = len(frames_tar)
    onehot = lambda m: torch.nn.functional.one_hot(torch.from_numpy(m.astype(np.int64)), label_nc).permute(2, 0, 1).float()
    img = lambda b: torch.from_numpy(b.astype(np.float32) - IMG_MEAN).permute(2, 0, 1)
    src_img = [img(b).unsqueeze(0).repeat(B, 1, 1, 1) for b, _, _ in frames_src]
    src_lbl = [onehot(l).unsqueeze(0).repeat(B, 1, 1, 1) for _, l, _ in frames_src]
    src_bbox = [torch.from_numpy(x.astype(np.float32)).unsqueeze(0).repeat(B, 1, 1) for _, _, x in frames_src]
    tar_lbl = torch.stack([onehot(l) for _, l, _ in frames_tar])
    tar_bbox = torch.stack([torch.from_numpy(x.astype(np.float32)) for _, _, x in frames_tar])
    return src_img, src_lbl, src_bbox, tar_lbl, tar_bbox


# (name, model, source clip, source frames, driving clip, driving frames, weight seed)
PAIRS = [("g10_face_test114_to_val024_b1", "face", "test114", (0, 13, 26), "val024", (5,), 140),
         ("g10_face_val024_to_test114_b2", "face", "val024", (0, 12, 24), "test114", (7, 20), 141),
         ("g10_pose_00110_to_00164_b1", "pose", "00110", (0, 10, 20), "00164", (5,), 150),
         ("g10_pose_00164_to_00110_b2", "pose", "00164", (0, 10, 20), "00110", (3, 17), 151)]


def main():
    torch.set_num_threads(CG.THREADS)
    ref_face, ref_pose = CG.import_reference()
    from oracle import tsnet_oracle as O
    z7 = np.load(os.path.join(CG.GOLD, "g7_raster_face.npz")); m7 = json.loads(str(z7["meta"]))
    z9 = np.load(os.path.join(CG.GOLD, "g9_raster_pose.npz")); m9 = json.loads(str(z9["meta"]))
    metas = []
    for name, model, sclip, sfr, tclip, tfr, wseed in PAIRS:
        if model == "face":
            kw = dict(label_nc=2, n_blocks=4, n_source=3)
            fs = [face_frame(z7, m7, sclip, f) for f in sfr]; ft = [face_frame(z7, m7, tclip, f) for f in tfr]
        else:
            kw = dict(label_nc=25, n_blocks=4, n_source=3, pose=True)
            fs = [pose_frame(z9, m9, sclip, f) for f in sfr]; ft = [pose_frame(z9, m9, tclip, f) for f in tfr]
        cfg = O.TSNetConfig(**kw)
        inp = tensors(fs, ft, kw["label_nc"])
        lo, hi = min(float(x.min()) for x in inp[0]), max(float(x.max()) for x in inp[0])
        print(f"[{name}] image range [{lo:.1f}, {hi:.1f}], label pixels {[int((l > 0).sum()) for _, l, _ in fs + ft]}", flush=True)
        extra = {"in_src_bgr": np.stack([b for b, _, _ in fs]), "in_src_lbl": np.stack([l for _, l, _ in fs]).astype(np.uint8),
                 "in_src_bbox": np.packbits(np.stack([x for _, _, x in fs]).astype(np.uint8), axis=-1),
                 "in_tar_lbl": np.stack([l for _, l, _ in ft]).astype(np.uint8),
                 "in_tar_bbox": np.packbits(np.stack([x for _, _, x in ft]).astype(np.uint8), axis=-1)}
        em = dict(inputs="stored", source=[sclip, list(sfr)], driving=[tclip, list(tfr)], image_range=[lo, hi], img_mean_bgr=[float(v) for v in IMG_MEAN])
        metas.append(CS.capture_pair(ref_face, ref_pose, O, "demo", cfg, kw, len(tfr), 256, 256, 0.02, wseed, -1, "demo", name=name, inp=inp,
                                     extra_arrays=extra, extra_meta=em))
    mpath = os.path.join(CG.GOLD, "MANIFEST.json")
    names = {m["name"] for m in metas}
    old = [x for x in json.load(open(mpath)) if x["name"] not in names]
    with open(mpath, "w") as f:
        json.dump(old + metas, f, indent=1)


if __name__ == "__main__":
    main()
