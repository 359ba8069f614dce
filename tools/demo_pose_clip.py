"""Pose clip harness (the caller pattern of demo/demo_pose.py:110-247) on one MI355X:

    checkpoint dict -> TSNetPose(n_blocks=4, use_mask) -> OpenPose-format points of a synthetic dancer in a 1920 x 1080 frame -> skeleton labels
    ON THE DEVICE (class-index skeleton, crop, bounding box, 128 x 256 nearest resize, 256 x 256 padding, one-hot: csrc/raster.hpp)
    -> set_sources once -> per driving frame: forward_target (batch 1, fixed-background composite) + device post-processing -> strips + GIF

and the demo-shaped throughput figure of the pose model (B = 1, label_nc = 25, n_blocks = 4, K = 3, clip mode).  No pretrained checkpoint is
reachable from this image, so the generator is randomly initialised and goes through the reference's .pth schema: the frames are noise, the
path is the real one.

    python tools/demo_pose_clip.py --out gpurun_out/demo_pose --frames 16
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from wacv23_tsnet_amd import demo, raster
from wacv23_tsnet_amd.model import TSNetPose


def synthetic_dancer(n_frames: int) -> np.ndarray:
    """(F,137,2): BODY_25 | 70 face points | 21 + 21 hand points of a schematic figure swinging its arms and shifting its weight."""
    out = np.zeros((n_frames, 137, 2))
    for f, ph in enumerate(np.linspace(0, 2 * np.pi, n_frames, endpoint=False)):
        cx, top, s = 960 + 40 * np.sin(ph), 170.0, 1.0
        sway = 25 * np.sin(ph)
        neck = np.array([cx, top + 120])
        hip = np.array([cx + sway * 0.3, top + 420])
        body = np.zeros((25, 2))
        body[0] = [cx, top + 40]; body[1] = neck; body[8] = hip
        body[15] = [cx - 16, top + 28]; body[16] = [cx + 16, top + 28]; body[17] = [cx - 36, top + 40]; body[18] = [cx + 36, top + 40]
        for side, (sh, el, wr, hp, kn, an, toe, small, heel) in ((-1, (2, 3, 4, 9, 10, 11, 22, 23, 24)), (1, (5, 6, 7, 12, 13, 14, 19, 20, 21))):
            a = 0.9 * np.sin(ph + (0 if side < 0 else np.pi))
            body[sh] = neck + [side * 95, 10]
            body[el] = body[sh] + 130 * np.array([side * np.cos(a) * 0.45, np.sin(a) * 0.2 + 0.85])
            body[wr] = body[el] + 120 * np.array([side * np.cos(a + 0.5) * 0.6, 0.7 + 0.3 * np.sin(a)])
            body[hp] = hip + [side * 55, 5]
            body[kn] = body[hp] + [side * (15 + 10 * np.sin(ph)), 190]
            body[an] = body[kn] + [side * 5 - sway * 0.2, 185]
            body[toe] = body[an] + [side * 45, 22]; body[small] = body[an] + [side * 62, 16]; body[heel] = body[an] + [-side * 14, 14]
        face = np.zeros((70, 2))
        ang = np.linspace(np.pi * 0.05, np.pi * 0.95, 17)
        face[0:17] = np.stack([cx - 38 * np.cos(ang), top + 30 + 46 * np.sin(ang)], 1)
        for k, x0 in enumerate((-30, 6)):
            face[17 + 5 * k:22 + 5 * k] = np.stack([cx + x0 + np.linspace(0, 24, 5), top + 14 - 4 * np.sin(np.linspace(0, np.pi, 5))], 1)
        face[27:31] = np.stack([np.full(4, cx + 0.5), top + 22 + np.linspace(0, 22, 4)], 1)
        face[31:36] = np.stack([cx + np.linspace(-9, 9, 5), top + 50 + 2 * np.sin(np.linspace(0, np.pi, 5))], 1)
        for k, ex in enumerate((-18, 18)):
            a6 = np.linspace(np.pi, -np.pi, 6, endpoint=False)
            face[36 + 6 * k:42 + 6 * k] = np.stack([cx + ex + 8 * np.cos(a6), top + 26 - 4 * np.sin(a6)], 1)
        a12 = np.linspace(np.pi, -np.pi, 12, endpoint=False)
        face[48:60] = np.stack([cx + 16 * np.cos(a12), top + 62 - (5 + 3 * np.sin(2 * ph)) * np.sin(a12)], 1)
        a8 = np.linspace(np.pi, -np.pi, 8, endpoint=False)
        face[60:68] = np.stack([cx + 10 * np.cos(a8), top + 62 - 2.5 * np.sin(a8)], 1)
        face[68] = [cx - 18, top + 26]; face[69] = [cx + 18, top + 26]
        hands = []
        for side, wr in ((1, 7), (-1, 4)):                         # left hand hangs off the left wrist (7), right off the right (4)
            h = np.zeros((21, 2))
            h[0] = body[wr]
            for fi in range(5):
                d = np.array([side * 0.9 * np.cos(0.5 * fi - 1.0), 1.0 + 0.25 * np.sin(0.5 * fi)])
                d /= np.linalg.norm(d)
                for j in range(4):
                    h[1 + 4 * fi + j] = h[0] + d * (14 + 9 * j + 2 * fi)
            hands.append(h)
        out[f] = np.concatenate([body, face, hands[0], hands[1]]) + 0.137          # OpenPose coordinates are not integers
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/demo_pose")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--n-blocks", type=int, default=4)
    ap.add_argument("--timing-frames", type=int, default=200)
    ap.add_argument("--clip", default=None, help="OpenPose points of a real demo clip (demo/dance_example/labels/<clip>, stored with the raster golden "
                    "tests/golden/g9_raster_pose.npz: 00110 or 00164) instead of the synthetic dancer; the frames' pixels stay synthetic")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.makedirs(args.out, exist_ok=True)

    torch.manual_seed(0)
    kw = dict(is_train=False, label_nc=25, n_blocks=args.n_blocks, n_downsampling=3, n_source=3)
    m0 = TSNetPose(**kw)
    ckpt = {net: getattr(m0, net).state_dict() for net in ("img_enc", "lbl_enc", "dec", "fuse_net")}       # demo_pose.py:127-131 schema
    path = os.path.join(args.out, "TSNet_pose_B0004_S000000.pth")
    torch.save(ckpt, path)
    model = TSNetPose(**kw)
    model.load_checkpoint(torch.load(path, map_location="cpu"))
    model = model.cuda()

    K, F = 3, args.frames
    size = (1920, 1080)
    if args.clip:
        import json
        z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g9_raster_pose.npz"))
        meta = json.loads(str(z["meta"]))["clips"][args.clip]
        pts, size = z[f"{args.clip}_pts"], tuple(meta["size"])
        F = min(F, pts.shape[0] - K)
        pts = pts[:F + K]
    else:
        pts = synthetic_dancer(F + K)
    pr, fr = raster.PoseRasteriser(dev), raster.FaceRasteriser(dev)
    t0 = time.perf_counter()
    cls, box, crop = pr.clip_labels(list(pts), size=size)                   # (F+K,256,256) class indices / 0-1 masks on the device
    if args.clip:
        assert list(crop) == meta["crop"]                                   # the crop arithmetic reproduces the reference's on the real clip
    lbl = fr.vl2ch(cls, 25)                                                # vl2ch(label map, "pose") (demo_pose.py:164,170)
    torch.cuda.synchronize()
    t_raster = time.perf_counter() - t0
    present = sorted(int(c) for c in torch.unique(cls).tolist())
    g = torch.Generator().manual_seed(1)
    src_img = [(torch.rand((1, 3, 256, 256), generator=g) * 255.0 - torch.from_numpy(demo.IMG_MEAN).view(1, 3, 1, 1)) for _ in range(K)]
    runner = demo.ClipRunner(model, src_img, [lbl[i:i + 1] for i in range(K)], [box[i:i + 1] for i in range(K)])
    frames = runner.run(lbl[K:], box[K:], out_dir=args.out, name=args.clip or "synthetic_pose")
    print(f"[demo_pose_clip] {frames.shape[0]} frames written to {args.out} (crop {tuple(crop)}, classes present {present}); "
          f"labels of {F + K} frames from the key points: {t_raster * 1e3:.2f} ms")

    for _ in range(20):
        runner.frame(lbl[K:K + 1], box[K:K + 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.timing_frames):
        j = K + i % F
        runner.frame(lbl[j:j + 1], box[j:j + 1])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"[demo_pose_clip] clip mode, B=1, label_nc=25, n_blocks={args.n_blocks}, K=3: {args.timing_frames / dt:.1f} frames/s "
          f"({dt / args.timing_frames * 1e3:.3f} ms per driving frame, composite and device post-processing included)")


if __name__ == "__main__":
    main()
