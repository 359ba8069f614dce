"""Capture golden bytes of the demo scripts' per-frame post-processing from the REAL reference -- authoring container only.

demo/demo_face.py cannot be imported (it parses argv and runs main() on absolute paths at import), so this script takes the
reference's own STATEMENTS out of its source with `ast` and executes exactly those:
    * the module-level `IMG_MEAN = ...` assignment (:27) and the whole `sample_img` function (:94-103);
    * from main(): the three statements computing renorm_ref_img / ref_mean / ref_std (:180-182) and the five statements computing
      gen_mean, gen_std, norm_rec_tar_imgs, rec_tar_imgs (re-normalised) and rec_tar_img (:195-199).
Nothing is retyped: the AST nodes of the reference file are compiled and run, on PRNG frames, with two stand-ins for things this
image lacks -- `.cuda()` is a no-op and `cv2.cvtColor(x, cv2.COLOR_BGR2RGB)` is the channel reversal it is for a float HxWx3 array.
The uint8 conversion is the reference's `.astype('uint8')` (:222).  Stored: the frames' bytes (B,H,W,3) and the four statistics.

    python oracle/capture_demo_goldens.py
"""
from __future__ import annotations

import ast
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_FILE = "/root/reference/demo/demo_face.py"
GOLD = os.path.join(ROOT, "tests", "golden")
CASES = [("a", 2, 64, 48, 31), ("b", 1, 33, 17, 32), ("c", 2, 256, 256, 41), ("d", 1, 256, 256, 42)]    # (tag, B, H, W, seed): tests/test_demo_post.py


def reference_pieces():
    if not os.path.exists(REF_FILE):
        raise SystemExit("reference not mounted; goldens can only be captured in the authoring container")
    tree = ast.parse(open(REF_FILE).read(), REF_FILE)
    mod_nodes, ref_stats, frame = [], [], []
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "IMG_MEAN" for t in node.targets):
            mod_nodes.append(node)
        if isinstance(node, ast.FunctionDef) and node.name == "sample_img":
            mod_nodes.append(node)
    want_ref = ["renorm_ref_img", "ref_mean", "ref_std"]
    want_frame = ["gen_mean", "gen_std", "norm_rec_tar_imgs", "rec_tar_imgs", "rec_tar_img"]
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    for node in ast.walk(main):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
            src = ast.unparse(node)
            if name in want_ref:
                ref_stats.append(node)
            elif name in want_frame and "model." not in src:         # not `rec_tar_imgs = model.rec_tar_img.data.cpu()` (:194)
                frame.append(node)
    assert [n.targets[0].id for n in ref_stats] == want_ref, [ast.unparse(n) for n in ref_stats]
    assert [n.targets[0].id for n in frame] == want_frame, [ast.unparse(n) for n in frame]
    return mod_nodes, ref_stats, frame


def compile_nodes(nodes, name):
    m = ast.Module(body=list(nodes), type_ignores=[])
    ast.fix_missing_locations(m)
    return compile(m, name, "exec")


def main():
    from wacv23_tsnet_amd import prng
    mod_nodes, ref_stats, frame = reference_pieces()
    print("reference statements:")
    for n in mod_nodes + ref_stats + frame:
        print("   ", ast.unparse(n).splitlines()[0])
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2RGB = 4
    cv2.cvtColor = lambda x, code: np.ascontiguousarray(x[:, :, ::-1])
    torch.Tensor.cuda = lambda self, *a, **k: self
    ns = {"np": np, "torch": torch, "cv2": cv2}
    exec(compile_nodes(mod_nodes, "demo_face.py:module"), ns)
    code_ref, code_frame = compile_nodes(ref_stats, "demo_face.py:180-182"), compile_nodes(frame, "demo_face.py:195-199")
    arrays, meta = {}, {"cases": {}}
    for tag, B, H, W, seed in CASES:
        g = prng.normal(seed, "rec", (B, 3, H, W))
        rec = (torch.tanh(0.6 * g + torch.tensor([0.1, -0.2, 0.3]).view(1, 3, 1, 1)) * torch.tensor([0.5, 0.35, 0.6]).view(1, 3, 1, 1)).float().contiguous()
        ref = (prng.uniform01(seed, "ref", (1, 3, H, W)) * 255.0 - torch.from_numpy(ns["IMG_MEAN"]).view(1, 3, 1, 1)).float().contiguous()
        ns["ref_img_list"] = [ref]                                   # :180 reads ref_img_list[0]
        exec(code_ref, ns)
        out, gm, gs = [], [], []
        for b in range(B):
            ns["rec_tar_imgs"] = rec[b:b + 1].clone()                # what :194 leaves in rec_tar_imgs
            exec(code_frame, ns)
            out.append(ns["rec_tar_img"].astype("uint8"))            # :222
            gm.append(ns["gen_mean"].view(3).numpy()); gs.append(ns["gen_std"].view(3).numpy())
        arrays[f"{tag}_rgb"] = np.stack(out)
        arrays[f"{tag}_gen_mean"], arrays[f"{tag}_gen_std"] = np.stack(gm), np.stack(gs)
        arrays[f"{tag}_ref_mean"], arrays[f"{tag}_ref_std"] = ns["ref_mean"].view(3).numpy(), ns["ref_std"].view(3).numpy()
        meta["cases"][tag] = dict(B=B, H=H, W=W, seed=seed)
        print(f"[{tag}] {B}x{H}x{W}: bytes min {arrays[f'{tag}_rgb'].min()} max {arrays[f'{tag}_rgb'].max()}")
    meta["torch"] = torch.__version__
    np.savez_compressed(os.path.join(GOLD, "g8_demo_post.npz"), meta=json.dumps(meta), **arrays)
    print("saved", os.path.getsize(os.path.join(GOLD, "g8_demo_post.npz")), "bytes")


if __name__ == "__main__":
    main()
