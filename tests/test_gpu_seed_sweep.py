"""GPU tier: the 1e-3 parity gate of BASELINE.json configs[1] over a SWEEP of (weight seed, input seed) pairs -- and over two / two /
one pairs at the shapes of configs[2], [3], [4] (face checkpoint n_blocks = 4; pose model with the composite; 512 x 512, K = 5) -- and the
same outputs against the reference run in fp64.

For every pair captured by oracle/capture_seed_sweep.py (tests/golden/g6_cfg1_w*_i*.npz: crops, a 64 x 64 lattice over the
frame, per-row checksums and flows of the REAL reference in fp32 and in fp64) this test
  1. gates the HIP forward against the stored fp32 reference data:  max|GPU - ref32| <= 1e-3 on the crops and the lattice
     (67 k pixels per pair), every row checksum within W x 1e-3, flows <= 1e-4;
  2. re-runs the oracle on this host in fp64 -- pinned to the stored fp64 reference data to 1e-9, fp64 does not depend on the
     host -- and separates "GPU wrong" from "fp32 reference noisy" over the WHOLE image:
         max|GPU - ref64| <= max|ref32 - ref64| + 1e-4,
     the right-hand side being the reference's own fp32 rounding noise on this pair, recorded at capture;
  3. re-runs the oracle in fp32 on this host (what bench.py's parity figure uses) and gates max|GPU - oracle32| <= 1e-3 over
     the whole image.  An fp32 CPU forward depends on the host (oneDNN picks other blockings on another ISA: measured
     0.8 - 2.1e-4 between the authoring container's Xeon and the GPU box's EPYC), so this run is only required to sit within
     the reference's own noise class of the stored data, not to reproduce it.
The margins are printed and, when the directory exists, written to gpurun_out/seed_sweep_margins.json."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import tsnet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL_REC, TOL_FLOW = 1e-3, 1e-4
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(Hh.GOLD, "g6_cfg*_w*_i*.npz")))
# ... and pairs on REALISTIC inputs (oracle/capture_demo_input_goldens.py): frames of the reference's demo clips as demo_face.py:150-192 /
# demo_pose.py feed them (BGR - IMG_MEAN in [-112, 154], edge-map / skeleton labels, box masks; n_blocks = 4, K = 3, B = 1 and 2), the
# reference itself run on them in fp32 and fp64.  Same gates.
NAMES += sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(Hh.GOLD, "g10_*.npz")))
_rows = []


def test_sweep_has_at_least_eight_pairs():
    assert sum(n.startswith("g6_cfg1_") for n in NAMES) >= 8
    assert sum(n.startswith("g10_face_") for n in NAMES) >= 2 and sum(n.startswith("g10_pose_") for n in NAMES) >= 2
    for other in ("g6_cfg2_", "g6_cfg3_", "g6_cfg4_"):          # the shapes of BASELINE.json configs[2..4]: n_blocks = 4; pose model; 512 x 512 with K = 5
        assert any(n.startswith(other) for n in NAMES), other


@pytest.mark.parametrize("name", NAMES)
def test_cfg1_pair(name):
    meta, z, cfg, sd, inputs = Hh.golden_case(name)
    CROPS = {"c": (slice(96, 128), slice(96, 128)), "tl": (slice(0, 16), slice(0, 16)),
             "br": (slice(meta["H"] - 16, meta["H"]), slice(meta["W"] - 16, meta["W"]))}
    torch.set_num_threads(min(16, max(1, len(os.sched_getaffinity(0)))))
    o32 = O.tsnet_forward(sd, cfg, *inputs)
    i64 = [[t.double() for t in x] if isinstance(x, list) else x.double() for x in inputs]
    o64 = O.tsnet_forward({k: v.double() for k, v in sd.items()}, cfg, *i64)
    r32, r64 = o32["rec_tar_img"], o64["rec_tar_img"]
    views = dict(CROPS)
    if "rec32_sub4" in z.files:
        views["sub4"] = (slice(None, None, 4), slice(None, None, 4))
    noise = meta["ref32_vs_ref64"]
    # the fp64 checker is pinned to the stored fp64 reference data; the fp32 run of this host stays in the reference's noise class
    d_o32 = 0.0
    for tag, (ys, xs) in views.items():
        assert np.abs(r64[:, :, ys, xs].numpy() - z[f"rec64_{tag}"]).max() <= 1e-9, tag
        d_o32 = max(d_o32, float(np.abs(r32[:, :, ys, xs].numpy() - z[f"rec32_{tag}"]).max()))
    assert np.abs(r64.sum(dim=3).numpy() - z["rec64_rowsum"]).max() <= 1e-7
    assert d_o32 <= 2 * noise["max"] + 1e-4, f"oracle fp32 on this host is {d_o32:.3e} from the stored reference"
    # the HIP path
    eng = Hh.make_engine(cfg, sd, meta["H"], meta["W"], meta["B"], DEV)
    has_flow = meta.get("has_flow", True)
    rec, flows = Hh.run_engine(eng, inputs, DEV, return_flow=has_flow)
    eng.close()
    d32 = (rec - r32).abs().max().item()
    d64 = (rec.double() - r64).abs().max().item()
    d64_mean = (rec.double() - r64).abs().mean().item()
    d_gold = max(float(np.abs(rec[:, :, ys, xs].numpy() - z[f"rec32_{tag}"]).max()) for tag, (ys, xs) in views.items())
    d_rows = float(np.abs(rec.double().sum(dim=3).numpy() - z["rec32_rowsum"]).max())
    d_flow = max(float(np.abs(flows[i].numpy() - z[f"flow32_{i}"]).max()) for i in range(cfg.n_source)) if has_flow else 0.0
    row = dict(pair=name, mask=meta["mask_mode"], gpu_vs_ref32_stored=d_gold, gpu_vs_oracle32_here=d32, gpu_vs_ref64=d64,
               gpu_vs_ref64_mean=d64_mean, ref32_vs_ref64=noise["max"], ref32_vs_ref64_mean=noise["mean"],
               oracle32_here_vs_ref32_stored=d_o32, rowsum=d_rows, flow=d_flow)
    _rows.append(row)
    print("[sweep] " + json.dumps(row))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "seed_sweep_margins.json"), "w") as f:
            json.dump(_rows, f, indent=1)
    assert d_flow <= TOL_FLOW
    assert d_gold <= TOL_REC, f"max|GPU - stored ref32| = {d_gold:.3e}"
    assert d_rows <= meta["W"] * TOL_REC
    assert d32 <= TOL_REC, f"max|GPU - oracle32 on this host| = {d32:.3e} over the whole image"
    assert d64 <= noise["max"] + 1e-4, f"GPU is further from the fp64 reference ({d64:.3e}) than the fp32 reference is ({noise['max']:.3e})"
